"""NetworkWrapper (rl_coach/architectures/network_wrapper.py:30-240) over HipArchitecture: the online
network, the optional slow target network, and the calls the agents make on the pair.  The global
(parameter-server) network of the reference's multi-process mode does not exist here — data parallelism
is one gradient all-reduce per update (coach_amd.distributed) — so `global_network` is always None and
every "…to_global_network" call lands on the online network."""
from .hip_architecture import HipArchitecture


class NetworkWrapper(object):
    def __init__(self, agent_parameters, has_target, has_global, name, spaces, replicated_device=None,
                 worker_device=None):
        if has_global:
            raise NotImplementedError("has_global: the shared parameter-server network is replaced by "
                                      "GradientSync (one flat all-reduce per update)")
        self.ap, self.name = agent_parameters, name
        self.network_parameters = agent_parameters.network_wrappers[name]
        self.has_target, self.has_global = bool(has_target), False
        self.sess = self.global_network = None
        scope = "{}/{}".format(getattr(agent_parameters, "full_name_id", "agent"), name)

        def build(role, trainable, **extra):
            return HipArchitecture.construct(
                scope, [worker_device] if worker_device is not None else [], agent_parameters=agent_parameters,
                name="%s/%s" % (name, role), spaces=spaces, global_network=None, network_is_local=True,
                network_is_trainable=trainable, **extra)

        self.online_network = build("online", True)
        # the target network is a second view of the SAME device allocation ([2, size] weights)
        self.target_network = build("target", False, shared_with=self.online_network) if has_target else None

    def _members(self):
        return [n for n in (self.online_network, self.target_network) if n is not None]

    # ---- weight movement ------------------------------------------------------------------
    def update_target_network(self, rate=1.0):
        """online >>> target: target = rate * online + (1 - rate) * target."""
        tgt = self.target_network
        if tgt is not None:
            tgt.set_weights(self.online_network.get_weights(), rate)

    def update_online_network(self, rate=1.0):
        """global >>> online; a no-op without a global network."""

    def sync(self):
        self.update_online_network()
        self.update_target_network()

    # ---- training --------------------------------------------------------------------------
    def apply_gradients_to_online_network(self, gradients=None, additional_inputs=None):
        """network_wrapper.py:143-154 (additional_inputs: batch-norm update ops need none here — the statistics of the
        tracked pass are committed with the step, nn/graph.py BatchNorm)."""
        net = self.online_network
        net.apply_gradients(net.accumulated_gradients if gradients is None else gradients)

    def apply_gradients_to_global_network(self, gradients=None, additional_inputs=None):
        """network_wrapper.py:127-141: with a shared optimizer the GLOBAL network takes the step; without a global
        network (always, here: workers exchange gradients, not parameters) the online network does."""
        self.apply_gradients_to_online_network(gradients, additional_inputs)

    def collect_savers(self, parent_path_suffix):
        """network_wrapper.py:250-270: savers of the copy that holds the most recent parameters — the online network
        (a global network would take precedence; the target network never does)."""
        return self.online_network.collect_savers(parent_path_suffix)

    def apply_gradients_and_sync_networks(self, reset_gradients=True, additional_inputs=None):
        net = self.online_network
        step = net.apply_and_reset_gradients if reset_gradients else net.apply_gradients
        step(net.accumulated_gradients)

    def train_and_sync_networks(self, inputs, targets, additional_fetches=[], importance_weights=None,
                                use_inputs_for_apply_gradients=False):
        """accumulate (overwriting) + apply; returns accumulate_gradients' tuple
        (total_loss, losses, norm_unclipped_grads, fetched)."""
        out = self.online_network.accumulate_gradients(
            inputs, targets, additional_fetches=additional_fetches, importance_weights=importance_weights,
            no_accumulation=True)
        self.apply_gradients_and_sync_networks(reset_gradients=False)
        return out

    # ---- inference / bookkeeping --------------------------------------------------------------
    def parallel_prediction(self, network_input_tuples):
        return HipArchitecture.parallel_predict(self.sess, network_input_tuples)

    def set_is_training(self, state):
        for n in self._members():
            n.set_is_training(state)

    def set_session(self, sess):
        self.sess = sess
        for n in self._members():
            n.set_session(sess)

    def __str__(self):
        roles = ["online network"] + (["target network"] if self.target_network is not None else [])
        return "Network: {}, Copies: {} ({})".format(self.name, len(roles), ' | '.join(roles))
