"""NetworkWrapper (rl_coach/architectures/network_wrapper.py:30-240) over HipArchitecture: the online
network, the optional slow target network, and the calls the agents make on the pair.  The global
(parameter-server) network of the reference's multi-process mode does not exist here — data parallelism
is one gradient all-reduce per update (coach_amd.distributed)."""
from .hip_architecture import HipArchitecture


class NetworkWrapper(object):
    def __init__(self, agent_parameters, has_target, has_global, name, spaces, replicated_device=None,
                 worker_device=None):
        if has_global:
            raise NotImplementedError("has_global: the shared parameter-server network is replaced by "
                                      "GradientSync (one flat all-reduce per update)")
        self.ap = agent_parameters
        self.network_parameters = self.ap.network_wrappers[name]
        self.has_target, self.has_global, self.name = has_target, has_global, name
        self.sess = None
        scope = "{}/{}".format(getattr(agent_parameters, "full_name_id", "agent"), name)
        devices = [worker_device] if worker_device is not None else []
        self.global_network = None
        self.online_network = HipArchitecture.construct(
            scope, devices, agent_parameters=agent_parameters, name='{}/online'.format(name), spaces=spaces,
            global_network=None, network_is_local=True, network_is_trainable=True)
        self.target_network = None
        if has_target:
            self.target_network = HipArchitecture.construct(
                scope, devices, agent_parameters=agent_parameters, name='{}/target'.format(name), spaces=spaces,
                global_network=None, network_is_local=True, network_is_trainable=False,
                shared_with=self.online_network)

    def sync(self):
        self.update_online_network()
        self.update_target_network()

    def update_target_network(self, rate=1.0):
        if self.target_network:
            self.target_network.set_weights(self.online_network.get_weights(), rate)

    def update_online_network(self, rate=1.0):
        if self.global_network:
            self.online_network.set_weights(self.global_network.get_weights(), rate)

    def apply_gradients_to_global_network(self, gradients=None):
        """No global network here: the update lands on the online network (shared_optimizer = False path)."""
        self.apply_gradients_to_online_network(gradients)

    def apply_gradients_to_online_network(self, gradients=None):
        if gradients is None:
            gradients = self.online_network.accumulated_gradients
        self.online_network.apply_gradients(gradients)

    def train_and_sync_networks(self, inputs, targets, additional_fetches=[], importance_weights=None):
        result = self.online_network.accumulate_gradients(inputs, targets, additional_fetches=additional_fetches,
                                                          importance_weights=importance_weights,
                                                          no_accumulation=True)
        self.apply_gradients_and_sync_networks(reset_gradients=False)
        return result

    def apply_gradients_and_sync_networks(self, reset_gradients=True):
        if reset_gradients:
            self.online_network.apply_and_reset_gradients(self.online_network.accumulated_gradients)
        else:
            self.online_network.apply_gradients(self.online_network.accumulated_gradients)

    def parallel_prediction(self, network_input_tuples):
        return type(self.online_network).parallel_predict(self.sess, network_input_tuples)

    def set_is_training(self, state):
        self.online_network.set_is_training(state)
        if self.has_target:
            self.target_network.set_is_training(state)

    def set_session(self, sess):
        self.sess = sess
        self.online_network.set_session(sess)
        if self.target_network:
            self.target_network.set_session(sess)

    def __str__(self):
        sub = ["online network"] + (["target network"] if self.target_network else [])
        return "Network: {}, Copies: {} ({})".format(self.name, len(sub), ' | '.join(sub))
