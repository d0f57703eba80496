"""`Architecture` on librlx for the two network families of the BASELINE configs C1-C3, i.e. what
GeneralTensorFlowNetwork builds (architectures/tensorflow_components/general_network.py:228-405) behind
the calls of architecture.py:26-237 and tensorflow_components/architecture.py:312-385, 469-521, 598-607:
  * DQN / DDQN: embedder -> FC middleware -> QHead or DuelingQHead, MSE / Huber regression against
    explicit [B, A] targets (`DQNNetworkParameters`);
  * Clipped PPO: two separate embedder + middleware copies, head 0 = VHead, head 1 = PPOHead, fed the way
    ClippedPPOAgent.train_network feeds it (clipped_ppo_agent.py:226-266): inputs 'observation',
    'output_1_0' = actions, 'output_1_1' [, 'output_1_2'] = old policy distribution, the last
    'output_1_n' = clip_param_rescaler; targets [value targets, advantages]; additional_fetches taken
    from `online_network.output_heads[1]` (`ClippedPPONetworkParameters`);
  * DDPG / TD3 actor (`*ActorNetworkParameters`): predict -> actions; predict(outputs=
    weighted_gradients[0], initial_feed_dict={gradients_weights_ph[0]: w}) -> d sum(actions * w)/d theta
    (ddpg_agent.py:183-193);
  * DDPG / TD3 critic (`*CriticNetworkParameters`, inputs 'observation' + 'action'): predict ->
    [Q, mean Q] or [Q1, Q2, min(Q1, Q2), mean Q1]; predict(outputs=gradients_wrt_inputs[k]['action']) with
    k = the mean output -> d mean(Q1)/d action (ddpg_agent.py:169-173, td3_agent.py:194-198);
    accumulate_gradients(inputs, TD targets) for the loss sum_i mean((y - Q_i)^2);
  * SAC (soft_actor_critic_agent.py:168-280): the policy network (`SACPolicyNetworkParameters`: predict ->
    [mu, log_std, raw actions, actions, logprob, mean logprob] with a FRESH noise draw per call, like
    every sess.run of the TF graph; weighted_gradients[5] / [3] with gradients_weights_ph[5] / [3]), the
    twin-Q network (`SACCriticNetworkParameters`, inputs 'observation' + 'output_0_0': predict ->
    [min(Q1, Q2), its mean], outputs=[q_head.q1_output, q_head.q2_output],
    gradients_wrt_inputs[1]['output_0_0'], train_on_batch with q1_loss / q2_loss fetches) and the value
    network (`SACValueNetworkParameters`).  Gradients of the policy network are handed out as a list of
    per-variable tensors, because the agent text combines them variable by variable.

One device network (`coach_amd.nn.networks.DQNNet`) holds the online and the target weights in one
allocation; the `…/online` and `…/target` HipArchitecture objects of a NetworkWrapper are two views of
it.  Inputs are numpy arrays (or device tensors) owned by the caller, outputs fresh numpy arrays, as in
the reference; `get_weights()` hands out a backend-native handle that is only good for `set_weights`
of a sibling network (network_wrapper.py:109-125).  `accumulated_gradients` is one flat device buffer
instead of a list of per-variable arrays.
"""
import numpy as np
import torch

from .. import _rlx
from ..nn.actor_critic_nets import ActorNet, CriticNet, SACPolicyNet, SACQNet, SACValueNet
from ..nn.networks import ClippedPPONet, DQNNet
from .architecture import Architecture
from .head_parameters import DuelingQHeadParameters


def squeeze_list(var):
    """utils.py squeeze_list: a one-element list is returned as its element."""
    return var[0] if isinstance(var, (list, tuple)) and len(var) == 1 else var


class _HeadFetches(object):
    """`online_network.output_heads[i].<name>`: the handles agents put into additional_fetches."""

    def __init__(self, head_idx, names):
        for n in names:
            setattr(self, n, "output_heads/%d/%s" % (head_idx, n))


def _adam_args(np_):
    return (np_.learning_rate, np_.adam_optimizer_beta1, np_.adam_optimizer_beta2, np_.optimizer_epsilon)


class HipArchitecture(Architecture):
    """Common part of every network family: device plumbing, the online / target views of one device
    network, gradient accumulation and application, weights, variables, savers.  `construct` picks
    the family from the class of the network parameters (the role `GeneralTensorFlowNetwork` plays for
    the TF backend); the families below add what their heads need."""

    inputs = ['observation']

    @staticmethod
    def construct(variable_scope, devices, *args, **kwargs):
        """devices: [torch.device | 'cuda:N'] — the first entry is used (one process per GPU)."""
        agent_parameters = kwargs["agent_parameters"] if "agent_parameters" in kwargs else args[0]
        name = kwargs["name"] if "name" in kwargs else (args[2] if len(args) > 2 else "")
        cls = _family_of(agent_parameters.network_wrappers[name.split('/')[0]])
        return cls(*args, device=devices[0] if devices else None, variable_scope=variable_scope, **kwargs)

    def __init__(self, agent_parameters, spaces, name="", global_network=None, network_is_local=True,
                 network_is_trainable=True, device=None, variable_scope="", shared_with=None):
        super().__init__(agent_parameters, spaces, name)
        if type(self) is HipArchitecture:
            raise TypeError("use HipArchitecture.construct(...): the family is chosen from the network parameters")
        if global_network is not None:
            raise NotImplementedError("parameter-server (global network) mode is replaced by the gradient "
                                      "all-reduce of coach_amd.distributed.GradientSync")
        if not torch.cuda.is_available():
            raise _rlx.RlxUnavailable("HipArchitecture needs a GPU: the HIP path has no CPU fallback")
        self.variable_scope = variable_scope
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.network_is_local, self.network_is_trainable = network_is_local, network_is_trainable
        self.is_target = shared_with is not None
        if shared_with is None:
            self.net = self._build(agent_parameters, spaces, self.network_parameters,
                                   getattr(agent_parameters, "seed", 0) or 0)
        else:
            self.net = shared_with.net                    # the target view of the same device network
        self._handles()
        self.accumulated_gradients = None
        self.sess = None
        self.current_learning_rate = self.learning_rate

    # ---- what a family provides ------------------------------------------------------------
    def _build(self, agent_parameters, spaces, np_, seed):
        raise NotImplementedError

    def _handles(self):
        """Fetch / placeholder handles (output_heads, gradients_wrt_inputs, ...) and input names."""

    def _queue(self, inputs, tag):
        """Queue the forward pass on the device stream; returns whatever _fetch needs."""
        raise NotImplementedError

    def _fetch(self, queued):
        """-> the list of numpy outputs of predict()."""
        return [t.cpu().numpy() for t in queued]

    def _predict_fetch(self, inputs, outputs, feed):
        raise NotImplementedError("outputs={} is not a fetch of this network".format(outputs))

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        """Forward + loss + backward into params.grads (+ self.net.norm); returns
        (total_loss, losses, norm_unclipped_grads, fetched)."""
        raise NotImplementedError

    # ------------------------------------------------------------------------------ plumbing
    def _weights_buffer(self):
        return self.net.target if self.is_target else self.net.params.weights

    def _to_device(self, array, dtype=None):
        if isinstance(array, torch.Tensor):
            t = array.to(self.device)
        else:
            t = torch.from_numpy(np.ascontiguousarray(array)).to(self.device)
        return t if dtype is None or t.dtype == dtype else t.to(dtype)

    def _observation(self, inputs):
        for k in inputs:
            if isinstance(k, str) and k not in self.inputs:
                raise ValueError('input name {} was provided to create a feed dictionary, but there is no '
                                 'placeholder with that name. placeholder names available include: {}'
                                 .format(k, ', '.join(self.inputs)))
        obs = inputs['observation']
        B = int(obs.shape[0])
        obs = self._to_device(obs, torch.uint8 if self.net.image else torch.float32).contiguous()
        shape = tuple(getattr(self.net, "obs_shape", None) or (self.net.obs_dim,))
        if tuple(obs.shape[1:]) != shape:
            raise ValueError("observation shape {} does not match the network input {}"
                             .format(tuple(obs.shape[1:]), shape))
        return obs, B

    def _action(self, inputs, B, key='action'):
        if key not in inputs:
            raise ValueError("the critic needs the input {}".format(key))
        a = self._to_device(inputs[key], torch.float32).contiguous()
        if tuple(a.shape) != (B, self.net.A):
            raise ValueError("action shape {} does not match {}".format(tuple(a.shape), (B, self.net.A)))
        return a

    def _per_sample(self, array, B, what):
        t = self._to_device(array[0] if isinstance(array, (list, tuple)) else array, torch.float32)
        t = t.reshape(-1).contiguous()
        if t.numel() != B:
            raise ValueError("{} must hold one value per sample".format(what))
        return t

    # ------------------------------------------------------------------------------ inference
    def predict(self, inputs, outputs=None, squeeze_output=True, initial_feed_dict=None):
        """One entry per head output (general_network.py outputs), fresh numpy arrays; `outputs` selects
        one of the family's fetch handles instead."""
        if outputs is not None:
            return self._predict_fetch(inputs, outputs, initial_feed_dict or {})
        out = self._fetch(self._queue(inputs, "predict"))
        return squeeze_list(out) if squeeze_output else out

    @staticmethod
    def parallel_predict(sess, network_input_tuples):
        """The reference runs the listed networks in one session.run; here every forward pass is queued
        on the device stream back to back and the results are fetched afterwards."""
        queued = [(net, net._queue(inputs, "pp%d" % i)) for i, (net, inputs) in enumerate(network_input_tuples)]
        return tuple(squeeze_list(net._fetch(q)) for net, q in queued)

    # ------------------------------------------------------------------------------- training
    def reset_accumulated_gradients(self):
        if self.accumulated_gradients is None:
            self.accumulated_gradients = torch.zeros_like(self.net.params.grads)
        else:
            self.accumulated_gradients.zero_()

    def accumulate_gradients(self, inputs, targets, additional_fetches=None, importance_weights=None,
                             no_accumulation=False):
        if not self.network_is_trainable or self.is_target:
            raise ValueError("{} is not trainable".format(self.name))
        if self.accumulated_gradients is None:
            self.reset_accumulated_gradients()
        result = self._accumulate_impl(inputs, targets, additional_fetches or [], importance_weights)
        net = self.net
        lib, s, n = net.lib, _rlx.current_stream(), net.params.size
        if no_accumulation:
            lib.axpby(self.accumulated_gradients, 1.0, net.params.grads, 0.0, None, n, s)
        else:
            lib.axpby(self.accumulated_gradients, 1.0, self.accumulated_gradients, 1.0, net.params.grads, n, s)
        return result

    def apply_gradients(self, gradients, scaler=1.):
        """The gradients are MULTIPLIED by scaler (tensorflow_components/architecture.py:469-521)."""
        p = self.net.params
        if isinstance(gradients, (list, tuple)):
            # per-variable tensors (the SAC policy text combines gradients variable by variable)
            if len(gradients) != len(p.entries) or any(e[2] != 1 for e in p.entries.values()):
                raise ValueError("a gradient list must hold one tensor per variable of a single-tower network")
            for name, g in zip(p.entries, gradients):
                dst = p.g(name)
                dst.copy_(self._to_device(g, torch.float32).reshape(dst.shape))
            gradients = p.grads
        if not isinstance(gradients, torch.Tensor) or gradients.numel() != p.size:
            raise ValueError("gradients must be the accumulated_gradients buffer of an identical network")
        self.net.adam.step(float(scaler), lr=self.current_learning_rate, grads=gradients)

    def apply_and_reset_gradients(self, gradients, scaler=1.):
        self.apply_gradients(gradients, scaler)
        self.reset_accumulated_gradients()

    def train_on_batch(self, inputs, targets, scaler=1., additional_fetches=None, importance_weights=None):
        result = self.accumulate_gradients(inputs, targets, additional_fetches, importance_weights)
        self.apply_and_reset_gradients(self.accumulated_gradients, scaler)
        return result

    # -------------------------------------------------------------------------------- weights
    def get_weights(self):
        return self._weights_buffer()

    def set_weights(self, weights, rate=1.0):
        """new = rate * given + (1 - rate) * old (architecture.py:598-607)."""
        own = self._weights_buffer()
        if not isinstance(weights, torch.Tensor) or weights.numel() != own.numel():
            raise ValueError("weights must come from get_weights() of an identical network")
        self.net.lib.mix_weights(own, weights, own.numel(), float(rate), _rlx.current_stream())

    def get_variable_value(self, variable):
        """variable: 'learning_rate' or a parameter name ('main/q_head/dense/kernel'; tower 0)."""
        if variable == 'learning_rate':
            return np.float32(self.current_learning_rate)
        return self.net.params.w(variable, 0, self._weights_buffer()).cpu().numpy()

    def set_variable_value(self, assign_op, value, placeholder=None):
        if assign_op == 'learning_rate':
            self.current_learning_rate = float(value)
            return
        dst = self.net.params.w(assign_op, 0, self._weights_buffer())
        dst.copy_(self._to_device(np.asarray(value, dtype=np.float32)).view_as(dst))

    def collect_savers(self, parent_path_suffix):
        from ..checkpoint import NetworkSaver
        return [NetworkSaver("%s.%s" % (parent_path_suffix, self.name.replace('/', '.')), self.net)]

    # ---- de-facto members the reference agents touch
    def set_session(self, sess):
        self.sess = sess

    def set_is_training(self, state):
        self.is_training = bool(state)

    def reset_internal_memory(self):
        pass


# ================================================================================ DQN / DDQN
class QArchitecture(HipArchitecture):
    """embedder -> FC middleware -> QHead | DuelingQHead; regression against explicit [B, A] targets."""

    def _build(self, agent_parameters, spaces, np_, seed):
        head = np_.heads_parameters[0]
        return DQNNet(
            self.device, tuple(int(x) for x in spaces.state['observation'].shape), len(spaces.action.actions),
            activation=np_.activation_function, embedder=np_.embedder_scheme, middleware=np_.middleware_scheme,
            learning_rate=np_.learning_rate, adam_beta1=np_.adam_optimizer_beta1,
            adam_beta2=np_.adam_optimizer_beta2, optimizer_epsilon=np_.optimizer_epsilon,
            replace_mse_with_huber_loss=np_.replace_mse_with_huber_loss, seed=seed,
            dueling=isinstance(head, DuelingQHeadParameters), head_activation=head.activation_function,
            head_gradient_rescale=head.rescale_gradient_from_head_by_factor,
            clip_gradients=getattr(np_, "clip_gradients", None))

    def _queue(self, inputs, tag):
        obs, B = self._observation(inputs)
        net = self.net
        return [net.q_values(obs, B, use_target=self.is_target, tag="%s_%d" % (tag, B)).data.view(B, net.A)]

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        net = self.net
        obs, B = self._observation(inputs)
        target = targets[0] if isinstance(targets, (list, tuple)) else targets
        target = self._to_device(target, torch.float32).contiguous()
        if tuple(target.shape) != (B, net.A):
            raise ValueError("targets shape {} does not match the head output {}"
                             .format(tuple(target.shape), (B, net.A)))
        w = importance_weights[0] if isinstance(importance_weights, (list, tuple)) else importance_weights
        if w is not None:
            w = self._per_sample(np.asarray(w) if not isinstance(w, torch.Tensor) else w, B, "importance_weights")
        net.accumulate_regression(obs, B, target, w)
        total_loss = float(net.loss.item())
        net.check_status()
        return total_loss, [total_loss], float(net.norm.item()), []


# =============================================================================== Clipped PPO
class PPOArchitecture(HipArchitecture):
    """Two embedder + middleware copies, head 0 = VHead, head 1 = PPOHead (discrete or continuous)."""

    def _build(self, agent_parameters, spaces, np_, seed):
        alg = agent_parameters.algorithm
        continuous = not hasattr(spaces.action, "actions")
        n_act = int(spaces.action.shape[0]) if continuous else len(spaces.action.actions)
        return ClippedPPONet(
            self.device, tuple(int(x) for x in spaces.state['observation'].shape), n_act,
            activation=np_.activation_function, embedder=np_.embedder_scheme, middleware=np_.middleware_scheme,
            learning_rate=np_.learning_rate, adam_beta1=np_.adam_optimizer_beta1,
            adam_beta2=np_.adam_optimizer_beta2, optimizer_epsilon=np_.optimizer_epsilon,
            clip_likelihood_ratio_using_epsilon=alg.clip_likelihood_ratio_using_epsilon,
            beta_entropy=alg.beta_entropy, seed=seed, continuous=continuous)

    def _handles(self):
        self.n_dist = 2 if self.net.continuous else 1           # old policy: [probs] or [mean, std]
        self.inputs = ['observation'] + ['output_1_%d' % i for i in range(self.n_dist + 2)]
        self.output_heads = [_HeadFetches(0, ['output']),
                             _HeadFetches(1, ['kl_divergence', 'entropy', 'likelihood_ratio',
                                              'clipped_likelihood_ratio', 'output'])]

    def _queue(self, inputs, tag):
        """[V (B, 1), policy probabilities (B, A)] or [V, policy_mean, policy_std]."""
        obs, B = self._observation(inputs)
        net, tag = self.net, "%s_%d" % (tag, B)
        w = net.target if self.is_target else None
        acts = net.torso.forward(net.ctx, net.obs_tensor(obs, B), tag=tag + "v", weights=w, t0=0, nt=1)
        v = net.v_head.forward(net.ctx, acts[-1], tag=tag + "v", weights=w).data.view(B, 1)
        if net.continuous:
            mean, std = net.policy_mean_std(obs, B, use_target=self.is_target, tag=tag)
            return [v, mean, std]
        return [v, net.policy_probs(obs, B, use_target=self.is_target, tag=tag)]

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        """The feed of ClippedPPOAgent.train_network (clipped_ppo_agent.py:226-266)."""
        net, n_dist = self.net, self.n_dist
        obs, B = self._observation(inputs)
        for k in ['output_1_%d' % i for i in range(n_dist + 1)]:
            if k not in inputs:
                raise ValueError("the PPO head needs the input {}".format(k))
        if not isinstance(targets, (list, tuple)) or len(targets) != 2:
            raise ValueError("targets must be [value targets, advantages]")
        f32 = torch.float32
        actions = self._to_device(inputs['output_1_0'], f32 if net.continuous else torch.int32).contiguous()
        old = [self._to_device(inputs['output_1_%d' % (i + 1)], f32).contiguous() for i in range(n_dist)]
        rescaler = float(np.asarray(inputs.get('output_1_%d' % (n_dist + 1), 1.0)).reshape(-1)[0])
        value_targets = self._per_sample(targets[0], B, "value targets")
        advantages = self._per_sample(targets[1], B, "advantages")
        if actions.shape[0] != B or any(tuple(o.shape) != (B, net.A) for o in old):
            raise ValueError("head inputs do not match the batch size {}".format(B))
        ratio = torch.empty(B, dtype=f32, device=self.device)
        clipped = torch.empty(B, dtype=f32, device=self.device)
        net.forward_backward(obs, B, actions, advantages, value_targets, tuple(old) if net.continuous else old[0],
                             rescaler, ratio, clipped)
        net.grad_norm()
        sc = net.scalars.cpu().numpy()      # [surrogate, entropy, kl, policy-head total, value loss, grad norm]
        net.check_status()
        head = self.output_heads[1]
        table = {head.kl_divergence: lambda: sc[2], head.entropy: lambda: sc[1],
                 head.likelihood_ratio: lambda: ratio.cpu().numpy(),
                 head.clipped_likelihood_ratio: lambda: clipped.cpu().numpy()}
        if any(f not in table for f in additional_fetches):
            raise ValueError("unknown fetch in {}".format(additional_fetches))
        losses = [float(sc[4]), float(sc[3])]                   # head 0 (V), head 1 (PPO)
        return float(sc[4] + sc[3]), losses, float(sc[5]), [table[f]() for f in additional_fetches]


# ================================================================================ DDPG / TD3
def _no_batchnorm(np_):
    if getattr(np_, "batchnorm", False):
        # the moving averages step on the batch fed WITH apply_gradients (additional_inputs, ddpg_agent.py:178-193); this
        # interface keeps no such pass — coach_amd.agents.ddpg_agent.DDPGAgent runs the batch-normalised networks
        raise NotImplementedError("batch-normalised DDPG networks are driven by coach_amd.agents.ddpg_agent.DDPGAgent, "
                                  "not through the Architecture plug-in interface")


class ActorArchitecture(HipArchitecture):
    def _build(self, agent_parameters, spaces, np_, seed):
        scale = float(np.maximum(np.abs(spaces.action.low), np.abs(spaces.action.high)).max())
        _no_batchnorm(np_)
        return ActorNet(self.device, int(spaces.state['observation'].shape[0]), int(spaces.action.shape[0]), scale,
                        np_.observation_embedder_scheme, np_.middleware_scheme, np_.activation_function,
                        *_adam_args(np_), seed)

    def _handles(self):
        self.gradients_weights_ph = ["gradients_weights_ph/0"]
        self.weighted_gradients = ["weighted_gradients/0"]

    def _queue(self, inputs, tag):
        obs, B = self._observation(inputs)
        return [self.net.forward(obs, B, use_target=self.is_target, tag="%s_%d" % (tag, B))[0]]

    def _predict_fetch(self, inputs, outputs, feed):
        """weighted_gradients[0]: d sum(actions * w) / d theta with w = feed[gradients_weights_ph[0]]."""
        if squeeze_list(outputs) != self.weighted_gradients[0]:
            return super()._predict_fetch(inputs, outputs, feed)
        net = self.net
        obs, B = self._observation(inputs)
        if self.is_target or self.gradients_weights_ph[0] not in feed:
            raise ValueError("weighted_gradients needs initial_feed_dict[gradients_weights_ph[0]] on the "
                             "online actor")
        w = self._to_device(np.asarray(feed[self.gradients_weights_ph[0]], dtype=np.float32)).contiguous()
        if tuple(w.shape) != (B, net.A):
            raise ValueError("gradient weights shape {} does not match {}".format(tuple(w.shape), (B, net.A)))
        _, saved = net.forward(obs, B, tag="wgrad")
        net.backward(saved, w, B)
        return net.params.grads.clone()

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        raise NotImplementedError("the actor has no loss head: its gradients come from "
                                  "predict(outputs=weighted_gradients[0])")


class CriticArchitecture(HipArchitecture):
    """DDPG critic (one stream: outputs [Q, mean Q]) / TD3 twin critic ([Q1, Q2, min, mean Q1])."""

    def _build(self, agent_parameters, spaces, np_, seed):          # the agents seed the critic with seed + 1
        _no_batchnorm(np_)
        return CriticNet(self.device, int(spaces.state['observation'].shape[0]), int(spaces.action.shape[0]),
                         np_.observation_embedder_scheme, np_.middleware_scheme, np_.num_streams,
                         np_.activation_function, np_.head_initializer, *_adam_args(np_), seed + 1)

    def _handles(self):
        self.inputs = ['observation', 'action']
        n_out = 2 if self.net.T == 1 else 4
        self.gradients_wrt_inputs = [{k: "gradients_wrt_inputs/%d/%s" % (i, k) for k in self.inputs}
                                     for i in range(n_out)]

    def _queue(self, inputs, tag):
        obs, B = self._observation(inputs)
        q, _ = self.net.forward(obs, self._action(inputs, B), B, use_target=self.is_target, tag="%s_%d" % (tag, B))
        return q, B

    def _fetch(self, queued):
        q, B = queued
        q = q.cpu().numpy()                                         # [T, B]
        if self.net.T == 1:
            return [q[0].reshape(B, 1), q[0].mean()]
        return [q[0].reshape(B, 1), q[1].reshape(B, 1), np.minimum(q[0], q[1]).reshape(B, 1), q[0].mean()]

    def _predict_fetch(self, inputs, outputs, feed):
        """gradients_wrt_inputs[mean output]['action'] = d mean(Q1) / d action."""
        mean_output = len(self.gradients_wrt_inputs) - 1
        if squeeze_list(outputs) != self.gradients_wrt_inputs[mean_output]['action'] or self.is_target:
            raise NotImplementedError("only gradients_wrt_inputs[{}]['action'] of the online critic is "
                                      "available (d mean(Q1) / d action)".format(mean_output))
        net = self.net
        obs, B = self._observation(inputs)
        _, saved = net.forward(obs, self._action(inputs, B), B, tag="agrad")
        g = torch.empty(B, net.A, dtype=torch.float32, device=self.device)
        net.action_gradient(saved, B, g, scale=1.0)
        return g.cpu().numpy()

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        net = self.net
        obs, B = self._observation(inputs)
        actions = self._action(inputs, B)
        target = self._per_sample(targets, B, "TD targets")
        _, saved = net.forward(obs, actions, B, tag="train")
        net.train_backward(saved, target, B)
        net.grad_norm()
        losses = [float(x) for x in net.loss[:net.T].cpu().numpy()]
        return float(sum(losses)), losses, float(net.norm.item()), []


# ======================================================================================= SAC
class SACPolicyArchitecture(HipArchitecture):
    """predict -> [mu, log_std, raw actions, actions, logprob, mean logprob]; every call draws fresh
    N(0, 1) noise from np.random, like every sess.run of the TF graph's sampling op."""

    def _build(self, agent_parameters, spaces, np_, seed):
        return SACPolicyNet(self.device, int(spaces.state['observation'].shape[0]), int(spaces.action.shape[0]),
                            np_.embedder_scheme, np_.middleware_scheme, *_adam_args(np_), seed=seed)

    def _handles(self):
        self.gradients_weights_ph = ["gradients_weights_ph/%d" % i for i in range(6)]
        self.weighted_gradients = ["weighted_gradients/%d" % i for i in range(6)]

    def _sample(self, inputs, tag):
        obs, B = self._observation(inputs)
        normals = self._to_device(np.random.standard_normal((B, self.net.A)), torch.float64).contiguous()
        return self.net.forward(obs, B, normals, tag=tag), B

    def _queue(self, inputs, tag):
        (o, _), B = self._sample(inputs, tag)
        return o

    def _fetch(self, o):
        lp = o["logprob"].cpu().numpy()
        return [o["mean"].cpu().numpy(), o["log_std"].cpu().numpy(), o["raw_actions"].cpu().numpy(),
                o["actions"].cpu().numpy(), lp, lp.mean()]

    def _grad_list(self):
        """params.grads as a list of per-variable tensors (views of one fresh flat copy)."""
        p = self.net.params
        flat = p.grads.clone()
        return [p.view(flat, name) for name in p.entries]

    def _predict_fetch(self, inputs, outputs, feed):
        fetch = squeeze_list(outputs)
        if fetch not in (self.weighted_gradients[5], self.weighted_gradients[3]):
            raise NotImplementedError("only weighted_gradients[5] (mean log-prob) and [3] (actions) exist")
        idx = 5 if fetch == self.weighted_gradients[5] else 3
        if self.gradients_weights_ph[idx] not in feed:
            raise ValueError("weighted_gradients[{0}] needs initial_feed_dict[gradients_weights_ph[{0}]]".format(idx))
        w = feed[self.gradients_weights_ph[idx]]
        net = self.net
        (_, saved), B = self._sample(inputs, "wgrad")
        if idx == 5:
            net.backward(saved, B, logprob_mean_weight=float(np.asarray(w)))
        else:
            wd = self._to_device(np.asarray(w, dtype=np.float32)).contiguous()
            if tuple(wd.shape) != (B, net.A):
                raise ValueError("gradient weights shape {} does not match {}".format(tuple(wd.shape), (B, net.A)))
            net.backward(saved, B, action_weights=wd)
        return self._grad_list()

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        raise NotImplementedError("the SAC policy has no loss head: its gradients come from "
                                  "predict(outputs=weighted_gradients[k])")


class SACQArchitecture(HipArchitecture):
    """Twin Q head: predict -> [min(Q1, Q2), its mean]; inputs 'observation' + 'output_0_0' (actions)."""

    def _build(self, agent_parameters, spaces, np_, seed):          # the agent's seeds: policy, q + 1, v + 2
        return SACQNet(self.device, int(spaces.state['observation'].shape[0]), int(spaces.action.shape[0]),
                       np_.network_layers_sizes, *_adam_args(np_), seed=seed + 1)

    def _handles(self):
        self.inputs = ['observation', 'output_0_0']
        self.output_heads = [_HeadFetches(0, ['q1_output', 'q2_output', 'q1_loss', 'q2_loss', 'q_output'])]
        self.gradients_wrt_inputs = [{k: "gradients_wrt_inputs/%d/%s" % (i, k) for k in self.inputs}
                                     for i in range(2)]           # outputs: [q_output, q_output_mean]

    def _queue(self, inputs, tag):
        obs, B = self._observation(inputs)
        q, saved = self.net.forward(obs, self._action(inputs, B, 'output_0_0'), B, tag="%s_%d" % (tag, B))
        return q, saved, B

    def _fetch(self, queued):
        q, _, B = queued
        q = q.cpu().numpy()
        qmin = np.minimum(q[0], q[1])
        return [qmin.reshape(B, 1), qmin.mean()]

    def _predict_fetch(self, inputs, outputs, feed):
        net, head = self.net, self.output_heads[0]
        if isinstance(outputs, (list, tuple)):
            q, _, B = self._queue(inputs, "predict")
            q = q.cpu().numpy()
            table = {head.q1_output: q[0].reshape(B, 1), head.q2_output: q[1].reshape(B, 1),
                     head.q_output: np.minimum(q[0], q[1]).reshape(B, 1)}
            if any(f not in table for f in outputs):
                raise NotImplementedError("unknown fetch in {}".format(outputs))
            return [table[f] for f in outputs]
        if outputs != self.gradients_wrt_inputs[1]['output_0_0']:
            raise NotImplementedError("only gradients_wrt_inputs[1]['output_0_0'] (d mean(min Q) / d action) exists")
        _, saved, B = self._queue(inputs, "agrad")
        g = torch.empty(B, net.A, dtype=torch.float32, device=self.device)
        net.action_gradient(saved, B, g)
        return g.cpu().numpy()

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        net, head = self.net, self.output_heads[0]
        obs, B = self._observation(inputs)
        target = self._per_sample(targets, B, "targets")
        _, saved = net.forward(obs, self._action(inputs, B, 'output_0_0'), B, tag="train")
        net.train_backward(saved, target, B)
        net.grad_norm()
        loss = net.loss.cpu().numpy()
        table = {head.q1_loss: loss[0], head.q2_loss: loss[1]}
        if any(f not in table for f in additional_fetches):
            raise ValueError("unknown fetch in {}".format(additional_fetches))
        total = float(loss[:2].sum())
        return total, [total], float(net.norm.item()), [table[f] for f in additional_fetches]


class SACValueArchitecture(HipArchitecture):
    def _build(self, agent_parameters, spaces, np_, seed):
        return SACValueNet(self.device, int(spaces.state['observation'].shape[0]), np_.embedder_scheme,
                           np_.middleware_scheme, *_adam_args(np_), seed=seed + 2)

    def _queue(self, inputs, tag):
        obs, B = self._observation(inputs)
        return [self.net.forward(obs, B, use_target=self.is_target, tag="%s_%d" % (tag, B))[0].view(B, 1)]

    def _accumulate_impl(self, inputs, targets, additional_fetches, importance_weights):
        net = self.net
        obs, B = self._observation(inputs)
        target = self._per_sample(targets, B, "targets")
        _, saved = net.forward(obs, B, tag="train")
        net.train_backward(saved, target, B)
        net.grad_norm()
        total = float(net.loss.sum().item())
        return total, [total], float(net.norm.item()), []


_BY_PARAMETER_CLASS = {"SACPolicyNetworkParameters": SACPolicyArchitecture,
                       "SACCriticNetworkParameters": SACQArchitecture,
                       "SACValueNetworkParameters": SACValueArchitecture,
                       "ClippedPPONetworkParameters": PPOArchitecture}


def _family_of(network_parameters):
    """The family is decided by the class of `agent_parameters.network_wrappers[name]`, as
    GeneralTensorFlowNetwork decides by the head parameter classes it is given."""
    for cls in type(network_parameters).__mro__:
        if cls.__name__ in _BY_PARAMETER_CLASS:
            return _BY_PARAMETER_CLASS[cls.__name__]
        if "Actor" in cls.__name__:
            return ActorArchitecture
        if "Critic" in cls.__name__:
            return CriticArchitecture
    return QArchitecture
