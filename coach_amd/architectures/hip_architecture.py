"""`Architecture` on librlx: the value-network family of the hot path (embedder -> FC middleware ->
QHead or DuelingQHead, MSE / Huber regression against explicit targets), i.e. what
GeneralTensorFlowNetwork builds for DQN / DDQN (architectures/tensorflow_components/general_network.py:
228-405) behind the calls of architecture.py:26-237 and tensorflow_components/architecture.py:312-385,
469-521, 598-607.

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
from ..nn.networks import DQNNet
from .architecture import Architecture
from .head_parameters import DuelingQHeadParameters


def squeeze_list(var):
    """utils.py squeeze_list: a one-element list is returned as its element."""
    return var[0] if isinstance(var, (list, tuple)) and len(var) == 1 else var


class HipArchitecture(Architecture):
    @staticmethod
    def construct(variable_scope, devices, *args, **kwargs):
        """devices: [torch.device | 'cuda:N'] — the first entry is used (one process per GPU)."""
        dev = devices[0] if devices else None
        return HipArchitecture(*args, device=dev, variable_scope=variable_scope, **kwargs)

    def __init__(self, agent_parameters, spaces, name="", global_network=None, network_is_local=True,
                 network_is_trainable=True, device=None, variable_scope="", shared_with=None):
        super().__init__(agent_parameters, spaces, name)
        if global_network is not None:
            raise NotImplementedError("parameter-server (global network) mode is replaced by the gradient "
                                      "all-reduce of coach_amd.distributed.GradientSync")
        if not torch.cuda.is_available():
            raise _rlx.RlxUnavailable("HipArchitecture needs a GPU: the HIP path has no CPU fallback")
        self.variable_scope = variable_scope
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.network_is_local, self.network_is_trainable = network_is_local, network_is_trainable
        self.is_target = shared_with is not None
        np_ = self.network_parameters
        if shared_with is None:
            obs_shape = tuple(int(x) for x in spaces.state['observation'].shape)
            head = np_.heads_parameters[0]
            self.net = DQNNet(
                self.device, obs_shape, len(spaces.action.actions), activation=np_.activation_function,
                embedder=np_.embedder_scheme, middleware=np_.middleware_scheme,
                learning_rate=np_.learning_rate, adam_beta1=np_.adam_optimizer_beta1,
                adam_beta2=np_.adam_optimizer_beta2, optimizer_epsilon=np_.optimizer_epsilon,
                replace_mse_with_huber_loss=np_.replace_mse_with_huber_loss,
                seed=getattr(agent_parameters, "seed", 0) or 0,
                dueling=isinstance(head, DuelingQHeadParameters), head_activation=head.activation_function,
                head_gradient_rescale=head.rescale_gradient_from_head_by_factor,
                clip_gradients=getattr(np_, "clip_gradients", None))
        else:
            self.net = shared_with.net                    # the target view of the same device network
        self.inputs = ['observation']
        self.accumulated_gradients = None
        self.sess = None
        self.current_learning_rate = self.learning_rate

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
        if tuple(obs.shape[1:]) != tuple(self.net.obs_shape):
            raise ValueError("observation shape {} does not match the network input {}"
                             .format(tuple(obs.shape[1:]), tuple(self.net.obs_shape)))
        return obs, B

    # ------------------------------------------------------------------------------ inference
    def predict(self, inputs, outputs=None, squeeze_output=True, initial_feed_dict=None):
        obs, B = self._observation(inputs)
        q = self.net.q_values(obs, B, use_target=self.is_target, tag="predict%d" % B)
        out = [q.data.view(B, self.net.A).cpu().numpy()]
        return squeeze_list(out) if squeeze_output else out

    @staticmethod
    def parallel_predict(sess, network_input_tuples):
        """The reference runs the listed networks in one session.run; here every forward pass is queued
        on the device stream back to back and the results are fetched afterwards."""
        queued = []
        for net, inputs in network_input_tuples:
            obs, B = net._observation(inputs)
            q = net.net.q_values(obs, B, use_target=net.is_target, tag="pp%d_%d" % (len(queued), B))
            queued.append((q, B, net.net.A))
        return tuple(q.data.view(B, A).cpu().numpy() for q, B, A in queued)

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
        obs, B = self._observation(inputs)
        target = targets[0] if isinstance(targets, (list, tuple)) else targets
        target = self._to_device(target, torch.float32).contiguous()
        if tuple(target.shape) != (B, self.net.A):
            raise ValueError("targets shape {} does not match the head output {}"
                             .format(tuple(target.shape), (B, self.net.A)))
        w = importance_weights[0] if isinstance(importance_weights, (list, tuple)) else importance_weights
        if w is not None:
            w = self._to_device(np.asarray(w).reshape(-1) if not isinstance(w, torch.Tensor) else w.reshape(-1),
                                torch.float32).contiguous()
            if w.numel() != B:
                raise ValueError("importance_weights must hold one value per sample")
        net = self.net
        net.accumulate_regression(obs, B, target, w)
        lib, s, n = net.lib, _rlx.current_stream(), net.params.size
        if no_accumulation:
            lib.axpby(self.accumulated_gradients, 1.0, net.params.grads, 0.0, None, n, s)
        else:
            lib.axpby(self.accumulated_gradients, 1.0, self.accumulated_gradients, 1.0, net.params.grads, n, s)
        total_loss = float(net.loss.item())
        net.check_status()
        return total_loss, [total_loss], float(net.norm.item()), []

    def apply_gradients(self, gradients, scaler=1.):
        """The gradients are MULTIPLIED by scaler (tensorflow_components/architecture.py:469-521)."""
        if not isinstance(gradients, torch.Tensor) or gradients.numel() != self.net.params.size:
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
        """variable: 'learning_rate' or a parameter name ('main/q_head/dense/kernel')."""
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
