"""The backend interface of rl_coach (architectures/architecture.py:26-237): what NetworkWrapper and
the agents call on a network, independent of the framework underneath.  `HipArchitecture`
(hip_architecture.py) is the gfx950 implementation; the agents of this package drive the same device
networks through fused, graph-captured update paths, this interface is the drop-in boundary for code
written against the reference's Architecture."""


class Architecture(object):
    @staticmethod
    def construct(variable_scope, devices, *args, **kwargs):
        """Build a network under `variable_scope` on `devices`; the remaining arguments go to the
        class initializer."""
        raise NotImplementedError

    def __init__(self, agent_parameters, spaces, name=""):
        self.spaces = spaces
        self.name = name
        self.network_wrapper_name = self.name.split('/')[0]          # 'main/online' -> 'main'
        self.full_name = "{}/{}".format(getattr(agent_parameters, "full_name_id", "agent"), name)
        self.network_parameters = agent_parameters.network_wrappers[self.network_wrapper_name]
        self.batch_size = self.network_parameters.batch_size
        self.learning_rate = self.network_parameters.learning_rate
        self.optimizer = None
        self.ap = agent_parameters

    def predict(self, inputs, outputs=None, squeeze_output=True, initial_feed_dict=None):
        raise NotImplementedError

    @staticmethod
    def parallel_predict(sess, network_input_tuples):
        raise NotImplementedError

    def train_on_batch(self, inputs, targets, scaler=1., additional_fetches=None, importance_weights=None):
        raise NotImplementedError

    def get_weights(self):
        raise NotImplementedError

    def set_weights(self, weights, rate=1.0):
        raise NotImplementedError

    def reset_accumulated_gradients(self):
        raise NotImplementedError

    def accumulate_gradients(self, inputs, targets, additional_fetches=None, importance_weights=None,
                             no_accumulation=False):
        raise NotImplementedError

    def apply_and_reset_gradients(self, gradients, scaler=1.):
        raise NotImplementedError

    def apply_gradients(self, gradients, scaler=1.):
        raise NotImplementedError

    def get_variable_value(self, variable):
        raise NotImplementedError

    def set_variable_value(self, assign_op, value, placeholder):
        raise NotImplementedError

    def collect_savers(self, parent_path_suffix):
        raise NotImplementedError
