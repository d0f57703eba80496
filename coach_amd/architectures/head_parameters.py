"""Head parameter objects of the network wrappers this engine implements — the subset of
rl_coach/architectures/head_parameters.py the hot-path presets put into
``network_wrappers[...].heads_parameters`` (QHeadParameters :120-130, DuelingQHeadParameters
:132-139).  Only the fields the device networks read are kept."""


class HeadParameters(object):
    head_type = None

    def __init__(self, activation_function='relu', name='head', rescale_gradient_from_head_by_factor=1.0,
                 loss_weight=1.0):
        self.activation_function = activation_function
        self.name = name
        # the gradient entering the middleware from this head is multiplied by this factor
        # (general_network.py:296-303: head_input = (1 - f) * stop_gradient(x) + f * x)
        self.rescale_gradient_from_head_by_factor = rescale_gradient_from_head_by_factor
        self.loss_weight = loss_weight


class QHeadParameters(HeadParameters):
    head_type = "QHead"

    def __init__(self, activation_function='relu', name='q_head_params',
                 rescale_gradient_from_head_by_factor=1.0, loss_weight=1.0):
        super().__init__(activation_function, name, rescale_gradient_from_head_by_factor, loss_weight)


class DuelingQHeadParameters(HeadParameters):
    head_type = "DuelingQHead"

    def __init__(self, activation_function='relu', name='dueling_q_head_params',
                 rescale_gradient_from_head_by_factor=1.0, loss_weight=1.0):
        super().__init__(activation_function, name, rescale_gradient_from_head_by_factor, loss_weight)
