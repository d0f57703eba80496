"""`network_wrappers[...].input_embedders_parameters['observation'].scheme = [Dense(400)]`,
`.middleware_parameters.scheme = ...`, `.heads_parameters[0].network_layers_sizes = ...` — the way reference
presets reach into NetworkParameters (base_parameters.py:226-305, embedder_parameters.py, middleware_parameters.py).
The device network parameter classes keep flat fields (`embedder_scheme`, `middleware_scheme`, ...); this mixin
gives them the reference's nested access path as live views onto those fields."""
from .layers import scheme_to_native


class _SchemeView(object):
    def __init__(self, owner, field, as_tuple):
        object.__setattr__(self, "_owner", owner)
        object.__setattr__(self, "_field", field)
        object.__setattr__(self, "_as_tuple", as_tuple)
        object.__setattr__(self, "_extra", owner.__dict__.setdefault("_view_extras", {}).setdefault(field, {}))

    @property
    def scheme(self):
        return getattr(self._owner, self._field)

    @scheme.setter
    def scheme(self, value):
        setattr(self._owner, self._field, scheme_to_native(value, self._as_tuple))

    @property
    def activation_function(self):
        return self._owner.activation_function

    @activation_function.setter
    def activation_function(self, value):
        self._owner.activation_function = value

    def __getattr__(self, name):                      # dropout_rate, ...: remembered, not used
        extra = object.__getattribute__(self, "_extra")
        if name in extra:
            return extra[name]
        owner = object.__getattribute__(self, "_owner")
        if name == "batchnorm" and hasattr(owner, "batchnorm"):
            return owner.batchnorm
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in ("scheme", "activation_function"):
            object.__setattr__(self, name, value)
        else:
            self._extra[name] = value
            if name == "batchnorm" and hasattr(self._owner, "batchnorm"):
                # the device DDPG networks switch batch normalisation on for the whole network (what
                # DDPGAgentParameters(use_batchnorm=True) does, ddpg_agent.py:37-60), not per component
                self._owner.batchnorm = bool(value)


class SchemeViews(object):
    """Mix into a network-parameters class.  _EMBEDDER_FIELDS: {'observation': field[, 'action': field]};
    _TUPLE_SCHEMES: the fields hold tuples of unit counts (actor-critic nets) instead of names / lists."""
    _EMBEDDER_FIELDS = {"observation": "embedder_scheme"}
    _MIDDLEWARE_FIELD = "middleware_scheme"
    _TUPLE_SCHEMES = False

    @property
    def input_embedders_parameters(self):
        return {k: _SchemeView(self, f, self._TUPLE_SCHEMES) for k, f in self._EMBEDDER_FIELDS.items()
                if hasattr(self, f)}

    @property
    def middleware_parameters(self):
        return _SchemeView(self, self._MIDDLEWARE_FIELD, self._TUPLE_SCHEMES)
