"""Layer descriptors presets put into embedder / middleware schemes (rl_coach/architectures/layers.py:
`Dense(units)` :168-185, `Conv2d(num_filters, kernel_size, strides)` :108-121).  Framework-neutral records; the
device networks (coach_amd/nn) read `.units` / `(num_filters, kernel_size, strides)`."""


class Dense(object):
    def __init__(self, units):
        self.units = int(units)

    def __repr__(self):
        return "Dense({})".format(self.units)

    def __eq__(self, other):
        return isinstance(other, Dense) and other.units == self.units


class Conv2d(object):
    def __init__(self, num_filters, kernel_size, strides):
        self.num_filters, self.kernel_size, self.strides = int(num_filters), int(kernel_size), int(strides)

    def __repr__(self):
        return "Conv2d({}, {}, {})".format(self.num_filters, self.kernel_size, self.strides)


def scheme_to_native(scheme, as_tuple):
    """A preset's scheme value -> what the device network builders take: an EmbedderScheme / MiddlewareScheme
    member -> its name ('Medium', ...; 'Empty' -> () for the tuple-style actor-critic networks), a list of layer
    descriptors -> unit counts (Dense) or (filters, kernel, stride) triples (Conv2d)."""
    from enum import Enum
    if isinstance(scheme, Enum):
        if scheme.name == "Empty" and as_tuple:
            return ()
        if as_tuple:
            raise ValueError("this network takes explicit layer lists, not the named scheme {}".format(scheme))
        return scheme.name
    if isinstance(scheme, str):
        return scheme
    out = []
    for layer in scheme:
        if isinstance(layer, Dense):
            out.append(layer.units)
        elif isinstance(layer, Conv2d):
            out.append((layer.num_filters, layer.kernel_size, layer.strides))
        elif isinstance(layer, (int, tuple)):
            out.append(layer)
        else:
            raise ValueError("unsupported layer in a scheme: {!r}".format(layer))
    return tuple(out) if as_tuple else out
