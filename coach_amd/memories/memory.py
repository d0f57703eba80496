"""What every device replay has in common: the size unit enum the presets use
(`(MemoryGranularity.Transitions, n)`, rl_coach/memories/memory.py:24-26) and the handful of calls the
agents make on a memory (`store`, `sample`, `num_transitions`, `length`, `clean`; memory.py:41-77).
The reference's memory *backend* hook (Redis pub/sub of transitions) is out of scope."""
import enum


class MemoryGranularity(enum.Enum):
    Transitions = 0
    Episodes = 1


class MemoryParameters(object):
    """Parameter holder resolved through `path`, like every rl_coach Parameters object."""
    max_size = None

    @property
    def path(self):
        return 'coach_amd.memories.memory:Memory'


class Memory(object):
    """Capacity bookkeeping; the storage itself lives in the subclasses' device tensors."""

    def __init__(self, max_size):
        unit, amount = max_size
        if not isinstance(unit, MemoryGranularity):
            raise ValueError("max_size must be (MemoryGranularity, count), got {!r}".format(max_size))
        self.max_size = (unit, int(amount))

    def _unsupported(self, what):
        raise NotImplementedError("{} does not implement {}".format(type(self).__name__, what))

    def store(self, *columns, **kw):
        self._unsupported("store")

    def sample(self, size):
        self._unsupported("sample")

    def num_transitions(self):
        self._unsupported("num_transitions")

    def length(self):
        return self.num_transitions()

    def clean(self):
        self._unsupported("clean")
