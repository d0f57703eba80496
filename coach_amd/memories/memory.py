"""Memory base types — mirror of rl_coach/memories/memory.py:24-77 (MemoryGranularity, the Memory
method set the agents call: store / sample / length / num_transitions / clean / get)."""
from enum import Enum


class MemoryGranularity(Enum):                           # memory.py:24-26
    Transitions = 0
    Episodes = 1


class MemoryParameters(object):                          # memory.py:29-38
    def __init__(self):
        self.max_size = None
        self.shared_memory = False
        self.load_memory_from_file_path = None

    @property
    def path(self):
        return 'coach_amd.memories.memory:Memory'


class Memory(object):                                    # memory.py:41-77
    def __init__(self, max_size):
        self.max_size = max_size
        self._length = 0
        self.memory_backend = None

    def store(self, obj):
        raise NotImplementedError("")

    def get(self, index):
        raise NotImplementedError("")

    def length(self):
        raise NotImplementedError("")

    def sample(self, size):
        raise NotImplementedError("")

    def clean(self):
        raise NotImplementedError("")

    def num_transitions(self):
        raise NotImplementedError("")
