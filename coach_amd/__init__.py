"""coach_amd — MI355X-native hot path behind the rl_coach Agent / Memory / Architecture API.

Only what the hot path of BASELINE.json's north_star needs lives here:
``csrc/`` holds the hand-written gfx950 kernels and the C ABI (``include/rlx.h``), the Python
modules mirror the reference's plug-in interfaces (memories, filters, agents, architecture).
"""
__version__ = "0.1.0"
