"""Import harness for running the REFERENCE's own numpy code (from /root/reference) in the build
container, where tensorflow / gym / redis / ... are not installed.

Used only by tests/golden/make_golden.py (fixture generation) and by the optional
`reference`-marked tests that run when /root/reference is present.  It never travels to the GPU
box: nothing in the `-m gpu` tests, smoke() or bench.py imports this module.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("COACH_REFERENCE_ROOT", "/root/reference")
_STUBBED = {"tensorflow", "redis", "pygame", "skimage", "minio", "kubernetes", "annoy", "bokeh",
            "OpenGL", "mxnet", "gym", "horovod", "mujoco_py", "roboschool", "pybullet_envs",
            "vizdoom", "carla", "pysc2", "dm_control", "robosuite"}


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        mod = sys.modules.get(full)
        if mod is None:
            mod = _Stub(full)
            sys.modules[full] = mod
        return mod

    def __call__(self, *a, **k):
        return None

    def __mro_entries__(self, bases):   # `class X(gym.Wrapper)` -> plain object subclass
        return (object,)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "rl_coach"))


def install():
    """Make `import rl_coach...` resolve to the reference tree with absent packages stubbed."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True      # /root/reference is read-only
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
