"""`import rl_coach...` for preset files written against IntelLabs/coach.

BASELINE.json's north_star keeps "the rl_coach Agent/GraphManager/preset API surface so existing presets drop
in".  The package mirrors the reference's module layout (agents/, memories/, filters/, exploration_policies/,
graph_managers/, environments/, base_parameters, core_types, schedules, architectures/), so dropping a preset in
is an import-name question: `install()` registers a meta-path finder that serves `rl_coach.<module>` from
`coach_amd.<module>` — the same module objects, no copies.  The reference splits its filters into one module per
class (`filters/reward/reward_rescale_filter.py` ...); those paths resolve to the module that holds the class here.

    import coach_amd.compat; coach_amd.compat.install()
    exec(open("rl_coach/presets/CartPole_DQN.py").read())        # unchanged preset text
    graph_manager.improve()                                       # runs on the MI355X engine

Nothing of the reference is imported: if a real `rl_coach` is already importable, install() refuses.
tests/test_preset_dropin.py executes the text of the reference's hot-path presets against this layer."""
import importlib
import importlib.abc
import importlib.machinery
import sys

# reference module path (below rl_coach.) -> module below coach_amd. holding the same names
_FOLDED = {
    "filters.reward": "filters.reward", "filters.observation": "filters.observation",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        mod = importlib.import_module(self.target)
        return mod

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "rl_coach" and not fullname.startswith("rl_coach."):
            return None
        rest = fullname[len("rl_coach"):].lstrip(".")
        for ref_prefix, mine in _FOLDED.items():          # one-module-per-class packages of the reference
            if rest == ref_prefix or rest.startswith(ref_prefix + "."):
                rest = mine
                break
        name = "coach_amd" + ("." + rest if rest else "")
        try:
            mod = importlib.import_module(name)
        except ImportError:
            return None
        spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(name), is_package=True)
        spec.submodule_search_locations = list(getattr(mod, "__path__", []))
        return spec


_installed = None


def install():
    """Make `rl_coach.*` importable as an alias of `coach_amd.*` (idempotent)."""
    global _installed
    if _installed is not None:
        return _installed
    existing = sys.modules.get("rl_coach")
    if existing is not None and not getattr(existing, "__name__", "").startswith("coach_amd"):
        raise ImportError("a different rl_coach package is already imported: {}".format(existing))
    _installed = _Finder()
    sys.meta_path.insert(0, _installed)
    return _installed


def uninstall():
    global _installed
    if _installed is not None:
        sys.meta_path.remove(_installed)
        _installed = None
    for k in [k for k in sys.modules if k == "rl_coach" or k.startswith("rl_coach.")]:
        del sys.modules[k]


def resolve_reference_style(agent_params, env_params):
    """Fold what a reference preset expresses through filter objects / environment classes into the flat fields
    the device agents read: reward filters of `agent_params.input_filter` (or the Atari default,
    gym_environment.py:106-113) -> algorithm.reward_rescale / reward_clipping; an
    ObservationNormalizationFilter in `pre_network_filter` -> algorithm.normalize_observations."""
    from .filters.observation import ObservationNormalizationFilter
    from .filters.reward import RewardClippingFilter, RewardRescaleFilter
    alg = agent_params.algorithm
    flt = getattr(agent_params, "input_filter", None)
    if flt is not None and hasattr(flt, "_reward_filters"):
        alg.reward_rescale, alg.reward_clipping = 1.0, None
        for f in flt._reward_filters.values():
            if isinstance(f, RewardRescaleFilter):
                alg.reward_rescale = float(f.rescale_factor)
            elif isinstance(f, RewardClippingFilter):
                alg.reward_clipping = (float(f.clipping_low), float(f.clipping_high))
            else:
                raise ValueError("reward filter {!r} has no device implementation".format(f))
    elif hasattr(env_params, "is_atari") and hasattr(alg, "reward_clipping"):
        # the level's default input filter: the Atari chain clips, GymVectorEnvironment has none
        # (gym_environment.py:76-81,106-113)
        alg.reward_clipping = (-1.0, 1.0) if env_params.is_atari else None
    pre = getattr(agent_params, "pre_network_filter", None)
    if pre is not None and hasattr(pre, "_observation_filters"):
        for d in pre._observation_filters.values():
            for f in d.values():
                if isinstance(f, ObservationNormalizationFilter):
                    alg.normalize_observations = True
                else:
                    raise ValueError("pre-network filter {!r} has no device implementation".format(f))
