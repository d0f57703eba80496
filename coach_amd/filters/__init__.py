"""Input filters on the device — the ``InputFilter`` plug point
(rl_coach/filters/filter.py:295-350: ordered observation filters per key, then reward filters) with
the reference's filter class names, batched over the n_env envs of a vector step (the reference calls
non-batching filters once per data point, filter.py:331-335, after a copy.deepcopy of the response,
:308)."""
from .observation import (ObservationNormalizationFilter, ObservationRescaleToSizeFilter,  # noqa: F401
                          ObservationRGBToYFilter, ObservationStackingFilter, ObservationToUInt8Filter)
from .reward import RewardClippingFilter, RewardRescaleFilter  # noqa: F401
from .filter import InputFilter, NoInputFilter  # noqa: F401
