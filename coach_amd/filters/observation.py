"""Observation filters (rl_coach/filters/observation/*) driving librlx kernels; every filter takes
and returns device tensors with a leading env/batch dimension."""
import numpy as np
import torch

from .. import _rlx


class ObservationFilter(object):
    def reset(self):
        pass


class ObservationRescaleToSizeFilter(ObservationFilter):
    """observation_rescale_to_size_filter.py:62-79 — bilinear resize of uint8 images [n,H,W,C]."""

    def __init__(self, output_shape):
        self.out_hw = tuple(output_shape[:2])

    def filter(self, observation, update_internal_state=True):
        n, H, W, C = observation.shape
        out = torch.empty((n,) + self.out_hw + (C,), dtype=torch.uint8, device=observation.device)
        _rlx.lib().resize_bilinear_u8(observation, out, n, H, W, C, self.out_hw[0], self.out_hw[1],
                                      _rlx.current_stream())
        return out


class ObservationRGBToYFilter(ObservationFilter):
    """observation_rgb_to_y_filter.py:41-47 fused with the ObservationToUInt8Filter(0, 255) that
    follows it in the Atari chain (observation_to_uint8_filter.py:51-60): [n,H,W,3] u8 -> [n,H,W] u8."""

    def __init__(self, input_low=0.0, input_high=255.0):
        self.low, self.high = float(input_low), float(input_high)

    def filter(self, observation, update_internal_state=True):
        n = observation.numel() // 3
        out = torch.empty(observation.shape[:-1], dtype=torch.uint8, device=observation.device)
        _rlx.lib().rgb_to_y_u8(observation, out, n, self.low, self.high, _rlx.current_stream())
        return out


class ObservationToUInt8Filter(ObservationFilter):
    """Stand-alone ObservationToUInt8Filter is the identity after the fused RGB->Y->uint8 kernel;
    kept so that the reference's filter chains can be written down one to one."""

    def __init__(self, input_low, input_high):
        self.input_low, self.input_high = input_low, input_high

    def filter(self, observation, update_internal_state=True):
        if observation.dtype != torch.uint8:
            raise ValueError("place ObservationToUInt8Filter after ObservationRGBToYFilter (fused kernel)")
        return observation


class ObservationStackingFilter(ObservationFilter):
    """observation_stacking_filter.py:27-101.  On the device the stack is never materialised per
    step: frames go into the replay's frame ring (rlx_imgreplay_append) and a stacked state is
    gathered on demand (LazyStack semantics); this class only carries the parameters."""

    def __init__(self, stack_size, stacking_axis=-1):
        if stacking_axis != -1:
            raise ValueError("the device ring stacks along the last axis (the reference default)")
        self.stack_size = stack_size

    def filter(self, observation, update_internal_state=True):
        return observation


class ObservationNormalizationFilter(ObservationFilter):
    """observation_normalization_filter.py:71-78 + NumpySharedRunningStats
    (utilities/shared_running_stats.py:115-164): fp64 running sum / sum of squares on the device."""

    def __init__(self, dim=None, device=None, clip_min=-5.0, clip_max=5.0, epsilon=1e-2, name='observation_stats'):
        """dim / device None: a preset-level declaration (`ObservationNormalizationFilter(name=...)` in
        Mujoco_ClippedPPO.py); the agent builds the bound instance once the observation space is known."""
        self.dim, self.device, self.eps, self.name = dim, device, epsilon, name
        self.clip = (clip_min, clip_max)
        if dim is None:
            return
        f64 = torch.float64
        self.sum = torch.zeros(dim, dtype=f64, device=device)
        self.sum_squares = torch.full((dim,), epsilon, dtype=f64, device=device)
        self.count = torch.full((1,), epsilon, dtype=f64, device=device)
        self.mean = torch.zeros(dim, dtype=f64, device=device)
        self.std = torch.full((dim,), float(np.sqrt(epsilon)), dtype=f64, device=device)

    def push_shared(self, observation, dist):
        """Data-parallel state update: every rank's batch enters every rank's statistics (the reference
        publishes each push to the other workers, shared_running_stats.py:46-67).  One all-reduce of
        [sum | sum_squares | count] = 2*dim+1 doubles; not capturable in a hipGraph."""
        n, D = observation.shape[0], self.dim
        s = _rlx.current_stream()
        delta = torch.zeros(2 * D + 1, dtype=torch.float64, device=self.device)
        scratch = torch.empty(2 * D, dtype=torch.float64, device=self.device)
        _rlx.lib().running_stats_push(observation, int(observation.dtype == torch.float64), n, D, delta[:D],
                                      delta[D:2 * D], delta[2 * D:], scratch[:D], scratch[D:], self.eps, s)
        dist.all_reduce_sum(delta)
        _rlx.lib().running_stats_merge(delta, D, self.sum, self.sum_squares, self.count, self.mean, self.std,
                                       self.eps, s)

    def push(self, observation):
        """running_observation_stats.push alone: the batch enters the statistics, nothing is normalised."""
        _rlx.lib().running_stats_push(observation, int(observation.dtype == torch.float64), observation.shape[0],
                                      self.dim, self.sum, self.sum_squares, self.count, self.mean, self.std, self.eps,
                                      _rlx.current_stream())

    def filter(self, observation, update_internal_state=True, out=None):
        n = observation.shape[0]
        is64 = observation.dtype == torch.float64
        s = _rlx.current_stream()
        if update_internal_state:
            _rlx.lib().running_stats_push(observation, int(is64), n, self.dim, self.sum, self.sum_squares,
                                          self.count, self.mean, self.std, self.eps, s)
        if out is None:
            out = torch.empty(n, self.dim, dtype=torch.float32, device=observation.device)
        _rlx.lib().running_stats_normalize(observation, int(is64), n, self.dim, self.mean, self.std,
                                           self.clip[0], self.clip[1], out, None, s)
        return out
