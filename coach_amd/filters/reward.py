"""Reward filters (rl_coach/filters/reward/*) over the n_env rewards of a vector step."""
import torch

from .. import _rlx


class RewardRescaleFilter(object):
    """reward_rescale_filter.py:37-39: reward * rescale_factor."""

    def __init__(self, rescale_factor):
        if rescale_factor == 0:
            raise ValueError("The reward rescale value can not be set to 0")          # :32-33
        self.rescale_factor = rescale_factor

    def filter(self, reward, out=None):
        out = out if out is not None else torch.empty_like(reward)
        _rlx.lib().reward_filter(reward, out, reward.numel(), float(self.rescale_factor), 0, 0.0, 0.0,
                                 _rlx.current_stream())
        return out


class RewardClippingFilter(object):
    """reward_clipping_filter.py:41-49, including the truthiness quirk (a bound of 0 is ignored)."""

    def __init__(self, clipping_low=float('-inf'), clipping_high=float('inf')):
        self.clipping_low, self.clipping_high = clipping_low, clipping_high

    def filter(self, reward, out=None):
        out = out if out is not None else torch.empty_like(reward)
        _rlx.lib().reward_filter(reward, out, reward.numel(), 1.0, 1, float(self.clipping_low),
                                 float(self.clipping_high), _rlx.current_stream())
        return out
