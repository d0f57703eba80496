"""Filter adapter classes (coach_amd/filters) and the bilinear resize kernel.

Mirrors rl_coach/tests/filters/test_filters_stacking.py:19-66 (the Atari chain on all-ones frames:
rescale -> RGB-to-Y -> uint8 -> stack; shapes and values) and
tests/filters/observation/test_observation_rescale_to_size_filter.py:16-47 (shape / range of the
rescaled observation), plus the resize kernel vs the oracle restatement on random images, bit for bit (the oracle is
pinned bit for bit to scipy.ndimage.zoom, the routine scikit-image >= 0.19 executes this call with —
tests/test_filters.py; scikit-image itself is absent).
"""
import numpy as np
import pytest

from oracle import filters as OF


def test_oracle_resize_reference_properties():
    # all-ones stays all-ones, output shape follows the request (reference test :16-47)
    obs = np.ones((20, 30, 3), dtype=np.uint8)
    out = OF.resize_bilinear_u8(obs, (10, 10))
    assert out.shape == (10, 10, 3) and out.dtype == np.uint8 and (out == 1).all()
    # identity when the size does not change
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(12, 9, 3)).astype(np.uint8)
    np.testing.assert_array_equal(OF.resize_bilinear_u8(img, (12, 9)), img)
    # exact 2x down-scaling samples the centre of each 2x2 block: the mean of its 4 pixels
    img = rng.randint(0, 256, size=(8, 8)).astype(np.uint8)
    exp = (img.reshape(4, 2, 4, 2).astype(np.float64).transpose(0, 2, 1, 3).reshape(4, 4, 4).mean(-1)).astype(np.uint8)
    np.testing.assert_array_equal(OF.resize_bilinear_u8(img, (4, 4)), exp)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,out", [((210, 160, 3), (84, 84)), ((64, 48, 1), (84, 84)), ((20, 30, 3), (10, 10))])
def test_resize_kernel_matches_oracle(dev, shape, out):
    import torch
    from coach_amd.filters import ObservationRescaleToSizeFilter
    rng = np.random.RandomState(3)
    x = rng.randint(0, 256, size=(5,) + shape).astype(np.uint8)
    f = ObservationRescaleToSizeFilter(out + (shape[2],))
    y = f.filter(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = np.stack([OF.resize_bilinear_u8(x[i], out) for i in range(5)])
    np.testing.assert_array_equal(y, ref)                      # uint8 results: bit-exact


@pytest.mark.gpu
def test_atari_filter_chain_on_ones(dev):
    """tests/filters/test_filters_stacking.py:19-66: rescale(84,84) -> RGBToY -> ToUInt8(0,255) -> stack(4)."""
    import torch
    from coach_amd.filters import (InputFilter, ObservationRescaleToSizeFilter, ObservationRGBToYFilter,
                                   ObservationStackingFilter, ObservationToUInt8Filter, RewardClippingFilter,
                                   RewardRescaleFilter)
    flt = InputFilter()
    flt.add_observation_filter('observation', 'rescaling', ObservationRescaleToSizeFilter((84, 84, 3)))
    flt.add_observation_filter('observation', 'rgb_to_y', ObservationRGBToYFilter())
    flt.add_observation_filter('observation', 'to_uint8', ObservationToUInt8Filter(0, 255))
    flt.add_observation_filter('observation', 'stacking', ObservationStackingFilter(4))
    flt.add_reward_filter('rescale', RewardRescaleFilter(2.0))
    flt.add_reward_filter('clip', RewardClippingFilter(-1.0, 1.0))
    n = 3
    obs = torch.ones((n, 210, 160, 3), dtype=torch.uint8, device=dev)
    out = flt.filter_observation('observation', obs)
    assert tuple(out.shape) == (n, 84, 84) and out.dtype == torch.uint8
    # 0.2989 + 0.5870 + 0.1140 = 0.9999 -> truncates to 0 exactly like the numpy chain
    ref = OF.to_uint8(OF.rgb_to_y(np.ones((84, 84, 3))), 0, 255)
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref)
    r = torch.tensor([-3.0, 0.2, 0.75, 4.0], dtype=torch.float32, device=dev)
    np.testing.assert_allclose(flt.filter_reward(r).cpu().numpy(), [-1.0, 0.4, 1.0, 1.0], rtol=1e-6)


@pytest.mark.gpu
def test_normalization_filter_matches_oracle(dev, golden):
    import torch
    from coach_amd.filters import ObservationNormalizationFilter
    rng = np.random.RandomState(1)
    f = ObservationNormalizationFilter(7, dev)
    o = OF.RunningStatsOracle((7,))
    for _ in range(3):
        x = rng.randn(40, 7) * 3 + 1
        y = f.filter(torch.from_numpy(x).to(dev)).cpu().numpy()
        o.push(x)
        np.testing.assert_allclose(y, o.normalize(x).astype(np.float32), rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(f.mean.cpu().numpy(), o._mean)
    np.testing.assert_array_equal(f.std.cpu().numpy(), o._std)
