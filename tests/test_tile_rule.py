"""Host logic of the fused-convolution path (no GPU): nn.graph._kw2_tiling mirrors rlx_gemm's tile rule (csrc/gemm.hip
gemm_impl) with the library's LIVE thresholds — rlx_conv23_forward is taken exactly where the two tiled launches would run
on 32 x 64 tiles with two wave groups per K slab (there its sums are bit-identical to theirs)."""
import ctypes


def test_kw2_tiling_matches_the_documented_batch_window():
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    lib = _rlx.lib()
    below, least, xcd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.gemm_tuning_get(ctypes.byref(below), ctypes.byref(least), ctypes.byref(xcd))
    assert (below.value, least.value, xcd.value) == (192, 192, -1)          # rlx_gemm_tuning's defaults
    ok = [B for B in range(1, 200) if G._kw2_tiling(B * 81, 64, 2) and G._kw2_tiling(B * 49, 64, 2)]
    assert ok == list(range(63, 76))            # two towers: 63 .. 75 images (the C2 minibatch of 64 is inside)
    assert not G._kw2_tiling(64 * 81, 64, 1)    # one tower (acting): rlx_gemm takes 32 x 32 tiles there ...
    assert G._tiled_wave_groups(64 * 81, 64, 1) == 4 and G._tiled_wave_groups(64 * 49, 64, 1) == 4    # ... with four wave groups
    assert G._tiled_wave_groups(32 * 81, 64, 2) == 4 and G._tiled_wave_groups(32 * 49, 64, 2) == 4    # (the DQN update: 32 x 2)
    assert G._tiled_wave_groups(16 * 49, 64, 2) == 1 and G._tiled_wave_groups(256 * 81, 64, 2) == 1
    assert not G._kw2_tiling(64 * 400, 32, 2)   # N <= 32: the narrow 128 x 32 tiling
    assert lib.conv23_forward_supported(20, 20, 32, 4, 2, 64, 3, 1, 64) == 1
    assert lib.conv23_forward_supported(20, 20, 32, 4, 2, 64, 3, 1, 32) == 0
    assert lib.conv23_forward_supported(9, 9, 32, 4, 2, 64, 3, 1, 64) == 0
