"""The C-ABI boundary: librlx.so loads on a GPU-less box and exports exactly what include/rlx.h
declares (no compute calls here)."""
import ctypes
import subprocess

import pytest

from coach_amd import _rlx


def test_library_loads_and_exports_every_declared_symbol():
    lib = _rlx.lib()                      # raises RlxUnavailable if a declared symbol is missing
    assert len(lib.protos) >= 10
    assert lib.raw("rlx_abi_version")() == _rlx.ABI_VERSION      # (the loader refuses a library of another version)
    assert lib.build_arch() == "gfx950"


def test_no_undeclared_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _rlx.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "rlx_" in ln}
    exported = {s for s in exported if s.startswith("rlx_")}
    declared = set(_rlx.parse_header())
    assert exported == declared, (exported ^ declared)


def test_signatures_have_no_torch_types():
    text = open(_rlx.HEADER).read()
    assert "torch" not in text and "at::" not in text and "Tensor" not in text


def test_device_count_without_gpu_is_not_an_error():
    n = ctypes.c_int(-1)
    _rlx.lib().device_count(ctypes.byref(n))
    assert n.value >= 0


def test_status_and_message_on_invalid_argument():
    lib = _rlx.lib()
    with pytest.raises(_rlx.RlxError, match="null"):
        lib.device_count(None)
    assert "null" in lib.last_error()


def test_argument_validation_matches_reference_error_cases():
    """Host-side validation runs before any launch, so these hold on a GPU-less box.  The messages
    mirror the reference's ValueErrors where it has them."""
    lib = _rlx.lib()
    P = ctypes.c_void_p
    fake = P(0x1000)                      # never dereferenced: validation fails first
    # SegmentTree size must be a power of two (prioritized_experience_replay.py:62-63)
    with pytest.raises(_rlx.RlxError, match="power of 2"):
        lib.per_init(fake, fake, fake, 5, fake, None)
    with pytest.raises(_rlx.RlxError, match="power of two"):
        lib.per_sample(fake, fake, 6, fake, 4, 8.0, 0.4, fake, fake, None, 0, 0, None, None)
    with pytest.raises(_rlx.RlxError, match="batch"):
        lib.per_sample(fake, fake, 8, fake, 0, 8.0, 0.4, fake, fake, None, 0, 0, None, None)
    with pytest.raises(_rlx.RlxError, match="out of range"):
        lib.per_store(fake, fake, fake, 8, 9, 1, 0.6, fake, fake, None)
    # GEMM descriptor checks
    d = _rlx.GemmDesc()
    with pytest.raises(_rlx.RlxError, match="bad shape"):
        lib.gemm(ctypes.byref(d), None)
    d.M = d.N = d.K = d.batch = 4
    with pytest.raises(_rlx.RlxError, match="null operand"):
        lib.gemm(ctypes.byref(d), None)
    # narrow dense: N limited to 16 outputs
    with pytest.raises(_rlx.RlxError, match="N <= 16"):
        lib.dense_small_forward(fake, 0, fake, 0, None, 0, fake, 0, 1, 8, 8, 17, 0, None)
    # frame ring / gather geometry
    with pytest.raises(_rlx.RlxError, match="multiple of 4"):
        lib.imgreplay_reset(fake, fake, fake, fake, 2, 8, 7054, None)
    with pytest.raises(_rlx.RlxError, match="too small"):
        lib.imgreplay_append(fake, fake, fake, fake, fake, fake, None, fake, 2, 4, 7056, 4, 0, 64, 1, None)
    with pytest.raises(_rlx.RlxError, match="columns"):
        lib.copy_columns(None, 0, None, None, 0, 0, 8, 8, 4, fake, None)
    with pytest.raises(_rlx.RlxError, match="bad geometry|bad convolution"):
        lib.conv_tables(fake, fake, 1, 4, 4, 1, 8, 8, 1, None)
    with pytest.raises(_rlx.RlxError, match="kind must be"):
        lib.regression_loss(fake, 1, fake, 1, None, 4, 1, 7, 1.0, 1.0, None, 1, None, None)
    with pytest.raises(_rlx.RlxError, match="16-byte aligned"):
        lib.adam_tf1(P(0x1004), fake, fake, fake, 16, 1e-3, 0.9, 0.99, 1e-4, fake, 1.0, None)
    with pytest.raises(_rlx.RlxError, match="minibatch must be"):
        lib.ppo_continuous_loss(fake, 4, fake, fake, fake, fake, fake, 4, 5000, 4, 0.2, 0.0, 1.0, None, 4,
                                None, None, None, None, None, None)


def test_wrong_argument_count_is_a_type_error():
    with pytest.raises(TypeError, match="expects"):
        _rlx.lib().gae(None, None)


def test_product_path_does_not_import_the_oracle():
    """coach_amd/ must never reach into oracle/ (the oracle is the checker, not a fallback)."""
    import os
    import re
    root = os.path.dirname(_rlx.__file__)
    bad = []
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
