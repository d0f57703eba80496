"""The C-ABI boundary: librlx.so loads on a GPU-less box and exports exactly what include/rlx.h
declares (no compute calls here)."""
import ctypes
import subprocess

import pytest

from coach_amd import _rlx


def test_library_loads_and_exports_every_declared_symbol():
    lib = _rlx.lib()                      # raises RlxUnavailable if a declared symbol is missing
    assert len(lib.protos) >= 10
    assert lib.raw("rlx_abi_version")() >= 1
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
