"""ctypes binding of librlx.so — the only way the Python adapters reach the HIP kernels.

The prototypes are read from ``include/rlx.h`` (the single source of truth for the C ABI),
so a symbol that is declared but not exported — or the other way round — fails at load time.
There is deliberately NO fallback: if the library cannot be loaded the product path raises
(``RlxUnavailable``); nothing under ``oracle/`` is ever imported from here.
"""
import ctypes
import os
import re
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "rlx.h")
LIB_PATH = os.path.join(_HERE, "librlx.so")
CSRC = os.path.join(_HERE, "csrc")


class RlxUnavailable(RuntimeError):
    """librlx.so is missing or cannot be loaded; the HIP hot path cannot run."""


class RlxError(RuntimeError):
    """A librlx entry point returned a non-zero rlx_status."""


_CTYPE = {
    "int": ctypes.c_int,
    "unsigned": ctypes.c_uint,
    "unsigned int": ctypes.c_uint,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "long long": ctypes.c_longlong,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "size_t": ctypes.c_size_t,
}

_VALUE_RETURNING = {"rlx_abi_version"}  # return a value, not an rlx_status

_DECL = re.compile(r"^\s*(int|const char \*)\s*(rlx_\w+)\s*\(([^;{]*?)\)\s*;", re.M | re.S)


def parse_header(path=HEADER):
    """Return {name: (restype, [(ctype, param_name), ...])} for every declaration in rlx.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for ret, name, args in _DECL.findall(text):
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)(\w+)$", a)
                ctype_s, pname = m.group(1).strip(), m.group(2)
                if "*" in ctype_s:
                    params.append((ctypes.c_void_p, pname))
                else:
                    key = ctype_s.replace("const ", "").strip()
                    if key not in _CTYPE:
                        raise ValueError("rlx.h: unknown parameter type %r in %s" % (ctype_s, name))
                    params.append((_CTYPE[key], pname))
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, params)
    return protos


def build_library(verbose=False):
    """Compile every csrc/*.hip for gfx950 into coach_amd/librlx.so (hipcc cross-compiles on CPU)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise RlxUnavailable("hipcc not found at %s; cannot build librlx.so" % hipcc)
    jobs = str(max(1, min(16, os.cpu_count() or 1)))
    res = subprocess.run(["make", "-C", CSRC, "-j", jobs, "HIPCC=" + hipcc],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RlxUnavailable("building librlx.so failed (see output above)")
    return LIB_PATH


def _as_arg(a):
    # torch tensors (and anything with data_ptr) are passed as raw device pointers
    if hasattr(a, "data_ptr"):
        if hasattr(a, "is_contiguous") and not a.is_contiguous():
            raise ValueError("librlx expects contiguous tensors")
        return a.data_ptr()
    return a


class _Lib:
    def __init__(self):
        # torch must load ITS libamdhip64.so.7 first so that device pointers, streams and events
        # are shared between torch and librlx (same soname -> the dynamic linker reuses it).
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            if os.path.exists("/opt/rocm/bin/hipcc"):
                build_library()
            else:
                raise RlxUnavailable(
                    "%s is missing and hipcc is not available to build it. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        try:
            self._dll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover - depends on the box
            raise RlxUnavailable("cannot load %s: %s" % (LIB_PATH, e))
        self.protos = parse_header()
        self._fns = {}
        missing = []
        for name, (restype, params) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = restype
            fn.argtypes = [t for t, _ in params]
            self._fns[name] = fn
        if missing:
            raise RlxUnavailable("librlx.so does not export: " + ", ".join(missing))

    def raw(self, name):
        return self._fns[name]

    def last_error(self):
        return self._fns["rlx_last_error"]().decode()

    def __getattr__(self, name):
        fns = self.__dict__.get("_fns", {})
        full = name if name.startswith("rlx_") else "rlx_" + name
        if full not in fns:
            raise AttributeError(name)
        fn = fns[full]
        nparams = len(self.protos[full][1])
        if fn.restype is ctypes.c_char_p:
            def call_str(*args):
                return fn(*args).decode()
            return call_str
        if full in _VALUE_RETURNING:
            return fn

        def call(*args):
            if len(args) != nparams:
                raise TypeError("%s expects %d arguments, got %d" % (full, nparams, len(args)))
            rc = fn(*[_as_arg(a) for a in args])
            if rc != 0:
                raise RlxError("%s failed (%d): %s" % (full, rc, self.last_error()))
            return rc
        call.__name__ = full
        setattr(self, name, call)
        return call


_lock = threading.Lock()
_lib = None


def lib():
    """The loaded library (singleton). Raises RlxUnavailable — never falls back."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = _Lib()
    return _lib


def current_stream():
    """hipStream_t of torch's current stream, as an int usable for the `stream` parameters."""
    import torch
    return torch.cuda.current_stream().cuda_stream
