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

_VALUE_RETURNING = {"rlx_abi_version", "rlx_ppo_fc_heads_supported", "rlx_ppo_fc_rows_supported", "rlx_td3_fused_supported", "rlx_sac_fused_supported", "rlx_mlp_dqn_supported", "rlx_mlp_q_act_supported", "rlx_conv23_forward_supported",
                    "rlx_conv123_forward_supported", "rlx_conv32_input_grad_supported", "rlx_conv_dw_u8_supported", "rlx_conv_dw_f32_supported"}  # return a value, not an rlx_status

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
        built = self._fns["rlx_abi_version"]()
        if built != ABI_VERSION:             # struct layouts / buffer contracts of this package and of the binary differ
            raise RlxUnavailable("%s is ABI version %d, this package drives version %d: rebuild it "
                                 "(python -c 'import __graft_entry__ as g; g.build()')" % (LIB_PATH, built, ABI_VERSION))

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
            global CALL_COUNT
            CALL_COUNT += 1
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


class KernelTimer(object):
    """`with KernelTimer(max_records) as t: <eager librlx calls>` -> t.records = [(kernel name, microseconds), ...] in
    launch order: every kernel the library launched inside the block, timed by its own dispatch timestamps
    (rlx_profile_*, include/rlx.h).  Not usable inside a stream capture."""

    def __init__(self, max_records=4096):
        self.max_records, self.records = int(max_records), []

    def __enter__(self):
        lib().profile_begin(self.max_records)
        return self

    def __exit__(self, *exc):
        n = ctypes.c_int()
        try:
            lib().profile_end(ctypes.byref(n))       # raises when launches overflowed max_records (an undercounting trace)
        except Exception:
            if exc[0] is None:
                raise
        if exc[0] is None:
            name, ms = ctypes.c_char_p(), ctypes.c_float()
            for i in range(n.value):
                lib().profile_read(i, ctypes.byref(name), ctypes.byref(ms))
                self.records.append((name.value.decode().strip("()"), 1e3 * ms.value))
        return False


# --------------------------------------------------------------------------- ABI structures
class Column(ctypes.Structure):
    """rlx_column (include/rlx.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("row_bytes", ctypes.c_longlong)]


def make_columns(pairs):
    """[(src_tensor, dst_tensor), ...] -> rlx_column[]; rows run along dim 0 of every tensor."""
    arr = (Column * len(pairs))()
    for i, (s, d) in enumerate(pairs):
        rb = (s[0].numel() if s.dim() > 1 else 1) * s.element_size()
        arr[i] = Column(s.data_ptr(), d.data_ptr(), rb)
    return arr


class GemmDesc(ctypes.Structure):
    """rlx_gemm_desc (include/rlx.h) — field order must match the header."""
    _fields_ = [
        ("A", ctypes.c_void_p), ("a_row_tab", ctypes.c_void_p), ("a_k_tab", ctypes.c_void_p),
        ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("deriv_aux", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("colsum_out", ctypes.c_void_p),
        ("a_row_stride", ctypes.c_longlong), ("a_k_stride", ctypes.c_longlong),
        ("a_batch_stride", ctypes.c_longlong),
        ("b_k_stride", ctypes.c_longlong), ("b_n_stride", ctypes.c_longlong),
        ("b_batch_stride", ctypes.c_longlong),
        ("ldc", ctypes.c_longlong), ("c_batch_stride", ctypes.c_longlong),
        ("bias_batch_stride", ctypes.c_longlong),
        ("aux_ld", ctypes.c_longlong), ("aux_batch_stride", ctypes.c_longlong),
        ("workspace_floats", ctypes.c_longlong), ("colsum_batch_stride", ctypes.c_longlong),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("batch", ctypes.c_int),
        ("a_is_u8", ctypes.c_int), ("a_vec_along_k", ctypes.c_int), ("a_tab_vec_ok", ctypes.c_int),
        ("activation", ctypes.c_int), ("deriv_kind", ctypes.c_int), ("accumulate", ctypes.c_int),
        ("a_div", ctypes.c_float),
        ("batch_inner", ctypes.c_int),
        ("a_batch_stride2", ctypes.c_longlong), ("b_batch_stride2", ctypes.c_longlong),
        ("bias_batch_stride2", ctypes.c_longlong),
        ("n_fold", ctypes.c_int),
        ("row_heads", ctypes.c_void_p), ("n_row_heads", ctypes.c_int),
        ("kw_min_tiles", ctypes.c_int),
    ]


class SplitkJob(ctypes.Structure):
    """rlx_splitk_job (include/rlx.h) — field order must match the header."""
    _fields_ = [("partials", ctypes.c_void_p), ("colsum_partials", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("colsum_out", ctypes.c_void_p), ("ldc", ctypes.c_longlong), ("c_batch_stride", ctypes.c_longlong),
                ("colsum_batch_stride", ctypes.c_longlong), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("batch", ctypes.c_int), ("splits", ctypes.c_int), ("n_fold", ctypes.c_int)]

    def workspace_floats(self):
        """floats of the workspace the outstanding reduction still reads (0 when nothing is outstanding)."""
        return (self.M * self.N + self.N) * self.batch * self.splits if self.splits > 1 else 0


ADAM_TICKET_WORDS = 1056        # rlx.h RLX_ADAM_TICKET_WORDS
ABI_VERSION = 10                # rlx_abi_version() of the library this module's structures and buffer sizes match
MAX_SPLITK_JOBS = 8


def splitk_reduce_jobs(jobs, stream=None, ppo_tail=None, per_tail=None):
    """sum the partials of the deferred products `jobs` (SplitkJob list) in one launch.
    ppo_tail: a PpoRowsDesc whose all-rows part (rlx_ppo_heads_tail) rides on the (last) launch as extra workgroups.
    per_tail: the argument tuple of rlx_per_update (without the stream) — the prioritized replay's priority update rides on
    the (last) launch as one workgroup (rlx_splitk_reduce_jobs_per_update)."""
    live = [j for j in jobs if j.splits > 1]
    s = current_stream() if stream is None else stream
    for i in range(0, len(live), MAX_SPLITK_JOBS):
        chunk = live[i:i + MAX_SPLITK_JOBS]
        arr = (SplitkJob * len(chunk))(*chunk)
        if per_tail is not None and ppo_tail is None and i + MAX_SPLITK_JOBS >= len(live):
            run = lambda a=arr, n=len(chunk), t=per_tail: lib().splitk_reduce_jobs_per_update(ctypes.byref(a), n, *t, s)
            per_tail = None
        elif ppo_tail is not None and i + MAX_SPLITK_JOBS >= len(live):
            run = lambda a=arr, n=len(chunk), t=ppo_tail: lib().splitk_reduce_jobs_ppo_tail(ctypes.byref(a), n, ctypes.byref(t), s)
            ppo_tail = None
        else:
            run = lambda a=arr, n=len(chunk): lib().splitk_reduce_jobs(ctypes.byref(a), n, s)
        _record((), run, flops=0.0)                # part of the GEMM family's time, no products of its own
        run()
    if ppo_tail is not None:                       # nothing to reduce: the tail as a launch of its own
        lib().ppo_heads_tail(ctypes.byref(ppo_tail), s)
    if per_tail is not None:
        lib().per_update(*per_tail, s)


def conv_input_grad(dz, weights, dx, x_out, deriv, tables, B, H, W, C, KH, KW, S, Co, towers, dy_stride, w_stride,
                    dx_stride, stream=None):
    """rlx_conv_input_grad; recorded with the flops of the product it replaces (dcol = dz W^T: 2*M*K*Co per tower)."""
    s = current_stream() if stream is None else stream
    OH, OW = (H - KH) // S + 1, (W - KW) // S + 1
    run = lambda: lib().conv_input_grad(dz, weights, dx, x_out, ACT[deriv], tables, B, H, W, C, KH, KW, S, Co, towers,
                                        dy_stride, w_stride, dx_stride, s)
    if GEMM_HOOK is not None:
        d = GemmDesc()
        d.M, d.N, d.K, d.batch = B * OH * OW, KH * KW * C, Co, towers
        _record((d,), run)
    run()


class PerUpdateDesc(ctypes.Structure):
    """rlx_per_update_desc (include/rlx.h): the arguments of rlx_per_update for a rider launch."""
    _fields_ = [("sum_tree", ctypes.c_void_p), ("min_tree", ctypes.c_void_p), ("max_tree", ctypes.c_void_p),
                ("capacity", ctypes.c_int), ("idx", ctypes.c_void_p), ("td_errors", ctypes.c_void_p), ("n", ctypes.c_int),
                ("alpha", ctypes.c_double), ("epsilon", ctypes.c_double), ("max_priority", ctypes.c_void_p),
                ("status", ctypes.c_void_p)]


def per_update_desc(args):
    """PrioritizedExperienceReplay.priority_update_args() -> PerUpdateDesc"""
    sum_t, min_t, max_t, cap, idx, err, n, alpha, eps, maxp, status = args
    d = PerUpdateDesc()
    d.sum_tree, d.min_tree, d.max_tree, d.capacity = sum_t.data_ptr(), min_t.data_ptr(), max_t.data_ptr(), int(cap)
    d.idx, d.td_errors, d.n, d.alpha, d.epsilon = idx.data_ptr(), err.data_ptr(), int(n), float(alpha), float(eps)
    d.max_priority, d.status = maxp.data_ptr(), status.data_ptr()
    return d


class ConvDwItem(ctypes.Structure):
    """rlx_conv_dw_item (include/rlx.h) — field order must match the header."""
    _fields_ = [("x", ctypes.c_void_p), ("x_tower_stride", ctypes.c_longlong), ("x_is_u8", ctypes.c_int), ("a_div", ctypes.c_float),
                ("dz", ctypes.c_void_p), ("dz_tower_stride", ctypes.c_longlong)] + \
               [(n, ctypes.c_int) for n in ("B", "H", "W", "C", "KH", "KW", "S", "filters", "towers")] + \
               [("dw", ctypes.c_void_p), ("dw_tower_stride", ctypes.c_longlong), ("db", ctypes.c_void_p),
                ("db_tower_stride", ctypes.c_longlong), ("workspace", ctypes.c_void_p), ("workspace_floats", ctypes.c_longlong)]


def conv_dw_item(x, x_tower_stride, x_is_u8, a_div, dz, dz_tower_stride, B, H, W, C, KH, KW, S, Co, towers, dw, dw_tower_stride, db,
                 db_tower_stride, workspace):
    it = ConvDwItem()
    it.x, it.x_tower_stride, it.x_is_u8, it.a_div = x.data_ptr(), x_tower_stride, int(bool(x_is_u8)), float(a_div)
    it.dz, it.dz_tower_stride = dz.data_ptr(), dz_tower_stride
    it.B, it.H, it.W, it.C, it.KH, it.KW, it.S, it.filters, it.towers = B, H, W, C, KH, KW, S, Co, towers
    it.dw, it.dw_tower_stride, it.db, it.db_tower_stride = dw.data_ptr(), dw_tower_stride, _ptr(db), db_tower_stride
    it.workspace, it.workspace_floats = workspace.data_ptr(), workspace.numel()
    return it


def conv_dw_multi(items, jobs, stream=None):
    """rlx_conv_dw_multi: the convolution layers' weight gradients of one backward pass as ONE launch; jobs[i] receives
    item i's outstanding reduction (the SAME SplitkJob objects the caller commits)."""
    s = current_stream() if stream is None else stream
    n = len(items)
    arr = (ConvDwItem * n)(*items)
    jarr = (SplitkJob * n)()

    def run():
        lib().conv_dw_multi(arr, jarr, n, s)
    if GEMM_HOOK is not None:
        descs = []
        for it in items:
            OH, OW = (it.H - it.KH) // it.S + 1, (it.W - it.KW) // it.S + 1
            d = GemmDesc()
            d.M, d.K = it.KH * it.KW * it.C, it.B * OH * OW
            d.N, d.batch, d.a_is_u8 = (it.towers * it.filters, 1, 1) if it.x_is_u8 else (it.filters, it.towers, 0)
            descs.append(d)
        _record(tuple(descs), run)
    run()
    for j, out in zip(jarr, jobs):
        ctypes.memmove(ctypes.byref(out), ctypes.byref(j), ctypes.sizeof(SplitkJob))


class SmallDenseProblem(ctypes.Structure):
    """rlx_small_dense_problem (include/rlx.h) — field order must match the header."""
    _fields_ = [
        ("x", ctypes.c_void_p), ("x_tower_stride", ctypes.c_longlong),
        ("w", ctypes.c_void_p), ("w_tower_stride", ctypes.c_longlong),
        ("bias", ctypes.c_void_p), ("bias_tower_stride", ctypes.c_longlong),
        ("y", ctypes.c_void_p), ("y_tower_stride", ctypes.c_longlong),
        ("dy", ctypes.c_void_p), ("dy_tower_stride", ctypes.c_longlong),
        ("dw", ctypes.c_void_p), ("dw_tower_stride", ctypes.c_longlong),
        ("db", ctypes.c_void_p), ("db_tower_stride", ctypes.c_longlong),
        ("dx", ctypes.c_void_p), ("dx_tower_stride", ctypes.c_longlong),
        ("towers", ctypes.c_int), ("M", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int),
        ("activation", ctypes.c_int), ("lower_activation", ctypes.c_int),
    ]


class MlpDqnDesc(ctypes.Structure):
    """rlx_mlp_dqn_desc (include/rlx.h) — field order must match the header."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "weights", "target_weights", "adam_m", "adam_v", "adam_state", "states", "next_states", "actions",
        "rewards", "game_overs", "importance_weights", "workspace", "sync_words", "loss_out", "norm_out",
        "td_errors", "status")] + [(n, ctypes.c_longlong) for n in (
            "workspace_floats", "off_w1", "off_b1", "off_w2", "off_b2", "off_w3", "off_b3")] + [
        ("discount", ctypes.c_double)] + [(n, ctypes.c_int) for n in (
            "batch", "obs_dim", "h1", "h2", "n_actions", "huber", "double_dqn")] + [(n, ctypes.c_float) for n in (
                "learning_rate", "beta1", "beta2", "epsilon", "grad_scale")]


class Mlp3(ctypes.Structure):
    """rlx_mlp3 (include/rlx.h) — field order must match the header."""
    _fields_ = [(n, ctypes.c_longlong) for n in ("off_w1", "off_b1", "off_w2", "off_b2", "off_w3", "off_b3",
                                                 "tower_stride1", "tower_stride2", "tower_stride3")] + \
               [(n, ctypes.c_int) for n in ("d_in", "h1", "h2", "d_out")]


class FusedNet(ctypes.Structure):
    """rlx_fused_net (include/rlx.h) — field order must match the header."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("weights", "target_weights", "adam_m", "adam_v", "adam_state", "grads",
                                               "norm_out", "ticket")] + \
               [(n, ctypes.c_float) for n in ("learning_rate", "beta1", "beta2", "epsilon", "grad_scale", "mix_rate")]


class Td3FusedDesc(ctypes.Structure):
    """rlx_td3_fused_desc (include/rlx.h) — field order must match the header."""
    _fields_ = [("actor", FusedNet), ("critic", FusedNet), ("actor_mlp", Mlp3), ("critic_mlp", Mlp3)] + \
               [(n, ctypes.c_void_p) for n in ("obs", "next_obs", "actions", "rewards", "game_overs", "noise",
                                               "action_low", "action_high")] + \
               [(n, ctypes.c_double) for n in ("noise_clip", "discount", "clip_low", "clip_high")] + \
               [("use_non_zero_discount_for_terminal_states", ctypes.c_int), ("has_clip", ctypes.c_int),
                ("actor_scale", ctypes.c_float), ("batch", ctypes.c_int), ("obs_dim", ctypes.c_int),
                ("act_dim", ctypes.c_int), ("workspace", ctypes.c_void_p), ("workspace_floats", ctypes.c_longlong)] + \
               [(n, ctypes.c_void_p) for n in ("td_targets", "q_min", "loss", "neg_action_grad")]


class SacFusedDesc(ctypes.Structure):
    """rlx_sac_fused_desc (include/rlx.h) — field order must match the header."""
    _fields_ = [("policy", FusedNet), ("q", FusedNet), ("v", FusedNet), ("policy_mlp", Mlp3), ("v_mlp", Mlp3)] + \
               [(n, ctypes.c_longlong) for n in ("q_off_obs_w", "q_off_obs_b", "q_off_act_w", "q_off_act_b", "q_off_fc_w",
                                                 "q_off_fc_b", "q_off_out_w", "q_off_out_b", "q_stride_obs", "q_stride_act",
                                                 "q_stride_fc", "q_stride_out")] + \
               [(n, ctypes.c_void_p) for n in ("obs", "next_obs", "actions", "rewards", "game_overs", "normals")] + \
               [("discount", ctypes.c_double)] + \
               [(n, ctypes.c_int) for n in ("resample_noise_per_pass", "batch", "obs_dim", "act_dim", "q_hidden", "reserved")] + \
               [("workspace", ctypes.c_void_p), ("workspace_floats", ctypes.c_longlong)] + \
               [(n, ctypes.c_void_p) for n in ("value_targets", "log_target", "td_targets", "dq_da", "q_loss", "v_loss")]


class PpoFcHeadsDesc(ctypes.Structure):
    """rlx_ppo_fc_heads_desc (include/rlx.h) — field order must match the header."""
    P, LL, F, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_int
    _fields_ = [("x", P), ("x_tower_stride", LL), ("weights", P), ("weight_tower_stride", LL), ("bias", P),
                ("bias_tower_stride", LL), ("value_w", P), ("value_b", P), ("policy_w", P), ("policy_b", P),
                ("value_targets", P), ("advantages", P), ("old_probs", P), ("ld_old", LL), ("actions", P), ("clip_scale", P),
                ("clip_epsilon", F), ("beta_entropy", F), ("grad_scale", F), ("batch", I), ("in_features", I), ("units", I),
                ("n_actions", I), ("activation", I), ("h", P), ("dz", P), ("values", P), ("logits", P), ("dvalues", P),
                ("dlogits", P), ("d_value_w", P), ("d_value_b", P), ("d_policy_w", P), ("d_policy_b", P), ("scalars", P),
                ("likelihood_ratio", P), ("clipped_likelihood_ratio", P), ("status", P), ("workspace", P),
                ("workspace_floats", LL), ("tickets", P)]


class PpoRowsDesc(ctypes.Structure):
    """rlx_ppo_rows_desc (include/rlx.h) — field order must match the header."""
    P, LL, F, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_int
    _fields_ = [("value_head", SmallDenseProblem), ("policy_head", SmallDenseProblem), ("value_targets", P), ("actions", P),
                ("advantages", P), ("old_probs", P), ("ld_old", LL), ("clip_scale", P), ("clip_epsilon", F),
                ("beta_entropy", F), ("grad_scale", F), ("batch", I), ("row_terms", P), ("scalars", P),
                ("likelihood_ratio", P), ("clipped_likelihood_ratio", P), ("status", P)]


class ObserveDesc(ctypes.Structure):
    """rlx_observe_desc (include/rlx.h) — field order must match the header."""
    P, LL, D, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_double, ctypes.c_int
    _fields_ = [("reward", P), ("filtered_reward", P), ("reward_rescale", D), ("has_clip", I), ("clip_low", D),
                ("clip_high", D), ("game_over", P), ("stored_game_over", P), ("ep_return", P), ("ep_len", P),
                ("acc", P), ("last_return", P), ("last_len", P), ("actions", P), ("action_row_bytes", LL),
                ("cur_state", P), ("next_obs", P), ("reset_obs", P), ("obs_row_bytes", LL), ("mem_action", P),
                ("mem_reward", P), ("mem_game_over", P), ("mem_obs", P), ("mem_next_obs", P), ("dst_rows", P),
                ("mem_rows", LL), ("status", P), ("n_env", I)]


ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2}
GEMM_HOOK = None   # bench.py sets this to record the descriptors issued by one update
CALL_COUNT = 0     # librlx entry points called so far (bench.py: launches of one eager update)


def _ptr(t):
    return None if t is None else (t.data_ptr() if hasattr(t, "data_ptr") else t)


def gemm(M, N, K, A, B, C, *, a_strides=None, a_tabs=None, a_u8=False, a_div=1.0,
         a_vec_along_k=1, a_tab_vec_ok=0, b_strides=None, ldc=None, bias=None, activation=None,
         deriv_aux=None, aux_ld=None, deriv_kind=None, accumulate=False, batch=1,
         a_batch_stride=0, b_batch_stride=0, c_batch_stride=0, bias_batch_stride=0,
         aux_batch_stride=0, workspace=None, colsum_out=None, colsum_batch_stride=0, stream=None,
         batch_inner=0, a_batch_stride2=0, b_batch_stride2=0, bias_batch_stride2=0, n_fold=0, launch=True,
         defer=None, row_heads=None, kw_min_tiles=0):
    """Thin wrapper building an rlx_gemm_desc.  a_strides=(row, k), b_strides=(k, n) in elements.
    launch=False: return the descriptor (for gemm_pair) instead of running it.
    defer: a SplitkJob — if K gets split over workgroups the reduction of the partials is left to
    splitk_reduce_jobs (rlx_gemm_defer); `workspace` must then stay untouched until that ran."""
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = int(M), int(N), int(K), int(batch)
    d.A, d.B, d.C = _ptr(A), _ptr(B), _ptr(C)
    if a_tabs is not None:
        d.a_row_tab, d.a_k_tab = _ptr(a_tabs[0]), _ptr(a_tabs[1])
        d.a_row_stride, d.a_k_stride = 0, 0
    else:
        rs, ks = a_strides if a_strides is not None else (K, 1)
        d.a_row_stride, d.a_k_stride = int(rs), int(ks)
    ks, ns = b_strides if b_strides is not None else (N, 1)
    d.b_k_stride, d.b_n_stride = int(ks), int(ns)
    d.a_batch_stride, d.b_batch_stride = int(a_batch_stride), int(b_batch_stride)
    d.ldc = int(N if ldc is None else ldc)
    d.c_batch_stride, d.bias_batch_stride = int(c_batch_stride), int(bias_batch_stride)
    d.bias, d.deriv_aux = _ptr(bias), _ptr(deriv_aux)
    d.aux_ld = int(N if aux_ld is None else aux_ld)
    d.aux_batch_stride = int(aux_batch_stride)
    d.workspace = _ptr(workspace)
    d.colsum_out, d.colsum_batch_stride = _ptr(colsum_out), int(colsum_batch_stride)
    d.workspace_floats = int(workspace.numel()) if workspace is not None else 0
    d.a_is_u8, d.a_div = int(bool(a_u8)), float(a_div)
    d.a_vec_along_k, d.a_tab_vec_ok = int(a_vec_along_k), int(a_tab_vec_ok)
    d.activation, d.deriv_kind = ACT[activation], ACT[deriv_kind]
    d.accumulate = int(bool(accumulate))
    d.batch_inner = int(batch_inner)
    d.a_batch_stride2, d.b_batch_stride2 = int(a_batch_stride2), int(b_batch_stride2)
    d.bias_batch_stride2 = int(bias_batch_stride2)
    d.n_fold = int(n_fold)
    d.kw_min_tiles = int(kw_min_tiles)
    if row_heads is not None:                 # a ctypes array of SmallDenseProblem (kept alive by the descriptor)
        d._row_heads = row_heads
        d.row_heads, d.n_row_heads = ctypes.addressof(row_heads), len(row_heads)
    if not launch:
        return d
    s = current_stream() if stream is None else stream
    if defer is not None:
        run = lambda: lib().gemm_defer(ctypes.byref(d), ctypes.byref(defer), s)
    else:
        run = lambda: lib().gemm(ctypes.byref(d), s)
    _record((d,), run)
    run()


def _record(descs, run, flops=None):
    """bench.py's recorder: the products a launch stands for (their 2*M*N*K*batch are its algorithmic flops unless
    `flops` says otherwise) and a thunk that issues the same launch again."""
    if GEMM_HOOK is not None:
        GEMM_HOOK({"descs": tuple(GemmDesc.from_buffer_copy(x) for x in descs), "run": run, "flops": flops})


def gemm_pair_or_single(desc, stream=None):
    """run one descriptor built with gemm(..., launch=False)."""
    s = current_stream() if stream is None else stream
    run = lambda: lib().gemm(ctypes.byref(desc), s)
    _record((desc,), run)
    run()


def gemm_multi(descs, jobs, stream=None):
    """descriptors built with gemm(..., launch=False) of up to three weight-gradient products -> ONE launch
    (rlx_gemm_multi_defer); jobs: their SplitkJob objects (filled where a split-K reduction stays outstanding)."""
    s = current_stream() if stream is None else stream
    n = len(descs)
    darr = (GemmDesc * n)(*descs)
    jarr = (SplitkJob * n)()

    def run():
        lib().gemm_multi_defer(darr, n, jarr, s)
        for i, j in enumerate(jobs):
            ctypes.memmove(ctypes.addressof(j), ctypes.addressof(jarr[i]), ctypes.sizeof(SplitkJob))
    _record(tuple(descs), run)                     # one entry: the products go out as one launch
    run()


def gemm_pair(weight_grad, input_grad, stream=None, defer=None):
    """two descriptors built with gemm(..., launch=False): a layer's dW and dX products as one launch.
    defer: a SplitkJob for the weight gradient's split-K reduction (see gemm)."""
    s = current_stream() if stream is None else stream
    if defer is not None:
        run = lambda: lib().gemm_pair_defer(ctypes.byref(weight_grad), ctypes.byref(input_grad), ctypes.byref(defer), s)
    else:
        run = lambda: lib().gemm_pair(ctypes.byref(weight_grad), ctypes.byref(input_grad), s)
    _record((weight_grad, input_grad), run)        # one entry: the two products go out as one launch
    run()
