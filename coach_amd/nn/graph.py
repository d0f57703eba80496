"""Layer graph executed through the C ABI (librlx.so): the MI355X backend behind the reference's
``Architecture`` plug point (rl_coach/architectures/architecture.py:26-237).

Design (MI355X-first, not a translation of the TF1 graph/session code):
  * every network keeps weights / gradients / Adam moments in ONE flat fp32 HBM buffer each
    (16-byte aligned tensors): Adam, target mixing, gradient norm and the data-parallel RCCL
    all-reduce are each a single launch over that buffer;
  * identical towers (Clipped-PPO's ``use_separate_networks_per_head`` value / policy copies,
    clipped_ppo_agent.py:50; TD3's twin critic streams) run as ONE batched GEMM launch per layer
    (blockIdx.z = tower) instead of one launch per copy;
  * convolutions are implicit-im2col GEMMs reading the uint8 frame stack directly;
  * PyTorch is used only to own device memory and streams.

Semantics restated from rl_coach/architectures/tensorflow_components/
(general_network.py:228-405, layers.py:108-185, embedders/embedder.py:100-125,
middlewares/fc_middleware.py:40-70): activation after every layer, VALID conv padding, NHWC,
Xavier-uniform kernels, zero biases (layers.py:180-181).
"""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _rlx

ALIGN = 4  # floats (16 bytes)
SMALL_N = 16  # widest layer served by the narrow-dense kernels (csrc/dense_small.hip)
# towers that read the same input run their first layer as ONE GEMM over the concatenated output
# columns (rlx_gemm_desc.n_fold)
FOLD_SHARED_INPUT = True


def _thin_takes(M, N, K, batch):
    return K <= 1024 and N * K <= (1 << 18) and -(-M // 64) * -(-N // 64) * batch <= 96


def _fold(M, N, K, T):
    """... unless the per-tower problem fits the thin GEMM kernel (csrc/gemm.hip: K <= 1024, N * K <= 2^18, <= 96 tiles
    of 64 x 64) AND the folded product would have to split K over workgroups (K >= 64): T batched thin launches on
    16 x 16 tiles then beat one folded tiled GEMM + its reduce launch (SAC's Q towers, C5 +14 %; TD3's critic with
    K = 23 stays folded, C4 -3 % otherwise — profiles/r03_ab_candidates.txt)."""
    return FOLD_SHARED_INPUT and not (_thin_takes(M, N, K, T) and K >= 64)


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class FlatParams:
    """Named fp32 tensors carved out of one flat buffer; `towers` identical copies are laid out
    tower-major so that tensor `name` of tower t sits at offset(name) + t * tower_stride."""

    def __init__(self):
        self.entries = {}       # name -> (offset, shape, towers, tower_stride)
        self.size = 0
        self.weights = self.grads = None

    def add_group(self, specs, towers=1):
        per_tower = sum(_align(int(np.prod(s))) for _, s in specs)
        base, off = self.size, 0
        for name, shape in specs:
            if name in self.entries:
                raise ValueError("duplicate parameter name %s" % name)
            self.entries[name] = (base + off, tuple(shape), towers, per_tower)
            off += _align(int(np.prod(shape)))
        self.size += per_tower * towers
        return per_tower

    def finalize(self, device):
        # online and target copies live in ONE allocation [2, size]: a "paired" forward pass runs the
        # online network on s and the target network on s' as two towers of the same launches
        # (weight tower stride = size), which is what NetworkWrapper.parallel_prediction
        # (network_wrapper.py:188-213) asks the backend for
        self.both = torch.zeros(2, self.size, dtype=torch.float32, device=device)
        self.weights = self.both[0]
        self.target_weights = self.both[1]
        self.grads = torch.zeros(self.size, dtype=torch.float32, device=device)
        return self

    def view(self, buf, name, tower=0):
        off, shape, towers, stride = self.entries[name]
        n = int(np.prod(shape))
        return buf[off + tower * stride: off + tower * stride + n].view(*shape)

    def w(self, name, tower=0, buf=None):
        return self.view(self.weights if buf is None else buf, name, tower)

    def g(self, name, tower=0):
        return self.view(self.grads, name, tower)

    def stride(self, name):
        return self.entries[name][3]

    def named_arrays(self, buf=None):
        """{name: [numpy array per tower]} — used by the parity tests to feed the oracle."""
        out = {}
        for name, (off, shape, towers, stride) in self.entries.items():
            out[name] = [self.view(self.weights if buf is None else buf, name, t).cpu().numpy().copy()
                         for t in range(towers)]
        return out


def xavier_uniform(rng, fan_in, fan_out, shape):
    """tf.glorot_uniform_initializer (the variable-scope default, tensorflow_components/
    architecture.py:100): U(-l, l), l = sqrt(6 / (fan_in + fan_out))."""
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def normalized_columns(std):
    """heads/head.py:27-33 — consumes the GLOBAL np.random stream exactly like the reference."""
    def init(shape):
        out = np.random.randn(*shape).astype(np.float32)
        out *= std / np.sqrt(np.square(out).sum(axis=0, keepdims=True))
        return out
    return init


# a layer's dW and dX products (independent, same dz) go out as ONE launch (rlx_gemm_pair); the tests switch it off to
# compare with the two-launch sequence
PAIR_GRADIENT_GEMMS = True
# Input gradient of a convolution as one windowed-gather product (rlx_conv_input_grad) instead of dcol = dz W^T + col2im:
# True = where its extra products stay < 1.3x, "always", or False.  Off by default: same-box A/Bs of the C2 update
# (profiles/r03_ab_gemm_changes.txt) show -0 .. +5 us per update on the pool's fast boxes and +25 us on its slow ones
# (the gather's dependent loads pay the slower memory system twice); it halves the HBM bytes of the convolutions'
# backward pass, so it stays available for configurations where that is the limit.
DIRECT_CONV_INPUT_GRAD = False
# The Atari torso's second and third convolution as ONE launch with the second layer's output in LDS
# (rlx_conv23_forward, csrc/conv_fused.hip) — taken only where the two tiled launches it replaces would run on 32 x 64
# tiles with two wave groups per K slab (_kw2_tiling below: rlx_gemm's own rule), because exactly there the fused
# kernel's sums are bit-identical to theirs.  The tests flip it to compare the two paths.
FUSE_CONV_PAIR = True
# ... and the first convolution in front of them in the same launch (rlx_conv123_forward), where rlx_gemm_describe says its
# own launch would sum in one of the two orders that kernel reproduces (Sequential._fused_conv_triple)
FUSE_CONV_FIRST = True
# The input gradients of the third and second convolution in the backward pass as ONE launch with both column matrices in
# LDS (rlx_conv32_input_grad, csrc/conv_bwd_fused.hip) instead of a dW + dcol pair launch and a col2im launch per layer; the
# two weight-gradient products then run on their own.  Taken where rlx_gemm_describe says the dcol products would sum in
# the order that kernel reproduces, and only when there are enough half images (workgroups) to fill the chip.
# Round 5 left it OFF (-2 us per C2 update: the weight-gradient products it leaves behind cost 14 us each as tiled launches of
# their own, profiles/r05_ab_conv32.txt); ON since those products run through rlx_conv_dw_f32 (11.5 / 12.9 us):
# 182.3 against 188.8 us per update (profiles/r06_ab_conv_dw_f32.txt).  bench.py --fuse-conv-bwd 0 / 1 is the A/B.
# At 64 half-image workgroups (the DQN update: B = 32, one tower) the chain alone is no faster than the two tiled pairs + col2im
# it replaces (24.3 against 33.7 us, but it leaves the weight gradients behind); with all three weight gradients in ONE
# LDS-resident launch behind it (conv1 through conv_dw_u8_body_half) the C3 update is 143.3 against 150.1 us
# (gpurun_out/r06_call34, profiles/r06_ab_c3_backward.txt).
FUSE_CONV_INPUT_GRADS = True
FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS = 64
# a pending PrioritizedExperienceReplay priority update (Context.per_tail) rides on the fused input-gradient launch while
# that launch leaves CUs idle (one workgroup per CU: 115 KB of LDS each); behind a full chip it would start last
PER_RIDER_MAX_WORKGROUPS = 256
# Wave groups per K slab of the fused forward launch (rlx_conv23_forward / rlx_conv123_forward).  None: what the tiled launches
# would have used at this batch (2 for two towers of 63 .. 75 images — the third convolution then runs on four of the
# eight waves, as the tiled kernel split its K chain — 4 for acting and the DQN update): the fused launch is then
# bit-identical to the tiled ones.  4: every fused forward launch runs with four groups (conv3 on all eight waves); the
# sums of the two-tower minibatch then differ from the tiled launches' in the last bits (another grouping of the k-quads),
# inside tests/tolerances.py against the oracle.
DENSE_DX_KW_MIN_TILES = 96       # rlx_gemm_desc.kw_min_tiles of a dense layer's input gradient (0: the library's 192)
CONV1_CHUNKS = None      # A/B (bench.py --conv1-chunks): accumulator chains of conv1's K = 256 in the fused forward launch (None: as the tiled launch would split K)
CONV_FORWARD_WAVE_GROUPS = 4           # (-0.9 us on the C2 update's forward launch, profiles/r06_ab_conv_fwd_groups_split_cap.txt)


def _fused_operands_aligned(args):
    """what RLX_REQUIRE checks inside rlx_conv23_forward / rlx_conv123_forward, asked BEFORE the fused launch is chosen: every
    tensor operand 16-byte aligned, every tower stride (the integer behind a tensor) a multiple of 4 floats — a misaligned
    caller-supplied weights view or output buffer takes the tiled launches instead of raising."""
    prev_tensor = False
    for a in args:
        if hasattr(a, "data_ptr"):
            if a.data_ptr() % 16:
                return False
            prev_tensor = True
        else:
            if prev_tensor and isinstance(a, int) and a % 4:
                return False
            prev_tensor = False
    return True


def _tiled_wave_groups(M, N, batch):
    """How rlx_gemm would tile an M x N x K product of `batch` towers (csrc/gemm.hip gemm_impl, with the LIVE thresholds of
    rlx_gemm_tuning: 192 / 192 unless an A/B tool changed them): 2 = 32 x 64 tiles with the K slab split over two wave
    groups, 4 = 32 x 32 tiles with four, 1 = anything else (64 x 64 tiles, narrow tiles, K split over workgroups)."""
    if N <= 32:
        return 1
    below, least = ctypes.c_int(), ctypes.c_int()
    _rlx.lib().gemm_tuning_get(ctypes.byref(below), ctypes.byref(least), None)
    t64 = -(-M // 64) * -(-N // 64) * batch
    if t64 >= below.value:
        return 1
    if -(-M // 32) * -(-N // 64) * batch >= least.value:
        return 2
    if -(-M // 32) * -(-N // 32) * batch >= least.value:
        return 4
    return 1


def _kw2_tiling(M, N, batch):
    return _tiled_wave_groups(M, N, batch) == 2


# The convolution layers' weight-gradient products of a backward pass as ONE launch behind the input-gradient chain
# (rlx_gemm_multi_defer) instead of one per layer inside its dW + dX pair.  Same tilings and K splits: bit-identical sums.
# OFF: measured +8 us per C2 update (profiles/r05_ab_multi_dw.txt) — the one launch takes as long as the three products one
# after the other; co-resident independent products do not overlap on this chip at these sizes.
MULTI_DW = False
# most K splits a deferred product may get room for (rlx_gemm_split_cap's largest useful setting; its default is 64)
SPLIT_CAP_BOUND = 128
# conv1's weight gradient from the uint8 frames by rlx_conv_dw_u8 (one workgroup per image and pair of kernel rows) instead of
# the register-staged implicit-im2col product of rlx_gemm; the tests flip it to compare the two
CONV_DW_U8 = True
# the inner convolutions' weight gradients, when they run on their own, by rlx_conv_dw_f32 (csrc/conv_dw_f32.hip)
CONV_DW_F32 = True
# ... and all of a backward pass's LDS-resident weight-gradient products as ONE launch (rlx_conv_dw_multi)
CONV_DW_ONE_LAUNCH = True


class Workspace:
    def __init__(self, device, floats=1 << 24):
        self.splitk = torch.empty(floats, dtype=torch.float32, device=device)
        self.splitk_side = torch.empty(floats, dtype=torch.float32, device=device)   # side-stream GEMMs
        self.small = torch.empty(1 << 18, dtype=torch.float32, device=device)


class Context:
    """Scratch owned by one network: cached activation / gradient buffers, conv offset tables."""

    def __init__(self, device, ws=None):
        self.device = device
        self.lib = _rlx.lib()
        self.ws = ws or Workspace(device)
        self.cache = {}
        self.buffers = {}
        # (weight-gradient GEMMs on a second HIP stream, concurrent with the input-gradient chain, measured slower inside
        # hipGraph replays on MI355X / ROCm 7.2 — C2 126 vs 116 ms, C3 376 vs 318 ms per bench step: the cross-stream edges
        # cost more than the concurrency returns at these launch sizes; removed)
        self.overlap = False
        self.deferred, self.arena, self._arena_off = None, None, 0

    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def join(self):
        pass

    # ---- deferred split-K reductions of the weight gradients (rlx_gemm_defer): every dW product of a backward pass
    # leaves its partial sums in its own slice of an arena; ONE launch sums them all when the pass ends.
    # ONE arena per device, shared by every network's context: a backward pass flushes its reductions before it returns
    # and the passes of one process queue on one stream, so two contexts never have partial sums outstanding together
    # (128 MB once, instead of once per network).  Deferral changes the GROUPING of a reduction's additions for products
    # with <= 16 splits (the one launch always sums 16 split groups, the immediate reduction 4 for those): same partials,
    # last-bit differences in such a weight gradient (tests/test_gemm.py states the tolerance).  Making the two groupings
    # equal was tried and undone: it moves every network's trajectory by those last bits, and the device-vs-recorded-
    # reference loop tests (tests/test_reference_image_loops.py) pin action sequences that contain fp32 near-ties.
    # The sharing rests on "one stream": a context whose passes run on a SIDE stream next to another context's
    # (SoftActorCriticAgent.parallel_branches) sets private_arena = True and gets an arena of its own.
    ARENA_FLOATS = 1 << 25
    _arenas = {}
    private_arena = False

    def begin_deferring(self):
        """-> True if this call opened the deferral (its caller must flush)."""
        if self.deferred is not None:
            return False
        if self.arena is None or (self.private_arena and self.arena is Context._arenas.get(str(self.device))):
            if self.private_arena:
                # owned by this context and freed with it (a class-level entry keyed by id(self) would outlive the
                # context and could be picked up by a later one with the same id)
                self.arena = torch.empty(self.ARENA_FLOATS, dtype=torch.float32, device=self.device)
            else:
                key = str(self.device)
                if key not in Context._arenas:
                    Context._arenas[key] = torch.empty(self.ARENA_FLOATS, dtype=torch.float32, device=self.device)
                self.arena = Context._arenas[key]
        self.deferred, self._arena_off = [], 0
        return True

    def deferred_workspace(self, M, N, batch):
        """(job, workspace slice) for one deferred product with an M x N output per batch entry."""
        bound = (M * N + N) * batch * SPLIT_CAP_BOUND
        room = self.ARENA_FLOATS - self._arena_off
        if self.deferred is None or len(self.deferred) >= _rlx.MAX_SPLITK_JOBS or room < (M * N + N) * batch * 2:
            return None, self.ws.splitk
        return _rlx.SplitkJob(), self.arena[self._arena_off:self._arena_off + min(bound, room)]

    def commit_deferred(self, job, reserved=False):
        """reserved: the job's slice was set aside by reserve_deferred_workspace (the arena offset already moved)."""
        if job is not None and job.splits > 1:
            self.deferred.append(job)
            if not reserved:
                self._arena_off += (job.workspace_floats() + 3) // 4 * 4

    def reserve_deferred_workspace(self, M, N, batch):
        """deferred_workspace for a product that is launched LATER, together with others (rlx_gemm_multi_defer): its
        slice — the most its split-K partials can take — is set aside now, so that the next product's slice lies behind it."""
        job, ws = self.deferred_workspace(M, N, batch)
        if job is not None:
            self._arena_off += (ws.numel() + 3) // 4 * 4
        return job, ws

    def flush_deferred(self):
        jobs, self.deferred = self.deferred, None
        tail, self.ppo_tail = self.ppo_tail, None      # (ppo_fc_rows: the heads' all-rows part rides on this launch)
        per, self.per_tail = self.per_tail, None       # (prioritized replay: update_priorities rides on this launch)
        if jobs or tail is not None or per is not None:
            _rlx.splitk_reduce_jobs(jobs or [], self.stream, ppo_tail=tail, per_tail=per)

    ppo_tail = None
    per_tail = None            # set by the agent in front of an update: PrioritizedExperienceReplay.priority_update_args()

    def flush_ppo_tail(self):
        """the pending all-rows part of ppo_fc_rows as a launch of its own (no deferred-reduction launch took it along)."""
        tail, self.ppo_tail = self.ppo_tail, None
        if tail is not None:
            self.lib.ppo_heads_tail(ctypes.byref(tail), self.stream)

    def buffer(self, name, shape, dtype=torch.float32, tag=""):
        key = (name, tuple(shape), dtype, tag)
        buf = self.buffers.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, device=self.device)
            self.buffers[key] = buf
        return buf


class Tensor:
    """An activation `data` [towers, rows, cols]; towers == 0 marks an input shared by all towers.
    For NHWC images `rows` counts samples and cols = H*W*C."""

    def __init__(self, data, rows, cols, towers, u8=False, div=1.0, grad_key=None, act=None):
        self.data, self.rows, self.cols, self.towers = data, rows, cols, towers
        self.u8, self.div = u8, div
        self.act = act                      # activation of the layer that produced this tensor
        self.grad_is_dz = False             # grad already carries act'(data) (applied by the consumer)
        self.grad = None
        self.grad_key = grad_key            # (ctx, name, tag): gradient lives in a cached buffer
        self.bn_saved = None                # BatchNorm output: (batch mean, batch variance) of the pass, or (None, None)

    def tower_stride(self):
        return 0 if self.towers == 0 else self.rows * self.cols

    def ensure_grad(self):
        if self.grad is None:
            if self.grad_key is not None:
                ctx, name, tag = self.grad_key
                self.grad = ctx.buffer(name + ":grad", tuple(self.data.shape), tag=tag)
            else:
                self.grad = torch.empty(self.data.shape, dtype=torch.float32, device=self.data.device)
        return self.grad

    def tower(self, t):
        """One tower as a single-tower tensor; data and gradient alias this tensor's memory."""
        g = self.ensure_grad()
        v = Tensor(self.data[t:t + 1], self.rows, self.cols, 1, u8=self.u8, div=self.div, act=self.act)
        v.grad = g[t:t + 1]
        v.bn_saved = self.bn_saved
        return v

    def slice_towers(self, t0, n, with_grad=True):
        """towers [t0, t0 + n) as a tensor of n towers aliasing this one (data and gradient)."""
        v = Tensor(self.data[t0:t0 + n], self.rows, self.cols, n, u8=self.u8, div=self.div, act=self.act)
        if with_grad:
            v.grad = self.ensure_grad()[t0:t0 + n]
        v.bn_saved = self.bn_saved
        return v

    def tower_view(self, t):
        """tower t without touching gradients (inputs)."""
        return Tensor(self.data[t:t + 1], self.rows, self.cols, 1, u8=self.u8, div=self.div, act=self.act)


def input_tensor(data, rows, cols, u8=False, div=1.0):
    return Tensor(data, rows, cols, 0, u8=u8, div=div)


class Layer:
    def _range(self, t0, nt):
        nt = self.T - t0 if nt is None else nt
        assert 0 <= t0 and t0 + nt <= self.T
        return t0, nt


class Dense(Layer):
    """y = act(x W + b), tf.layers.dense (layers.py:168-185).  `towers` copies in one launch."""

    def __init__(self, params, name, in_features, units, activation=None, towers=1, init=None):
        self.name, self.K, self.N, self.act, self.T = name, in_features, units, activation, towers
        self.init, self.params = init, params
        self.kname, self.bname = name + "/kernel", name + "/bias"
        params.add_group([(self.kname, (in_features, units)), (self.bname, (units,))], towers)

    def initialize(self, rng):
        for t in range(self.T):
            w = self.init((self.K, self.N)) if self.init is not None else \
                xavier_uniform(rng, self.K, self.N, (self.K, self.N))
            self.params.w(self.kname, t).copy_(torch.from_numpy(np.ascontiguousarray(w)))

    def forward(self, ctx, x, tag="", weights=None, t0=0, nt=None, pair=False, launch=True, row_heads=None):
        """pair=True: the online and the target copy of the layer in one launch — towers [0, T) use
        the online weights, [T, 2T) the target weights (FlatParams.both); x carries either one tower
        per copy (shared by the T streams of a copy) or one per (copy, stream).
        launch=False: -> (output Tensor, GEMM descriptor not yet run); the descriptor is None when the layer took a
        path that is not a plain batched GEMM (it has then been run)."""
        t0, T = self._range(t0, nt)
        p = self.params
        M = x.rows
        assert x.cols == self.K, (self.name, x.cols, self.K)
        wstride, bstride = p.stride(self.kname), p.stride(self.bname)
        kw = {}
        if pair:
            assert weights is None and t0 == 0 and T == self.T
            if x.towers == 2 * T:
                a_stride, a_stride2 = M * self.K, T * M * self.K
            else:
                assert x.towers == 2, (self.name, x.towers)
                a_stride, a_stride2 = 0, M * self.K
            kw = dict(batch_inner=T, a_batch_stride2=a_stride2, b_batch_stride2=p.size, bias_batch_stride2=p.size)
            TT = 2 * T
        else:
            assert x.towers in (0, T), (self.name, x.towers, T)
            a_stride, TT = x.tower_stride(), T
        y = ctx.buffer(self.name, (TT, M, self.N), tag=tag)
        desc, head_outs = None, None
        if self.N <= SMALL_N and not x.u8:
            # heads: coalesced FMA kernel, no MFMA tile / split-K round trip (csrc/dense_small.hip)
            if not pair:
                ctx.lib.dense_small_forward(x.data, a_stride, p.w(self.kname, t0, weights), wstride,
                                            p.w(self.bname, t0, weights), bstride, y, M * self.N, T, M,
                                            self.K, self.N, _rlx.ACT[self.act], ctx.stream)
            else:                                             # online and target copy: 2 problems, 1 launch
                import ctypes
                arr = (_rlx.SmallDenseProblem * 2)()
                for c in range(2):
                    q = arr[c]
                    q.x, q.x_tower_stride = x.data.data_ptr() + c * a_stride2 * 4, a_stride
                    q.w, q.w_tower_stride = p.w(self.kname).data_ptr() + c * p.size * 4, wstride
                    q.bias, q.bias_tower_stride = p.w(self.bname).data_ptr() + c * p.size * 4, bstride
                    q.y, q.y_tower_stride = y.data_ptr() + c * T * M * self.N * 4, M * self.N
                    q.towers, q.M, q.K, q.N, q.activation = T, M, self.K, self.N, _rlx.ACT[self.act]
                ctx.lib.dense_small_forward_multi(ctypes.byref(arr), 2, ctx.stream)
        elif launch and not pair and x.towers == 0 and T > 1 and self.N % 4 == 0 and _fold(M, self.N, self.K, T):
            # the T towers read the SAME input: one GEMM over T*N columns loads (gathers) it once
            _rlx.gemm(M, T * self.N, self.K, x.data, p.w(self.kname, t0, weights), y,
                      b_strides=(self.N, 1), ldc=self.N, bias=p.w(self.bname, t0, weights),
                      activation=self.act, batch=1, b_batch_stride=wstride, c_batch_stride=M * self.N,
                      bias_batch_stride=bstride, workspace=ctx.ws.splitk, n_fold=self.N)
        else:
            if row_heads:
                heads_arr, head_outs = row_head_problems(ctx, row_heads, y, M, self.N, tag, weights)
                kw = dict(kw, row_heads=heads_arr)
            desc = _rlx.gemm(M, self.N, self.K, x.data, p.w(self.kname, t0, weights), y,
                             bias=p.w(self.bname, t0, weights), activation=self.act, batch=TT,
                             a_batch_stride=a_stride, b_batch_stride=wstride,
                             c_batch_stride=M * self.N, bias_batch_stride=bstride,
                             workspace=ctx.ws.splitk, launch=launch, **kw)
        T = TT
        out = Tensor(y, M, self.N, T, grad_key=(ctx, self.name, tag), act=self.act)
        if row_heads:
            assert head_outs is not None and launch, "row_heads: only on the tiled-GEMM path of a wide layer"
            return out, head_outs
        return out if launch else (out, desc)

    def backward(self, ctx, x, y, need_dx=True, weights=None, t0=0, nt=None, need_dw=True, overlap=False):
        """y.grad holds dL/dy (post-activation); writes dW, db and (optionally) x.grad.
        need_dw=False only propagates to the input (gradients_wrt_inputs, architecture.py:187-220).
        overlap: issue the dW GEMM on the context's side stream (the caller joins)."""
        t0, T = self._range(t0, nt)
        M, p = x.rows, self.params
        overlap = overlap and ctx.overlap
        dz = y.grad
        own_act = self.act if (self.act is not None and not y.grad_is_dz) else None
        lower = x.act if need_dx else None           # fuse the lower layer's act' into dx
        shared_dx = need_dx and x.towers == 0 and T > 1     # T streams fed by ONE input: dx sums them
        if need_dx and not shared_dx:
            assert x.towers == T, "input gradients need a per-tower input"
        # one launch for dW + db + dx while the batch is small; beyond ~1k outputs its few workgroups
        # lose to the GEMM path (M=256, N=16: 56 us measured vs ~30)
        if self.N <= SMALL_N and not x.u8 and M * self.N <= 1024 and not shared_dx:
            dx = x.ensure_grad() if need_dx else None
            ctx.lib.dense_small_backward(
                x.data, x.tower_stride(), p.w(self.kname, t0, weights), p.stride(self.kname),
                dz, M * self.N, y.data if own_act else None, M * self.N,
                p.g(self.kname, t0) if need_dw else None, p.stride(self.kname),
                p.g(self.bname, t0) if need_dw else None, p.stride(self.bname),
                dx, M * self.K, T, M, self.K, self.N, _rlx.ACT[own_act], _rlx.ACT[lower], ctx.stream)
            if need_dx:
                x.grad_is_dz = lower is not None
            return
        if own_act is not None:                       # dz = dy * act'(y), in place
            ctx.lib.act_backward(dz, y.data, T * M * self.N, _rlx.ACT[own_act], ctx.stream)
        # dW[K,N] = x^T dz : A(k, m) = x[m, k]
        # (db = column sums of dz, accumulated by the same launch from the staged B slabs)
        if need_dw:
            fold = x.towers == 0 and T > 1 and self.N % 4 == 0 and _fold(self.K, self.N, M, T)
            job, dws = ctx.deferred_workspace(self.K, T * self.N if fold else self.N, 1 if fold else T) \
                if not overlap else (None, None)

            def dw(ws, launch=True):
                ws = dws if job is not None else ws
                dj = job if launch else None          # a pair launch passes the job itself
                if fold:       # shared input: dW of all towers = x^T [dz_0 | dz_1 | ...] in one GEMM
                    return _rlx.gemm(self.K, T * self.N, M, x.data, dz, p.g(self.kname, t0), a_strides=(1, self.K),
                                     b_strides=(self.N, 1), ldc=self.N, batch=1, b_batch_stride=M * self.N,
                                     c_batch_stride=p.stride(self.kname), workspace=ws,
                                     colsum_out=p.g(self.bname, t0), colsum_batch_stride=p.stride(self.bname),
                                     n_fold=self.N, launch=launch, defer=dj)
                return _rlx.gemm(self.K, self.N, M, x.data, dz, p.g(self.kname, t0), a_strides=(1, self.K), batch=T,
                                 a_batch_stride=x.tower_stride(), b_batch_stride=M * self.N,
                                 c_batch_stride=p.stride(self.kname), workspace=ws,
                                 colsum_out=p.g(self.bname, t0), colsum_batch_stride=p.stride(self.bname),
                                 launch=launch, defer=dj)
            pair_dw = need_dx and not shared_dx and not overlap and PAIR_GRADIENT_GEMMS   # dW rides with the dX launch
            if not pair_dw:
                dw(ctx.ws.splitk)
        if shared_dx:
            # dx = sum_t dz_t W_t^T, each term already multiplied by the lower layer's act'(x)
            dx = x.ensure_grad()
            for t in range(T):
                _rlx.gemm(M, self.K, self.N, dz[t], p.w(self.kname, t0 + t, weights), dx, b_strides=(1, self.N),
                          workspace=ctx.ws.splitk, deriv_aux=x.data if lower else None, aux_ld=self.K,
                          deriv_kind=lower, accumulate=t > 0)
            x.grad_is_dz = lower is not None
        elif need_dx:
            dx = x.ensure_grad()
            # dx[M,K] = dz W^T : B(n, k) = W[k, n]; the epilogue multiplies by the lower layer's
            # activation derivative, so dx IS that layer's dz (no separate act_backward launch)
            # (kw_min_tiles: at batch 32 the product is 98 tiles of 32 x 32 — K over the waves of a workgroup instead of a
            # K split over workgroups + a reduce launch in the chain: -3.4 us per DQN update, DENSE_DX_KW_MIN_TILES)
            dxd = _rlx.gemm(M, self.K, self.N, dz, p.w(self.kname, t0, weights), dx, b_strides=(1, self.N),
                            batch=T, a_batch_stride=M * self.N, b_batch_stride=p.stride(self.kname),
                            c_batch_stride=M * self.K, workspace=ctx.ws.splitk,
                            deriv_aux=x.data if lower else None, aux_ld=self.K, deriv_kind=lower,
                            aux_batch_stride=M * self.K, launch=False, kw_min_tiles=DENSE_DX_KW_MIN_TILES)
            if need_dw and not overlap and PAIR_GRADIENT_GEMMS:
                # dW and dX are independent products of the same dz: one launch (own split-K workspaces)
                _rlx.gemm_pair(dw(ctx.ws.splitk_side, launch=False), dxd, defer=job)
            else:
                _rlx.gemm_pair_or_single(dxd)
            x.grad_is_dz = lower is not None
        if need_dw:
            ctx.commit_deferred(job)


class BatchNorm(Layer):
    """tf.layers.batch_normalization(x, training=is_training) + the activation, after a Dense layer built without one
    (BatchnormActivationDropout, tensorflow_components/layers.py:26-55, 138-152): momentum 0.99, epsilon 1e-3, gamma / beta
    trainable, moving_mean / moving_variance not.  csrc/batchnorm.hip.

    The four vectors live in the network's flat parameter buffer: gamma and beta get gradients like any weight; the
    gradient slots of the moving statistics stay zero, so TF1-Adam leaves them alone (m = v = 0 -> a zero step) and only
    commit() — the UPDATE_OPS tied to apply_gradients (architecture.py:273-277) — moves them.  Context.bn_training is the
    network's is_training variable (architecture.py:627-633): batch statistics inside Agent.train, moving ones when acting.
    """
    MOMENTUM, EPSILON = 0.99, 1e-3
    TRACKED_TAG = "train"         # the pass whose batch feeds the moving averages at apply time (ddpg_agent.py:178-193)

    def __init__(self, params, name, channels, activation=None):
        self.name, self.C, self.act, self.T, self.params = name, channels, activation, 1, params
        self.gname, self.bname = name + "/gamma", name + "/beta"
        self.mname, self.vname = name + "/moving_mean", name + "/moving_variance"
        params.add_group([(self.gname, (channels,)), (self.bname, (channels,)), (self.mname, (channels,)),
                          (self.vname, (channels,))], 1)

    def initialize(self, rng):
        self.params.w(self.gname).fill_(1.0)
        self.params.w(self.vname).fill_(1.0)

    def _saved(self, ctx, tag):
        return (ctx.buffer(self.name + "/batch_mean", (self.C,), tag=tag),
                ctx.buffer(self.name + "/batch_variance", (self.C,), tag=tag))

    def forward(self, ctx, x, tag="", weights=None, t0=0, nt=None, pair=False):
        assert not pair and x.towers == 1 and x.cols == self.C and x.act is None, (self.name, x.towers, x.cols, x.act)
        p, M = self.params, x.rows
        training = bool(getattr(ctx, "bn_training", False))
        y = ctx.buffer(self.name, (1, M, self.C), tag=tag)
        mean, var = self._saved(ctx, tag) if training else (None, None)
        ctx.lib.bn_forward(x.data, p.w(self.gname, 0, weights), p.w(self.bname, 0, weights),
                           p.w(self.mname, 0, weights), p.w(self.vname, 0, weights), M, self.C, self.EPSILON,
                           int(training), _rlx.ACT[self.act], y, mean, var, ctx.stream)
        out = Tensor(y, M, self.C, 1, grad_key=(ctx, self.name, tag), act=self.act)
        out.bn_saved = (mean, var)
        return out

    def backward(self, ctx, x, y, need_dx=True, weights=None, t0=0, nt=None, need_dw=True, overlap=False):
        """y.grad = dL/dy (or dL/du when the layer above already applied act'); writes x.grad = dL/d(dense output) and
        the gamma / beta gradients.  Batch-statistics form only (the reference differentiates inside Agent.train)."""
        mean, var = y.bn_saved
        if mean is None:
            raise ValueError("%s: backward through a batch-norm pass that ran on the moving statistics" % self.name)
        p, M = self.params, x.rows
        own = self.act if (self.act is not None and not y.grad_is_dz) else None
        ctx.lib.bn_backward(y.grad, y.data if own else None, x.data, p.w(self.gname, 0, weights), mean, var, M, self.C,
                            self.EPSILON, _rlx.ACT[own], x.ensure_grad(), p.g(self.gname) if need_dw else None,
                            p.g(self.bname) if need_dw else None, ctx.stream)
        x.grad_is_dz = False

    def commit(self, ctx):
        """moving statistics <- the batch statistics of the tracked pass (run with the pre-update weights)."""
        p = self.params
        mean, var = self._saved(ctx, self.TRACKED_TAG)
        ctx.lib.bn_update_moving(p.w(self.mname), p.w(self.vname), mean, var, self.C, self.MOMENTUM, ctx.stream)


class Conv2d(Layer):
    """tf.layers.conv2d, VALID, NHWC (layers.py:108-121) as an implicit-im2col GEMM."""

    def __init__(self, params, name, in_hwc, filters, kernel, stride, activation=None, towers=1):
        self.name, self.T, self.act, self.params = name, towers, activation, params
        self.H, self.W, self.C = in_hwc
        self.KH = self.KW = kernel
        self.S, self.Co = stride, filters
        self.OH = (self.H - kernel) // stride + 1
        self.OW = (self.W - kernel) // stride + 1
        self.K = kernel * kernel * self.C
        self.kname, self.bname = name + "/kernel", name + "/bias"
        params.add_group([(self.kname, (self.K, filters)), (self.bname, (filters,))], towers)

    @property
    def out_hwc(self):
        return (self.OH, self.OW, self.Co)

    def initialize(self, rng):
        fan_in, fan_out = self.K, self.KH * self.KW * self.Co
        for t in range(self.T):
            w = xavier_uniform(rng, fan_in, fan_out, (self.K, self.Co))
            self.params.w(self.kname, t).copy_(torch.from_numpy(w))

    def _tables(self, ctx, B):
        key = ("convtab", self.name, B)
        if key not in ctx.cache:
            rb = torch.empty(B * self.OH * self.OW, dtype=torch.int32, device=ctx.device)
            ko = torch.empty(self.K, dtype=torch.int32, device=ctx.device)
            ctx.lib.conv_tables(rb, ko, B, self.H, self.W, self.C, self.KH, self.KW, self.S, ctx.stream)
            ctx.cache[key] = (rb, ko)
        return ctx.cache[key]

    def forward(self, ctx, x, tag="", weights=None, t0=0, nt=None, pair=False):
        y, T, args, kw = self.forward_product(ctx, x, tag, weights, t0, nt, pair)
        _rlx.gemm(*args, **kw)
        # same memory read as [T, B, OH*OW*Co]: flattening (embedder.py:120-121) is free
        return Tensor(y, x.rows, self.OH * self.OW * self.Co, T, grad_key=(ctx, self.name, tag), act=self.act)

    def forward_product(self, ctx, x, tag="", weights=None, t0=0, nt=None, pair=False):
        """-> (output buffer, towers, _rlx.gemm's positional and keyword arguments) of the layer's forward product."""
        t0, T = self._range(t0, nt)
        B, p = x.rows, self.params
        wstride, bstride = p.stride(self.kname), p.stride(self.bname)
        if pair:
            assert self.T == 1 and weights is None
            T, wstride, bstride = 2, p.size, p.size
        assert x.cols == self.H * self.W * self.C, (self.name, x.cols)
        assert x.towers in (0, T)
        M = B * self.OH * self.OW
        rb, ko = self._tables(ctx, B)
        y = ctx.buffer(self.name, (T, M, self.Co), tag=tag)
        if not pair and x.towers == 0 and T > 1 and self.Co % 4 == 0 and self.C % 4 == 0 and FOLD_SHARED_INPUT:
            # the towers convolve the SAME frames: one implicit-im2col GEMM over T*Co output channels
            # gathers (and converts) every patch once instead of T times
            return y, T, (M, T * self.Co, self.K, x.data, p.w(self.kname, t0, weights), y), dict(
                a_tabs=(rb, ko), a_u8=x.u8, a_div=x.div, a_vec_along_k=1, a_tab_vec_ok=1, b_strides=(self.Co, 1),
                ldc=self.Co, bias=p.w(self.bname, t0, weights), activation=self.act, batch=1,
                b_batch_stride=wstride, c_batch_stride=M * self.Co, bias_batch_stride=bstride,
                workspace=ctx.ws.splitk, n_fold=self.Co)
        return y, T, (M, self.Co, self.K, x.data, p.w(self.kname, t0, weights), y), dict(
            a_tabs=(rb, ko), a_u8=x.u8, a_div=x.div, a_vec_along_k=1, a_tab_vec_ok=int(self.C % 4 == 0),
            bias=p.w(self.bname, t0, weights), activation=self.act, batch=T,
            a_batch_stride=x.tower_stride(), b_batch_stride=wstride,
            c_batch_stride=M * self.Co, bias_batch_stride=bstride,
            workspace=ctx.ws.splitk)

    def _dx_tables(self, ctx, B):
        key = ("convdxtab", self.name, B)
        if key not in ctx.cache:
            import ctypes
            n = ctypes.c_longlong()
            ctx.lib.conv_input_grad_tables_ints(B, self.H, self.W, self.C, self.KH, self.KW, self.S, self.Co,
                                                ctypes.byref(n))
            tab = torch.empty(n.value, dtype=torch.int32, device=ctx.device)
            ctx.lib.conv_input_grad_tables(tab, B, self.H, self.W, self.C, self.KH, self.KW, self.S, self.Co, ctx.stream)
            ctx.cache[key] = tab
        return ctx.cache[key]

    def direct_input_grad(self):
        """the input gradient as ONE product gathering dY (rlx_conv_input_grad) instead of the column matrix
        dcol = dz W^T + col2im: needs the kernel to be a multiple of the stride and 4-aligned channel counts."""
        if not (DIRECT_CONV_INPUT_GRAD and self.KH % self.S == 0 and self.KW % self.S == 0 and self.C % 4 == 0 and
                self.Co % 4 == 0 and (self.KH // self.S) * (self.KW // self.S) * self.Co <= 1024):
            return False
        # The gather multiplies every input position by every tap of its phase, the border ones by zeros: worth it while
        # that is < 1.3x the products of the column-matrix form (Atari conv2, 4x4 stride 2 on 20x20: 1.23x, -5 us per
        # update; conv3, 3x3 stride 1 on 9x9 -> 7x7: 1.65x, +3 us — profiles/r03_ab_gemm_changes.txt)
        s = self.S
        executed = s * s * (-(-self.H // s)) * (-(-self.W // s)) * (self.KH // s) * (self.KW // s)
        return DIRECT_CONV_INPUT_GRAD == "always" or executed <= 1.3 * self.OH * self.OW * self.KH * self.KW

    def backward(self, ctx, x, y, need_dx=True, weights=None, t0=0, nt=None, need_dw=True, overlap=False, dw_later=None,
                 dw_collect=None):
        """dw_later (a list): the weight-gradient product is not launched — its (descriptor, split-K job) is appended for
        Sequential.backward to issue together with the other convolution layers' (rlx_gemm_multi_defer).  dw_collect (a
        list): the same for a product that goes through the LDS-resident kernels — its (rlx_conv_dw_item, job) is appended
        for ONE rlx_conv_dw_multi launch at the end of the backward pass."""
        t0, T = self._range(t0, nt)
        B, p = x.rows, self.params
        overlap = overlap and ctx.overlap
        direct = need_dx and self.direct_input_grad()
        pairing = need_dx and need_dw and not overlap and PAIR_GRADIENT_GEMMS and not direct and dw_later is None
        M = B * self.OH * self.OW
        rb, ko = self._tables(ctx, B)
        dz = y.grad
        if self.act is not None and not y.grad_is_dz:
            ctx.lib.act_backward(dz, y.data, dz.numel(), _rlx.ACT[self.act], ctx.stream)
        dz = dz.view(T, M, self.Co)
        # dW[K,Co] = cols^T dz : A(k, m) gathered with outer table = koff, reduction table = rowbase
        if need_dw:
            fold = x.towers == 0 and T > 1 and self.Co % 4 == 0 and self.C % 4 == 0 and FOLD_SHARED_INPUT
            # the LDS-resident weight-gradient kernels (csrc/conv_dw_u8.hip: the first convolution of an image torso, frame
            # rows + the image's dz in LDS; csrc/conv_dw_f32.hip: an inner convolution on its own — its input gradient went
            # through rlx_conv32_input_grad): one deferred split per image / image pair
            geo = (B, self.H, self.W, self.C, self.KH, self.KW, self.S, self.Co, T)
            # (one tower of 32 filters — the DQN update — has a body of its own: two splits per image, conv_dw_u8_body_half)
            half = T == 1 and x.towers in (0, 1) and self.Co == 32
            lds_u8 = bool(not pairing and not overlap and dw_later is None and (fold or half) and x.u8 and CONV_DW_U8 and
                          ctx.lib.conv_dw_u8_supported(*geo))
            lds_f32 = bool(not pairing and not overlap and dw_later is None and not fold and not x.u8 and CONV_DW_F32 and
                           x.towers == T and ctx.lib.conv_dw_f32_supported(*geo))
            collecting = dw_collect is not None and (lds_u8 or lds_f32)
            take = ctx.reserve_deferred_workspace if dw_later is not None or collecting else ctx.deferred_workspace
            job, dws = take(self.K, T * self.Co if fold else self.Co, 1 if fold else T) if not overlap else (None, None)
            collecting = collecting and job is not None

            def dw(ws, launch=True):
                ws = dws if job is not None else ws
                dj = job if launch else None          # a pair launch passes the job itself
                if fold:
                    return _rlx.gemm(self.K, T * self.Co, M, x.data, dz, p.g(self.kname, t0), a_tabs=(ko, rb),
                                     a_u8=x.u8, a_div=x.div, a_vec_along_k=0, a_tab_vec_ok=1, b_strides=(self.Co, 1),
                                     ldc=self.Co, batch=1, b_batch_stride=M * self.Co,
                                     c_batch_stride=p.stride(self.kname), workspace=ws,
                                     colsum_out=p.g(self.bname, t0), colsum_batch_stride=p.stride(self.bname),
                                     n_fold=self.Co, launch=launch, defer=dj)
                return _rlx.gemm(self.K, self.Co, M, x.data, dz, p.g(self.kname, t0), a_tabs=(ko, rb), a_u8=x.u8,
                                 a_div=x.div, a_vec_along_k=0, a_tab_vec_ok=int(self.C % 4 == 0), batch=T,
                                 a_batch_stride=x.tower_stride(), b_batch_stride=M * self.Co,
                                 c_batch_stride=p.stride(self.kname), workspace=ws,
                                 colsum_out=p.g(self.bname, t0), colsum_batch_stride=p.stride(self.bname),
                                 launch=launch, defer=dj)
            if dw_later is not None and job is not None:
                dw_later.append((dw(None, launch=False), job))
            elif (lds_u8 or lds_f32) and job is not None:
                item = _rlx.conv_dw_item(x.data, 0 if lds_u8 else x.tower_stride(), lds_u8, x.div if lds_u8 else 1.0, dz,
                                         M * self.Co, *geo, p.g(self.kname, t0), p.stride(self.kname), p.g(self.bname, t0),
                                         p.stride(self.bname), dws)
                if collecting:
                    dw_collect.append((item, job))
                else:
                    _rlx.conv_dw_multi([item], [job], ctx.stream)
            elif not pairing:
                dw(ctx.ws.splitk)
        if direct:
            assert x.towers == T
            dx, lower = x.ensure_grad(), x.act
            _rlx.conv_input_grad(dz, p.w(self.kname, t0, weights), dx, x.data if lower else None, lower,
                                 self._dx_tables(ctx, B), B, self.H, self.W, self.C, self.KH, self.KW, self.S, self.Co, T,
                                 M * self.Co, p.stride(self.kname), B * self.H * self.W * self.C, ctx.stream)
            x.grad_is_dz = lower is not None
        elif need_dx:
            assert x.towers == T
            dcol = ctx.buffer(self.name + "/dcol", (T, M, self.K))
            dxd = _rlx.gemm(M, self.K, self.Co, dz, p.w(self.kname, t0, weights), dcol, b_strides=(1, self.Co),
                            batch=T, a_batch_stride=M * self.Co, b_batch_stride=p.stride(self.kname),
                            c_batch_stride=M * self.K, workspace=ctx.ws.splitk, launch=False)
            if pairing:
                # dW and dcol are independent products of the same dz: one launch (own split-K workspaces)
                _rlx.gemm_pair(dw(ctx.ws.splitk_side, launch=False), dxd, defer=job)
            else:
                _rlx.gemm_pair_or_single(dxd)
            dx = x.ensure_grad()
            # the gather also applies the producing layer's activation derivative: dx is its dz
            lower = x.act
            ctx.lib.col2im(dcol, dx, x.data if lower else None, _rlx.ACT[lower], T * B, self.H, self.W,
                           self.C, self.KH, self.KW, self.S, ctx.stream)
            x.grad_is_dz = lower is not None
        if need_dw and not ((dw_later is not None or collecting) and job is not None):
            ctx.commit_deferred(job)


def row_head_problems(ctx, row_heads, y, M, N, tag, weights=None):
    """rlx_gemm_desc.row_heads for [(narrow Dense layer, tower index of the wide layer's output it reads)]: the ctypes
    array and the heads' output Tensors (what small_dense_forward_multi would have returned)."""
    if row_heads and row_heads[0][1] == "pair":
        # [(layer, "pair")]: the wide layer ran the online and the target copy as batch entries 0 and 1 (Dense.forward
        # pair=True); the head does the same — ONE output Tensor with two towers, as its own pair launch returns
        (l, _), = row_heads
        assert l.N <= SMALL_N and l.K == N and l.T == 1
        p = l.params
        hy = ctx.buffer(l.name, (2, M, l.N), tag=tag)
        arr = (_rlx.SmallDenseProblem * 2)()
        for c in range(2):
            q = arr[c]
            q.x, q.x_tower_stride = y.data_ptr() + c * M * N * 4, 0
            q.w, q.w_tower_stride = p.w(l.kname).data_ptr() + c * p.size * 4, p.stride(l.kname)
            q.bias, q.bias_tower_stride = p.w(l.bname).data_ptr() + c * p.size * 4, p.stride(l.bname)
            q.y, q.y_tower_stride = hy.data_ptr() + c * M * l.N * 4, M * l.N
            q.towers, q.M, q.K, q.N, q.activation = 1, M, l.K, l.N, _rlx.ACT[l.act]
        return arr, [Tensor(hy, M, l.N, 2, grad_key=(ctx, l.name, tag), act=l.act)]
    arr = (_rlx.SmallDenseProblem * len(row_heads))()
    outs = []
    for i, item in enumerate(row_heads):
        # (layer, tower) or (layer, tower, out): `out` — a contiguous fp32 tensor of M * N elements — receives the head's
        # output instead of the context's buffer (a step's rows of a rollout column: no copy launch behind the head)
        l, t = item[0], item[1]
        assert l.N <= SMALL_N and l.K == N and l.T == 1
        p = l.params
        hy = item[2] if len(item) > 2 and item[2] is not None else None
        if hy is not None:
            assert hy.is_contiguous() and hy.dtype == torch.float32 and hy.numel() == l.T * M * l.N
            hy = hy.view(l.T, M, l.N)
        else:
            hy = ctx.buffer(l.name, (l.T, M, l.N), tag=tag)
        q = arr[i]
        q.x, q.x_tower_stride = y.data_ptr() + t * M * N * 4, 0
        q.w, q.w_tower_stride = p.w(l.kname, 0, weights).data_ptr(), p.stride(l.kname)
        q.bias, q.bias_tower_stride = p.w(l.bname, 0, weights).data_ptr(), p.stride(l.bname)
        q.y, q.y_tower_stride = hy.data_ptr(), M * l.N
        q.towers, q.M, q.K, q.N, q.activation = l.T, M, l.K, l.N, _rlx.ACT[l.act]
        outs.append(Tensor(hy, M, l.N, l.T, grad_key=(ctx, l.name, tag), act=l.act))
    return arr, outs


def small_dense_forward_multi(ctx, items, tag="", weights=None):
    """[(Dense layer, input Tensor)] -> [output Tensor], all layers in ONE launch (each N <= SMALL_N).
    weights: parameter buffer to read instead of the online weights (the target copy)."""
    import ctypes
    arr = (_rlx.SmallDenseProblem * len(items))()
    outs = []
    for i, (l, x) in enumerate(items):
        assert l.N <= SMALL_N and not x.u8 and x.cols == l.K and x.towers in (0, l.T)
        M, p = x.rows, l.params
        y = ctx.buffer(l.name, (l.T, M, l.N), tag=tag)
        q = arr[i]
        q.x, q.x_tower_stride = x.data.data_ptr(), x.tower_stride()
        q.w, q.w_tower_stride = p.w(l.kname, 0, weights).data_ptr(), p.stride(l.kname)
        q.bias, q.bias_tower_stride = p.w(l.bname, 0, weights).data_ptr(), p.stride(l.bname)
        q.y, q.y_tower_stride = y.data_ptr(), M * l.N
        q.towers, q.M, q.K, q.N, q.activation = l.T, M, l.K, l.N, _rlx.ACT[l.act]
        outs.append(Tensor(y, M, l.N, l.T, grad_key=(ctx, l.name, tag), act=l.act))
    ctx.lib.dense_small_forward_multi(ctypes.byref(arr), len(items), ctx.stream)
    return outs


def _small_backward_problem(q, l, x, y):
    """rlx_small_dense_problem of one narrow layer's backward pass (y.grad = gradient w.r.t. its output)."""
    M, p = x.rows, l.params
    assert l.N <= SMALL_N and M * l.N <= 1024 and x.towers == l.T
    own = l.act if (l.act is not None and not y.grad_is_dz) else None
    lower = x.act
    dx = x.ensure_grad()
    q.x, q.x_tower_stride = x.data.data_ptr(), x.tower_stride()
    q.w, q.w_tower_stride = p.w(l.kname).data_ptr(), p.stride(l.kname)
    q.dy, q.dy_tower_stride = y.grad.data_ptr(), M * l.N
    q.y, q.y_tower_stride = (y.data.data_ptr() if own else None), M * l.N
    q.dw, q.dw_tower_stride = p.g(l.kname).data_ptr(), p.stride(l.kname)
    q.db, q.db_tower_stride = p.g(l.bname).data_ptr(), p.stride(l.bname)
    q.dx, q.dx_tower_stride = dx.data_ptr(), M * l.K
    q.towers, q.M, q.K, q.N = l.T, M, l.K, l.N
    q.activation, q.lower_activation = _rlx.ACT[own], _rlx.ACT[lower]
    x.grad_is_dz = lower is not None


def small_dense_backward_multi(ctx, items):
    """[(Dense layer, input Tensor x, output Tensor y with y.grad set)]: dW, db and dx (with the lower
    layer's activation derivative) of every layer in ONE launch."""
    import ctypes
    arr = (_rlx.SmallDenseProblem * len(items))()
    for i, (l, x, y) in enumerate(items):
        _small_backward_problem(arr[i], l, x, y)
    ctx.lib.dense_small_backward_multi(ctypes.byref(arr), len(items), ctx.stream)


def ppo_heads_loss_backward(ctx, value, policy, value_targets, actions, advantages, old_probs, ld_old, clip_epsilon,
                            clip_scale, beta, scalars, ratio_out, clipped_out, status):
    """value / policy = (Dense head, input Tensor x, output Tensor y): both head losses of discrete Clipped PPO and the
    heads' backward pass in ONE launch (rlx_ppo_heads_loss_backward); y.grad receives dV / dlogits."""
    import ctypes
    arr = (_rlx.SmallDenseProblem * 2)()
    for q, (l, x, y) in zip(arr, (value, policy)):
        y.ensure_grad()
        _small_backward_problem(q, l, x, y)
    B = value[1].rows
    ctx.lib.ppo_heads_loss_backward(ctypes.byref(arr[0]), ctypes.byref(arr[1]), value[2].data, value_targets,
                                    policy[2].data, actions, advantages, old_probs, ld_old, B, clip_epsilon, clip_scale,
                                    beta, 1.0, scalars, ratio_out, clipped_out, status, ctx.stream)


def ppo_fc_rows(ctx, layer, x, v_head, pi_head, value_targets, actions, advantages, old_probs, ld_old, clip_epsilon,
                clip_scale, beta, scalars, ratio_out, clipped_out, status, tag="", tail_now=False):
    """`layer` (the torso's last Dense, two towers: value / policy) on x with everything of the discrete heads that is local
    to a row inside its K-split reduction (rlx_ppo_fc_rows: heads forward, the rows' loss terms, dV / dlogits, dz of the
    layer) -> (mid, v, logits) with their gradients set, or None where the layer's product does not take that reduction
    (nothing launched).  The all-rows part (the heads' dW / db, the loss scalars) is left pending on the context:
    Context.flush_deferred takes it along with the backward pass's reductions; tail_now: launched here (a caller that
    reads the heads' gradients before the torso's backward pass has ended)."""
    p, hp = layer.params, v_head.params
    if not (layer.T == 2 and x.towers == 2 and not x.u8 and x.cols == layer.K and v_head.T == 1 and pi_head.T == 1 and
            v_head.N == 1 and 2 <= pi_head.N <= SMALL_N and v_head.K == layer.N and pi_head.K == layer.N and
            v_head.act is None and pi_head.act is None and x.rows <= 256):
        return None
    M, N, K = x.rows, layer.N, layer.K
    y = ctx.buffer(layer.name, (2, M, N), tag=tag)
    d = _rlx.gemm(M, N, K, x.data, p.w(layer.kname), y, bias=p.w(layer.bname), activation=layer.act, batch=2,
                  a_batch_stride=x.tower_stride(), b_batch_stride=p.stride(layer.kname), c_batch_stride=M * N,
                  bias_batch_stride=p.stride(layer.bname), workspace=ctx.ws.splitk, launch=False)
    if not ctx.lib.ppo_fc_rows_supported(ctypes.byref(d), pi_head.N):
        return None
    mid = Tensor(y, M, N, 2, grad_key=(ctx, layer.name, tag), act=layer.act)
    dz = mid.ensure_grad()
    r = _rlx.PpoRowsDesc()
    outs = []
    for t, (q, l) in enumerate(((r.value_head, v_head), (r.policy_head, pi_head))):
        hy = ctx.buffer(l.name, (1, M, l.N), tag=tag)
        out = Tensor(hy, M, l.N, 1, grad_key=(ctx, l.name, tag), act=None)
        q.x, q.x_tower_stride = y.data_ptr() + t * M * N * 4, 0
        q.w, q.w_tower_stride = hp.w(l.kname).data_ptr(), hp.stride(l.kname)
        q.bias, q.bias_tower_stride = hp.w(l.bname).data_ptr(), hp.stride(l.bname)
        q.y, q.y_tower_stride = hy.data_ptr(), M * l.N
        q.dy, q.dy_tower_stride = out.ensure_grad().data_ptr(), M * l.N
        q.dw, q.dw_tower_stride = hp.g(l.kname).data_ptr(), hp.stride(l.kname)
        q.db, q.db_tower_stride = hp.g(l.bname).data_ptr(), hp.stride(l.bname)
        q.dx, q.dx_tower_stride = dz.data_ptr() + t * M * N * 4, M * N
        q.towers, q.M, q.K, q.N = 1, M, l.K, l.N
        q.activation, q.lower_activation = 0, _rlx.ACT[layer.act]
        outs.append(out)
    r.value_targets, r.actions, r.advantages = _rlx._ptr(value_targets), _rlx._ptr(actions), _rlx._ptr(advantages)
    r.old_probs, r.ld_old, r.clip_scale = _rlx._ptr(old_probs), int(ld_old), _rlx._ptr(clip_scale)
    r.clip_epsilon, r.beta_entropy, r.grad_scale, r.batch = float(clip_epsilon), float(beta), 1.0, M
    r.row_terms = ctx.buffer("ppo_fc_rows/terms", (M, 4), tag=tag).data_ptr()
    r.scalars, r.status = _rlx._ptr(scalars), _rlx._ptr(status)
    r.likelihood_ratio, r.clipped_likelihood_ratio = _rlx._ptr(ratio_out), _rlx._ptr(clipped_out)
    r._keep = (d,)
    run = lambda: ctx.lib.ppo_fc_rows(ctypes.byref(d), ctypes.byref(r), ctx.stream)
    _rlx._record((d,), run)
    run()
    mid.grad_is_dz = layer.act is not None
    ctx.flush_ppo_tail()                          # (a tail still pending from a pass whose backward never ran)
    ctx.ppo_tail = r
    if tail_now:
        ctx.flush_ppo_tail()
    return mid, outs[0], outs[1]


def ppo_fc_heads_supported(ctx, layer, x, v_head, pi_head):
    """Can rlx_ppo_fc_heads run `layer` (the torso's last Dense, two towers) + both discrete heads + losses + the heads'
    backward pass as one launch?"""
    return (isinstance(layer, Dense) and layer.T == 2 and x.towers == 2 and not x.u8 and v_head.T == 1 and pi_head.T == 1 and
            v_head.N == 1 and v_head.K == layer.N and pi_head.K == layer.N and v_head.act is None and pi_head.act is None and
            x.data.data_ptr() % 16 == 0 and layer.params.stride(layer.kname) % 4 == 0 and
            bool(ctx.lib.ppo_fc_heads_supported(x.rows, layer.K, layer.N, pi_head.N)))


def ppo_fc_heads(ctx, layer, x, v_head, pi_head, value_targets, actions, advantages, old_probs, ld_old, clip_epsilon,
                 clip_scale, beta, scalars, ratio_out, clipped_out, status, tag=""):
    """-> (mid, v, logits): the layer's output Tensor with mid.grad = d loss / d pre-activation (grad_is_dz), the heads'
    outputs with their gradients; the heads' weight gradients are in params.grads (rlx_ppo_fc_heads: one launch)."""
    import ctypes
    p, B, N, K = layer.params, x.rows, layer.N, layer.K
    y = ctx.buffer(layer.name, (2, B, N), tag=tag)
    mid = Tensor(y, B, N, 2, grad_key=(ctx, layer.name, tag), act=layer.act)
    dz = mid.ensure_grad()
    v = Tensor(ctx.buffer(v_head.name, (1, B, 1), tag=tag), B, 1, 1, grad_key=(ctx, v_head.name, tag), act=None)
    lg = Tensor(ctx.buffer(pi_head.name, (1, B, pi_head.N), tag=tag), B, pi_head.N, 1, grad_key=(ctx, pi_head.name, tag), act=None)
    key = ("ppo_fc_heads/ws", N)
    ws = ctx.cache.get(key)
    if ws is None:
        floats, words = ctypes.c_longlong(), ctypes.c_longlong()
        ctx.lib.ppo_fc_heads_workspace(N, ctypes.byref(floats), ctypes.byref(words))
        ws = ctx.cache[key] = (torch.zeros(floats.value, dtype=torch.float32, device=ctx.device),
                               torch.zeros(words.value, dtype=torch.int32, device=ctx.device))
    d = _rlx.PpoFcHeadsDesc()
    d.x, d.x_tower_stride = x.data.data_ptr(), B * K
    d.weights, d.weight_tower_stride = p.w(layer.kname).data_ptr(), p.stride(layer.kname)
    d.bias, d.bias_tower_stride = p.w(layer.bname).data_ptr(), p.stride(layer.bname)
    hp = v_head.params
    d.value_w, d.value_b = hp.w(v_head.kname).data_ptr(), hp.w(v_head.bname).data_ptr()
    d.policy_w, d.policy_b = hp.w(pi_head.kname).data_ptr(), hp.w(pi_head.bname).data_ptr()
    d.value_targets, d.advantages, d.old_probs, d.ld_old = value_targets.data_ptr(), advantages.data_ptr(), old_probs.data_ptr(), ld_old
    d.actions = actions.data_ptr()
    d.clip_scale = clip_scale.data_ptr() if clip_scale is not None else None
    d.clip_epsilon, d.beta_entropy, d.grad_scale = float(clip_epsilon), float(beta), 1.0
    d.batch, d.in_features, d.units, d.n_actions, d.activation = B, K, N, pi_head.N, _rlx.ACT[layer.act]
    d.h, d.dz = y.data_ptr(), dz.data_ptr()
    d.values, d.logits = v.data.data_ptr(), lg.data.data_ptr()
    d.dvalues, d.dlogits = v.ensure_grad().data_ptr(), lg.ensure_grad().data_ptr()
    d.d_value_w, d.d_value_b = hp.g(v_head.kname).data_ptr(), hp.g(v_head.bname).data_ptr()
    d.d_policy_w, d.d_policy_b = hp.g(pi_head.kname).data_ptr(), hp.g(pi_head.bname).data_ptr()
    d.scalars = scalars.data_ptr()
    d.likelihood_ratio = ratio_out.data_ptr() if ratio_out is not None else None
    d.clipped_likelihood_ratio = clipped_out.data_ptr() if clipped_out is not None else None
    d.status = status.data_ptr()
    d.workspace, d.workspace_floats, d.tickets = ws[0].data_ptr(), ws[0].numel(), ws[1].data_ptr()
    run = lambda: ctx.lib.ppo_fc_heads(ctypes.byref(d), ctx.stream)
    _rlx._record((), run, flops=2.0 * 2 * B * N * K)        # bench.py's recorder: the two towers' dense products
    run()
    mid.grad_is_dz = True
    return mid, v, lg


class Sequential:
    """A chain of Dense / Conv2d layers replicated over `towers` identical copies."""

    def __init__(self, layers):
        self.layers = layers

    def initialize(self, rng):
        for l in self.layers:
            l.initialize(rng)

    def forward(self, ctx, x, tag="", weights=None, t0=0, nt=None, pair=False, row_heads=None, skip_last=False):
        """row_heads: [(narrow Dense layer, tower index)] reading the LAST layer's output — computed by that layer's
        launch where it can (rlx_gemm_desc.row_heads); returns (acts, head outputs) then.
        skip_last: stop in front of the last layer (its caller runs it fused with what follows: ppo_fc_heads)."""
        acts = [x]
        kw = {"pair": True} if pair else {}
        done = 0                     # layers [0, done) are computed
        for i, l in enumerate(self.layers):
            if i < done:
                continue
            fused = (self._fused_conv_triple(ctx, i, acts[-1], tag, weights, t0, nt, pair) or
                     self._fused_conv_pair(ctx, i, acts[-1], tag, weights, t0, nt, pair))
            if fused is not None:
                acts.extend(fused)
                done = i + len(fused)
                continue
            if skip_last and i == len(self.layers) - 1:
                return acts
            if row_heads and i == len(self.layers) - 1:
                y, heads = l.forward(ctx, acts[-1], tag=tag, weights=weights, t0=t0, nt=nt, row_heads=row_heads, **kw)
                acts.append(y)
                return acts, heads
            acts.append(l.forward(ctx, acts[-1], tag=tag, weights=weights, t0=t0, nt=nt, **kw))
        return acts

    def _fused_conv_pair(self, ctx, i, x, tag, weights, t0, nt, pair):
        """layers i and i + 1 as one launch (rlx_conv23_forward) -> [y_i, y_{i+1}], or None: both convolutions, the
        geometry the kernel is compiled for, fp32 per-tower input, and batch / towers for which the two tiled launches
        would BOTH have run with two (or both with four) wave groups per K slab — the fused kernel then sums in exactly
        their order (FUSE_CONV_PAIR).  Two towers of 63 .. 75 images: two groups (the Clipped-PPO minibatch); one tower of
        64 images or 32 images x (online, target): four (acting; the DQN update)."""
        if not FUSE_CONV_PAIR or i + 1 >= len(self.layers):
            return None
        a, b = self.layers[i], self.layers[i + 1]
        if not (isinstance(a, Conv2d) and isinstance(b, Conv2d)) or x.u8 or a.act != b.act or a.T != b.T:
            return None
        t0_, T = a._range(t0, nt)
        p = a.params
        ws2, bs2, ws3, bs3 = p.stride(a.kname), p.stride(a.bname), p.stride(b.kname), p.stride(b.bname)
        if pair:          # online and target copy of a one-tower network as two towers (Conv2d.forward pair=True)
            if a.T != 1 or weights is not None:
                return None
            T, ws2, bs2, ws3, bs3 = 2, p.size, p.size, p.size, p.size
        if x.towers != T or not ctx.lib.conv23_forward_supported(a.H, a.W, a.C, a.KH, a.S, a.Co, b.KH, b.S, b.Co):
            return None
        B = x.rows
        M2, M3 = B * a.OH * a.OW, B * b.OH * b.OW
        groups = _tiled_wave_groups(M2, a.Co, T)
        if groups not in (2, 4) or _tiled_wave_groups(M3, b.Co, T) != groups:
            return None
        groups = CONV_FORWARD_WAVE_GROUPS or groups
        y2 = ctx.buffer(a.name, (T, M2, a.Co), tag=tag)
        y3 = ctx.buffer(b.name, (T, M3, b.Co), tag=tag)
        stream = ctx.stream
        args = (x.data, x.tower_stride(), p.w(a.kname, t0_, weights), ws2, p.w(a.bname, t0_, weights), bs2,
                p.w(b.kname, t0_, weights), ws3, p.w(b.bname, t0_, weights), bs3, y2, M2 * a.Co, y3, M3 * b.Co, B, T,
                _rlx.ACT[a.act], groups, stream)
        if not _fused_operands_aligned(args):    # (a caller-supplied `weights` view: the tiled launches take anything)
            return None
        run = lambda: ctx.lib.conv23_forward(*args)
        if _rlx.GEMM_HOOK is not None:       # bench.py's recorder: the two products this launch stands for
            d2, d3 = _rlx.GemmDesc(), _rlx.GemmDesc()
            d2.M, d2.N, d2.K, d2.batch = M2, a.Co, a.K, T
            d3.M, d3.N, d3.K, d3.batch = M3, b.Co, b.K, T
            _rlx._record((d2, d3), run)
        run()
        return [Tensor(y2, B, a.OH * a.OW * a.Co, T, grad_key=(ctx, a.name, tag), act=a.act),
                Tensor(y3, B, b.OH * b.OW * b.Co, T, grad_key=(ctx, b.name, tag), act=b.act)]

    def _fused_conv_triple(self, ctx, i, x, tag, weights, t0, nt, pair):
        """layers i, i + 1, i + 2 as one launch (rlx_conv123_forward: the first convolution from the uint8 frames in front of
        the fused pair) -> [y_i, y_{i+1}, y_{i+2}], or None.  On top of _fused_conv_pair's conditions for the last two:
        rlx_gemm_describe must say that the first layer's own launch would be the register-staged tiled kernel summing
        K = 256 in one chain (64 x 64 tiles: the two towers of the PPO update folded into N) or in three 96-long chunks
        on 128 x 32 tiles (one tower of 64 frames, 32 frames x (online, target)) — the two orders the kernel reproduces."""
        if not (FUSE_CONV_PAIR and FUSE_CONV_FIRST) or i + 2 >= len(self.layers):
            return None
        f, a, b = self.layers[i], self.layers[i + 1], self.layers[i + 2]
        if not all(isinstance(l, Conv2d) for l in (f, a, b)) or not x.u8 or not (f.act == a.act == b.act) or \
                not (f.T == a.T == b.T):
            return None
        t0_, T = f._range(t0, nt)
        p = f.params
        if pair:
            if f.T != 1 or weights is not None:
                return None
            T = 2
        stride = (lambda name: p.size) if pair else p.stride
        lib = ctx.lib
        if x.towers not in (0, T) or f.out_hwc != (a.H, a.W, a.C) or a.out_hwc != (b.H, b.W, b.C) or \
                x.data.data_ptr() % 16 or x.tower_stride() % 16 or \
                not lib.conv123_forward_supported(f.H, f.W, f.C, f.KH, f.S, f.Co) or \
                not lib.conv23_forward_supported(a.H, a.W, a.C, a.KH, a.S, a.Co, b.KH, b.S, b.Co):
            return None
        B = x.rows
        M1, M2, M3 = B * f.OH * f.OW, B * a.OH * a.OW, B * b.OH * b.OW
        groups = _tiled_wave_groups(M2, a.Co, T)
        if groups not in (2, 4) or _tiled_wave_groups(M3, b.Co, T) != groups:
            return None
        groups = CONV_FORWARD_WAVE_GROUPS or groups
        y1, T1, gargs, gkw = f.forward_product(ctx, x, tag, weights, t0, nt, pair)
        assert T1 == T
        d1 = _rlx.gemm(*gargs, launch=False, **gkw)
        q = (ctypes.c_int * 8)()
        lib.gemm_describe(ctypes.byref(d1), q)
        fast, bm, bn, kw, splits, kchunk, ring, thin = list(q)
        if not fast or ring or thin or kw != 1:
            return None
        if (bm, bn) not in ((64, 64), (128, 32)):
            return None
        if splits == 1:
            chunks = 1
        elif splits == 3 and kchunk == 96:       # (<= 16 chunks of a 4-aligned N: splitk_reduce4_kernel<4> adds them)
            chunks = 3
        else:
            return None
        chunks = CONV1_CHUNKS or chunks
        y2 = ctx.buffer(a.name, (T, M2, a.Co), tag=tag)
        y3 = ctx.buffer(b.name, (T, M3, b.Co), tag=tag)
        w = lambda l: p.w(l.kname, t0_, weights)
        bi = lambda l: p.w(l.bname, t0_, weights)
        args = (x.data, x.tower_stride(), float(x.div), w(f), stride(f.kname), bi(f), stride(f.bname), y1, M1 * f.Co,
                w(a), stride(a.kname), bi(a), stride(a.bname), w(b), stride(b.kname), bi(b), stride(b.bname),
                y2, M2 * a.Co, y3, M3 * b.Co, B, T, _rlx.ACT[f.act], groups, chunks, ctx.stream)
        if not _fused_operands_aligned(args[3:]):
            return None
        run = lambda: lib.conv123_forward(*args)
        if _rlx.GEMM_HOOK is not None:       # bench.py's recorder: the three products this launch stands for
            d2, d3 = _rlx.GemmDesc(), _rlx.GemmDesc()
            d2.M, d2.N, d2.K, d2.batch = M2, a.Co, a.K, T
            d3.M, d3.N, d3.K, d3.batch = M3, b.Co, b.K, T
            _rlx._record((d1, d2, d3), run)
        run()
        return [Tensor(y, B, l.OH * l.OW * l.Co, T, grad_key=(ctx, l.name, tag), act=l.act)
                for y, l in ((y1, f), (y2, a), (y3, b))]

    def _fused_conv_input_grads(self, ctx, acts, i, weights, t0, nt, kw, need_dx_below):
        """The input gradients of convolution layers i and i - 1 as one launch (rlx_conv32_input_grad: both column matrices
        stay in LDS), then the two weight-gradient products as launches of their own -> True; or False (nothing done):
        the geometry the kernel is compiled for, per-tower fp32 activations, enough half images to fill the chip, and
        rlx_gemm_describe saying that the two dcol products would run on the LDS-DMA ring without a K split — the order of
        sums the kernel reproduces (FUSE_CONV_INPUT_GRADS)."""
        b, a = self.layers[i], self.layers[i - 1]
        if not FUSE_CONV_INPUT_GRADS or not (isinstance(a, Conv2d) and isinstance(b, Conv2d)) or ctx.overlap or \
                not need_dx_below or a.T != b.T or a.direct_input_grad() or b.direct_input_grad():
            return False
        y3, x2, x1 = acts[i + 1], acts[i], acts[i - 1]
        t0_, T = b._range(t0, nt)
        B, p, lib = x2.rows, b.params, ctx.lib
        if x2.towers != T or x1.towers != T or x1.u8 or x2.act != x1.act or a.out_hwc != (b.H, b.W, b.C) or \
                2 * B * T < FUSE_CONV_INPUT_GRADS_MIN_WORKGROUPS or \
                not lib.conv32_input_grad_supported(a.H, a.W, a.C, a.KH, a.S, a.Co, b.KH, b.S, b.Co):
            return False
        M3, M2, M1 = B * b.OH * b.OW, B * a.OH * a.OW, B * a.H * a.W
        dz3 = y3.grad
        dz2, dz1 = x2.ensure_grad(), x1.ensure_grad()
        descs = []
        for l, M, dz in ((b, M3, dz3), (a, M2, dz2)):        # the dcol products Conv2d.backward would launch
            dcol = ctx.buffer(l.name + "/dcol", (T, M, l.K))
            d = _rlx.gemm(M, l.K, l.Co, dz, p.w(l.kname, t0_, weights), dcol, b_strides=(1, l.Co), batch=T,
                          a_batch_stride=M * l.Co, b_batch_stride=p.stride(l.kname), c_batch_stride=M * l.K,
                          workspace=ctx.ws.splitk, launch=False)
            q = (ctypes.c_int * 8)()
            lib.gemm_describe(ctypes.byref(d), q)
            fast, bm, bn, kwg, splits, kchunk, ring, thin = list(q)
            if not (fast and ring and not thin and kwg == 1 and splits == 1):
                return False
            descs.append(d)
        if b.act is not None and not y3.grad_is_dz:
            lib.act_backward(dz3, y3.data, dz3.numel(), _rlx.ACT[b.act], ctx.stream)
            y3.grad_is_dz = True
        args = (dz3, M3 * b.Co, p.w(b.kname, t0_, weights), p.stride(b.kname), x2.data, M2 * a.Co, dz2, M2 * a.Co,
                p.w(a.kname, t0_, weights), p.stride(a.kname), x1.data, M1 * a.C, dz1, M1 * a.C, B, T,
                _rlx.ACT[x2.act], ctx.stream)
        per, ctx.per_tail = ctx.per_tail, None
        if per is not None and 2 * B * T + 1 <= PER_RIDER_MAX_WORKGROUPS:
            # the prioritized replay's priority update as one more workgroup of this launch (a CU it leaves idle takes it)
            pd = _rlx.per_update_desc(per)
            run = lambda: lib.conv32_input_grad_per_update(*args[:-1], ctypes.byref(pd), args[-1])
        else:
            ctx.per_tail = per
            run = lambda: lib.conv32_input_grad(*args)
        _rlx._record(tuple(descs), run)
        run()
        x2.grad_is_dz, x1.grad_is_dz = x2.act is not None, x1.act is not None
        if kw.get("need_dw", True):
            for l, x, y in ((b, x2, y3), (a, x1, x2)):
                l.backward(ctx, x, y, need_dx=False, weights=weights, t0=t0, nt=nt, **kw)
        return True

    def backward(self, ctx, acts, need_input_grad=False, weights=None, t0=0, nt=None, need_dw=True,
                 layers=None):
        """layers=(lo, hi): only layers lo <= i < hi (the data-parallel path runs the last layers
        first, starts their gradient all-reduce, then the rest)."""
        lo, hi = layers if layers is not None else (0, len(self.layers))
        mine = need_dw and ctx.begin_deferring()      # the layers' split-K reductions: one launch at the end
        try:
            # MULTI_DW: the convolution layers' weight gradients wait until the input-gradient chain has passed all of
            # them, then go out as ONE launch (up to three products) instead of one per layer inside its dW + dX pair
            later = [] if (MULTI_DW and need_dw and ctx.deferred is not None and not ctx.overlap and
                           sum(isinstance(self.layers[i], Conv2d) for i in range(lo, hi)) >= 2) else None
            # CONV_DW_ONE_LAUNCH: the LDS-resident weight-gradient products of the convolution layers go out as ONE launch
            # (rlx_conv_dw_multi) once the input-gradient chain has passed all of them
            collect = [] if (CONV_DW_ONE_LAUNCH and need_dw and later is None and ctx.deferred is not None and
                             not ctx.overlap) else None
            i = hi
            while i > lo:
                i -= 1
                kw = {"overlap": True} if need_dw else {"need_dw": False}
                if collect is not None and isinstance(self.layers[i], Conv2d):
                    kw["dw_collect"] = collect
                if i - 1 >= lo and later is None and \
                        self._fused_conv_input_grads(ctx, acts, i, weights, t0, nt, kw, i - 1 > 0 or need_input_grad):
                    i -= 1               # layers i and i - 1 are done
                    continue
                if later is not None and isinstance(self.layers[i], Conv2d):
                    kw["dw_later"] = later
                self.layers[i].backward(ctx, acts[i], acts[i + 1], need_dx=(i > 0 or need_input_grad),
                                        weights=weights, t0=t0, nt=nt, **kw)
            while collect:
                # longest workgroups first: the uint8 layer, then the fp32 layers by reduction length
                group = sorted(collect[:3], key=lambda e: (not e[0].x_is_u8, -e[0].H))
                collect = collect[3:]
                _rlx.conv_dw_multi([it for it, _ in group], [j for _, j in group], ctx.stream)
                for _, j in group:
                    ctx.commit_deferred(j, reserved=True)
            while later:
                group, later = later[:3], later[3:]
                _rlx.gemm_multi([d for d, _ in group], [j for _, j in group])
                for _, j in group:
                    ctx.commit_deferred(j, reserved=True)
            ctx.join()                   # weight gradients issued on the side stream are complete
        finally:
            if mine:
                ctx.flush_deferred()


class AdamState:
    """tf.train.AdamOptimizer slots for one flat parameter buffer (general_network.py:390-394)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.99, epsilon=1e-4):
        dev = params.weights.device
        self.params, self.lr, self.beta1, self.beta2, self.eps = params, lr, beta1, beta2, epsilon
        self.m = torch.empty_like(params.weights)
        self.v = torch.empty_like(params.weights)
        self.state = torch.empty(2, dtype=torch.float32, device=dev)
        self.ticket = torch.zeros(_rlx.ADAM_TICKET_WORDS, dtype=torch.int32, device=dev)   # rlx_adam_tf1_step's last-arriver count
        self.one_launch = True
        _rlx.lib().adam_init(self.m, self.v, params.size, self.state, beta1, beta2, _rlx.current_stream())

    def step(self, grad_scale=1.0, lr=None, norm_out=None, workspace=None, acc=None, grads=None, mix_target=None,
             mix_rate=0.0):
        """norm_out given: the same pass also returns tf.global_norm of the gradients; acc =
        (src, dst, n): the finish adds src[:n] onto dst[:n] (signal accumulation); grads: a
        flat gradient buffer to consume instead of params.grads; mix_target: the target copy of the weights, updated
        as rate * w_new + (1 - rate) * target in the same pass (a soft target update due right after this step)."""
        p = self.params
        lr = self.lr if lr is None else lr
        g = p.grads if grads is None else grads
        if self.one_launch:
            _rlx.lib().adam_tf1_step(p.weights, g, self.m, self.v, p.size, lr, self.beta1, self.beta2, self.eps,
                                     self.state, grad_scale, norm_out, workspace if norm_out is not None else None,
                                     workspace.numel() if norm_out is not None else 0,
                                     acc[0] if acc else None, acc[1] if acc else None, int(acc[2]) if acc else 0,
                                     mix_target, float(mix_rate), self.ticket, _rlx.current_stream())
            return
        if mix_target is not None:
            raise ValueError("the two-launch Adam path does not fuse the target update")
        if norm_out is not None:
            _rlx.lib().adam_tf1_norm(p.weights, g, self.m, self.v, p.size, lr, self.beta1,
                                     self.beta2, self.eps, self.state, grad_scale, norm_out, workspace,
                                     workspace.numel(), acc[0] if acc else None, acc[1] if acc else None,
                                     int(acc[2]) if acc else 0, _rlx.current_stream())
        else:
            _rlx.lib().adam_tf1(p.weights, g, self.m, self.v, p.size, lr, self.beta1, self.beta2,
                                self.eps, self.state, grad_scale, _rlx.current_stream())
