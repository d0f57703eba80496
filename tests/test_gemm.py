"""fp32 MFMA GEMM family (dense NN/TN/NT, implicit-im2col conv, uint8 input, split-K, batched
towers) against fp64 numpy.  Tolerance: fp32 accumulation over K terms -> rtol 2e-5 * sqrt(K)-ish,
stated per test."""
import numpy as np
import pytest

from tests.util import dev_tensor


@pytest.fixture(autouse=True, params=[2, 0], ids=["lds-dma-ring", "register-staged"])
def _gemm_pipeline(request):
    """every test of this file runs with both main loops of the fast tiled kernels (rlx_gemm_pipeline): the LDS-DMA ring
    (mode 2: uint8 operands included; the default, mode 1, sends those to the other loop) and the register-staged two-set
    pipeline."""
    from coach_amd import _rlx
    lib = _rlx.lib()
    lib.gemm_pipeline(request.param)
    yield
    lib.gemm_pipeline(1)


def _tol(K):
    return dict(rtol=2e-5, atol=2e-6 * max(1.0, np.sqrt(K)))


SHAPES = [  # (M, N, K) — the BASELINE layer shapes plus ragged edges
    (64, 512, 3136), (64, 6, 512), (64, 1, 512), (100, 400, 23), (100, 300, 400), (256, 256, 376),
    (256, 34, 256), (32, 2, 512), (1, 1, 1), (33, 65, 31), (129, 33, 130), (25600, 32, 256),
    (5184, 64, 512), (7, 700, 9),
    # mid-sized: too few 64 x 64 tiles for the chip -> 32 x 64 / 32 x 32 tiles with the K slab split over the waves
    (3136, 128, 576), (3001, 68, 132), (6272, 64, 576),
]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nn_bias_act(rlx, dev, M, N, K):
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(np.float32)
    B = (rng.randn(K, N) / np.sqrt(K)).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    for act, f in (("none", lambda x: x), ("relu", lambda x: np.maximum(x, 0)), ("tanh", np.tanh)):
        C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
        gemm(M, N, K, dev_tensor(A, dev), dev_tensor(B, dev), C, bias=dev_tensor(bias, dev),
             activation=act, workspace=ws)
        ref = f(A.astype(np.float64) @ B.astype(np.float64) + bias)
        np.testing.assert_allclose(C.cpu().numpy(), ref, **_tol(K))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 512, 3136), (100, 400, 23), (25600, 32, 256), (33, 65, 31), (5184, 64, 512),
                                   (128, 512, 3136), (60, 132, 3004)])
def test_gemm_weight_and_input_grads(rlx, dev, M, N, K):
    """dW = X^T dY (TN, reduction over the batch M, split-K) and dX = (dY W^T) * act'(X) (NT)."""
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M * 3 + N)
    X = np.tanh(rng.randn(M, K)).astype(np.float32)        # an activation output
    W = (rng.randn(K, N) / np.sqrt(K)).astype(np.float32)
    dY = rng.randn(M, N).astype(np.float32)
    ws = torch.empty(1 << 23, dtype=torch.float32, device=dev)
    Xd, Wd, dYd = dev_tensor(X, dev), dev_tensor(W, dev), dev_tensor(dY, dev)
    dW = torch.empty(K, N, dtype=torch.float32, device=dev)
    # C[K,N] = A[K,M] * B[M,N] with A(k,m) = X[m,k]: row stride 1, k stride K
    gemm(K, N, M, Xd, dYd, dW, a_strides=(1, K), workspace=ws)
    np.testing.assert_allclose(dW.cpu().numpy(), X.astype(np.float64).T @ dY.astype(np.float64),
                               rtol=2e-5, atol=2e-6 * np.sqrt(M) * 4)
    dX = torch.empty(M, K, dtype=torch.float32, device=dev)
    # C[M,K] = A[M,N] * B[N,K] with B(n,k) = W[k,n]: k(red)=n stride 1, n(out)=k stride N
    gemm(M, K, N, dYd, Wd, dX, b_strides=(1, N), deriv_aux=Xd, aux_ld=K, deriv_kind="tanh", workspace=ws)
    ref = (dY.astype(np.float64) @ W.astype(np.float64).T) * (1 - X.astype(np.float64) ** 2)
    np.testing.assert_allclose(dX.cpu().numpy(), ref, **_tol(N))
    # accumulate=True adds onto the previous contents
    gemm(M, K, N, dYd, Wd, dX, b_strides=(1, N), deriv_aux=Xd, aux_ld=K, deriv_kind="tanh",
         workspace=ws, accumulate=True)
    np.testing.assert_allclose(dX.cpu().numpy(), 2 * ref, **_tol(N))
    db = torch.empty(N, dtype=torch.float32, device=dev)
    rlx.colsum(dYd, M, N, N, db, 0, ws, ws.numel(), 0)
    np.testing.assert_allclose(db.cpu().numpy(), dY.astype(np.float64).sum(0), rtol=2e-5,
                               atol=2e-6 * np.sqrt(M) * 4)


def _im2col(x, KH, KW, s):
    B, H, W, C = x.shape
    OH, OW = (H - KH) // s + 1, (W - KW) // s + 1
    cols = np.zeros((B, OH, OW, KH, KW, C), dtype=x.dtype)
    for ky in range(KH):
        for kx in range(KW):
            cols[:, :, :, ky, kx, :] = x[:, ky:ky + s * OH:s, kx:kx + s * OW:s, :]
    return cols.reshape(B * OH * OW, KH * KW * C), OH, OW


CONVS = [  # (B, H, W, C, KH, KW, stride, Cout, u8)
    (8, 84, 84, 4, 8, 8, 4, 32, True),      # Atari conv1 on the uint8 frame stack
    (8, 20, 20, 32, 4, 4, 2, 64, False),    # conv2
    (8, 9, 9, 64, 3, 3, 1, 64, False),      # conv3
    (3, 11, 13, 4, 3, 5, 2, 7, True),       # ragged
    (2, 7, 7, 8, 7, 7, 1, 5, False),        # single output position
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,KH,KW,s,Co,u8", CONVS)
def test_conv_forward_backward_implicit_im2col(rlx, dev, B, H, W, C, KH, KW, s, Co, u8):
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(H * W + C)
    if u8:
        x = rng.randint(0, 256, size=(B, H, W, C)).astype(np.uint8)
        xf = x.astype(np.float32) / np.float32(255.0)
    else:
        x = np.tanh(rng.randn(B, H, W, C)).astype(np.float32)
        xf = x
    K = KH * KW * C
    Wt = (rng.randn(K, Co) / np.sqrt(K)).astype(np.float32)
    bias = rng.randn(Co).astype(np.float32)
    cols, OH, OW = _im2col(xf.astype(np.float64), KH, KW, s)
    M = B * OH * OW
    rowbase = torch.empty(M, dtype=torch.int32, device=dev)
    koff = torch.empty(K, dtype=torch.int32, device=dev)
    rlx.conv_tables(rowbase, koff, B, H, W, C, KH, KW, s, 0)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    xd, Wd = dev_tensor(x, dev), dev_tensor(Wt, dev)
    y = torch.empty(M, Co, dtype=torch.float32, device=dev)
    vec_ok = int(C % 4 == 0)
    gemm(M, Co, K, xd, Wd, y, a_tabs=(rowbase, koff), a_u8=u8, a_div=255.0, a_vec_along_k=1,
         a_tab_vec_ok=vec_ok, bias=dev_tensor(bias, dev), activation="relu", workspace=ws)
    ref_y = np.maximum(cols @ Wt.astype(np.float64) + bias, 0)
    np.testing.assert_allclose(y.cpu().numpy(), ref_y, **_tol(K))
    # weight gradient: dW[K,Co] = cols^T dY, A(k_out, m) gathered: outer tab = koff, red tab = rowbase
    dY = rng.randn(M, Co).astype(np.float32)
    dYd = dev_tensor(dY, dev)
    dW = torch.empty(K, Co, dtype=torch.float32, device=dev)
    gemm(K, Co, M, xd, dYd, dW, a_tabs=(koff, rowbase), a_u8=u8, a_div=255.0, a_vec_along_k=0,
         a_tab_vec_ok=vec_ok, workspace=ws)
    np.testing.assert_allclose(dW.cpu().numpy(), cols.T @ dY.astype(np.float64), rtol=3e-5,
                               atol=4e-6 * np.sqrt(M) * 4)
    if not u8:
        # input gradient: dcol = dY W^T, then col2im gather with the previous layer's tanh'
        dcol = torch.empty(M, K, dtype=torch.float32, device=dev)
        gemm(M, K, Co, dYd, Wd, dcol, b_strides=(1, Co), workspace=ws)
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        rlx.col2im(dcol, dx, xd, 2, B, H, W, C, KH, KW, s, 0)
        dcol_ref = (dY.astype(np.float64) @ Wt.astype(np.float64).T).reshape(B, OH, OW, KH, KW, C)
        dx_ref = np.zeros((B, H, W, C))
        for ky in range(KH):
            for kx in range(KW):
                dx_ref[:, ky:ky + s * OH:s, kx:kx + s * OW:s, :] += dcol_ref[:, :, :, ky, kx, :]
        dx_ref *= 1 - x.astype(np.float64) ** 2
        np.testing.assert_allclose(dx.cpu().numpy(), dx_ref, **_tol(Co * 16))


@pytest.mark.gpu
def test_gemm_matches_torch_fp32_reference(rlx, dev):
    """Same op in plain PyTorch fp32 on the GPU (rocBLAS): agreement at fp32-roundoff level."""
    import torch
    from coach_amd._rlx import gemm
    torch.manual_seed(0)
    A = torch.randn(64, 3136, device=dev)
    B = torch.randn(3136, 512, device=dev) / 56
    C = torch.empty(64, 512, device=dev)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    gemm(64, 512, 3136, A, B, C, workspace=ws)
    torch.testing.assert_close(C, A @ B, rtol=2e-5, atol=2e-4)


@pytest.mark.gpu
def test_gemm_batched_towers(rlx, dev):
    """Two independent towers (Clipped-PPO value / policy copies) in one launch: shared input,
    per-tower weights, biases and outputs addressed through batch strides."""
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(4)
    M, N, K = 64, 512, 3136
    A = rng.randn(M, K).astype(np.float32)
    W = (rng.randn(2, K, N) / 56).astype(np.float32)
    b = rng.randn(2, N).astype(np.float32)
    C = torch.empty(2, M, N, dtype=torch.float32, device=dev)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    gemm(M, N, K, dev_tensor(A, dev), dev_tensor(W, dev), C, bias=dev_tensor(b, dev), activation="tanh",
         batch=2, a_batch_stride=0, b_batch_stride=K * N, c_batch_stride=M * N, bias_batch_stride=N,
         workspace=ws)
    for t in range(2):
        np.testing.assert_allclose(C[t].cpu().numpy(), np.tanh(A.astype(np.float64) @ W[t] + b[t]), **_tol(K))


@pytest.mark.gpu
def test_gemm_rejects_bad_arguments(rlx, dev):
    import torch
    from coach_amd._rlx import RlxError, gemm
    t = torch.zeros(4, 4, device=dev)
    with pytest.raises(RlxError, match="bad shape"):
        gemm(0, 4, 4, t, t, t)
    with pytest.raises(RlxError, match="contiguous"):
        gemm(4, 4, 4, t, t, t, a_strides=(8, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,T", [(256, 64, 32, 2), (100, 23, 20, 2), (4096, 128, 64, 3)])
def test_gemm_n_fold_shared_input(rlx, dev, M, K, N, T):
    """rlx_gemm_desc.n_fold: T towers reading the SAME A as one GEMM over T*N columns (forward with
    bias + activation, and the weight gradient with column sums) — including a shape the fast path
    cannot take (K = 23), which must fall back to the equivalent batched problem."""
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M + K)
    A = rng.randn(M, K).astype(np.float32)
    W = (rng.randn(T, K, N) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(T, N).astype(np.float32)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    Ad, Wd, bd = (torch.from_numpy(x).to(dev) for x in (A, W, b))
    y = torch.empty(T, M, N, device=dev)
    gemm(M, T * N, K, Ad, Wd, y, b_strides=(N, 1), ldc=N, bias=bd, activation="tanh", batch=1,
         b_batch_stride=K * N, c_batch_stride=M * N, bias_batch_stride=N, workspace=ws, n_fold=N)
    ref = np.tanh(np.einsum("mk,tkn->tmn", A.astype(np.float64), W.astype(np.float64)) + b[:, None, :])
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    dz = torch.from_numpy(rng.randn(T, M, N).astype(np.float32)).to(dev)
    dW = torch.empty(T, K, N, device=dev)
    db = torch.empty(T, N, device=dev)
    gemm(K, T * N, M, Ad, dz, dW, a_strides=(1, K), b_strides=(N, 1), ldc=N, batch=1, b_batch_stride=M * N,
         c_batch_stride=K * N, workspace=ws, colsum_out=db, colsum_batch_stride=N, n_fold=N)
    dzn = dz.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(dW.cpu().numpy(), np.einsum("mk,tmn->tkn", A.astype(np.float64), dzn),
                               rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), dzn.sum(1), rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(32, 256, 256), (100, 300, 400), (37, 45, 130), (256, 64, 1000), (5, 3, 1)])
def test_gemm_thin_all_layouts(rlx, dev, M, N, K):
    """MLP-sized problems take the single-launch thin kernel (K split over the waves of a workgroup):
    all four operand layouts, two towers with two-level batch offsets, bias + activation, derivative
    epilogue, accumulate, and the fused column sums of B, against fp64 numpy."""
    import torch
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M * 7 + N + K)
    T = 2
    A = rng.randn(T, M, K).astype(np.float32)
    B = (rng.randn(T, K, N) / np.sqrt(K)).astype(np.float32)
    bias = rng.randn(T, N).astype(np.float32)
    aux = np.tanh(rng.randn(T, M, N)).astype(np.float32)
    ref = np.einsum("tmk,tkn->tmn", A.astype(np.float64), B.astype(np.float64))
    ws = torch.empty(1 << 20, dtype=torch.float32, device=dev)
    for a_t in (False, True):
        for b_t in (False, True):
            Ad = dev_tensor(np.ascontiguousarray(A.transpose(0, 2, 1)) if a_t else A, dev)
            Bd = dev_tensor(np.ascontiguousarray(B.transpose(0, 2, 1)) if b_t else B, dev)
            C = torch.full((T, M, N), 3.0, dtype=torch.float32, device=dev)
            cs = torch.empty(T, N, dtype=torch.float32, device=dev)
            kw = dict(a_strides=(1, M) if a_t else (K, 1), b_strides=(1, K) if b_t else (N, 1), batch=T,
                      a_batch_stride=M * K, b_batch_stride=K * N, c_batch_stride=M * N, workspace=ws)
            gemm(M, N, K, Ad, Bd, C, bias=dev_tensor(bias, dev), bias_batch_stride=N, activation="relu",
                 colsum_out=cs, colsum_batch_stride=N, **kw)
            np.testing.assert_allclose(C.cpu().numpy(), np.maximum(ref + bias[:, None, :], 0), **_tol(K))
            np.testing.assert_allclose(cs.cpu().numpy(), B.astype(np.float64).sum(1), rtol=2e-5,
                                       atol=2e-6 * np.sqrt(K) * 4)
            gemm(M, N, K, Ad, Bd, C, deriv_aux=dev_tensor(aux, dev), aux_ld=N, aux_batch_stride=M * N,
                 deriv_kind="tanh", **kw)
            gemm(M, N, K, Ad, Bd, C, deriv_aux=dev_tensor(aux, dev), aux_ld=N, aux_batch_stride=M * N,
                 deriv_kind="tanh", accumulate=True, **kw)
            np.testing.assert_allclose(C.cpu().numpy(), 2 * ref * (1 - aux.astype(np.float64) ** 2), **_tol(K))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "conv", "mlp"])
def test_weight_and_input_gradient_as_one_launch_equals_two(rlx, dev, kind):
    """rlx_gemm_pair (a layer's dW and dX products in ONE launch) against the two launches it replaces, on the C2
    layer shapes with two towers: weight gradients, bias gradients and input gradients bit for bit.  The layer's
    `overlap=True` path with stream overlap switched off issues exactly the old two-launch sequence."""
    import torch
    from coach_amd.nn import graph as G
    rng = np.random.RandomState(3)
    out = {}
    for paired in (True, False):
        params = G.FlatParams()
        if kind in ("dense", "mlp"):
            B, K, N, T = (64, 3136, 512, 2) if kind == "dense" else (100, 400, 300, 2)   # C2 FC layer / TD3 critic (thin kernel)
            layer = G.Dense(params, "fc", K, N, "relu", T)
            x_np = rng.randn(T, B, K).astype(np.float32) if paired else x_np
        else:
            B, T = 64, 2
            layer = G.Conv2d(params, "c3", (9, 9, 64), 64, 3, 1, "relu", T)
            x_np = np.maximum(rng.randn(T, B, 9 * 9 * 64), 0).astype(np.float32) if paired else x_np
        params.finalize(dev)
        layer.initialize(np.random.RandomState(5))
        ctx = G.Context(dev)
        assert not ctx.overlap
        x = G.Tensor(dev_tensor(x_np, dev), B, x_np.shape[2], T, act="relu")
        y = layer.forward(ctx, x, tag="t")
        if paired:
            dy_np = rng.randn(*y.data.shape).astype(np.float32)
        y.ensure_grad().copy_(dev_tensor(dy_np, dev))
        G.PAIR_GRADIENT_GEMMS = paired
        try:
            layer.backward(ctx, x, y, need_dx=True)
        finally:
            G.PAIR_GRADIENT_GEMMS = True
        torch.cuda.synchronize()
        out[paired] = (params.grads.clone(), x.grad.clone())
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    assert float(out[True][0].abs().sum()) > 0 and float(out[True][1].abs().sum()) > 0


@pytest.mark.gpu
def test_deferred_weight_gradient_reductions_equal_immediate_ones(rlx, dev):
    """Sequential.backward leaves the split-K partials of every layer's weight gradient in an arena and sums them with
    ONE rlx_splitk_reduce_jobs launch at its end; the gradients must equal those of the per-layer reductions (same
    partials; the grouping of the additions differs only for <= 16 splits: tolerance 1e-6 relative)."""
    import torch
    from coach_amd.nn import graph as G
    rng = np.random.RandomState(11)
    B, T = 64, 2
    out, launches = {}, {}
    for deferred in (True, False):
        params = G.FlatParams()
        layers = [G.Conv2d(params, "c2", (20, 20, 32), 64, 4, 2, "tanh", T), G.Conv2d(params, "c3", (9, 9, 64), 64, 3, 1, "tanh", T),
                  G.Dense(params, "fc", 7 * 7 * 64, 512, "tanh", T)]
        params.finalize(dev)
        seq = G.Sequential(layers)
        seq.initialize(np.random.RandomState(5))
        ctx = G.Context(dev)
        if not deferred:
            ctx.begin_deferring = lambda: False
        if deferred:
            x_np = np.tanh(rng.randn(T, B, 20 * 20 * 32)).astype(np.float32)
        x = G.Tensor(dev_tensor(x_np, dev), B, x_np.shape[2], T, act="tanh")
        acts = seq.forward(ctx, x, tag="t")
        if deferred:
            dy_np = rng.randn(*acts[-1].data.shape).astype(np.float32)
        acts[-1].ensure_grad().copy_(dev_tensor(dy_np, dev))
        from coach_amd import _rlx
        before = _rlx.CALL_COUNT
        seq.backward(ctx, acts, need_input_grad=True)
        launches[deferred] = _rlx.CALL_COUNT - before
        torch.cuda.synchronize()
        assert ctx.deferred is None
        out[deferred] = (params.grads.clone().cpu().numpy(), x.grad.clone().cpu().numpy())
    scale = np.abs(out[False][0]).max()
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_array_equal(out[True][1], out[False][1])
    assert np.abs(out[True][0]).sum() > 0
    # host calls: + the one reduce call, - one where the convolution layers' weight gradients go out as one launch
    # (CONV_DW_ONE_LAUNCH needs the deferred arena); device launches: fewer either way
    assert launches[True] <= launches[False] + 1


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,KH,KW,s,Co,T", [
    (64, 9, 9, 64, 3, 3, 1, 64, 2),      # Atari conv3 (C2 minibatch, two towers): 32 x 64 tiles, slab split over the waves
    (64, 20, 20, 32, 4, 4, 2, 64, 2),    # conv2: four phases, N = 32 -> 128 x 32 tiles
    (8, 9, 9, 64, 3, 3, 1, 64, 1),
    (5, 21, 19, 8, 4, 2, 2, 12, 1),      # odd sizes: phases of different extent, rows that do not exist, padding rows
    (3, 12, 12, 4, 6, 6, 3, 20, 2),      # stride 3
    (256, 9, 9, 64, 3, 3, 1, 64, 1),     # a value-pass sized batch: 64 x 64 tiles
])
def test_conv_input_gradient_direct(rlx, dev, B, H, W, C, KH, KW, s, Co, T):
    """rlx_conv_input_grad (the windowed gather of dY against the kernel taps, one product) against the fp64 definition
    dX = col2im(dY W^T) * tanh'(x), per tower."""
    import ctypes
    import torch
    rng = np.random.RandomState(H * W + C + B)
    OH, OW = (H - KH) // s + 1, (W - KW) // s + 1
    K = KH * KW * C
    x = np.tanh(rng.randn(T, B, H, W, C)).astype(np.float32)
    Wt = (rng.randn(T, K, Co) / np.sqrt(K)).astype(np.float32)
    dY = rng.randn(T, B * OH * OW, Co).astype(np.float32)
    n = ctypes.c_longlong()
    rlx.conv_input_grad_tables_ints(B, H, W, C, KH, KW, s, Co, ctypes.byref(n))
    tab = torch.empty(n.value, dtype=torch.int32, device=dev)
    rlx.conv_input_grad_tables(tab, B, H, W, C, KH, KW, s, Co, 0)
    xd, Wd, dYd = dev_tensor(x, dev), dev_tensor(Wt, dev), dev_tensor(dY, dev)
    dx = torch.full((T, B, H, W, C), 9.0, dtype=torch.float32, device=dev)
    rlx.conv_input_grad(dYd, Wd, dx, xd, 2, tab, B, H, W, C, KH, KW, s, Co, T, B * OH * OW * Co, K * Co,
                        B * H * W * C, 0)
    for t in range(T):
        dcol = (dY[t].astype(np.float64) @ Wt[t].astype(np.float64).T).reshape(B, OH, OW, KH, KW, C)
        ref = np.zeros((B, H, W, C))
        for ky in range(KH):
            for kx in range(KW):
                ref[:, ky:ky + s * OH:s, kx:kx + s * OW:s, :] += dcol[:, :, :, ky, kx, :]
        ref *= 1 - x[t].astype(np.float64) ** 2
        np.testing.assert_allclose(dx[t].cpu().numpy(), ref, **_tol(Co * 16))
    # without the derivative operand
    rlx.conv_input_grad(dYd, Wd, dx, None, 0, tab, B, H, W, C, KH, KW, s, Co, T, B * OH * OW * Co, K * Co,
                        B * H * W * C, 0)
    np.testing.assert_allclose(dx[T - 1].cpu().numpy(), ref / (1 - x[T - 1].astype(np.float64) ** 2), **_tol(Co * 16))


@pytest.mark.gpu
def test_conv_layer_backward_direct_equals_column_matrix_path(rlx, dev):
    """Conv2d.backward with the direct input gradient against the dcol + col2im path (weight gradients bit for bit: the
    same launch; input gradients to summation-order tolerance)."""
    import torch
    from coach_amd.nn import graph as G
    rng = np.random.RandomState(2)
    B, T = 64, 2
    out = {}
    for direct in (True, False):
        params = G.FlatParams()
        layer = G.Conv2d(params, "c2", (20, 20, 32), 64, 4, 2, "tanh", T)
        params.finalize(dev)
        layer.initialize(np.random.RandomState(5))
        ctx = G.Context(dev)
        if direct:
            x_np = np.tanh(rng.randn(T, B, 20 * 20 * 32)).astype(np.float32)
        x = G.Tensor(dev_tensor(x_np, dev), B, x_np.shape[2], T, act="tanh")
        y = layer.forward(ctx, x, tag="t")
        if direct:
            dy_np = rng.randn(*y.data.shape).astype(np.float32)
        y.ensure_grad().copy_(dev_tensor(dy_np, dev))
        G.DIRECT_CONV_INPUT_GRAD = "always" if direct else False
        try:
            layer.backward(ctx, x, y, need_dx=True)
        finally:
            G.DIRECT_CONV_INPUT_GRAD = True
        torch.cuda.synchronize()
        out[direct] = (params.grads.clone().cpu().numpy(), x.grad.clone().cpu().numpy())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-6 * np.abs(out[False][0]).max())
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=2e-5, atol=2e-5)
    assert np.abs(out[True][1]).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(3136, 128, 576), (5184, 64, 512), (64, 512, 3136), (3001, 68, 132), (25600, 32, 256)])
def test_tile_order_does_not_change_a_bit(rlx, dev, M, N, K):
    """rlx_gemm_tuning's xcd_mode decides which workgroup computes which tile, nothing else: every mode must give the
    plain order's output bit for bit (a remap that is not a bijection would leave tiles unwritten or written twice) —
    products, the two-problem pair launch and the windowed convolution input gradient, tile counts that are not
    multiples of 8 x the group size included."""
    import torch
    from coach_amd._rlx import gemm, gemm_pair
    rng = np.random.RandomState(M + K)
    A = dev_tensor(rng.randn(M, K).astype(np.float32), dev)
    B = dev_tensor((rng.randn(K, N) / np.sqrt(K)).astype(np.float32), dev)
    dY = dev_tensor(rng.randn(M, N).astype(np.float32), dev)
    ws, ws2 = (torch.empty(1 << 23, dtype=torch.float32, device=dev) for _ in range(2))
    outs = {}
    try:
        for mode in (0, 2, 8, 32, -1):
            rlx.gemm_tuning(192, 192, mode)
            C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
            gemm(M, N, K, A, B, C, workspace=ws)
            dW = torch.full((K, N), 7.0, dtype=torch.float32, device=dev)
            dX = torch.full((M, K), 7.0, dtype=torch.float32, device=dev)
            gemm_pair(gemm(K, N, M, A, dY, dW, a_strides=(1, K), workspace=ws, launch=False),
                      gemm(M, K, N, dY, B, dX, b_strides=(1, N), workspace=ws2, launch=False))
            outs[mode] = [t.cpu().numpy() for t in (C, dW, dX)]
    finally:
        rlx.gemm_tuning(192, 192, -1)
    for mode, got in outs.items():
        for a, b in zip(outs[0], got):
            assert np.array_equal(a, b), mode


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,batch,head_n,act,head_act", [
    (64, 512, 3136, 2, (1, 6), "relu", (None, None)),        # the Clipped-PPO middleware: value and policy head
    (7, 260, 4096, 1, (3,), "tanh", ("tanh",)),              # a partial second pass (65 float4 groups), a partial row count
    (100, 320, 4000, 3, (16, 1), "none", ("relu", None)),    # two row tiles; the third batch entry has no head
    (60, 1024, 4000, 1, (16,), "none", ("relu",)),           # four passes
    (1, 256, 2048, 2, (8, 8), "relu", (None, "tanh")),       # one row, one pass
    (33, 1984, 2100, 1, (5,), "relu", (None,)),              # eight passes (a row of 31 tiles: 17 splits)
    (64, 512, 320, 2, (1, 6), "relu", (None, None))])        # few splits (<= 16): not fused — rlx_gemm launches the heads behind the product
def test_row_heads_in_the_split_k_reduction_equal_the_separate_launches(rlx, dev, M, N, K, batch, head_n, act, head_act):
    """rlx_gemm_desc.row_heads against rlx_gemm followed by rlx_dense_small_forward_multi: the product's output AND the
    head outputs bit for bit, whatever the path rlx_gemm takes (fused reduction, or the heads launched behind it)."""
    import ctypes
    import torch
    from coach_amd import _rlx
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M + N + K + batch)
    A = dev_tensor(rng.randn(batch, M, K).astype(np.float32), dev)
    B = dev_tensor((rng.randn(batch, K, N) / np.sqrt(K)).astype(np.float32), dev)
    bias = dev_tensor(rng.randn(batch, N).astype(np.float32), dev)
    hw = [dev_tensor((rng.randn(N, n) / np.sqrt(N)).astype(np.float32), dev) for n in head_n]
    hb = [dev_tensor(rng.randn(n).astype(np.float32), dev) for n in head_n]
    ws = torch.empty(1 << 23, dtype=torch.float32, device=dev)
    lib, s_ = _rlx.lib(), _rlx.current_stream()

    def problems(C, ys):
        arr = (_rlx.SmallDenseProblem * len(head_n))()
        for i, n in enumerate(head_n):
            q = arr[i]
            q.x, q.x_tower_stride = C.data_ptr() + i * M * N * 4, 0
            q.w, q.w_tower_stride = hw[i].data_ptr(), 0
            q.bias, q.bias_tower_stride = hb[i].data_ptr(), 0
            q.y, q.y_tower_stride = ys[i].data_ptr(), M * n
            q.towers, q.M, q.K, q.N, q.activation = 1, M, N, n, _rlx.ACT[head_act[i]]
        return arr
    kw = dict(bias=bias, activation=act, batch=batch, a_batch_stride=M * K, b_batch_stride=K * N, c_batch_stride=M * N,
              bias_batch_stride=N, workspace=ws)
    C0 = torch.full((batch, M, N), 7.0, dtype=torch.float32, device=dev)
    y0 = [torch.full((M, n), 5.0, dtype=torch.float32, device=dev) for n in head_n]
    gemm(M, N, K, A, B, C0, **kw)
    arr0 = problems(C0, y0)
    lib.dense_small_forward_multi(ctypes.byref(arr0), len(head_n), s_)
    C1 = torch.full((batch, M, N), 7.0, dtype=torch.float32, device=dev)
    y1 = [torch.full((M, n), 5.0, dtype=torch.float32, device=dev) for n in head_n]
    with _rlx.KernelTimer(64) as timer:
        gemm(M, N, K, A, B, C1, row_heads=problems(C1, y1), **kw)
    names = [n for n, _ in timer.records]
    fused = any("splitk_reduce_rows_kernel" in n for n in names)
    assert fused == (K >= 2000), names                       # the cases built to be split > 16 ways take the fused reduction
    assert fused != any("dense_small_fwd" in n for n in names), names
    assert torch.equal(C0, C1)
    for a, b in zip(y0, y1):
        assert torch.equal(a, b)
    ref = np.maximum(C0[0].cpu().numpy().astype(np.float64), -np.inf) @ hw[0].cpu().numpy().astype(np.float64) + hb[0].cpu().numpy()
    if head_act[0] is None:
        np.testing.assert_allclose(y0[0].cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_kernel_timer_overflow_is_an_error(dev):
    """rlx_profile_*: a trace bounded at max_records must not silently drop launches (bench.py sums the records into the
    headline roofline) — the launches beyond the bound still run, and closing the trace fails."""
    import torch
    from coach_amd import _rlx
    lib = _rlx.lib()
    s_ = _rlx.current_stream()
    x = torch.zeros(1024, dtype=torch.float32, device=dev)
    y = torch.ones(1024, dtype=torch.float32, device=dev)
    with _rlx.KernelTimer(2) as timer:
        lib.mix_weights(x, y, 1024, 0.5, s_)
        lib.mix_weights(x, y, 1024, 0.5, s_)
    assert len(timer.records) == 2 and all(us > 0 for _, us in timer.records)
    with pytest.raises(Exception, match="beyond max_records = 1"):
        with _rlx.KernelTimer(1):                    # smaller than the event arrays allocated above: the bound is the request
            lib.mix_weights(x, y, 1024, 0.5, s_)
            lib.mix_weights(x, y, 1024, 0.5, s_)
            lib.mix_weights(x, y, 1024, 0.5, s_)
    torch.cuda.synchronize()
    assert float(x[0]) == 1.0 - 0.5 ** 5              # all five launches ran, timed or not
    with _rlx.KernelTimer(4) as timer:               # and the timer is usable again
        lib.mix_weights(x, y, 1024, 0.5, s_)
    assert len(timer.records) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,bt", [(4096, 4096, 512, False), (16384, 256, 384, False), (4100, 200, 1000, False),
                                      (131072, 64, 512, False), (70000, 36, 96, False), (8192, 640, 256, True),
                                      (33000, 132, 260, False), (16500, 260, 36, True)])
def test_big_wave_tiles_equal_the_64x64_tiling_bit_for_bit(rlx, dev, M, N, K, bt, request):
    """rlx_gemm_big_tiles: products of hundreds of 128-row tiles run with several accumulator tiles per wave (128 x 128 or
    128 x 64 per workgroup).  Every output element is the same chain over K as on 64 x 64 tiles: torch.equal, bias and
    activation included; bt: B given transposed (the input-gradient form dY W^T)."""
    import torch
    from coach_amd import _rlx
    from coach_amd._rlx import gemm
    rng = np.random.RandomState(M % 1000 + N + K)
    A = dev_tensor(rng.randn(M, K).astype(np.float32), dev)
    B = dev_tensor((rng.randn(N, K) if bt else rng.randn(K, N)).astype(np.float32) / np.float32(np.sqrt(K)), dev)
    bias = dev_tensor(rng.randn(N).astype(np.float32), dev)
    ws = torch.empty(1 << 22, dtype=torch.float32, device=dev)
    out, names = {}, {}
    ring = request.node.callspec.params["_gemm_pipeline"] != 0        # (the register-staged loop has no such tiling)
    for big in (0, 2):
        rlx.gemm_big_tiles(big)
        try:
            C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
            with _rlx.KernelTimer(16) as timer:
                gemm(M, N, K, A, B, C, bias=bias, activation="tanh", workspace=ws,
                     **({"b_strides": (1, K)} if bt else {}))
            out[big], names[big] = C, [n for n, _ in timer.records]
        finally:
            rlx.gemm_big_tiles(1)
    out[1], names[1] = out[2], names[2]
    assert torch.equal(out[0], out[1])
    assert not any("big_kernel" in n for n in names[0])
    t128 = -(-M // 128)
    expect_big = (N >= 128 and t128 * -(-N // 128) >= 256) or (N <= 64 and t128 >= 512)       # csrc/gemm.hip kBigMinTiles
    assert any("gemm_dma_big_kernel" in n for n in names[1]) == (expect_big and ring), names[1]
    if M <= 16384:
        ref = np.tanh(A.cpu().numpy().astype(np.float64) @ (B.cpu().numpy().astype(np.float64).T if bt else
                                                           B.cpu().numpy().astype(np.float64)) + bias.cpu().numpy())
        np.testing.assert_allclose(out[1].cpu().numpy(), ref, **_tol(K))


@pytest.mark.gpu
def test_big_wave_tiles_on_a_convolution_of_the_whole_dataset(rlx, dev, request):
    """conv2 of the Atari torso over 2048 images (the V(s) / old-policy passes of Clipped PPO: 165 888 rows through im2col
    tables) on 128 x 64 workgroup tiles against the 64 x 64 tiling: bit-identical."""
    import torch
    from coach_amd import _rlx
    from coach_amd.nn import graph as G
    params = G.FlatParams()
    conv = G.Conv2d(params, "c2", (20, 20, 32), 64, 4, 2, "tanh", 1)
    params.finalize(dev)
    conv.initialize(np.random.RandomState(0))
    B = 2048
    x = torch.from_numpy(np.random.RandomState(1).randn(1, B, 20 * 20 * 32).astype(np.float32)).to(dev)
    out, names = {}, {}
    ring = request.node.callspec.params["_gemm_pipeline"] != 0
    for big in (0, 2):
        rlx.gemm_big_tiles(big)
        try:
            ctx = G.Context(dev)
            conv.forward(ctx, G.Tensor(x, B, 20 * 20 * 32, 1), tag="w")       # (tables)
            with _rlx.KernelTimer(16) as timer:
                y = conv.forward(ctx, G.Tensor(x, B, 20 * 20 * 32, 1), tag="t")
            out[big], names[big] = y.data.clone(), [n for n, _ in timer.records]
        finally:
            rlx.gemm_big_tiles(1)
    assert torch.equal(out[0], out[2])
    assert any("gemm_dma_big_kernel" in n for n in names[2]) == ring, names[2]
    assert not any("big_kernel" in n for n in names[0])
