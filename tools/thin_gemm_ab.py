"""A/B of the thin GEMM kernels on the MLP layer shapes of C4 (TD3, B = 100, 400-300) and C5 (SAC, B = 256, 256-256):
forward (X W + b, relu), input gradient (dY W^T), weight gradient (X^T dY) and the dW / dX pair launch.

    python tools/thin_gemm_ab.py
(the prefetch-all variant this was written to compare — commit d1470f9, RLX_THIN_NO_PREFETCH_ALL / RLX_THIN_DBG — measured no
faster and was removed; the record is profiles/r02_ab_thin_prefetch_all.txt)

Prints one JSON line per shape: average device time per launch over a captured graph of 200 back-to-back launches
(launch overhead amortised the way the update graphs amortise it) and a checksum of the output bits — the two variants
must print identical checksums (same reduction order)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coach_amd import _rlx  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    t = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(dev)
    cases = []
    for tag, B, dims in (("c4", 100, (23, 400, 300)), ("c4a", 100, (17, 400, 300)), ("c5", 256, (376, 256, 256))):
        for i in range(2):
            K, N = dims[i], dims[i + 1]
            X, W, b, dY = t(B, K), t(K, N), t(N), t(B, N)
            Y, dX, dW, db = torch.zeros(B, N, device=dev), torch.zeros(B, K, device=dev), torch.zeros(K, N, device=dev), \
                torch.zeros(N, device=dev)
            cases.append(("%s fwd %dx%dx%d" % (tag, B, N, K), Y,
                          lambda X=X, W=W, b=b, Y=Y, B=B, N=N, K=K: _rlx.gemm(B, N, K, X, W, Y, bias=b, activation="relu")))
            cases.append(("%s dX  %dx%dx%d" % (tag, B, K, N), dX,
                          lambda dY=dY, W=W, dX=dX, B=B, N=N, K=K: _rlx.gemm(B, K, N, dY, W, dX, b_strides=(1, N))))
            cases.append(("%s dW  %dx%dx%d" % (tag, K, N, B), dW,
                          lambda X=X, dY=dY, dW=dW, db=db, B=B, N=N, K=K: _rlx.gemm(K, N, B, X, dY, dW, a_strides=(1, K),
                                                                                   colsum_out=db)))

            def pair(X=X, dY=dY, dW=dW, db=db, W=W, dX=dX, B=B, N=N, K=K):
                d1 = _rlx.gemm(K, N, B, X, dY, dW, a_strides=(1, K), colsum_out=db, launch=False)
                d2 = _rlx.gemm(B, K, N, dY, W, dX, b_strides=(1, N), launch=False)
                _rlx.gemm_pair(d1, d2)
            cases.append(("%s pair dW+dX K=%d N=%d" % (tag, K, N), dW, pair))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for name, out, fn in cases:
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(200):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000.0 / 1000
            bits = out.cpu().numpy().view(np.uint32)
            print(json.dumps({"case": name, "us_per_launch": round(us, 2),
                              "checksum": int(np.bitwise_xor.reduce(bits.ravel())) ^ int(bits.astype(np.uint64).sum() & 0xffffffff)}))


if __name__ == "__main__":
    main()
