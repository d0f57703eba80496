"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS library's access patterns (MI355X_MICROARCH.md: "FETCH_SIZE
reports 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated:
calibrate on a known byte count in your own access pattern").  Launches whose memory-side reads are known by
construction:

  copy   torch copy of 64 MiB fp32              reads 64 MiB once, writes 64 MiB          (16 B / lane stream)
  g_once rlx_gemm M=64 N=64 K=262144            ONE output tile: A and B are each read exactly once (split-K);
                                                 reads 2 x 64 x K x 4 B = 128 MiB, writes splits x 16 KiB
  g_fc   rlx_gemm M=64 N=512 K=3136 batch 2     the C2 FC forward: B (12.85 MB) once, A (1.6 MB) once per column tile
                                                 that misses in its XCD's L2: between 14.4 and 27.3 MB

    rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -- python tools/pmc_calibrate.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  ... -- python tools/pmc_calibrate.py
    python tools/pmc_calibrate.py --summarise fetch.csv write.csv out.json
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

REPS = 3
G_ONCE = dict(M=64, N=64, K=262144, batch=1)
G_FC = dict(M=64, N=512, K=3136, batch=2)
COPY_BYTES = 64 << 20


def run():
    import ctypes
    import torch
    from coach_amd import _rlx
    lib, dev = _rlx.lib(), torch.device("cuda:0")
    src = torch.randn(COPY_BYTES // 4, device=dev)
    dst = torch.empty_like(src)
    for _ in range(REPS):
        dst.copy_(src)
    for g in (G_ONCE, G_FC):
        M, N, K, b = g["M"], g["N"], g["K"], g["batch"]
        A = torch.randn(b, M, K, device=dev)
        B = torch.randn(b, K, N, device=dev)
        C = torch.empty(b, M, N, device=dev)
        fl = ctypes.c_longlong()
        lib.gemm_workspace_floats(M, N, K, b, ctypes.byref(fl))
        ws = torch.empty(max(int(fl.value), 1), device=dev)
        for _ in range(REPS):
            _rlx.gemm(M, N, K, A, B, C, batch=b, a_batch_stride=M * K, b_batch_stride=K * N, c_batch_stride=M * N,
                      workspace=ws)
    torch.cuda.synchronize()


def summarise(fetch_csv, write_csv, out):
    def rows(path):
        return sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    f, w = rows(fetch_csv), rows(write_csv)
    assert [r["Kernel_Name"] for r in f] == [r["Kernel_Name"] for r in w]
    table = [(r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0, float(x["Counter_Value"]) * 1024.0)
             for r, x in zip(f, w)]
    copies = [t for t in table if "elementwise" in t[0] or "copy" in t[0].lower()][-REPS:]
    gemms = [t for t in table if "gemm_fast_kernel" in t[0]]
    reduces = [t for t in table if "splitk_reduce" in t[0]]
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/pmc_calibrate.py",
           "copy_64MiB": [{"kernel": t[0][:80], "FETCH_SIZE_bytes_raw": t[1], "WRITE_SIZE_bytes": t[2],
                           "fetch_raw_over_known": round(t[1] / COPY_BYTES, 4),
                           "write_over_known": round(t[2] / COPY_BYTES, 4)} for t in copies]}
    once_known = 2 * G_ONCE["M"] * G_ONCE["K"] * 4
    fc_b = G_FC["batch"] * G_FC["K"] * G_FC["N"] * 4
    fc_a = G_FC["batch"] * G_FC["M"] * G_FC["K"] * 4
    res["gemm_one_tile_K262144"] = [{"kernel": t[0][:110], "FETCH_SIZE_bytes_raw": t[1], "WRITE_SIZE_bytes": t[2],
                                     "known_read_bytes": once_known,
                                     "fetch_raw_over_known": round(t[1] / once_known, 4)} for t in gemms[:REPS]]
    res["gemm_fc_forward"] = [{"kernel": t[0][:110], "FETCH_SIZE_bytes_raw": t[1], "WRITE_SIZE_bytes": t[2],
                               "B_bytes": fc_b, "A_bytes": fc_a,
                               "fetch_raw_over_B_plus_A": round(t[1] / (fc_b + fc_a), 4)} for t in gemms[REPS:2 * REPS]]
    res["reduces"] = [{"kernel": t[0][:80], "FETCH_SIZE_bytes_raw": t[1], "WRITE_SIZE_bytes": t[2]} for t in reduces]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(*sys.argv[2:5])
    else:
        run()
