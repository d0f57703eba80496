"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of tools/ppo_update_once.py (two separate passes) -> the
per-update HBM traffic of the GEMM family (profiles/r0N_pmc_gemm_traffic.json, read by bench.py `roofline.traffic`).
The LAST eager minibatch update of the run is summarised: every dispatch after the previous update's
Adam launch.  FETCH_SIZE is doubled (gfx950 counts 128-byte requests in 64-byte units,
MI355X_MICROARCH.md); both counters are in KB.

    python tools/pmc_summary.py gpurun_out/r02_final/pmc_FETCH_SIZE.csv gpurun_out/r02_final/pmc_WRITE_SIZE.csv out.json
"""
import csv
import json
import sys


def last_update(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    # an update ends with its Adam launch: adam_finish_norm_kernel (two-launch form) or, since the norm is finished inside
    # the Adam launch, the adam_step*_kernel itself
    last_of = lambda key: [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
    ends = last_of("adam_finish_norm_kernel") or last_of("adam_step")
    return rows[ends[-2] + 1:ends[-1] + 1]


def mfma_table(busy_csv, active_csv, out):
    """rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES pass (the dispatch records carry start / end timestamps)
    -> per kernel of the last update: MFMA busy cycles (summed over the chip's 1024 SIMDs; = MFMA instructions x their
    64 / 16 ... cycles) / (dispatch duration x 2.4 GHz x 1024 SIMDs) = the fraction of the matrix pipes' cycles spent
    on MFMAs while the kernel ran.  2.4 GHz is the top of the clock range, so the figure is a lower bound where the clock
    sat lower.  (GRBM_GUI_ACTIVE, the denominator of the gfx94x MfmaUtil formula, is collected too and kept in the rows:
    on gfx950 / ROCm 7.2 it does not scale with the dispatch duration — 260 k 'cycles' for a 7 us fill kernel — so it is
    not used.)  Durations are those of THIS pass: a counter-collection run serialises dispatches, kernels run cold."""
    b, a = last_update(busy_csv), last_update(active_csv)
    assert [r["Kernel_Name"] for r in b] == [r["Kernel_Name"] for r in a]
    rows, tb, tt = [], 0.0, 0.0
    for x, y in zip(b, a):
        busy, act = float(x["Counter_Value"]), float(y["Counter_Value"])
        dur_ns = float(x["End_Timestamp"]) - float(x["Start_Timestamp"])
        rows.append({"name": x["Kernel_Name"][:110], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                     "duration_us": round(dur_ns / 1e3, 2),
                     "mfma_util": round(busy / (dur_ns * 2.4 * 1024.0), 4) if dur_ns > 0 else None})
        if any(k in x["Kernel_Name"] for k in ("gemm_", "splitk_reduce", "conv23_", "conv32_", "conv_dw_")):
            tb += busy
            tt += dur_ns
    res = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES (dispatch timestamps of the same pass) on "
                     "tools/ppo_update_once.py, last eager minibatch update",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration [ns] * 2.4 cycles/ns * 256 CUs * 4 SIMDs)",
           "gemm_family_mfma_util": round(tb / (tt * 2.4 * 1024.0), 4) if tt else None,
           "gemm_family_us_in_this_pass": round(tt / 1e3, 1), "kernels": rows}
    json.dump(res, open(out, "w"), indent=1)
    print({k: v for k, v in res.items() if k != "kernels"})
    for r in rows:
        print("%-100s %s" % (r["name"][-100:], r["mfma_util"]))


def main():
    if sys.argv[1] == "--mfma":
        return mfma_table(*sys.argv[2:5])
    fetch, write, out = sys.argv[1:4]
    f, w = last_update(fetch), last_update(write)
    assert [r["Kernel_Name"] for r in f] == [r["Kernel_Name"] for r in w]
    gemm = lambda n: any(k in n for k in ("gemm_", "conv23_", "conv32_", "conv_dw_"))    # (the fused launches: 2-3 products each)
    red = lambda n: "splitk_reduce" in n
    kernels, fk, wk = [], 0.0, 0.0
    n_gemm = n_red = 0
    for a, b in zip(f, w):
        n = a["Kernel_Name"]
        if not (gemm(n) or red(n)):
            continue
        n_gemm += gemm(n)
        n_red += red(n)
        fv, wv = float(a["Counter_Value"]), float(b["Counter_Value"])
        fk += fv
        wk += wv
        kernels.append({"name": n, "FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": wv})
    def n_products(n):
        if "conv23_" in n:                                   # conv23_forward_kernel<D, S, G, C>: C > 0 = conv1 in front
            import re
            targs = re.search(r"conv23_forward_kernel<([^>]*)>", n).group(1).split(",")
            return 3 if len(targs) >= 4 and int(targs[3]) > 0 else 2
        if "conv_dw_multi" in n:                             # the three convolution weight gradients
            return 3
        return 2 if ("pair" in n or "conv32_" in n) else 1
    products = sum(n_products(k["name"]) for k in kernels if gemm(k["name"]))
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/ppo_update_once.py, "
                     "last eager minibatch update",
           "gemm_launches": n_gemm, "gemm_products": products, "reduce_launches": n_red,
           "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB": wk, "fetch_bytes_corrected": 2 * fk * 1024, "write_bytes": wk * 1024,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); WRITE_SIZE uncalibrated",
           "traffic_bytes_per_update": 2 * fk * 1024 + wk * 1024,
           "traffic_bytes_per_gemm_launch": (2 * fk * 1024 + wk * 1024) / products,
           "all_kernels_of_the_update": [{"name": a["Kernel_Name"][:100], "FETCH_SIZE_KB": float(a["Counter_Value"]),
                                          "WRITE_SIZE_KB": float(b["Counter_Value"])} for a, b in zip(f, w)],
           "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    print({k: v for k, v in res.items() if k not in ("kernels", "all_kernels_of_the_update")})


if __name__ == "__main__":
    main()
