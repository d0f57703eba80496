"""Evidence tables from ONE gpurun call's raw files (kernel traces without counters + rocprofv3 --pmc passes):

  * MFMA utilisation per kernel of a C2 minibatch update: SQ_VALU_MFMA_BUSY_CYCLES (PMC pass; = 64 cycles x the number of
    v_mfma_f32_32x32x2_f32 issued, summed over the chip's 1024 SIMDs) / (the kernel's average duration in the UNPROFILED
    kernel trace x 2.4 GHz x 1024 SIMDs).  Durations under counter collection are ~1.7x longer (the dispatches are
    serialised), so the busy cycles are taken from the PMC pass and the time from the plain trace.
  * GB/s of the streaming kernels: algorithmic bytes (SURVEY.md §8(d) per-unit figures x the units one launch
    processes, stated per row) / the kernel's average duration in the trace, against 8 TB/s.

    python tools/evidence_tables.py gpurun_out/r03_call8 profiles/r03
"""
import csv
import json
import sys


def stats(path):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(path))}


def last_update(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "adam_finish_norm_kernel" in r["Kernel_Name"]]
    return rows[ends[-2] + 1:ends[-1] + 1]


def main(src, dst):
    c2 = stats(src + "/c2_kernel_stats.csv")
    busy = last_update(src + "/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv")
    per_kernel, seen = [], {}
    for r in busy:
        n = r["Kernel_Name"]
        seen.setdefault(n, []).append(float(r["Counter_Value"]))
    tot_busy = tot_time = 0.0
    for n, vals in seen.items():
        if n not in c2:
            continue
        us = c2[n][1]
        b = sum(vals) / len(vals)
        util = b / (us * 1e-6 * 2.4e9 * 1024)
        per_kernel.append({"kernel": n[:120], "launches_per_update": len(vals), "mfma_busy_cycles_per_launch": b,
                           "avg_us_in_the_plain_trace": round(us, 2), "mfma_util": round(util, 4)})
        if "gemm_" in n or "splitk_reduce" in n:
            tot_busy += sum(vals)
            tot_time += us * len(vals)
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES on tools/ppo_update_once.py (busy cycles) + "
                     "rocprofv3 --kernel-trace --stats on bench.py (durations), same gpurun call",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (avg_us * 2.4e9 Hz * 256 CUs * 4 SIMDs)",
           "gemm_family": {"mfma_busy_cycles_per_update": tot_busy, "us_per_update": round(tot_time, 1),
                           "mfma_util": round(tot_busy / (tot_time * 1e-6 * 2.4e9 * 1024), 4)},
           "kernels": per_kernel}
    json.dump(out, open(dst + "_pmc_mfma_util.json", "w"), indent=1)
    print(json.dumps(out["gemm_family"]))
    # ---- streaming kernels
    c3 = stats(src + "/c3_kernel_stats.csv")
    PARAMS_C2, PARAMS_C3 = 2 * 1686086 + 3593, 1685670          # flat parameter buffers (floats, incl. alignment)
    rows = [
        ("adam_step_kernel<true, false, false>", c2, "C2 Adam + gradient norm, 3.37 M parameters", 28.0 * 3.3736e6,
         "28 B / parameter: read g, m, v, w, write m, v, w"),
        ("adam_step_kernel<true, false, false>", c3, "C3 Adam + gradient norm, 1.69 M parameters", 28.0 * 1.6859e6,
         "28 B / parameter"),
        ("img_gather4_kernel", c2, "C2 stacked-state gather (epoch: 2048 states; acting / value pass: 64-256)", None,
         "4 x 7 056 B read + 28 224 B written per state"),
        ("img_gather4_kernel", c3, "C3 minibatch gather: 32 transitions, state + next state", 32 * 91728.0,
         "5 x 7 056 B read + 2 x 28 224 B written per transition (SURVEY 8(d))"),
        ("col2im_kernel<4>", c2, "C2 col2im of conv3 / conv2 (2 towers, B = 64)", None, "see note"),
        ("per_sample_kernel", c3, "C3 stratified sum-tree descent, 32 draws, capacity 2^20", 32 * 168.0,
         "20 levels x 8 B + leaf, per draw: latency-bound (20 dependent loads)"),
        ("per_update_kernel", c3, "C3 priority update, 32 leaves x 3 trees x 20 levels", 32 * 1440.0,
         "1 440 B / priority: latency-bound"),
        ("reverse_scan_kernel", c2, "C2 GAE: 64 sequences x 32 steps (one launch per rollout)", 2048 * 17.0,
         "17 B / step: launch-latency-bound"),
        ("copy_columns_kernel<unsigned int>", c2, "C2 per-epoch column gather (2048 rows x ~44 B)", 2048 * 2 * 44.0,
         "read + write of action, advantage, value target, old probabilities, row index"),
        ("mix_kernel", c2, "C2 target-network copy (rate 1), 3.37 M parameters", 12.0 * 3.3736e6, "12 B / parameter"),
    ]
    table = []
    for key, st, what, nbytes, note in rows:
        hit = [(n, v) for n, v in st.items() if key in n]
        if not hit:
            continue
        n, (calls, us) = hit[0]
        row = {"kernel": key, "what": what, "calls_in_trace": calls, "avg_us": round(us, 2), "bytes_note": note}
        if nbytes is not None:
            gbps = nbytes / (us * 1e-6) / 1e9
            row.update({"algorithmic_bytes_per_launch": nbytes, "GB_per_s": round(gbps, 1),
                        "frac_of_8_TB_per_s": round(gbps / 8000.0, 4)})
        table.append(row)
    json.dump({"source": "rocprofv3 --kernel-trace --stats of bench.py (C2) and bench.py --workload c3, same gpurun call",
               "peak": "8 TB/s HBM3E (spec); 6.3 TB/s is what a float4 copy reaches (MI355X_MICROARCH.md)",
               "rows": table}, open(dst + "_stream_kernels_gbps.json", "w"), indent=1)
    for r in table:
        print("%-36s %8.2f us  %s" % (r["kernel"], r["avg_us"], r.get("GB_per_s", "-")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
