"""Where the waves of each kernel of one C2 update spend their cycles: rocprofv3 --pmc passes of tools/ppo_update_once.py
with the SQ counters of MI355X_MICROARCH.md's table (WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST_ANY =
issue stall, ACTIVE_INST_ANY = issuing; the three add up to WAVE_CYCLES) -> per kernel of the last eager update.

    python tools/pmc_wave_states.py passA.csv passB.csv out.json
"""
import csv
import json
import sys


def last_update(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: (int(r["Dispatch_Id"]), r["Counter_Name"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    name = {int(r["Dispatch_Id"]): r["Kernel_Name"] for r in rows}
    # an update ends with its Adam launch (adam_finish_norm_kernel in the two-launch form, else the adam_step*_kernel)
    ends = [i for i in ids if "adam_finish_norm_kernel" in name[i]] or [i for i in ids if "adam_step" in name[i]]
    keep = [i for i in ids if ends[-2] < i <= ends[-1]]
    out = {i: {"name": name[i]} for i in keep}
    for r in rows:
        i = int(r["Dispatch_Id"])
        if i in out:
            out[i][r["Counter_Name"]] = out[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [out[i] for i in keep]


def main():
    a, b, dst = sys.argv[1:4]
    ra, rb = last_update(a), last_update(b)
    assert [x["name"] for x in ra] == [x["name"] for x in rb]
    res = []
    for x, y in zip(ra, rb):
        x.update({k: v for k, v in y.items() if k != "name"})
        wc = x.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        row = {"name": x["name"][:120]}
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in x:
                row[k.replace("SQ_", "").lower() + "_frac"] = round(x[k] / wc, 3)
        for k in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVES"):
            if k in x:
                row[k] = x[k]
        if "SQ_LDS_BANK_CONFLICT" in x and x.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac"] = round(x["SQ_LDS_BANK_CONFLICT"] / x["SQ_LDS_IDX_ACTIVE"], 3)
        res.append(row)
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ counters> (two passes) on tools/ppo_update_once.py, last eager "
                         "minibatch update; fractions are of SQ_WAVE_CYCLES", "kernels": res}, open(dst, "w"), indent=1)
    print("%-74s %6s %6s %6s | %6s %7s %7s %6s | %5s" % ("kernel", "parked", "stall", "issue", "ldsStl", "valu/mf", "lds/mf", "vm/mf", "bankC"))
    for r in res:
        mf = r.get("SQ_INSTS_MFMA") or 0.0
        q = lambda k: ("%7.1f" % (r.get(k, 0.0) / mf)) if mf else "      -"
        print("%-74s %6.2f %6.2f %6.2f | %6.2f %s %s %s | %5s" % (
            r["name"][28:102], r.get("wait_any_frac", 0), r.get("wait_inst_any_frac", 0), r.get("active_inst_any_frac", 0),
            r.get("wait_inst_lds_frac", 0), q("SQ_INSTS_VALU"), q("SQ_INSTS_LDS"), q("SQ_INSTS_VMEM"),
            r.get("lds_bank_conflict_frac", "-")))


if __name__ == "__main__":
    main()
