"""Per-seed tables of the identical-history experiment (tools/loss_curve_c2.py --follow-hip-actions): when each follower leaves
the history it is given, and where its entropy ends relative to the on-policy run that produced the history.

    python tools/identical_histories_notes.py gpurun_out/lc_forced > profiles/r05_identical_histories_tables.txt
"""
import os
import sys

import numpy as np

FILES = ["hip.npz", "hip_forced.npz", "hip_forced_ulp.npz", "hip_forced_ulpall.npz", "hip_forced_cap16.npz",
         "oracle_forced.npz", "oracle_forced_ulp.npz"]
STEPS_PER_ITERATION, N_ENV, W = 32, 64, 7


def window_means(path, signal):
    r = np.load(path)["results"]
    n = len(r) // W * W
    return r[:n].mean(1)[:, signal].reshape(-1, W).mean(1)


def main(d):
    seeds = sorted(int(x[4:]) for x in os.listdir(d) if x.startswith("seed"))
    print("leader = hip.npz (the device's free run that sampled the action history); every other run trains on that history.")
    print("\n1. first iteration at which > 2 % of a follower's OWN samples differ from the history (-1: never), and the fraction "
          "that differs in its last iteration")
    print("seed | " + " | ".join(f[:-4] for f in FILES[1:]))
    below = total = 0
    for s in seeds:
        out = []
        for f in FILES[1:]:
            p = os.path.join(d, "seed%d" % s, f)
            if not os.path.exists(p):
                out.append("    -    ")
                continue
            z = np.load(p)
            own = z["own_actions"]
            n = len(own) // STEPS_PER_ITERATION * STEPS_PER_ITERATION
            frac = (own[:n] != z["actions"][:n]).reshape(-1, STEPS_PER_ITERATION * N_ENV).mean(1)
            k = np.nonzero(frac > 0.02)[0]
            out.append("%2d (%.2f)%s" % (k[0] if len(k) else -1, frac[-1], "" if len(frac) == 49 else " [%d it]" % len(frac)))
        print("  %d  | " % s + " | ".join(out))
    print("\n2. entropy, mean of the last two complete windows of %d iterations, leader and followers" % W)
    print("seed | " + " | ".join(f[:-4] for f in FILES))
    for s in seeds:
        vals = []
        for f in FILES:
            p = os.path.join(d, "seed%d" % s, f)
            if not os.path.exists(p) or len(np.load(p)["results"]) < 2 * W:
                vals.append("  -  ")
                continue
            vals.append("%.3f" % window_means(p, 1)[-2:].mean())
        print("  %d  | " % s + " | ".join(vals))
        lead = vals[0]
        for v in vals[1:]:
            if v.strip() != "-" and lead.strip() != "-":
                total += 1
                below += float(v) < float(lead)
    print("\nfollowers that end BELOW the leader's entropy: %d of %d" % (below, total))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lc_forced")
