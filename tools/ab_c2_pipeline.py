"""Same-box A/B of (build of the library, rlx_gemm_pipeline mode) pairs on the C2 minibatch update: every configuration
runs bench.py's agent in its own process, alternately, `rounds` times; us per update = hipGraph replay of 10 epochs x
32 minibatches / 320, the minimum over the rounds.
    python tools/ab_c2_pipeline.py rounds lib:mode [lib:mode ...]      e.g.  2 coach_amd/librlx.so:0 coach_amd/ab/librlx_d2.so:1 coach_amd/librlx.so:0:192,200,-1
(the optional third field is rlx_gemm_tuning's kw_below_tiles,kw_min_tiles,xcd_mode[,rlx_gemm_split_cap]; an optional fourth
field is a list of knob=value calls, e.g. lib:1:192,192,-1:adam_norm_in_kernel=0)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, random, sys
sys.path.insert(0, %r)
import coach_amd._rlx as _rlx
_rlx.LIB_PATH = sys.argv[1]
_rlx.lib().gemm_pipeline(int(sys.argv[2]))
import torch
sys.path.insert(0, os.path.join(%r, "tools"))
import ab_c2
tuning = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (192, 192, -1)
if len(tuning) > 3:                       # 4th value: rlx_gemm_split_cap
    _rlx.lib().gemm_split_cap(tuning[3])
    tuning = tuning[:3]
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):      # knob=value: lib.<knob>(int(value))
    k, v = kv.split("=")
    if k.startswith("graph."):               # a module switch of coach_amd.nn.graph, e.g. graph.DIRECT_CONV_INPUT_GRAD=1
        import coach_amd.nn.graph as _G
        setattr(_G, k[6:], {"0": False, "1": True}.get(v, v))
    elif k.startswith("net."):                 # a class switch of the network, e.g. net.HEADS_FORWARD_WITH_TORSO=0
        from coach_amd.nn.networks import ClippedPPONet
        setattr(ClippedPPONet, k[4:], bool(int(v)))
    else:
        getattr(_rlx.lib(), k)(int(v))
agent = ab_c2.build(False, True, tuning=tuning)
ts = [ab_c2.train_ms(agent) for _ in range(6)]
print(json.dumps({"us_per_update": [round(1e3 * t / 320, 1) for t in ts[2:]]}))
''' % (ROOT, ROOT)


def main():
    rounds = int(sys.argv[1])
    cfgs = []
    for a in sys.argv[2:]:                 # lib:mode[:kw_below,kw_min,xcd_mode]
        parts = a.split(":")
        cfgs.append((os.path.abspath(parts[0]), parts[1]) + tuple(parts[2:4]))
    res = {c: [] for c in cfgs}
    for _ in range(rounds):
        for c in cfgs:
            out = subprocess.run([sys.executable, "-c", CHILD] + list(c), capture_output=True, text=True, cwd=ROOT)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:])
                raise SystemExit(1)
            res[c] += json.loads(line[-1])["us_per_update"]
    for c, v in res.items():
        print("%-34s pipeline %s tuning %-12s %-24s: %.1f us per update (all: %s)" % (
            os.path.relpath(c[0], ROOT), c[1], c[2] if len(c) > 2 else "default", c[3] if len(c) > 3 else "", min(v), v))


if __name__ == "__main__":
    main()
