"""Same-box A/B of two BUILDS of the library on the C2 minibatch update: each build runs bench.py's agent in its own
process (a process loads one librlx), alternately, `rounds` times; us per update = hipGraph replay of 10 epochs x 32
minibatches / 320, the minimum over the rounds.

    bash tools/ab_lib.sh 2 && python tools/ab_c2_libs.py coach_amd/librlx.so coach_amd/ab/librlx_depth2.so [rounds]
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
import torch
sys.path.insert(0, os.path.join(%r, "tools"))
import ab_c2
agent = ab_c2.build(False, True)
ts = [ab_c2.train_ms(agent) for _ in range(6)]
print(json.dumps({"us_per_update": [round(1e3 * t / 320, 1) for t in ts[2:]]}))
''' % (ROOT, ROOT)


def main():
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    res = {p: [] for p in libs}
    for _ in range(rounds):
        for p in libs:
            out = subprocess.run([sys.executable, "-c", CHILD, p], capture_output=True, text=True, cwd=ROOT)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(out.stdout[-2000:], out.stderr[-2000:])
                raise SystemExit(1)
            res[p] += json.loads(line[-1])["us_per_update"]
    print(json.dumps({os.path.relpath(p, ROOT): {"us_per_update": min(v), "all": v} for p, v in res.items()}))


if __name__ == "__main__":
    main()
