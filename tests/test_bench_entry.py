"""`python bench.py --gpus N` from a bare shell must launch its own N ranks: the entry
point re-executes itself under torch.distributed.run on 127.0.0.1.  CPU flavour: --dry-run (gloo, no
kernels) exercises the launcher, the rendezvous, the barrier-bracketed timing, max-over-ranks /
sum-over-ranks and the single JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"] + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bare_shell_two_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["allreduce_check"] == 1.0            # sum over 2 ranks / 2
    assert out["value"] > 0 and out["scaling"] == "weak"


def test_single_rank_needs_no_launcher():
    out = _run([])
    assert out["n_gpus"] == 1
