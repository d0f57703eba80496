"""rlx::libm_pow (coach_amd/csrc/libm_pow.hpp) — the device restatement of glibc's pow that makes the
prioritized-replay leaves (`priority ** alpha`) and importance weights (`(N * P) ** -beta`) bit-identical
to the reference's CPython arithmetic (prioritized_experience_replay.py:197-198,236,248).

CPU: the header compiled as host C++ (g++, -ffp-contract=off) against the C library's pow() — i.e. the
function CPython's float.__pow__ calls — on 4 M inputs across the priority / weight ranges.
GPU: the same function as a kernel (rlx_libm_pow) against math.pow on 1 M inputs.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "coach_amd", "csrc")

HARNESS = r'''
#include "libm_pow.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char **argv) {
    long n = atol(argv[1]);
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(0, 1);
    const double ys[] = {0.6, -0.4, 0.5, 1.0, -1.0, 0.7, -0.52, 0.4, 2.0, 0.123456789};
    long bad = 0, out = 0;
    for (long i = 0; i < n; ++i) {
        double x, y;
        switch (i % 4) {
        case 0: x = exp((U(g) * 2 - 1) * 30); y = ys[(i / 4) % 10]; break;      // priorities 1e-13 .. 1e13
        case 1: x = U(g) * 10 + 1e-6; y = 0.6; break;                          // |TD error| + epsilon, alpha .6
        case 2: x = exp((U(g) * 2 - 1) * 300); y = (U(g) * 2 - 1) * 2; break;   // whole double range
        default: x = 1.0 + (U(g) * 2 - 1) * 1e-3; y = (U(g) * 2 - 1) * 40;      // around 1 (tiny log)
        }
        int dom = 0;
        volatile double vx = x, vy = y;                 // keep the libm call a real call
        double mine = rlx::libm_pow(x, y, &dom), ref = pow(vx, vy);
        out += dom;
        if (memcmp(&mine, &ref, 8)) {
            if (bad < 5) printf("MISMATCH x=%a y=%a mine=%a libm=%a\n", x, y, mine, ref);
            ++bad;
        }
    }
    int dom = 0;
    double one = rlx::libm_pow(1.0, 0.6, &dom);         // maximal_priority = 1.0 at the first store
    printf("n=%ld mismatches=%ld out_of_domain=%ld one=%a\n", n, bad, out, one);
    return bad != 0 || one != 1.0 || dom != 0;
}
'''


def test_host_flavour_is_bit_identical_to_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", CSRC, str(src), "-o", str(exe), "-lm"],
                   check=True)
    res = subprocess.run([str(exe), "4000000"], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stdout
    assert "mismatches=0" in res.stdout


def test_tables_match_this_hosts_libm(tmp_path):
    """The committed tables are the ones of the libm this test runs against (regenerating gives the
    same file) — otherwise the bit-exactness claim would be about some other glibc."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import gen_libm_pow_tables as gen
    out = tmp_path / "tables.inc"
    gen.main(str(out))
    assert out.read_text() == open(os.path.join(CSRC, "libm_pow_tables.inc")).read()


@pytest.mark.gpu
def test_device_pow_is_bit_identical_to_host_libm(rlx, dev):
    import torch
    rng = np.random.RandomState(7)
    n = 1 << 20
    x = np.concatenate([np.exp(rng.uniform(-30, 30, n // 2)), rng.uniform(0, 10, n // 4) + 1e-6,
                        1.0 + rng.uniform(-1e-3, 1e-3, n // 4)])
    y = np.concatenate([rng.choice([0.6, -0.4, 0.5, -1.0, 0.7, 0.4, 1.0], n // 2), np.full(n // 4, 0.6),
                        rng.uniform(-40, 40, n // 4)])
    x[:4], y[:4] = [1.0, 1.0, 1e-6, 2.0], [0.6, -0.4, 0.6, 0.5]
    out = torch.empty(n, dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rlx.libm_pow(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), out, n, status, 0)
    ref = np.array([math.pow(a, b) for a, b in zip(x.tolist(), y.tolist())])
    got = out.cpu().numpy()
    assert int(status.item()) == 0
    assert np.array_equal(got.view(np.int64), ref.view(np.int64)), \
        "%d of %d device pow results differ from the host libm" % ((got != ref).sum(), n)
