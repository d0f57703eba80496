"""rlx::libm_sin / rlx::libm_cos (coach_amd/csrc/libm_sincos.hpp) — the device restatement of glibc's sin / cos that
lets the device CartPole (csrc/cartpole.hip) follow gym's `math.cos(theta)` / `math.sin(theta)` bit for bit.

CPU: the header compiled as host C++ (g++, -ffp-contract=off) against the C library's sin() / cos() — the functions
CPython's math.sin / math.cos call — on 4 M inputs over the pole-angle range and the whole table domain.
GPU: tests/test_cartpole.py drives the same functions through the kernels against math.sin / math.cos.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "coach_amd", "csrc")

HARNESS = r'''
#include "libm_sincos.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char **argv) {
    long n = atol(argv[1]);
    std::mt19937_64 g(777);
    std::uniform_real_distribution<double> U(0, 1);
    long bad = 0, out = 0;
    for (long i = 0; i < n; ++i) {
        double x;
        switch (i % 4) {
        case 0: x = (U(g) * 2 - 1) * 0.25; break;                       // pole angles (episode ends beyond 0.2095)
        case 1: x = (U(g) * 2 - 1) * 0.855; break;                      // the whole first branch
        case 2: x = (U(g) * 2 - 1) * 0.5 * exp(-U(g) * 30); break;      // tiny angles down to 2^-44
        default: x = 0.126 + (U(g) * 2 - 1) * 1e-3; if (i & 4) x = -x;  // around the Taylor / table switch
        }
        int dom = 0;
        volatile double vx = x;                         // keep the libm calls real calls
        const double s = rlx::libm_sin(x, &dom), c = rlx::libm_cos(x, &dom), rs = sin(vx), rc = cos(vx);
        out += dom;
        if (memcmp(&s, &rs, 8) || memcmp(&c, &rc, 8)) {
            if (bad < 5) printf("MISMATCH x=%a sin %a / %a cos %a / %a\n", x, s, rs, c, rc);
            ++bad;
        }
    }
    int dom = 0;
    const double far = rlx::libm_sin(1.0, &dom);        // outside the domain: flagged, ordinary sin
    printf("n=%ld mismatches=%ld out_of_domain=%ld flagged=%d\n", n, bad, out, dom);
    return bad != 0 || out != 0 || dom != 1 || far != sin(1.0);
}
'''


def test_host_flavour_is_bit_identical_to_libm(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", CSRC, str(src), "-o", str(exe), "-lm"],
                   check=True)
    res = subprocess.run([str(exe), "4000000"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mismatches=0" in res.stdout
