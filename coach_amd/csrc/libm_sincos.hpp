// rlx::libm_sin / rlx::libm_cos — sin(x), cos(x) rounded exactly like the host libm's, on the device.
//
// Why it exists: gym 0.12.5's CartPole-v0 (gym/envs/classic_control/cartpole.py `step`; the simulator behind the
// reference's CartPole presets, rl_coach/environments/gym_environment.py:436,465, requirements.txt:10) computes its
// dynamics with math.cos(theta) / math.sin(theta): libm.  glibc's sin / cos (>= 2.28, sysdeps/ieee754/dbl-64/s_sin.c)
// are accurate to ~0.55 ULP but not correctly rounded, and the device's round differently again — one step in a few
// hundred would differ in the last bit and the trajectories part.  This header evaluates the SAME algorithm on the
// SAME table (libm_sincos_tables.inc, read from the libm by tools/gen_libm_sincos_tables.py) in the operation order of
// the x86-64 FMA build of glibc (the variant every FMA-capable CPU dispatches to; the fused operations below are the
// ones that build executes fused — read from its disassembly), so a device CartPole follows the CPU one bit for bit.
//
// Domain: |x| < 0.855469 (the first branch of __sin / __cos: no range reduction).  A pole angle beyond 12 degrees
// ends the episode, so the physics never leave it; outside, `*inexact_domain` is set and the ordinary sin() / cos() is
// returned (the caller reports it through its status word).
//
// Compiles as HIP device code and as plain host C++ (tests/test_libm_sincos.py builds the host flavour with g++ and
// compares it with the C library on millions of inputs).  Must be compiled with -ffp-contract=off: every fusion is explicit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define LIBM_SINCOS_FN __device__ inline
#define LIBM_SINCOS_CONST static __device__ const
#else
#define LIBM_SINCOS_FN inline
#define LIBM_SINCOS_CONST static const
#endif

namespace rlx {
namespace sincos_detail {
#include "libm_sincos_tables.inc"

LIBM_SINCOS_FN uint64_t bits(double v) {
    uint64_t u;
    memcpy(&u, &v, sizeof u);
    return u;
}
// |x| = xk + r with xk = k / 128 the nearest multiple (u = big + |x| rounds to it; k sits in u's low word)
LIBM_SINCOS_FN int split(double ax, double *r) {
    const double u = k_big + ax;
    *r = ax - (u - k_big);
    return (int)(uint32_t)bits(u);
}
}  // namespace sincos_detail

LIBM_SINCOS_FN double libm_sin(double x, int *inexact_domain) {
    using namespace sincos_detail;
    const uint32_t hx = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (hx < 0x3e500000u) return x;                                   // |x| < 2^-26
    if (hx >= 0x3feb6000u) {
        *inexact_domain = 1;
        return sin(x);
    }
    const double ax = fabs(x);
    if (ax < k_taylor_below) {                                        // TAYLOR_SIN(x*x, x, 0)
        const double xx = x * x;
        double p = fma(xx, k_s5, k_s4);
        p = fma(xx, p, k_s3);
        p = fma(xx, p, k_s2);
        p = fma(xx, p, k_s1);
        const double a = fma(x, p, -0.0);
        const double t = fma(a, xx, 0.0);
        return x + t;
    }
    double r;                                                         // do_sin(x, 0)
    const int k = split(ax, &r);
    const double sn = k_sincostab[4 * k], ssn = k_sincostab[4 * k + 1];
    const double cs = k_sincostab[4 * k + 2], ccs = k_sincostab[4 * k + 3];
    const double dx = x > 0.0 ? 0.0 : -0.0;
    const double xx = r * r;
    const double p = fma(xx, k_sn5, k_sn3);
    const double t = fma(r * xx, p, dx);
    const double s = r + t;
    double q = fma(xx, k_cs6, k_cs4);
    q = fma(xx, q, k_cs2);
    const double c = fma(r, dx, xx * q);
    const double a = fma(s, ccs, ssn);
    const double b = fma(-c, sn, a);
    const double cor = fma(s, cs, b);
    return copysign(sn + cor, x);
}

LIBM_SINCOS_FN double libm_cos(double x, int *inexact_domain) {
    using namespace sincos_detail;
    const uint32_t hx = (uint32_t)(bits(x) >> 32) & 0x7fffffffu;
    if (hx < 0x3e400000u) return 1.0;                                 // |x| < 2^-27
    if (hx >= 0x3feb6000u) {
        *inexact_domain = 1;
        return cos(x);
    }
    double r;                                                         // do_cos(x, 0)
    const int k = split(fabs(x), &r);
    r = r + (x < 0.0 ? -0.0 : 0.0);
    const double sn = k_sincostab[4 * k], ssn = k_sincostab[4 * k + 1];
    const double cs = k_sincostab[4 * k + 2], ccs = k_sincostab[4 * k + 3];
    const double xx = r * r;
    const double p = fma(xx, k_sn5, k_sn3);
    const double s = fma(r * xx, p, r);
    double q = fma(xx, k_cs6, k_cs4);
    q = fma(xx, q, k_cs2);
    const double c = xx * q;
    const double a = fma(-s, ssn, ccs);
    const double b = fma(-c, cs, a);
    const double cor = fma(-s, sn, b);
    return cs + cor;
}

}  // namespace rlx
