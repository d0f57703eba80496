// rlx::libm_pow — x**y rounded exactly like the host libm's pow(), on the device.
//
// Why it exists: the reference computes prioritized-replay leaves with CPython's `priority ** alpha`
// and the importance weights with `(N * P) ** -beta`
// (rl_coach/memories/non_episodic/prioritized_experience_replay.py:197-198,236,248), i.e. with libm's
// pow().  glibc's pow (>= 2.28; sysdeps/ieee754/dbl-64/e_pow.c, from ARM optimized-routines) is
// accurate to ~0.52 ULP but not correctly rounded, and ROCm's device pow rounds differently again, so
// one leaf in a few hundred would differ in the last bit — and with it, eventually, a sampled index.
// This header evaluates the SAME algorithm on the SAME tables, in the operation order of the x86-64 FMA
// build of glibc (the variant every FMA-capable CPU dispatches to; the fused operations below are the
// ones that build executes fused), so the device trees are bit-identical to the reference's.
//
// Domain: x positive and normal, 2^-65 <= |y| < 2^63, |y*log(x)| < 512 — the whole range a priority
// or an importance weight can take.  Outside it `*inexact_domain` is set and the ordinary pow() is
// returned (the caller reports it through its status word).
//
// Compiles as HIP device code and as plain host C++ (tests/test_libm_pow.py builds the host flavour with
// g++ and compares it with math.pow on millions of inputs; the -m gpu flavour does the same through the
// kernels).  Must be compiled with -ffp-contract=off: every fusion is explicit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define LIBM_POW_FN __device__ inline
#define LIBM_POW_CONST static __device__ const
#else
#define LIBM_POW_FN inline
#define LIBM_POW_CONST static const
#endif

namespace rlx {
namespace libm_detail {
#include "libm_pow_tables.inc"

LIBM_POW_FN uint64_t bits(double v) {
    uint64_t u;
    memcpy(&u, &v, sizeof u);
    return u;
}
LIBM_POW_FN double from_bits(uint64_t u) {
    double v;
    memcpy(&v, &u, sizeof v);
    return v;
}
}  // namespace libm_detail

// log_tab: kLogTab as [128 * 3] doubles, exp_tab: kExpTab [256] — the tables themselves, or a copy the caller keeps closer
// (rlx_per_sample: in LDS, filled in the round trip that fetches the tree's top levels: each pow is otherwise two more
// dependent loads on the kernel's critical path)
LIBM_POW_FN double libm_pow_tabs(double x, double y, int *inexact_domain, const double *log_tab,
                                 const unsigned long long *exp_tab) {
    using namespace libm_detail;
    const uint64_t ix = bits(x), iy = bits(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
    if (topx - 1u > 0x7fdu || ((topy & 0x7ffu) - 0x3beu) > 0x7fu) {
        *inexact_domain = 1;
        return pow(x, y);
    }
    // ---- log(x) = hi + tail, relative error ~2^-68 (log_inline)
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int64_t k = (int64_t)tmp >> 52;
    const double z = from_bits(ix - (tmp & (0xfffull << 52)));
    const double kd = (double)(int32_t)k;
    const double invc = log_tab[3 * i], logc = log_tab[3 * i + 1], logctail = log_tab[3 * i + 2];
    const double r = fma(z, invc, -1.0);
    const double t1 = fma(kd, kLn2Hi, logc);
    const double t2 = t1 + r;
    const double lo1 = fma(kd, kLn2Lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = kLogPoly[0] * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double q56 = fma(r, kLogPoly[6], kLogPoly[5]);
    const double q34 = fma(r, kLogPoly[4], kLogPoly[3]);
    const double q12 = fma(r, kLogPoly[2], kLogPoly[1]);
    double p = fma(q56, ar2, q34);
    p = fma(ar2, p, q12);
    double lo = ((lo1 + lo2) + lo3) + lo4;
    lo = fma(ar3, p, lo);
    const double lhi = hi + lo;
    const double ltail = (hi - lhi) + lo;
    // ---- y * log(x) = ehi + elo
    const double ehi = y * lhi;
    const double elo = fma(y, ltail, fma(lhi, y, -ehi));
    // ---- exp(ehi + elo) (exp_inline, sign_bias = 0 because x > 0)
    const uint32_t abstop = (uint32_t)(bits(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + ehi;      // |y log x| < 2^-54
        *inexact_domain = 1;                                      // overflow / underflow range
        return pow(x, y);
    }
    const double zz = fma(ehi, kInvLn2N, kShift);
    const uint64_t ki = bits(zz);
    const double kk = zz - kShift;
    double rr = fma(kk, kNegLn2HiN, ehi);
    rr = fma(kk, kNegLn2LoN, rr);
    rr = elo + rr;
    const int idx = 2 * (int)(ki & 127);
    const double tail = from_bits(exp_tab[idx]);
    const uint64_t sbits = exp_tab[idx + 1] + (ki << 45);
    const double r2 = rr * rr;
    const double p23 = fma(rr, kExpPoly[1], kExpPoly[0]);
    const double s = rr + tail;
    const double p45 = fma(rr, kExpPoly[3], kExpPoly[2]);
    const double t = fma(p23, r2, s);
    const double r4 = r2 * r2;
    const double tm = fma(p45, r4, t);
    const double scale = from_bits(sbits);
    return fma(tm, scale, scale);
}

LIBM_POW_FN double libm_pow(double x, double y, int *inexact_domain) {
    return libm_pow_tabs(x, y, inexact_domain, &libm_detail::kLogTab[0][0], libm_detail::kExpTab);
}
constexpr int kLibmPowLogDoubles = 128 * 3, kLibmPowExpWords = 256;

}  // namespace rlx
