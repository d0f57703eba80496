// Per-sample arithmetic of the head losses (see losses.hip), shared with the fused Clipped-PPO heads kernel
// (ppo_heads_fused.hip): one definition, one rounding.
#pragma once
#include "rlx_common.hpp"

namespace rlx_losses {

constexpr int kMaxBlock = 1024;

__device__ __forceinline__ float block_sum(float v, float *red) {
    const int tid = threadIdx.x, nt = blockDim.x;
    red[tid] = v;
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    float r = red[0];
    __syncthreads();
    return r;
}

// block_sum over the first nt threads of a LARGER block (the tree of a block of nt threads)
__device__ __forceinline__ float block_sum_first(float v, float *red, int nt) {
    const int tid = threadIdx.x;
    if (tid < nt) red[tid] = v;
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// three such sums through ONE tree (each component goes through exactly the additions of block_sum_first)
__device__ __forceinline__ void block_sum3_first(float &x, float &y, float &z, float (*red)[256], int nt) {
    const int tid = threadIdx.x;
    if (tid < nt) { red[0][tid] = x; red[1][tid] = y; red[2][tid] = z; }
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) {
            red[0][tid] += red[0][tid + d];
            red[1][tid] += red[1][tid + d];
            red[2][tid] += red[2][tid + d];
        }
        __syncthreads();
    }
    x = red[0][0]; y = red[1][0]; z = red[2][0];
    __syncthreads();
}

// scalars[0..3] of the discrete Clipped-PPO head from the batch sums (ppo_head.py:62-96)
__device__ __forceinline__ void ppo_discrete_scalars(float sur, float ent, float kl, float beta, int batch,
                                                     float *__restrict__ scalars) {
#pragma clang fp contract(off)
    const float inv = 1.f / (float)batch;
    scalars[0] = -sur * inv;                                      // surrogate_loss (:91)
    scalars[1] = ent * inv;                                       // entropy (:62)
    scalars[2] = kl * inv;                                        // kl_divergence (:66)
    scalars[3] = -sur * inv - beta * ent * inv;                   // + entropy_regularization (:95-96)
}

// One row of a regression head (MSE / Huber, head.py:143-186): loss term and gradient factor of one output
__device__ __forceinline__ void regression_terms(float e, int kind, float &l, float &g) {
#pragma clang fp contract(off)
    if (kind == 0) {
        l = e * e;
        g = 2.f * e;
    } else {
        const float a = fabsf(e);
        l = a <= 1.f ? 0.5f * e * e : a - 0.5f;
        g = fminf(fmaxf(e, -1.f), 1.f);
    }
}
__device__ __forceinline__ float regression_grad(float grad_scale, float w, float g, int batch) {
#pragma clang fp contract(off)
    return grad_scale * w * g / (float)batch;
}

// One sample of the discrete Clipped-PPO head (ppo_head.py:52-116): its surrogate / entropy / KL terms and the
// gradient of the head loss w.r.t. its logits.  Returns false for an out-of-range action.
struct PpoRowTerms { float sur, ent, kl; };
__device__ __forceinline__ bool
ppo_discrete_row(const float *__restrict__ z, const float *__restrict__ po, int a, int n, float adv, float clip_eps,
                 float beta, float grad_scale, int batch, float *__restrict__ dlogits_row,
                 float *__restrict__ ratio_dst, float *__restrict__ clipped_dst, PpoRowTerms &out) {
    // compiled into two kernels that must agree bit for bit (losses.hip, ppo_heads_bwd.hip): with contraction left to
    // the optimiser the two contexts fused different products of this function
#pragma clang fp contract(off)
    float mx = z[0];
    for (int j = 1; j < n; ++j) mx = fmaxf(mx, z[j]);
    float se = 0.f, so = 0.f;
    for (int j = 0; j < n; ++j) {
        se += expf(z[j] - mx);
        so += po[j];
    }
    const float lse = mx + logf(se);          // log-sum-exp of the new logits
    const float lso = logf(so);               // Categorical(probs=p) renormalises: log p - log sum p
    if (a < 0 || a >= n) return false;
    float ent = 0.f, kl = 0.f;
    for (int j = 0; j < n; ++j) {
        const float lp = z[j] - lse;
        const float p = expf(lp);
        const float lpo = logf(po[j]) - lso;
        ent -= p * lp;                                            // distribution.entropy()
        kl += (po[j] / so) * (lpo - lp);                          // kl_divergence(old, new)
    }
    const float logp = z[a] - lse;                                // log_prob(actions)  (:60)
    const float logp_old = logf(po[a]) - lso;                     // (:61)
    const float ratio = expf(logp - logp_old);                    // (:79)
    const float lo = 1.f - clip_eps, hi = 1.f + clip_eps;         // (:83-84)
    const float clipped = fminf(fmaxf(ratio, lo), hi);            // (:85)
    const float s1 = ratio * adv, s2 = clipped * adv;
    out.sur = fminf(s1, s2);                                      // (:86-87)
    out.ent = ent;
    out.kl = kl;
    if (ratio_dst) *ratio_dst = ratio;
    if (clipped_dst) *clipped_dst = clipped;
    if (dlogits_row) {
        // d(-mean min(s1,s2))/d logp : tf.minimum routes the gradient to s1 when s1 <= s2,
        // otherwise to s2, whose clip passes gradient only inside [lo, hi].
        float g_logp;
        if (s1 <= s2)
            g_logp = -adv * ratio;
        else
            g_logp = (ratio >= lo && ratio <= hi) ? -adv * ratio : 0.f;
        g_logp /= (float)batch;
        const float gb = beta / (float)batch;
        for (int j = 0; j < n; ++j) {
            const float lp = z[j] - lse;
            const float p = expf(lp);
            float g = g_logp * ((j == a ? 1.f : 0.f) - p);        // d logp / d z_j
            g += gb * p * (lp + ent);                             // d(-beta*H)/d z_j
            dlogits_row[j] = grad_scale * g;
        }
    }
    return true;
}

// ppo_discrete_row for ONE WAVE: lane j holds action j's terms, so the transcendental chains (expf / logf of every action:
// ~5 N dependent library calls for a single thread, 4-5 us at N = 6) run side by side; every SUM is still taken by adding
// the lanes' terms in ascending j from the same start value, every term is the same expression on the same inputs —
// the results are ppo_discrete_row's bit for bit (tests/test_ppo_fc_rows.py).  All 64 lanes of the wave must call it;
// z / po: the row's logits / old probabilities (n <= 64 values, readable by every lane); outputs as ppo_discrete_row's,
// written by lanes j < n (dlogits_row) and lane 0 (the rest).  Returns false (on every lane) for an out-of-range action.
__device__ __forceinline__ bool
ppo_discrete_row_wave(const float *__restrict__ z, const float *__restrict__ po, int a, int n, float adv, float clip_eps,
                      float beta, float grad_scale, int batch, float *__restrict__ dlogits_row,
                      float *__restrict__ ratio_dst, float *__restrict__ clipped_dst, PpoRowTerms &out) {
#pragma clang fp contract(off)
    const int j = threadIdx.x & 63;
    const bool live = j < n;
    const float zj = live ? z[j] : 0.f, poj = live ? po[j] : 1.f;
    float mx = z[0];
    for (int i = 1; i < n; ++i) mx = fmaxf(mx, z[i]);             // (exact: no rounding in a maximum)
    const float ej = expf(zj - mx);
    float se = 0.f, so = 0.f;
    for (int i = 0; i < n; ++i) {
        se += __shfl(ej, i, 64);
        so += __shfl(poj, i, 64);
    }
    const float lse = mx + logf(se);
    const float lso = logf(so);
    if (a < 0 || a >= n) return false;
    const float lp = zj - lse;
    const float p = expf(lp);
    const float lpo = logf(poj) - lso;
    const float ent_t = p * lp;
    const float kl_t = (poj / so) * (lpo - lp);
    float ent = 0.f, kl = 0.f;
    for (int i = 0; i < n; ++i) {
        ent -= __shfl(ent_t, i, 64);
        kl += __shfl(kl_t, i, 64);
    }
    const float logp = __shfl(lp, a, 64);
    const float logp_old = __shfl(lpo, a, 64);
    const float ratio = expf(logp - logp_old);
    const float lo = 1.f - clip_eps, hi = 1.f + clip_eps;
    const float clipped = fminf(fmaxf(ratio, lo), hi);
    const float s1 = ratio * adv, s2 = clipped * adv;
    out.sur = fminf(s1, s2);
    out.ent = ent;
    out.kl = kl;
    if (j == 0) {
        if (ratio_dst) *ratio_dst = ratio;
        if (clipped_dst) *clipped_dst = clipped;
    }
    if (dlogits_row) {
        float g_logp;
        if (s1 <= s2)
            g_logp = -adv * ratio;
        else
            g_logp = (ratio >= lo && ratio <= hi) ? -adv * ratio : 0.f;
        g_logp /= (float)batch;
        const float gb = beta / (float)batch;
        float g = g_logp * ((j == a ? 1.f : 0.f) - p);
        g += gb * p * (lp + ent);
        if (live) dlogits_row[j] = grad_scale * g;
    }
    return true;
}

}  // namespace rlx_losses
