// Clipped-PPO heads in ONE launch: value head + policy head forward, both head losses and their gradients, and the
// heads' backward pass (dW, db and the gradient into the two towers' last layer).
//
// Replaces three dependent launches of the minibatch update (profiles/r01_bench_c2_kernel_stats_v6.csv:
// dense_small_fwd_multi 5.4 us + ppo_value_losses 7.4 us + dense_small_bwd_multi 7.7 us per update), i.e.
//   heads/v_head.py:43-52, heads/ppo_head.py:52-116, head.py:143-186 and tf.gradients down to the middleware
// (clipped_ppo_agent.py:209-308 runs them once per minibatch, 320 times per rollout).
//
// Phase 1 — workgroup w takes rows w, w + G, ...: the two narrow dense layers of the row (dense_small_fwd_row, the
//   arithmetic of rlx_dense_small_forward), then the row's value-loss and PPO terms and the gradients w.r.t. its
//   head outputs (ppo_discrete_row / the MSE row of regression_loss_body): everything a row needs is its own.
// One agent-scope barrier among the G = 2 * ceil(K / 32) resident workgroups (mlp_fused.hip's form; 32 workgroups
//   ~1.5 us, profiles/r02_launch_cost_microbench.txt).
// Phase 2 — workgroup (problem, k-block): dense_small_bwd_body (dW = x^T dz, db, dx with the tower's activation
//   derivative), reading the dz rows of phase 1; workgroup 0 also reduces the per-row loss terms with the SAME
//   block_sum tree the stand-alone loss kernel uses.
// Same device functions, same summation orders: results are bit-identical to the three-launch path
// (tests/test_ppo_heads_fused.py).
#include "dense_small_body.hpp"
#include "losses_body.hpp"

namespace {
using namespace rlx_small;
using namespace rlx_losses;

constexpr int kSpinLimit = 1 << 22;

struct PpoHeadsDev {
    SmallDense fwd[2];            // 0 = value head (N = 1), 1 = policy head (N = A)
    SmallDenseBwd bwd[2];
    const int *actions;
    const float *advantages, *old_probs, *value_targets;
    long long ld_old;
    float clip_eps, beta, grad_scale;
    const float *clip_scale;
    float *scalars, *value_scalar, *ratio_out, *clipped_out;
    float *terms;                 // [M][4] per-row {surrogate, entropy, kl, squared value error}
    unsigned *sync;               // [2]: barrier, finish ticket (zero between launches)
    int *status;
    int kblocks;
};

template <int NN>
__global__ void __launch_bounds__(256) ppo_heads_fused_kernel(const PpoHeadsDev d) {
    extern __shared__ float smem[];
    __shared__ float part[4][NN];
    __shared__ float red[256];
    const int wg = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const int M = d.fwd[0].M, A = d.fwd[1].N;
    // ---- phase 1: forward + per-row losses
    for (int row = wg; row < M; row += G) {
        dense_small_fwd_row<NN>(d.fwd[0], row, 0, part);
        __syncthreads();
        dense_small_fwd_row<NN>(d.fwd[1], row, 0, part);
        __syncthreads();
        if (tid == 0) {                                          // VHead: MSE(target, V), loss weight 1
            const float e = d.fwd[0].y[row] - d.value_targets[row];
            const float w = 1.f * 1.f;
            const float g = 2.f * e;
            float rowl = 0.f;
            rowl += e * e;
            const_cast<float *>(d.bwd[0].dy)[row] = d.grad_scale * w * g / (float)M;
            d.terms[4 * row + 3] = w * rowl;
        }
        if (tid == 64) {                                         // PPOHead
            PpoRowTerms t{0.f, 0.f, 0.f};
            const bool ok = ppo_discrete_row(d.fwd[1].y + (size_t)row * A, d.old_probs + (size_t)row * d.ld_old,
                                             d.actions[row], A, d.advantages[row],
                                             d.clip_scale ? d.clip_eps * *d.clip_scale : d.clip_eps, d.beta, d.grad_scale, M,
                                             const_cast<float *>(d.bwd[1].dy) + (size_t)row * A,
                                             d.ratio_out ? d.ratio_out + row : nullptr,
                                             d.clipped_out ? d.clipped_out + row : nullptr, t);
            if (!ok) {
                atomicOr(d.status, 1);
                t = PpoRowTerms{0.f, 0.f, 0.f};
                for (int j = 0; j < A; ++j) const_cast<float *>(d.bwd[1].dy)[(size_t)row * A + j] = 0.f;
            }
            d.terms[4 * row + 0] = t.sur;
            d.terms[4 * row + 1] = t.ent;
            d.terms[4 * row + 2] = t.kl;
        }
        __syncthreads();
    }
    // ---- every row's dz is needed by every k-block: one barrier among the resident workgroups
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&d.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&d.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {
                atomicOr(d.status, 8);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // ---- phase 2: the heads' backward pass, one (problem, k-block) per workgroup
    const int prob = wg / d.kblocks, kb = wg - prob * d.kblocks;
    if (prob < 2 && kb * kKL < d.bwd[prob].K) dense_small_bwd_body<NN>(d.bwd[prob], kb, 0, smem);
    if (wg == 0) {                                               // batch means: the loss kernels' block_sum trees
        __syncthreads();
        const bool live = tid < M;
        const float sur = block_sum(live ? d.terms[4 * tid + 0] : 0.f, red);
        const float ent = block_sum(live ? d.terms[4 * tid + 1] : 0.f, red);
        const float kl = block_sum(live ? d.terms[4 * tid + 2] : 0.f, red);
        const float vl = block_sum(live ? d.terms[4 * tid + 3] : 0.f, red);
        if (tid == 0) {
            const float inv = 1.f / (float)M;
            if (d.scalars) {
                d.scalars[0] = -sur * inv;
                d.scalars[1] = ent * inv;
                d.scalars[2] = kl * inv;
                d.scalars[3] = -sur * inv - d.beta * ent * inv;
            }
            if (d.value_scalar) d.value_scalar[0] = vl / (float)M;
        }
    }
    // ---- the last workgroup to finish re-arms the barrier word for the next launch / graph replay
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&d.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned)G - 1) {
            __hip_atomic_store(&d.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&d.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The same three steps WITHOUT a grid-wide dependency: the value head and the policy head read different towers and
// have separate losses, so each gets ONE workgroup (grid = 2) that keeps its whole input [M][K] in LDS (M * K <= 32 K
// floats: C2's 64 x 512), runs forward (a wave per row), the per-row loss terms + batch means (the stand-alone loss
// kernels' block_sum tree), and the backward pass (a thread per input feature: dW, dx with the tower's activation
// derivative; db) — three launches' work behind workgroup barriers only.  Summation orders differ from the
// dense_small kernels' (fp32 noise level differences; the loss scalars' trees are the same).
template <int NN>
__global__ void __launch_bounds__(1024) ppo_heads_wg_kernel(const PpoHeadsDev d) {
    extern __shared__ float sm[];
    const int h = blockIdx.x;
    const SmallDense f = d.fwd[h];
    const SmallDenseBwd bw = d.bwd[h];
    const int M = f.M, K = f.K, N = f.N, tid = threadIdx.x, nt = blockDim.x;
    float *xs = sm;                       // [M][K]
    float *ws = xs + (size_t)M * K;       // [K][NN], zero-padded columns
    float *ys = ws + (size_t)K * NN;      // [M][NN]: head outputs, then their gradients
    float *red = ys + (size_t)M * NN;     // [1024]
    {
        const float4 *src = reinterpret_cast<const float4 *>(f.x);
        float4 *dst = reinterpret_cast<float4 *>(xs);
        for (int i = tid; i < (M * K) >> 2; i += nt) dst[i] = src[i];
        for (int i = tid; i < K * NN; i += nt) {
            const int k = i / NN, n = i - k * NN;
            ws[i] = n < N ? f.w[(size_t)k * N + n] : 0.f;
        }
    }
    __syncthreads();
    // ---- forward: a wave per row, lanes stride over the features
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int row = wave; row < M; row += nw) {
        float acc[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) acc[n] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float xv = xs[(size_t)row * K + k];
#pragma unroll
            for (int n = 0; n < NN; ++n) acc[n] = fmaf(xv, ws[k * NN + n], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            float v = acc[n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && n < N) {
                v += f.b ? f.b[n] : 0.f;
                ys[row * NN + n] = v;
                f.y[(size_t)row * N + n] = v;
            }
        }
    }
    __syncthreads();
    // ---- per-row loss terms and the gradient w.r.t. the head outputs
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    if (tid < M) {
        const int row = tid;
        if (h == 0) {                                            // VHead: MSE(target, V), loss weight 1
            const float e = ys[row * NN] - d.value_targets[row];
            const float g = d.grad_scale * (2.f * e) / (float)M;
            ys[row * NN] = g;
            const_cast<float *>(bw.dy)[row] = g;
            t0 = e * e;
        } else {                                                 // PPOHead
            float z[NN], dz[NN];
#pragma unroll
            for (int n = 0; n < NN; ++n) z[n] = ys[row * NN + n];
            PpoRowTerms t{0.f, 0.f, 0.f};
            const bool ok = ppo_discrete_row(z, d.old_probs + (size_t)row * d.ld_old, d.actions[row], N,
                                             d.advantages[row], d.clip_scale ? d.clip_eps * *d.clip_scale : d.clip_eps,
                                             d.beta, d.grad_scale, M, dz, d.ratio_out ? d.ratio_out + row : nullptr,
                                             d.clipped_out ? d.clipped_out + row : nullptr, t);
            if (!ok) {
                atomicOr(d.status, 1);
                t = PpoRowTerms{0.f, 0.f, 0.f};
#pragma unroll
                for (int n = 0; n < NN; ++n) dz[n] = 0.f;
            }
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                ys[row * NN + n] = n < N ? dz[n] : 0.f;
                if (n < N) const_cast<float *>(bw.dy)[(size_t)row * N + n] = dz[n];
            }
            t0 = t.sur; t1 = t.ent; t2 = t.kl;
        }
    }
    const float s0 = block_sum(t0, red);
    if (h == 0) {
        if (tid == 0 && d.value_scalar) d.value_scalar[0] = s0 / (float)M;
    } else {
        const float s1 = block_sum(t1, red);
        const float s2 = block_sum(t2, red);
        if (tid == 0 && d.scalars) {
            const float inv = 1.f / (float)M;
            d.scalars[0] = -s0 * inv;
            d.scalars[1] = s1 * inv;
            d.scalars[2] = s2 * inv;
            d.scalars[3] = -s0 * inv - d.beta * s1 * inv;
        }
    }
    __syncthreads();
    // ---- backward: a thread per input feature k: dW[k][:] = sum_b x[b][k] dz[b][:],  dx[b][k] = (dz[b] . W[k]) act'(x)
    for (int k = tid; k < K; k += nt) {
        float wv[NN], acc[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            wv[n] = ws[k * NN + n];
            acc[n] = 0.f;
        }
        for (int b = 0; b < M; ++b) {
            const float xv = xs[(size_t)b * K + k];
            float g = 0.f;
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                const float dzv = ys[b * NN + n];
                acc[n] = fmaf(xv, dzv, acc[n]);
                g = fmaf(dzv, wv[n], g);
            }
            if (bw.dx) bw.dx[(size_t)b * K + k] = bw.lower_act ? g * act_deriv_out(xv, bw.lower_act) : g;
        }
        if (bw.dw)
            for (int n = 0; n < N; ++n) bw.dw[(size_t)k * N + n] = acc[n];
    }
    if (tid < N && bw.db) {
        float sdb = 0.f;
        for (int b = 0; b < M; ++b) sdb += ys[b * NN + tid];
        bw.db[tid] = sdb;
    }
}

}  // namespace

extern "C" {

int rlx_ppo_discrete_heads_fused(const rlx_small_dense_problem *heads_host, const int *actions,
                                 const float *advantages, const float *old_probs, long long ld_old,
                                 const float *value_targets, float clip_epsilon, float beta_entropy, float grad_scale,
                                 float *scalars, float *value_loss_scalar, float *likelihood_ratio,
                                 float *clipped_likelihood_ratio, float *row_terms, unsigned int *sync_words,
                                 int *status, const float *clip_scale, void *stream) {
    RLX_REQUIRE(heads_host && actions && advantages && old_probs && value_targets && row_terms && sync_words && status,
                "rlx_ppo_discrete_heads_fused: null pointer");
    const rlx_small_dense_problem &v = heads_host[0], &pi = heads_host[1];
    RLX_REQUIRE(v.N == 1 && pi.N >= 1 && pi.N <= kMaxN && v.M == pi.M && v.M > 0 && v.M <= 256 && v.towers == 1 &&
                    pi.towers == 1 && v.K > 0 && pi.K > 0,
                "rlx_ppo_discrete_heads_fused: heads[0] is the value head (1 output), heads[1] the policy head, "
                "same minibatch of <= 256 rows");
    RLX_REQUIRE(v.activation == 0 && pi.activation == 0, "rlx_ppo_discrete_heads_fused: the heads are linear");
    RLX_REQUIRE(ld_old >= pi.N, "rlx_ppo_discrete_heads_fused: bad old-policy pitch");
    PpoHeadsDev d;
    const int nn = pi.N <= 4 ? 4 : pi.N <= 8 ? 8 : 16;
    int kblocks = 0;
    for (int i = 0; i < 2; ++i) {
        const rlx_small_dense_problem &q = heads_host[i];
        RLX_REQUIRE(q.x && q.w && q.y && q.dy && (q.dw || q.dx), "rlx_ppo_discrete_heads_fused: null pointer in head %d", i);
        RLX_REQUIRE(q.lower_activation >= 0 && q.lower_activation <= 2, "rlx_ppo_discrete_heads_fused: unknown activation");
        d.fwd[i] = SmallDense{q.x, q.x_tower_stride, q.w, q.w_tower_stride, q.bias, q.bias_tower_stride, q.y,
                              q.y_tower_stride, q.M, q.K, q.N, 0};
        d.bwd[i] = SmallDenseBwd{q.x, q.x_tower_stride, q.w, q.w_tower_stride, q.dy, q.dy_tower_stride, nullptr,
                                 q.y_tower_stride, q.dw, q.dw_tower_stride, q.db, q.db_tower_stride, q.dx,
                                 q.dx_tower_stride, q.M, q.K, q.N, 0, q.lower_activation};
        const int kb = (q.K + kKL - 1) / kKL;
        if (kb > kblocks) kblocks = kb;
    }
    d.actions = actions; d.advantages = advantages; d.old_probs = old_probs; d.value_targets = value_targets;
    d.ld_old = ld_old; d.clip_eps = clip_epsilon; d.beta = beta_entropy; d.grad_scale = grad_scale;
    d.clip_scale = clip_scale;
    d.scalars = scalars; d.value_scalar = value_loss_scalar; d.ratio_out = likelihood_ratio;
    d.clipped_out = clipped_likelihood_ratio; d.terms = row_terms; d.sync = sync_words; d.status = status;
    d.kblocks = kblocks;
    {
        // one workgroup per head, the head's input resident in LDS (no grid-wide dependency): the default
        static const bool grid_form = [] { const char *e = getenv("RLX_PPO_HEADS_GRID"); return e && e[0] == '1'; }();
        const int kmax = v.K > pi.K ? v.K : pi.K;
        const size_t wg_floats = (size_t)v.M * kmax + (size_t)kmax * nn + (size_t)v.M * nn + 1024;
        const bool aligned = (((uintptr_t)v.x | (uintptr_t)pi.x) & 15) == 0 && ((size_t)v.M * v.K) % 4 == 0 &&
                             ((size_t)pi.M * pi.K) % 4 == 0;
        if (!grid_form && aligned && v.M <= 1024 && wg_floats * sizeof(float) <= 160 * 1024) {
            const size_t lds = wg_floats * sizeof(float);
            hipStream_t s2 = rlx::as_stream(stream);
#define RLX_HEADS_WG(NN_)                                                                                       \
            {                                                                                                    \
                static size_t configured = 0;                                                                    \
                if (lds > configured) {                                                                          \
                    RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ppo_heads_wg_kernel<NN_>),        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
                    configured = lds;                                                                            \
                }                                                                                                \
                ppo_heads_wg_kernel<NN_><<<2, 1024, lds, s2>>>(d);                                               \
            }
            if (nn == 4) RLX_HEADS_WG(4)
            else if (nn == 8) RLX_HEADS_WG(8)
            else RLX_HEADS_WG(16)
#undef RLX_HEADS_WG
            RLX_LAUNCH_CHECK();
            return RLX_OK;
        }
    }
    const int G = 2 * kblocks;
    RLX_REQUIRE(G <= 128, "rlx_ppo_discrete_heads_fused: %d workgroups must be resident at once (K too large)", G);
    d.actions = actions; d.advantages = advantages; d.old_probs = old_probs; d.value_targets = value_targets;
    d.ld_old = ld_old; d.clip_eps = clip_epsilon; d.beta = beta_entropy; d.grad_scale = grad_scale;
    d.clip_scale = clip_scale;
    d.scalars = scalars; d.value_scalar = value_loss_scalar; d.ratio_out = likelihood_ratio;
    d.clipped_out = clipped_likelihood_ratio; d.terms = row_terms; d.sync = sync_words; d.status = status;
    d.kblocks = kblocks;
    const size_t smem = ((size_t)v.M * pi.N + (size_t)kRG * kKL * nn) * sizeof(float);
    RLX_REQUIRE(smem <= 64 * 1024, "rlx_ppo_discrete_heads_fused: batch x outputs exceeds the LDS budget");
    hipStream_t s = rlx::as_stream(stream);
    if (nn == 4) ppo_heads_fused_kernel<4><<<G, 256, smem, s>>>(d);
    else if (nn == 8) ppo_heads_fused_kernel<8><<<G, 256, smem, s>>>(d);
    else ppo_heads_fused_kernel<16><<<G, 256, smem, s>>>(d);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
