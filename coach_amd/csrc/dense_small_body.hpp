// Device bodies of the narrow dense layers (see dense_small.hip), shared with the fused Clipped-PPO heads
// kernel (ppo_heads_fused.hip): one definition, one summation order.
#pragma once
#include "rlx_common.hpp"

namespace rlx_small {
constexpr int kMaxN = 16;

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == RLX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RLX_ACT_TANH) return tanhf(v);
    return v;
}
__device__ __forceinline__ float act_deriv_out(float y, int kind) {
    if (kind == RLX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (kind == RLX_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

struct SmallDense {
    const float *x; long long x_ts;        // [T][M][K] (x_ts = 0: shared input)
    const float *w; long long w_ts;        // [T][K][N]
    const float *b; long long b_ts;        // [T][N]
    float *y; long long y_ts;              // [T][M][N]
    int M, K, N, act;
};

// y[row][0..N) of one row: the 4 waves each take a quarter of K (lanes stride over k), butterfly-reduce their N
// accumulators and combine the four partials through LDS (`part`, [4][NN] floats) in a fixed order.
template <int NN>
__device__ __forceinline__ void dense_small_fwd_row(const SmallDense &p, int row, int t, float (*part)[NN]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *x = p.x + (size_t)t * p.x_ts + (size_t)row * p.K;
    const float *w = p.w + (size_t)t * p.w_ts;
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = 0.f;
    const int kq = (p.K + 3) / 4;
    const int k1 = min(p.K, (wave + 1) * kq);
    for (int k = wave * kq + lane; k < k1; k += 64) {
        const float xv = x[k];
        const float *wr = w + (size_t)k * p.N;
#pragma unroll
        for (int n = 0; n < NN; ++n)
            if (n < p.N) acc[n] = fmaf(xv, wr[n], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wave][n] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < p.N) {
        const int n = threadIdx.x;
        float v = ((part[0][n] + part[1][n]) + part[2][n]) + part[3][n];
        v += p.b ? p.b[(size_t)t * p.b_ts + n] : 0.f;
        p.y[(size_t)t * p.y_ts + (size_t)row * p.N + n] = act_apply(v, p.act);
    }
}

struct SmallDenseBwd {
    const float *x; long long x_ts;        // [T][M][K] layer input (= lower layer's output)
    const float *w; long long w_ts;        // [T][K][N]
    const float *dy; long long dy_ts;      // [T][M][N] gradient w.r.t. this layer's OUTPUT
    const float *y; long long y_ts;        // [T][M][N] this layer's output (for its own act') or null
    float *dw; long long dw_ts;            // [T][K][N] or null
    float *db; long long db_ts;            // [T][N] or null
    float *dx; long long dx_ts;            // [T][M][K] or null
    int M, K, N, act, lower_act;           // lower_act != 0: dx *= act'(x)  (x is the lower layer's output)
};

// block = 256 threads: lane = k within the kKL-feature slab, g = row group (rows g, g+kRG, ...).
// 32 features x 8 row groups: twice the workgroups and half the per-thread row loop of a 64 x 4
// split (these launches are a handful of workgroups; latency, not bandwidth, sets their time).
constexpr int kKL = 32, kRG = 8;
// ---- several narrow layers in ONE launch (blockIdx.z = problem): the value and the policy head of
// Clipped PPO read different towers, have different widths (1 and A) and separate parameter groups,
// but the same depth in the dependency chain — one dispatch instead of two, forward and backward.
constexpr int kMaxProblems = 4;
struct MultiFwd { SmallDense p[kMaxProblems]; int towers[kMaxProblems]; int n; };
struct MultiBwd { SmallDenseBwd p[kMaxProblems]; int towers[kMaxProblems]; int n; };

// KL features x RG row groups per workgroup (KL * RG = 256).  32 x 8 is what the multi-problem launch uses; the
// single-problem launch (rlx_dense_small_backward) takes 16 x 16: twice the workgroups and, up to M = 128 rows, every x
// load of a thread in ONE chunk (one exposed round trip instead of two at M = 100) — a different grouping of the dW row
// sums (profiles/r03_ab_candidates.txt).
// PRESTAGED: the caller writes dz (the gradient w.r.t. the pre-activation output) into smem[0 .. M*N) itself (STAGE) —
// the Clipped-PPO heads launch computes it there from the head losses (ppo_heads_bwd.hip); p.dy / p.y are not read.
struct NoStage { __device__ __forceinline__ void operator()() const {} };
// STAGE (PRESTAGED only): called AFTER this thread's w and x loads have been issued and before the barrier that
// publishes dz — the caller computes dz there, under the loads' latency instead of in front of it.
template <int NN, int KL = kKL, int RG = kRG, bool PRESTAGED = false, typename STAGE = NoStage>
__device__ __forceinline__ void dense_small_bwd_body(const SmallDenseBwd &p, int kblock, int t, float *smem,
                                                     STAGE stage = STAGE()) {
    float *dz = smem;                              // [M][N]
    float *part = smem + (size_t)p.M * p.N;        // [RG][KL][NN] dW partials
    const int lane = threadIdx.x % KL, g = threadIdx.x / KL;
    const int k = kblock * KL + lane;
    if (!PRESTAGED) {
        const float *dy = p.dy + (size_t)t * p.dy_ts;
        const float *yy = p.y ? p.y + (size_t)t * p.y_ts : nullptr;
        for (int i = threadIdx.x; i < p.M * p.N; i += 256)
            dz[i] = dy[i] * (yy ? act_deriv_out(yy[i], p.act) : 1.f);
    }
    const bool live = k < p.K;
    const float *__restrict__ x = p.x + (size_t)t * p.x_ts;
    float wk[NN], acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        acc[n] = 0.f;
        wk[n] = (live && n < p.N) ? p.w[(size_t)t * p.w_ts + (size_t)k * p.N + n] : 0.f;
    }
    float *__restrict__ dx = p.dx ? p.dx + (size_t)t * p.dx_ts : nullptr;
    // rows g, g+8, ... in chunks of kCH: every x load of a chunk is issued before the first use (one
    // exposed memory latency per chunk instead of one per row); the first chunk's loads are in flight
    // while dz is being staged
    constexpr int kCH = 8;
    float xv[kCH];
    auto load_chunk = [&](int m0) {
#pragma unroll
        for (int u = 0; u < kCH; ++u) {
            const int mm = m0 + u * RG;
            xv[u] = (live && mm < p.M) ? x[(size_t)mm * p.K + k] : 0.f;
        }
    };
    load_chunk(g);
    if (PRESTAGED) stage();
    __syncthreads();
    for (int m0 = g; m0 < p.M; m0 += RG * kCH) {
        if (m0 != g) load_chunk(m0);
        if (live) {
#pragma unroll
            for (int u = 0; u < kCH; ++u) {
                const int mm = m0 + u * RG;
                if (mm < p.M) {
                    const float *dzr = dz + (size_t)mm * p.N;
                    float s = 0.f;
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        if (n < p.N) {
                            const float d = dzr[n];
                            acc[n] = fmaf(xv[u], d, acc[n]);
                            s = fmaf(d, wk[n], s);
                        }
                    if (dx) dx[(size_t)mm * p.K + k] = p.lower_act ? s * act_deriv_out(xv[u], p.lower_act) : s;
                }
            }
        }
    }
    if (p.dw) {
#pragma unroll
        for (int n = 0; n < NN; ++n) part[((size_t)g * KL + lane) * NN + n] = acc[n];
        __syncthreads();
        if (g == 0 && live) {
            float *dw = p.dw + (size_t)t * p.dw_ts + (size_t)k * p.N;
#pragma unroll
            for (int n = 0; n < NN; ++n)
                if (n < p.N) {
                    float v = part[lane * NN + n];
#pragma unroll
                    for (int u = 1; u < RG; ++u) v += part[(u * KL + lane) * NN + n];     // fixed order
                    dw[n] = v;
                }
        }
    }
    if (p.db && kblock == 0) {
        // bias gradient: column n is summed by R threads of one wave (rows r, r + R, ...), combined by a
        // fixed-order butterfly inside the wave
        constexpr int R = NN >= 4 ? 256 / NN : 64;        // 64 / 64 / 32 / 16 lanes for NN = 1 / 4 / 8 / 16
        const int n = threadIdx.x / R, r = threadIdx.x % R;
        float s = 0.f;
        if (n < p.N)
            for (int mm = r; mm < p.M; mm += R) s += dz[(size_t)mm * p.N + n];
#pragma unroll
        for (int o = R / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, R);
        if (r == 0 && n < p.N) p.db[(size_t)t * p.db_ts + n] = s;
    }
}

}  // namespace rlx_small
