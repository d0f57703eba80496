// Fused small-MLP DQN update — ONE launch for DQNAgent.learn_from_batch on a vector-observation Q network.
//
// Replaces, for the MLP Q networks of presets such as CartPole_DQN (embedder Dense(H1) -> middleware
// Dense(H2) -> QHead Dense(A), relu), the ~14 dependent launches of the layer-by-layer path:
//   rl_coach/agents/dqn_agent.py:81-113        TD targets, |TD error| (prioritized replay), TD-target fit
//   architectures/tensorflow_components/heads/q_head.py + head.py:143-186   MSE / Huber loss, importance weights
//   architectures/tensorflow_components/architecture.py:312-385,469-521     forward, tf.gradients, Adam apply
// A 133 k-parameter update at batch 32 is 43 MFLOP: 0.3 us of MFMA work at chip rate, so the layer-by-layer
// path is pure launch latency (profiles/r01: ~4.3 us per dependent launch inside a hipGraph).
//
// Structure ("tensor parallel over workgroups", G = H2 / 32 workgroups, one per CU):
//   * every workgroup computes the narrow first layer h1 = relu(x W1 + b1) in full (K = obs dim, tiny);
//   * workgroup g OWNS columns [32g, 32g+32) of the wide layer: h2[:, slice] = relu(h1 W2[:, slice] + b2),
//     its 32 rows of the head W3 and — in backward — the same slices of dW2 / db2 / dW3, whose Adam step it
//     applies straight from the MFMA accumulators (the gradients of the wide layer never touch HBM);
//   * two exchanges through global memory, each followed by an agent-scope barrier among the G resident
//     workgroups (bounded spin; cdna_hip_programming.md G16 "plain stores -> release -> counter -> acquire"):
//       (1) the head's partial sums q_g = h2[:, slice] W3[slice, :]  (B x A floats per tower),
//       (2) the partial input gradients of the wide layer dh1_g = dz2[:, slice] W2[:, slice]^T (B x H1),
//           reduce-scattered: workgroup g sums columns [g H1/G, (g+1) H1/G) and owns that slice of W1 / b1;
//   * loss, TD errors and dQ are computed redundantly by every workgroup from the summed head outputs
//     (fixed summation orders everywhere: the result is reproducible run to run);
//   * tf.global_norm partials and the Adam beta-power advance are finished by the LAST workgroup to arrive
//     at a ticket (no waiting), which also re-arms the barrier words for the next launch / graph replay.
// Measured on MI355X (profiles/r02_launch_cost_microbench.txt): such a barrier among 16 workgroups costs
// 1.2 us clean and 3.6 us after 16 KB of dirty lines per workgroup, against 4.3 us per launch boundary.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered fmaf chain) for the three wide products,
// plain fp32 FMAs for the narrow ones; TD targets in fp64 like the reference's Python loop.  Compiled with
// -ffp-contract=off: the Adam step rounds like adam_tf1_kernel (optim.hip).
#include "rlx_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int LD = 33;                 // LDS row pitch (words): conflict-free for row- and column-wise walks
constexpr int kMaxA = 16;
constexpr int kSpinLimit = 1 << 22;    // ~ a second of s_sleep polling; a stuck barrier reports status bit 8

struct MlpDqnDev {
    float *w;                 // online weights (flat)           [updated]
    const float *wt;          // target weights (flat)
    float *m, *v;             // Adam slots (flat)               [updated]
    float *adam_state;        // {beta1^t, beta2^t}              [advanced by the last workgroup]
    const float *s, *s_next;  // [B, D0]
    const int *actions;
    const float *rewards;
    const unsigned char *dones;
    const double *iw;         // importance weights or null
    float *ws_q, *ws_dh1, *ws_norm;
    long long *stamps;        // [12] s_memtime of workgroup 0 at the phase boundaries (tools/mlp_fused_phases.py)
    unsigned *sync;           // [3]: barrier 1, barrier 2, finish ticket (all zero between launches)
    float *loss, *norm;
    double *td;
    int *status;
    long long o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
    double discount;
    int B, D0, H1, H2, A;
    int huber, ddqn;
    float lr, beta1, beta2, eps, grad_scale;
};

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// All workgroups of the grid meet here.  Producer side: plain stores, __syncthreads, ONE lane releases at
// agent scope (L2 write-back), drains, and bumps the counter; consumer side: the same lane polls the counter
// relaxed with s_sleep, then ONE agent-scope acquire, __syncthreads, plain loads (G16's valid form).
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target, int *status) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {
                atomicOr(status, 8);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

struct Adam {
    float alpha, omb1, omb2, eps, gscale;
    float ss;
    float *w, *m, *v;
    __device__ __forceinline__ void step(long long i, float g) {
        ss += g * g;                                   // tf.global_norm of the raw gradient
        const float gr = g * gscale;
        float mi = m[i], vi = v[i];
        mi += (gr - mi) * omb1;
        vi += (gr * gr - vi) * omb2;
        m[i] = mi;
        v[i] = vi;
        w[i] -= (mi * alpha) / (sqrtf(vi) + eps);
    }
    // 16 accumulator elements of one MFMA tile: all 48 loads are issued before the first dependent use
    // (one memory latency per tile instead of sixteen)
    __device__ __forceinline__ void step16(const long long (&idx)[16], const f32x16 &g) {
        float mi[16], vi[16], wi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { mi[r] = m[idx[r]]; vi[r] = v[idx[r]]; wi[r] = w[idx[r]]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ss += g[r] * g[r];
            const float gr = g[r] * gscale;
            mi[r] += (gr - mi[r]) * omb1;
            vi[r] += (gr * gr - vi[r]) * omb2;
            wi[r] -= (mi[r] * alpha) / (sqrtf(vi[r]) + eps);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { m[idx[r]] = mi[r]; v[idx[r]] = vi[r]; w[idx[r]] = wi[r]; }
    }
};

// fixed-order sum of n values spaced `stride` apart, loads issued eight at a time
__device__ __forceinline__ float sum_strided(const float *src, size_t stride, int n, float s) {
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = src[(size_t)(i + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += t[j];
    }
    for (; i < n; ++i) s += src[(size_t)i * stride];
    return s;
}

// h1dst[n][row] = relu(x W1 + b1): wave w takes the 32-column tiles w, w+4, ...  K = D0 <= 16 (zero padded to
// even): the B operands of two tiles are fetched with 16 independent loads before the first MFMA.
__device__ __forceinline__ void dense1(const float *xs, const float *W, long long o_w1, long long o_b1,
                                       int D0, int H1, float *h1dst, int w, int l31, int hi) {
    const int D0p = (D0 + 1) & ~1;
    for (int nb = 32 * w; nb < H1; nb += 256) {
        float bv[2][8], bias[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n0 = nb + 128 * t;
            const bool tile = n0 < H1;
            bias[t] = tile ? W[o_b1 + n0 + l31] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 2 * j + hi;
                bv[t][j] = (tile && k < D0) ? W[o_w1 + (long long)k * H1 + n0 + l31] : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n0 = nb + 128 * t;
            if (n0 >= H1) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (2 * j < D0p) {
                    const float a = xs[(2 * j + hi) * LD + l31];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t][j], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] + bias[t];
                h1dst[(n0 + l31) * LD + mfma_row(r, hi)] = v > 0.f ? v : 0.f;
            }
        }
    }
}

// h2dst[c][row] = relu(h1 W2[:, slice] + b2[slice]); K = H1 split over the 4 waves, partials summed in fixed order
__device__ __forceinline__ void dense2(const float *h1s, const float *w2s, const float *W, long long o_b2,
                                       int col0, int H1, float *red, float *h2dst, int tid, int w, int l31,
                                       int hi) {
    const int kq = H1 / 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kk = w * kq; kk < (w + 1) * kq; kk += 2) {
        const float a = h1s[(kk + hi) * LD + l31];
        const float b = w2s[(kk + hi) * LD + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 32 + mfma_row(r, hi)) * LD + l31] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * kThreads;
        const int b = e >> 5, c = e & 31;
        const int o = b * LD + c;
        float v = ((red[o] + red[32 * LD + o]) + red[64 * LD + o]) + red[96 * LD + o];
        v += W[o_b2 + col0 + c];
        h2dst[c * LD + b] = v > 0.f ? v : 0.f;
    }
    __syncthreads();
}

__device__ __forceinline__ void load_w2_slice(const float *W, long long o_w2, int col0, int H1, int H2,
                                              float *w2s, int tid) {
    const int c = tid & 31;
#pragma unroll 8
    for (int i = tid >> 5; i < H1; i += kThreads / 32) w2s[i * LD + c] = W[o_w2 + (long long)i * H2 + col0 + c];
}

// partial head output of this workgroup's 32 hidden units -> ws_q[g][tower][b][a]
__device__ __forceinline__ void q_partial(const float *h2s, const float *w3s, int A, float *dst, int tid) {
    for (int e = tid; e < 32 * A; e += kThreads) {
        const int b = e & 31, a = e >> 5;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) s += h2s[c * LD + b] * w3s[c * kMaxA + a];
        dst[b * A + a] = s;
    }
}

__global__ void __launch_bounds__(kThreads) mlp_dqn_update_kernel(const MlpDqnDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.x, G = gridDim.x;
    const int B = p.B, D0 = p.D0, H1 = p.H1, H2 = p.H2, A = p.A;
    const int D0p = (D0 + 1) & ~1;
    const int col0 = 32 * g;
    int stamp_i = 0;
#define RLX_STAMP() do { if (g == 0 && tid == 0) p.stamps[stamp_i] = (long long)__builtin_readcyclecounter(); ++stamp_i; } while (0)
    RLX_STAMP();                                                   // 0: start

    // ---- LDS carve
    float *xs0 = smem;                       // [D0p][LD]   states, k-major
    float *xs1 = xs0 + D0p * LD;             // [D0p][LD]   next states
    float *h1a = xs1 + D0p * LD;             // [H1][LD]    online h1(s)           (kept for backward)
    float *h1b = h1a + H1 * LD;              // [H1][LD]    h1 of the s' passes    (transient)
    float *w2s = h1b + H1 * LD;              // [H1][LD]    W2[:, slice]: target copy first, then the online copy
    float *red = w2s + H1 * LD;              // [4][32][LD] K-split partials / staging
    float *h2a = red + 4 * 32 * LD;          // [32][LD]    online h2(s)[:, slice], c-major (kept)
    float *h2b = h2a + 32 * LD;              // [32][LD]    transient
    float *dz2 = h2b + 32 * LD;              // [32][LD]    [b][c]
    float *dz2t = dz2 + 32 * LD;             // [32][LD]    [c][b]
    float *qs = dz2t + 32 * LD;              // [3][32][A]  summed head outputs: online(s), target(s'), online(s')
    float *dqs = qs + 3 * 32 * kMaxA;        // [32][A]
    float *b3s = dqs + 32 * kMaxA;           // [2][A]      head bias online / target (read before anyone updates it)
    float *lred = b3s + 2 * kMaxA;           // [256]
    float *w3s = lred + kThreads;            // [2][32][A]  owned rows of the head: online / target (pre-update)

    // ---- inputs -> LDS (zero padded to 32 rows / even K)
    for (int e = tid; e < D0p * 32; e += kThreads) {
        const int k = e >> 5, b = e & 31;
        const bool in = b < B && k < D0;
        xs0[k * LD + b] = in ? p.s[(long long)b * D0 + k] : 0.f;
        xs1[k * LD + b] = in ? p.s_next[(long long)b * D0 + k] : 0.f;
    }
    if (tid < A) {
        b3s[tid] = p.w[p.o_b3 + tid];
        b3s[kMaxA + tid] = p.wt[p.o_b3 + tid];
    }
    for (int e = tid; e < 32 * A; e += kThreads) {
        const int c = e / A, a = e - c * A;
        w3s[c * kMaxA + a] = p.w[p.o_w3 + (long long)(col0 + c) * A + a];
        w3s[32 * kMaxA + c * kMaxA + a] = p.wt[p.o_w3 + (long long)(col0 + c) * A + a];
    }
    const float b1p = p.adam_state[0], b2p = p.adam_state[1];
    __syncthreads();
    RLX_STAMP();                                                   // 1: inputs staged

    float *wsq = p.ws_q + (size_t)g * 3 * 32 * A;
    // ---- target network on s'
    dense1(xs1, p.wt, p.o_w1, p.o_b1, D0, H1, h1b, w, l31, hi);
    load_w2_slice(p.wt, p.o_w2, col0, H1, H2, w2s, tid);
    __syncthreads();
    dense2(h1b, w2s, p.wt, p.o_b2, col0, H1, red, h2b, tid, w, l31, hi);
    q_partial(h2b, w3s + 32 * kMaxA, A, wsq + 1 * 32 * A, tid);
    __syncthreads();
    RLX_STAMP();                                                   // 2: target tower done
    // ---- online network: s' (Double DQN action selection, ddqn_agent.py:43) and s
    load_w2_slice(p.w, p.o_w2, col0, H1, H2, w2s, tid);
    if (p.ddqn) {
        dense1(xs1, p.w, p.o_w1, p.o_b1, D0, H1, h1b, w, l31, hi);
        __syncthreads();
        dense2(h1b, w2s, p.w, p.o_b2, col0, H1, red, h2b, tid, w, l31, hi);
        q_partial(h2b, w3s, A, wsq + 2 * 32 * A, tid);
    }
    dense1(xs0, p.w, p.o_w1, p.o_b1, D0, H1, h1a, w, l31, hi);
    __syncthreads();
    dense2(h1a, w2s, p.w, p.o_b2, col0, H1, red, h2a, tid, w, l31, hi);
    q_partial(h2a, w3s, A, wsq, tid);
    RLX_STAMP();                                                   // 3: online towers done

    grid_barrier(&p.sync[0], (unsigned)G, p.status);          // ---- exchange 1: head partial sums
    RLX_STAMP();                                                   // 4: barrier 1 passed

    const int towers = p.ddqn ? 3 : 2;
    for (int e = tid; e < towers * 32 * A; e += kThreads) {
        const int t = e / (32 * A), rem = e - t * 32 * A;
        const float s = sum_strided(p.ws_q + (size_t)t * 32 * A + rem, (size_t)3 * 32 * A, G,
                                    b3s[(t == 1 ? kMaxA : 0) + rem % A]);
        qs[t * 32 * kMaxA + (rem / A) * kMaxA + rem % A] = s;
    }
    __syncthreads();

    // ---- TD targets, |TD error|, loss, dQ (dqn_agent.py:92-111; the arithmetic of dqn_head_loss_kernel)
    float term = 0.f;
    if (tid < 32) {
        const int b = tid;
        for (int a = 0; a < A; ++a) dqs[b * kMaxA + a] = 0.f;
        if (b < B) {
            const float *q_on = qs + b * kMaxA;
            const float *q_tg = qs + 32 * kMaxA + b * kMaxA;
            const float *q_sel = p.ddqn ? qs + 2 * 32 * kMaxA + b * kMaxA : q_tg;
            int best = 0;
            float bv = q_sel[0];
            for (int a = 1; a < A; ++a)
                if (q_sel[a] > bv) { bv = q_sel[a]; best = a; }
            const int act = p.actions[b];
            if (act < 0 || act >= A) {
                atomicOr(p.status, 1);
            } else {
                const double y = (double)p.rewards[b] + (1.0 - (p.dones[b] ? 1.0 : 0.0)) * p.discount * (double)q_tg[best];
                const float qa = q_on[act];
                if (g == 0 && p.td) p.td[b] = fabs(y - (double)qa);
                const float e = qa - (float)y;
                const float iw = p.iw ? (float)p.iw[b] : 1.f;
                float l, gr;
                if (!p.huber) { l = e * e; gr = 2.f * e; }
                else { const float ae = fabsf(e); l = ae <= 1.f ? 0.5f * e * e : ae - 0.5f; gr = fminf(fmaxf(e, -1.f), 1.f); }
                term = iw * l;
                dqs[b * kMaxA + act] = iw * gr / (float)B;
            }
        }
    }
    lred[tid] = term;
    __syncthreads();
    for (int d = 32; d > 0; d >>= 1) {               // the 64-entry tree of dqn_head_loss_kernel at B <= 32
        if (tid < d) lred[tid] += lred[tid + d];
        __syncthreads();
    }
    if (g == 0 && tid == 0 && p.loss) p.loss[0] = lred[0] / (float)B;
    RLX_STAMP();                                                   // 5: loss / dQ

    Adam ad;
    ad.alpha = p.lr * sqrtf(1.f - b2p) / (1.f - b1p);
    ad.omb1 = 1.f - p.beta1; ad.omb2 = 1.f - p.beta2; ad.eps = p.eps; ad.gscale = p.grad_scale;
    ad.ss = 0.f; ad.w = p.w; ad.m = p.m; ad.v = p.v;

    // ---- backward through the head: dz2 = (dq W3[slice]^T) * relu'(h2), both layouts
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * kThreads;
        const int b = e >> 5, c = e & 31;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s += dqs[b * kMaxA + a] * w3s[c * kMaxA + a];
        const float v = h2a[c * LD + b] > 0.f ? s : 0.f;
        dz2[b * LD + c] = v;
        dz2t[c * LD + b] = v;
    }
    __syncthreads();
    // head weights of the owned hidden units: dW3[c][a] = sum_b h2[b][c] dq[b][a]
    for (int e = tid; e < 32 * A; e += kThreads) {
        const int c = e / A, a = e - c * A;
        float s = 0.f;
        for (int b = 0; b < 32; ++b) s += h2a[c * LD + b] * dqs[b * kMaxA + a];
        ad.step(p.o_w3 + (long long)(col0 + c) * A + a, s);
    }
    if (tid < 32) {                                   // db2[slice]
        float s = 0.f;
        for (int b = 0; b < 32; ++b) s += dz2[b * LD + tid];
        ad.step(p.o_b2 + col0 + tid, s);
    }
    if (g == 0 && tid >= 64 && tid < 64 + A) {        // db3 (workgroup 0)
        const int a = tid - 64;
        float s = 0.f;
        for (int b = 0; b < 32; ++b) s += dqs[b * kMaxA + a];
        ad.step(p.o_b3 + a, s);
    }
    // partial dh1 = dz2[:, slice] W2[:, slice]^T -> ws_dh1[g][b][i]   (uses the PRE-update W2 copy in LDS)
    float *wsd = p.ws_dh1 + (size_t)g * 32 * H1;
    for (int n0 = 32 * w; n0 < H1; n0 += 128) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a = dz2t[(kk + hi) * LD + l31];
            const float b = w2s[(n0 + l31) * LD + kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) wsd[(size_t)mfma_row(r, hi) * H1 + n0 + l31] = acc[r];
    }
    // dW2[:, slice] = h1^T dz2, Adam applied from the accumulators
    for (int i0 = 32 * w; i0 < H1; i0 += 128) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a = h1a[(i0 + l31) * LD + kk + hi];
            const float b = dz2[(kk + hi) * LD + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        long long idx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) idx[r] = p.o_w2 + (long long)(i0 + mfma_row(r, hi)) * H2 + col0 + l31;
        ad.step16(idx, acc);
    }

    RLX_STAMP();                                                   // 6: backward of the wide layer + its Adam
    grid_barrier(&p.sync[1], (unsigned)G, p.status);          // ---- exchange 2: dh1 partials (reduce-scatter)
    RLX_STAMP();                                                   // 7: barrier 2 passed

    const int S1 = H1 / G, i0 = g * S1;               // this workgroup's slice of the first layer
    float *dz1 = red;                                  // [32][S1]
    for (int e = tid; e < 32 * S1; e += kThreads) {
        const int b = e / S1, i = e - b * S1;
        const float s = sum_strided(p.ws_dh1 + (size_t)b * H1 + i0 + i, (size_t)32 * H1, G, 0.f);
        dz1[b * S1 + i] = h1a[(i0 + i) * LD + b] > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < (D0 + 1) * S1; e += kThreads) {      // rows 0..D0-1: dW1, row D0: db1
        const int d = e / S1, i = e - d * S1;
        float s = 0.f;
        if (d < D0) {
            for (int b = 0; b < 32; ++b) s += xs0[d * LD + b] * dz1[b * S1 + i];
            ad.step(p.o_w1 + (long long)d * H1 + i0 + i, s);
        } else {
            for (int b = 0; b < 32; ++b) s += dz1[b * S1 + i];
            ad.step(p.o_b1 + i0 + i, s);
        }
    }

    RLX_STAMP();                                                   // 8: first layer's gradient + Adam
    // ---- tf.global_norm partial; the last workgroup to take a ticket finishes norm + Adam state and re-arms
    __syncthreads();
    lred[tid] = ad.ss;
    __syncthreads();
    for (int d = kThreads >> 1; d > 0; d >>= 1) {
        if (tid < d) lred[tid] += lred[tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        p.ws_norm[g] = lred[0];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(&p.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned)G - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            float s = 0.f;
            for (int gg = 0; gg < G; ++gg) s += p.ws_norm[gg];
            if (p.norm) p.norm[0] = sqrtf(s);
            p.adam_state[0] = b1p * p.beta1;           // AdamOptimizer._finish: beta powers advance
            p.adam_state[1] = b2p * p.beta2;
            __hip_atomic_store(&p.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    RLX_STAMP();                                                   // 9: end
#undef RLX_STAMP
}

inline size_t lds_floats(int D0, int H1) {
    const int D0p = (D0 + 1) & ~1;
    return (size_t)2 * D0p * LD + (size_t)3 * H1 * LD + 4 * 32 * LD + 4 * 32 * LD + 3 * 32 * kMaxA + 32 * kMaxA +
           2 * kMaxA + kThreads + 2 * 32 * kMaxA;
}


// ---------------------------------------------------------------------------------------------------------------
// Acting with the same small MLP: Q(s) for a handful of envs and the epsilon-greedy choice, one launch of one
// workgroup (choose_action, agents/dqn_agent.py + exploration_policies/e_greedy.py:84-101).  A thread owns one hidden
// unit: the weight rows are read coalesced ([in][out] storage), every dot product runs in index order (plain fp32
// fmaf chain); the action values are one wave per (env, action) with a fixed shuffle tree.  The greedy choice
// restates egreedy_kernel (explore.hip) on the values still in LDS.
constexpr int kActEnvs = 8;
constexpr int kActThreads = 1024;
// slices of the wide layer's reduction index: as many as the threads allow, at most 2048 / H2 (LDS for the partials)
__host__ __device__ inline int act_slices(int threads, int n_env, int h2) {
    int s = threads / (h2 >> 2);
    const int cap = 8192 / (n_env * h2);          // 32 KB of partial sums
    if (s > cap) s = cap;
    return s > 0 ? s : 1;
}
inline size_t act_lds_floats(int n_env, int h1, int h2) {
    return (size_t)kActEnvs * (16 + h1 + h2 + kMaxA) + (size_t)n_env * act_slices(kActThreads, n_env, h2) * h2;
}

struct MlpActDev {
    const float *w; long long o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
    const float *states;
    const double *explore_u; const int *random_act; const double *tie_rand; double epsilon;
    float *q_out; int *actions;
    int n_env, D, H1, H2, A;
};

template <int E>
__global__ void __launch_bounds__(1024) mlp_q_act_kernel(const MlpActDev p) {
    extern __shared__ float sm[];
    float *xs = sm;                                  // [E][D]
    float *h1s = xs + kActEnvs * 16;                 // [E][H1]
    float *h2s = h1s + kActEnvs * p.H1;              // [E][H2]
    float *qs = h2s + kActEnvs * p.H2;               // [E][A]
    float *parts = qs + kActEnvs * kMaxA;            // [E][S][H2], S * H2 / 4 <= threads
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < E * p.D; i += nt) xs[i] = p.states[i];
    __syncthreads();
    const float *W1 = p.w + p.o_w1, *b1 = p.w + p.o_b1, *W2 = p.w + p.o_w2, *b2 = p.w + p.o_b2;
    const float *W3 = p.w + p.o_w3, *b3 = p.w + p.o_b3;
    for (int j = tid; j < p.H1; j += nt) {
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        for (int k = 0; k < p.D; ++k) {
            const float wv = W1[(size_t)k * p.H1 + j];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(xs[e * p.D + k], wv, acc[e]);
        }
        const float bv = b1[j];
#pragma unroll
        for (int e = 0; e < E; ++e) h1s[e * p.H1 + j] = fmaxf(acc[e] + bv, 0.f);
    }
    __syncthreads();
    // the wide layer: H1 x H2 weights through ONE compute unit.  A thread owns 4 adjacent hidden units (16-byte
    // weight loads, coalesced over the threads) and one of S slices of the reduction index; the S partial sums of
    // a unit meet in LDS and are added in slice order.
    {
        const int groups = p.H2 >> 2;
        const int S = act_slices(nt, E, p.H2);
        const int chunk = (((p.H1 + S - 1) / S) + 3) & ~3;
        for (int t = tid; t < S * groups; t += nt) {
            const int part = t / groups, c4 = (t - part * groups) << 2;
            const int k0 = part * chunk, k1 = min(p.H1, k0 + chunk);
            float4 acc[E];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
            for (int k = k0; k < k1; k += 4) {
                float4 wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[i] = *reinterpret_cast<const float4 *>(W2 + (size_t)(k + i) * p.H2 + c4);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float4 h = *reinterpret_cast<const float4 *>(h1s + e * p.H1 + k);
                    const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[e].x = fmaf(hv[i], wv[i].x, acc[e].x);
                        acc[e].y = fmaf(hv[i], wv[i].y, acc[e].y);
                        acc[e].z = fmaf(hv[i], wv[i].z, acc[e].z);
                        acc[e].w = fmaf(hv[i], wv[i].w, acc[e].w);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) *reinterpret_cast<float4 *>(parts + (size_t)(e * S + part) * p.H2 + c4) = acc[e];
        }
        __syncthreads();
        for (int j = tid; j < p.H2; j += nt) {
            const float bv = b2[j];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float v = parts[(size_t)(e * S) * p.H2 + j];
                for (int q = 1; q < S; ++q) v += parts[(size_t)(e * S + q) * p.H2 + j];
                h2s[e * p.H2 + j] = fmaxf(v + bv, 0.f);
            }
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int o = wave; o < E * p.A; o += nw) {
        const int e = o / p.A, a = o - e * p.A;
        float part = 0.f;
        for (int k = lane; k < p.H2; k += 64) part = fmaf(h2s[e * p.H2 + k], W3[(size_t)k * p.A + a], part);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) {
            const float v = part + b3[a];
            qs[o] = v;
            if (p.q_out) p.q_out[o] = v;
        }
    }
    __syncthreads();
    if (tid < E && p.actions) {
        const int e = tid;
        if (p.explore_u[e] < p.epsilon) {
            p.actions[e] = p.random_act[e];
        } else {
            const float *qe = qs + e * p.A;
            float mx = qe[0];
            for (int a = 1; a < p.A; ++a) mx = fmaxf(mx, qe[a]);
            const float tol = 1e-8f + 1e-5f * fabsf(mx);
            int best = 0;
            double bv = -1.0;
            for (int a = 0; a < p.A; ++a) {
                const bool close = fabsf(qe[a] - mx) <= tol;
                const double v = close ? p.tie_rand[(size_t)e * p.A + a] : 0.0;
                if (v > bv) {
                    bv = v;
                    best = a;
                }
            }
            p.actions[e] = best;
        }
    }
}

}  // namespace

extern "C" {

int rlx_mlp_q_act_supported(int n_env, int obs_dim, int h1, int h2, int n_actions) {
    if (n_env < 1 || n_env > kActEnvs || obs_dim < 1 || obs_dim > 16 || n_actions < 1 || n_actions > kMaxA) return 0;
    if (h1 < 4 || h2 < 4 || h1 > 1024 || h2 > 1024 || h1 % 4 || h2 % 4) return 0;
    return act_lds_floats(n_env, h1, h2) * sizeof(float) <= 64 * 1024 ? 1 : 0;      /* the default dynamic-LDS limit */
}

int rlx_mlp_q_act(const float *weights, long long off_w1, long long off_b1, long long off_w2, long long off_b2,
                  long long off_w3, long long off_b3, const float *states, int n_env, int obs_dim, int h1, int h2,
                  int n_actions, const double *explore_uniforms, const int *random_actions,
                  const double *tie_break_uniforms, double epsilon, float *q_out, int *actions, void *stream) {
    RLX_REQUIRE(weights && states && (q_out || actions), "rlx_mlp_q_act: null pointer");
    RLX_REQUIRE(rlx_mlp_q_act_supported(n_env, obs_dim, h1, h2, n_actions),
                "rlx_mlp_q_act: unsupported shape (n_env=%d obs=%d h1=%d h2=%d A=%d)", n_env, obs_dim, h1, h2, n_actions);
    RLX_REQUIRE(!actions || (explore_uniforms && random_actions && tie_break_uniforms),
                "rlx_mlp_q_act: the epsilon-greedy choice needs the host draws");
    MlpActDev p;
    p.w = weights; p.o_w1 = off_w1; p.o_b1 = off_b1; p.o_w2 = off_w2; p.o_b2 = off_b2; p.o_w3 = off_w3; p.o_b3 = off_b3;
    p.states = states; p.explore_u = explore_uniforms; p.random_act = random_actions; p.tie_rand = tie_break_uniforms;
    p.epsilon = epsilon; p.q_out = q_out; p.actions = actions;
    p.n_env = n_env; p.D = obs_dim; p.H1 = h1; p.H2 = h2; p.A = n_actions;
    const size_t lds = sizeof(float) * act_lds_floats(n_env, h1, h2);
    hipStream_t st = rlx::as_stream(stream);
    switch (n_env) {
        case 1: RLX_LAUNCH((mlp_q_act_kernel<1>), 1, kActThreads, lds, st, p); break;
        case 2: RLX_LAUNCH((mlp_q_act_kernel<2>), 1, kActThreads, lds, st, p); break;
        case 3: RLX_LAUNCH((mlp_q_act_kernel<3>), 1, kActThreads, lds, st, p); break;
        case 4: RLX_LAUNCH((mlp_q_act_kernel<4>), 1, kActThreads, lds, st, p); break;
        case 5: RLX_LAUNCH((mlp_q_act_kernel<5>), 1, kActThreads, lds, st, p); break;
        case 6: RLX_LAUNCH((mlp_q_act_kernel<6>), 1, kActThreads, lds, st, p); break;
        case 7: RLX_LAUNCH((mlp_q_act_kernel<7>), 1, kActThreads, lds, st, p); break;
        default: RLX_LAUNCH((mlp_q_act_kernel<8>), 1, kActThreads, lds, st, p); break;
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_mlp_dqn_supported(int batch, int obs_dim, int h1, int h2, int n_actions) {
    if (batch < 1 || batch > 32 || obs_dim < 1 || obs_dim > 16 || n_actions < 1 || n_actions > kMaxA) return 0;
    if (h1 % 32 || h2 % 32 || h1 < 32 || h2 < 32) return 0;
    const int G = h2 / 32;
    if (G > 64 || h1 % G || (h1 / 4) % 2) return 0;
    return lds_floats(obs_dim, h1) * sizeof(float) <= 160 * 1024 ? 1 : 0;
}

int rlx_mlp_dqn_workspace_floats(int h1, int h2, int n_actions, long long *floats_host) {
    RLX_REQUIRE(floats_host && h1 > 0 && h2 >= 32 && n_actions > 0, "rlx_mlp_dqn_workspace_floats: bad arguments");
    const long long G = h2 / 32;
    *floats_host = G * 3 * 32 * n_actions + G * 32 * h1 + G + 2 + 32;     /* + 12 int64 phase stamps */
    return RLX_OK;
}

int rlx_mlp_dqn_update(const rlx_mlp_dqn_desc *d_host, void *stream) {
    RLX_REQUIRE(d_host != nullptr, "rlx_mlp_dqn_update: null descriptor");
    const rlx_mlp_dqn_desc &d = *d_host;
    RLX_REQUIRE(rlx_mlp_dqn_supported(d.batch, d.obs_dim, d.h1, d.h2, d.n_actions),
                "rlx_mlp_dqn_update: unsupported shape (batch=%d <= 32, obs=%d <= 16, h1=%d, h2=%d multiples of 32, "
                "actions=%d <= 16)", d.batch, d.obs_dim, d.h1, d.h2, d.n_actions);
    RLX_REQUIRE(d.weights && d.target_weights && d.adam_m && d.adam_v && d.adam_state && d.states && d.next_states &&
                    d.actions && d.rewards && d.game_overs && d.workspace && d.sync_words && d.status,
                "rlx_mlp_dqn_update: null pointer");
    long long need = 0;
    rlx_mlp_dqn_workspace_floats(d.h1, d.h2, d.n_actions, &need);
    RLX_REQUIRE(d.workspace_floats >= need, "rlx_mlp_dqn_update: workspace of %lld floats, need %lld",
                d.workspace_floats, need);
    const int G = d.h2 / 32;
    MlpDqnDev p;
    p.w = d.weights; p.wt = d.target_weights; p.m = d.adam_m; p.v = d.adam_v; p.adam_state = d.adam_state;
    p.s = d.states; p.s_next = d.next_states; p.actions = d.actions; p.rewards = d.rewards; p.dones = d.game_overs;
    p.iw = d.importance_weights;
    p.ws_q = d.workspace;
    p.ws_dh1 = p.ws_q + (size_t)G * 3 * 32 * d.n_actions;
    p.ws_norm = p.ws_dh1 + (size_t)G * 32 * d.h1;
    p.stamps = reinterpret_cast<long long *>(p.ws_norm + ((G + 1) & ~1) + (((uintptr_t)(p.ws_norm + ((G + 1) & ~1))) & 4 ? 1 : 0));
    p.sync = d.sync_words;
    p.loss = d.loss_out; p.norm = d.norm_out; p.td = d.td_errors; p.status = d.status;
    p.o_w1 = d.off_w1; p.o_b1 = d.off_b1; p.o_w2 = d.off_w2; p.o_b2 = d.off_b2; p.o_w3 = d.off_w3; p.o_b3 = d.off_b3;
    p.discount = d.discount;
    p.B = d.batch; p.D0 = d.obs_dim; p.H1 = d.h1; p.H2 = d.h2; p.A = d.n_actions;
    p.huber = d.huber; p.ddqn = d.double_dqn;
    p.lr = d.learning_rate; p.beta1 = d.beta1; p.beta2 = d.beta2; p.eps = d.epsilon; p.grad_scale = d.grad_scale;
    const size_t lds = lds_floats(d.obs_dim, d.h1) * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_dqn_update_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    RLX_LAUNCH((mlp_dqn_update_kernel), G, kThreads, lds, rlx::as_stream(stream), p);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
