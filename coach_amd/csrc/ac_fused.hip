// TD3 learn_from_batch as eight short launches instead of ~36 (rl_coach/agents/td3_agent.py:148-209), and the tile-parallel
// weight-gradient + Adam launch every fused continuous-control update ends with.
//
// Replaces, for the MLP actor / twin-critic networks of TD3 (agents/td3_agent.py:36-68; heads/td3_v_head.py:40-60,
// heads/ddpg_actor_head.py:48-56) the layer-by-layer launch chain of coach_amd/agents/td3_agent.py.  A workgroup owns 4
// batch rows AND one of kSplit column slices of a network's wide middle layer: it computes the (narrow-input) first
// layer in full, its slice of the middle layer, and the slice's contribution to whatever follows — a head's output or a
// transposed product's input gradient, both SUMS over the middle layer's units — and leaves that partial sum in memory;
// the consumer (the next launch) adds the kSplit partials in slice order.  Why slices: a workgroup that streams a whole
// 400 x 300 layer is bound by what one CU can pull from L2, 6.6 us per layer however the loop is written
// (profiles/r06_td3_rowlocal_whole_layers.txt: the whole-layer version of this file ran the update in exactly the time of
// the launch chain it replaced); a quarter of the layer is 3 us, most of it fixed cost (profiles/r06_rowchain_slice_probe.txt).
//   critic step (every update):
//     td3_forward1_kernel    role A (rows, slice c): h2[:, c] of mu_target(s'), partial head sums
//                            role Q (rows, stream s, slice c): h1, h2[:, c] of Q_s(s, a) (kept), partial Q_s
//     td3_forward2_kernel    (rows, s, c): a' = tanh(sum of partials) -> clip(a' + clip(noise)) (:162-165) -> h2[:, c] of
//                            Q_target_s(s', a''), partial Q_target_s
//     td3_critic_backward_kernel  (rows, s, c): y = r + (1 - done) gamma min(Q_target_1, Q_target_2) (:168-180), loss terms,
//                            d loss / d Q_s, dz2[:, c], partial dz1 = dz2[:, c] W2[:, c]^T
//     mlp_dw_adam_kernel     dW / db of all six layers as 32 x 32 MFMA tiles over the batch (the first layer's gradient
//                            operand = the summed partials under relu'), TF1 Adam applied from the accumulators
//                            (+ tf.global_norm, + the soft target update when one is due)
//   actor step (every update_policy_every_x_episode_steps-th update, :186-207):
//     td3_actor_forward_kernel    (rows, c): h1, h2[:, c] of mu(s) (kept), partial head sums
//     td3_actor_q_kernel          (rows, c): mu(s) -> Q_1(s, mu(s)) with the UPDATED critic up to h2[:, c] -> dz2c[:, c] of
//                                 d mean(Q_1) (:194-198) -> partial dz1c
//     td3_actor_backward_kernel   (rows, c): d mean(Q_1) / d a -> -scale dQ/da into the tanh head (td3_v_head.py:57-58) ->
//                                 dz2a[:, c] -> partial dz1a
//     mlp_dw_adam_kernel          the actor's three layers
//
// Compiled with -ffp-contract=off: the Adam step rounds like adam_tf1_kernel (optim.hip); the layer products are MFMA
// chains (rowchain.hpp).
#include "rowchain.hpp"

namespace {

using namespace rlx_chain;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Mlp3 {
    long long o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
    long long s1, s2, s3;                 // tower strides of the three layers' parameter groups
    int d_in, h1, h2, d_out;
};


// LDS carve shared by the chain kernels: three activation rows sets, the partial-sum scratch, two weight tiles
struct Lds {
    float *x, *a, *b, *c, *d, *parts, *small, *wt;
    __device__ explicit Lds(float *base) {
        x = base;                              // [R][kPitch]
        a = x + R * kPitch;
        b = a + R * kPitch;
        c = b + R * kPitch;
        d = c + R * kPitch;
        parts = d + R * kPitch;                // kPartFloats
        small = parts + kPartFloats;           // 256
        wt = small + 256;                      // 2 * kTileFloats: only kernels with a transposed product ask for it
    }
};
// forward-only kernels leave the weight tiles out: two workgroups per CU (grids of up to 512 workgroups in one round)
constexpr size_t kFwdLdsBytes = sizeof(float) * (5 * R * kPitch + kPartFloats + 256);
constexpr size_t kChainLdsBytes = kFwdLdsBytes + sizeof(float) * 2 * kTileFloats;
static_assert(kChainLdsBytes <= 160 * 1024 && 2 * kFwdLdsBytes <= 160 * 1024, "chain kernels: LDS budget");

constexpr int kSplit = 4;                 // column slices of a middle layer (= workgroups that share a row block's layer)

// slice c of a layer of N columns (N % 4 == 0): whole 4-column groups, the remainder spread over the last slices
__device__ __host__ inline int slice_lo(int N, int c) { return 4 * (((N >> 2) * c) / kSplit); }

struct Td3Dev {
    const float *aw, *awt, *cw, *cwt;
    Mlp3 am, cm;
    const float *obs, *next_obs, *actions, *rewards;
    const unsigned char *dones;
    const double *noise;
    const float *low, *high;
    double noise_clip, discount, clip_lo, clip_hi;
    int nonzero_terminal, has_clip;
    float actor_scale;
    int B, D, A, nrb;
    // leading dimensions of the buffers the weight-gradient launch reads (mlp_dw_adam_kernel's operand contract: rows padded
    // to 128, columns to 64, zero outside): batch rows, merged critic input, critic h1 / h2, actor input / h1 / h2
    int Bp, ldx, ld1c, ld2c, ldxa, ld1a, ld2a;
    // critic step
    float *zT;        // [B][kSplit][A]      partial head sums of mu_target(s')
    float *xm;        // [Bp][ldx]           merged online critic input (a, s)
    float *h1c, *h2c; // [2][Bp][ld1c], [2][Bp][ld2c]
    float *qp, *qTp;  // [2][kSplit][B]      partial Q_s(s, a), partial Q_target_s(s', a'')
    float *dq, *dh2;  // [2][Bp][64] (column 0), [2][Bp][ld2c]
    float *dh1p;      // [2][kSplit][Bp][ld1c]  partial dz1 (before relu')
    float *loss_part; // [2][nrb]
    float *td_targets, *q_min;
    // actor step
    float *xa;        // [Bp][ldxa]          the states again, padded (the actor's first-layer operand)
    float *h1a, *h2a; // [Bp][ld1a], [Bp][ld2a]
    float *za;        // [B][kSplit][A]      partial head sums of mu(s)
    float *ya;        // [B][A]              tanh output
    float *h1q;       // [B][h1]             h1 of Q_1(s, mu(s))
    float *dh1qp;     // [kSplit][B][h1]     partial dz1 of the critic pass
    float *dz3, *dh2a;// [Bp][64], [Bp][ld2a]
    float *dh1ap;     // [kSplit][Bp][ld1a]
    float *neg_dq_da;
    long long *stamps;                    // [96] s_memtime of workgroup 0 at the phase boundaries, or null (tools/ac_fused_phases.py)
};
#define RLX_STAMP(slot) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[(slot)] = (long long)__builtin_readcyclecounter(); } while (0)

// the 4 rows' partial head sums of one column slice: out[r][j] = sum_{k in slice} h[r][k] W3[k][j]  (j < NO <= 16)
__device__ __forceinline__ void head_partial(const float *hs, int hp, const float *__restrict__ W3, int n_lo, int n_hi, int NO,
                                             float *__restrict__ out, long long ld, int row0, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < R * NO; o += T / 64) {
        const int r = o / NO, j = o - r * NO;
        float sum = 0.f;
        for (int k = n_lo + lane; k < n_hi; k += 64) sum = fmaf(hs[r * hp + (k - n_lo)], W3[(size_t)k * NO + j], sum);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane == 0 && row0 + r < B) out[(long long)(row0 + r) * ld + j] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(T) td3_forward1_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int tid = threadIdx.x, role = blockIdx.x % (3 * kSplit), rb = blockIdx.x / (3 * kSplit), row0 = rb * R;
    const int B = p.B, D = p.D, A = p.A, M = A + D;
    RLX_STAMP(0);
    if (role < kSplit) {
        // ---- mu_target(s'): first layer, slice c of the middle layer, the slice's share of the head's sums
        const int c = role;
        const Mlp3 &am = p.am;
        const int lo = slice_lo(am.h2, c), hi = slice_lo(am.h2, c + 1);
        load_rows(L.x, kPitch, p.next_obs, D, D, row0, B);
        __syncthreads();
        dense_fwd(L.x, kPitch, D, p.awt + am.o_w1, am.h1, p.awt + am.o_b1, am.h1, RLX_ACT_RELU, L.a, kPitch, L.parts, nullptr, 0, row0, B);
        RLX_STAMP(1);
        dense_fwd(L.a, kPitch, am.h1, p.awt + am.o_w2 + lo, am.h2, p.awt + am.o_b2 + lo, hi - lo, RLX_ACT_RELU, L.b, kPitch, L.parts,
                  nullptr, 0, row0, B);
        RLX_STAMP(2);
        head_partial(L.b, kPitch, p.awt + am.o_w3, lo, hi, A, p.zT + (size_t)c * A, (long long)kSplit * A, row0, B);
        RLX_STAMP(3);
    } else {
        // ---- Q_s(s, a): first layer (kept by slice 0), slice c of the middle layer (kept), the slice's share of Q_s
        const int s = (role - kSplit) / kSplit, c = (role - kSplit) % kSplit;
        const Mlp3 &cm = p.cm;
        const int lo = slice_lo(cm.h2, c), hi = slice_lo(cm.h2, c + 1);
        for (int e = tid; e < R * pad16(M); e += T) {
            const int r = e / pad16(M), col = e - r * pad16(M);
            float v = 0.f;
            if (row0 + r < B && col < M)
                v = col < A ? p.actions[(size_t)(row0 + r) * A + col] : p.obs[(size_t)(row0 + r) * D + (col - A)];
            L.x[r * kPitch + col] = v;
            if (s == 0 && c == 0 && row0 + r < B && col < M) p.xm[(size_t)(row0 + r) * p.ldx + col] = v;
        }
        __syncthreads();
        const float *w = p.cw;
        dense_fwd(L.x, kPitch, M, w + cm.o_w1 + s * cm.s1, cm.h1, w + cm.o_b1 + s * cm.s1, cm.h1, RLX_ACT_RELU, L.a, kPitch, L.parts,
                  c == 0 ? p.h1c + (size_t)s * p.Bp * p.ld1c : nullptr, p.ld1c, row0, B);
        dense_fwd(L.a, kPitch, cm.h1, w + cm.o_w2 + s * cm.s2 + lo, cm.h2, w + cm.o_b2 + s * cm.s2 + lo, hi - lo, RLX_ACT_RELU, L.b,
                  kPitch, L.parts, p.h2c + (size_t)s * p.Bp * p.ld2c + lo, p.ld2c, row0, B);
        head_partial(L.b, kPitch, w + cm.o_w3 + s * cm.s3, lo, hi, 1, p.qp + ((size_t)s * kSplit + c) * B, 1, row0, B);
    }
}

// a' = tanh(head sums + bias) * scale, smoothing, Q_target_s(s', a'') up to the slice's share of the head
__global__ void __launch_bounds__(T) td3_forward2_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int tid = threadIdx.x, role = blockIdx.x % (2 * kSplit), rb = blockIdx.x / (2 * kSplit), row0 = rb * R;
    const int s = role / kSplit, c = role % kSplit;
    const int B = p.B, D = p.D, A = p.A, M = A + D;
    const Mlp3 &am = p.am, &cm = p.cm;
    const int lo = slice_lo(cm.h2, c), hi = slice_lo(cm.h2, c + 1);
    RLX_STAMP(16);
    // merged critic input [a'' | s'] (general_network.py:251,270-277: sorted inputs, 'action' < 'observation')
    for (int e = tid; e < R * pad16(M); e += T) {
        const int r = e / pad16(M), col = e - r * pad16(M), i = row0 + r;
        float v = 0.f;
        if (i < B) {
            if (col < A) {
                const float *zp = p.zT + ((size_t)i * kSplit) * A + col;
                float z = zp[0];
#pragma unroll
                for (int q = 1; q < kSplit; ++q) z += zp[(size_t)q * A];
                const float mu = p.actor_scale * tanhf(z + p.awt[am.o_b3 + col]);
                if (p.noise) {
                    const double nz = fmin(fmax(p.noise[(size_t)i * A + col], -p.noise_clip), p.noise_clip);
                    double xv = (double)mu + nz;
                    xv = fmin(fmax(xv, (double)p.low[col]), (double)p.high[col]);          // spaces.py:379 np.clip
                    v = (float)xv;
                } else {
                    v = mu;
                }
            } else if (col < M) {
                v = p.next_obs[(size_t)i * D + (col - A)];
            }
        }
        L.x[r * kPitch + col] = v;
    }
    __syncthreads();
    RLX_STAMP(17);
    const float *w = p.cwt;
    dense_fwd(L.x, kPitch, M, w + cm.o_w1 + s * cm.s1, cm.h1, w + cm.o_b1 + s * cm.s1, cm.h1, RLX_ACT_RELU, L.a, kPitch, L.parts, nullptr, 0, row0, B);
    RLX_STAMP(18);
    dense_fwd(L.a, kPitch, cm.h1, w + cm.o_w2 + s * cm.s2 + lo, cm.h2, w + cm.o_b2 + s * cm.s2 + lo, hi - lo, RLX_ACT_RELU, L.b, kPitch,
              L.parts, nullptr, 0, row0, B);
    RLX_STAMP(19);
    head_partial(L.b, kPitch, w + cm.o_w3 + s * cm.s3, lo, hi, 1, p.qTp + ((size_t)s * kSplit + c) * B, 1, row0, B);
    RLX_STAMP(20);
}

// y, the loss terms of stream s, dz2[:, c] = dq W3[c]^T relu'(h2[:, c]) and the slice's partial dz1 = dz2[:, c] W2[:, c]^T
__global__ void __launch_bounds__(T) td3_critic_backward_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int tid = threadIdx.x, role = blockIdx.x % (2 * kSplit), rb = blockIdx.x / (2 * kSplit), row0 = rb * R;
    const int s = role / kSplit, c = role % kSplit;
    const int B = p.B;
    const Mlp3 &cm = p.cm;
    const int lo = slice_lo(cm.h2, c), hi = slice_lo(cm.h2, c + 1), wc = hi - lo;
    RLX_STAMP(32);
    load_rows(L.b, kPitch, p.h2c + (size_t)s * p.Bp * p.ld2c + lo, p.ld2c, wc, row0, B);
    if (tid < R) {
        const int i = row0 + tid;
        float dq = 0.f, term = 0.f;
        if (i < B) {
            float qt[2], qs = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {                                         // the partials in slice order, then the bias
                const float *pp = p.qTp + (size_t)t * kSplit * B + i;
                float v = pp[0];
#pragma unroll
                for (int q = 1; q < kSplit; ++q) v += pp[(size_t)q * B];
                qt[t] = v + p.cwt[cm.o_b3 + t * cm.s3];
            }
            {
                const float *pp = p.qp + (size_t)s * kSplit * B + i;
                float v = pp[0];
#pragma unroll
                for (int q = 1; q < kSplit; ++q) v += pp[(size_t)q * B];
                qs = v + p.cw[cm.o_b3 + s * cm.s3];
            }
            const float qn = qt[0] <= qt[1] ? qt[0] : qt[1];                      // output #2: min over the streams (:168)
            const double qd = (double)qn;
            double t;
            if (p.nonzero_terminal) t = (double)p.rewards[i] + p.discount * qd;
            else t = (double)p.rewards[i] + (1.0 - (p.dones[i] ? 1.0 : 0.0)) * p.discount * qd;     // :171-180
            if (p.has_clip) t = fmin(fmax(t, p.clip_lo), p.clip_hi);
            const float y = (float)t;
            if (s == 0 && c == 0) {
                p.td_targets[i] = y;
                p.q_min[i] = qn;
            }
            const float e = qs - y;                                               // head.py:143-186 (MSE): l = e^2, g = 2 e
            term = e * e;
            dq = 1.f * 1.f * (2.f * e) / (float)B;
            if (c == 0) p.dq[((size_t)s * p.Bp + i) * 64] = dq;
        }
        L.small[tid] = dq;
        L.small[R + tid] = term;
    }
    __syncthreads();
    RLX_STAMP(33);
    if (tid == 0 && c == 0)
        p.loss_part[(size_t)s * p.nrb + rb] = ((L.small[R] + L.small[R + 1]) + L.small[R + 2]) + L.small[R + 3];
    // dz2[r][k] = relu'(h2[r][k]) * dq[r] * W3[k]   (the head is Dense(1): its transposed product is an outer product)
    const float *w3 = p.cw + cm.o_w3 + s * cm.s3 + lo;
    for (int e = tid; e < R * pad16(wc); e += T) {
        const int r = e / pad16(wc), k = e - r * pad16(wc);
        float v = 0.f;
        if (k < wc && L.b[r * kPitch + k] > 0.f) v = L.small[r] * w3[k];
        L.c[r * kPitch + k] = v;
        if (k < wc && row0 + r < B) p.dh2[((size_t)s * p.Bp + row0 + r) * p.ld2c + lo + k] = v;
    }
    __syncthreads();
    RLX_STAMP(34);
    dense_bwdT(L.c, kPitch, wc, p.cw + cm.o_w2 + s * cm.s2 + lo, cm.h2, cm.h1, nullptr, 0, L.d, kPitch, L.wt,
               p.dh1p + ((size_t)s * kSplit + c) * p.Bp * p.ld1c, p.ld1c, row0, B);
    RLX_STAMP(35);
}

// mu(s): first layer (kept by slice 0), slice c of the middle layer (kept), partial head sums
__global__ void __launch_bounds__(T) td3_actor_forward_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int c = blockIdx.x % kSplit, rb = blockIdx.x / kSplit, row0 = rb * R;
    const int B = p.B, D = p.D, A = p.A;
    const Mlp3 &am = p.am;
    const int lo = slice_lo(am.h2, c), hi = slice_lo(am.h2, c + 1);
    RLX_STAMP(48);
    load_rows(L.x, kPitch, p.obs, D, D, row0, B);
    __syncthreads();
    if (c == 0)
        for (int e = threadIdx.x; e < R * D; e += T) {
            const int r = e / D, col = e - r * D;
            if (row0 + r < B) p.xa[(size_t)(row0 + r) * p.ldxa + col] = L.x[r * kPitch + col];
        }
    dense_fwd(L.x, kPitch, D, p.aw + am.o_w1, am.h1, p.aw + am.o_b1, am.h1, RLX_ACT_RELU, L.a, kPitch, L.parts,
              c == 0 ? p.h1a : nullptr, p.ld1a, row0, B);
    dense_fwd(L.a, kPitch, am.h1, p.aw + am.o_w2 + lo, am.h2, p.aw + am.o_b2 + lo, hi - lo, RLX_ACT_RELU, L.b, kPitch, L.parts,
              p.h2a + lo, p.ld2a, row0, B);
    head_partial(L.b, kPitch, p.aw + am.o_w3, lo, hi, A, p.za + (size_t)c * A, (long long)kSplit * A, row0, B);
    RLX_STAMP(49);
}

// mu(s) -> Q_1(s, mu(s)) with the updated critic up to h2[:, c]; dz2c[:, c] of d mean(Q_1); partial dz1c
__global__ void __launch_bounds__(T) td3_actor_q_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int tid = threadIdx.x, c = blockIdx.x % kSplit, rb = blockIdx.x / kSplit, row0 = rb * R;
    const int B = p.B, D = p.D, A = p.A, M = A + D;
    const Mlp3 &am = p.am, &cm = p.cm;
    const int lo = slice_lo(cm.h2, c), hi = slice_lo(cm.h2, c + 1), wc = hi - lo;
    RLX_STAMP(56);
    for (int e = tid; e < R * pad16(M); e += T) {
        const int r = e / pad16(M), col = e - r * pad16(M), i = row0 + r;
        float v = 0.f;
        if (i < B) {
            if (col < A) {
                const float *zp = p.za + ((size_t)i * kSplit) * A + col;
                float z = zp[0];
#pragma unroll
                for (int q = 1; q < kSplit; ++q) z += zp[(size_t)q * A];
                const float y = tanhf(z + p.aw[am.o_b3 + col]);
                if (c == 0) p.ya[(size_t)i * A + col] = y;
                v = p.actor_scale * y;
            } else if (col < M) {
                v = p.obs[(size_t)i * D + (col - A)];
            }
        }
        L.x[r * kPitch + col] = v;
    }
    __syncthreads();
    // critic stream 0 (td3_agent.py:188-192, output #3 = mean(Q_1))
    dense_fwd(L.x, kPitch, M, p.cw + cm.o_w1, cm.h1, p.cw + cm.o_b1, cm.h1, RLX_ACT_RELU, L.a, kPitch, L.parts,
              c == 0 ? p.h1q : nullptr, cm.h1, row0, B);
    dense_fwd(L.a, kPitch, cm.h1, p.cw + cm.o_w2 + lo, cm.h2, p.cw + cm.o_b2 + lo, wc, RLX_ACT_RELU, L.b, kPitch, L.parts, nullptr, 0, row0, B);
    // d mean_b(Q_1) / d Q_1 = 1 / B; dz2c = relu'(h2c) / B * W3
    const float inv_b = 1.0f / (float)B;
    const float *w3 = p.cw + cm.o_w3 + lo;
    for (int e = tid; e < R * pad16(wc); e += T) {
        const int r = e / pad16(wc), k = e - r * pad16(wc);
        L.c[r * kPitch + k] = (k < wc && L.b[r * kPitch + k] > 0.f) ? inv_b * w3[k] : 0.f;
    }
    __syncthreads();
    dense_bwdT(L.c, kPitch, wc, p.cw + cm.o_w2 + lo, cm.h2, cm.h1, nullptr, 0, L.d, kPitch, L.wt,
               p.dh1qp + (size_t)c * B * cm.h1, cm.h1, row0, B);
    RLX_STAMP(57);
}

// d mean(Q_1) / d a -> the tanh head's gradient -> dz2a[:, c] -> partial dz1a
__global__ void __launch_bounds__(T) td3_actor_backward_kernel(const Td3Dev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Lds L(smem);
    const int tid = threadIdx.x, c = blockIdx.x % kSplit, rb = blockIdx.x / kSplit, row0 = rb * R;
    const int B = p.B, A = p.A;
    const Mlp3 &am = p.am, &cm = p.cm;
    const int lo = slice_lo(am.h2, c), hi = slice_lo(am.h2, c + 1), wc = hi - lo;
    RLX_STAMP(64);
    // dz1c = (sum of the slices' partials) * relu'(h1 of the critic pass)
    for (int e = tid; e < R * pad16(cm.h1); e += T) {
        const int r = e / pad16(cm.h1), k = e - r * pad16(cm.h1), i = row0 + r;
        float v = 0.f;
        if (i < B && k < cm.h1 && p.h1q[(size_t)i * cm.h1 + k] > 0.f) {
            const float *pp = p.dh1qp + (size_t)i * cm.h1 + k;
            v = pp[0];
#pragma unroll
            for (int q = 1; q < kSplit; ++q) v += pp[(size_t)q * B * cm.h1];
        }
        L.a[r * kPitch + k] = v;
    }
    load_rows(L.b, kPitch, p.h2a + lo, p.ld2a, wc, row0, B);
    if (tid < R * 16) {
        const int r = tid >> 4, col = tid & 15;
        L.small[tid] = (col < A && row0 + r < B) ? p.ya[(size_t)(row0 + r) * A + col] : 0.f;
    }
    __syncthreads();
    // d / d action = the first A rows of the critic's first layer; straight into the actor head's gradient: -scale * dQ/da
    float *gy = L.small + 64;             // [R][16]
    dense_bwdT_few_rows(L.a, kPitch, cm.h1, p.cw + cm.o_w1, cm.h1, A, -p.actor_scale, gy, 16);
    // tanh head: dz3 = gy * (1 - y^2)
    float *dz3 = L.small + 128;           // [R][16]
    if (tid < R * 16) {
        const int r = tid >> 4, col = tid & 15;
        float v = 0.f;
        if (col < A) {
            const float y = L.small[r * 16 + col];
            v = gy[r * 16 + col] * (1.f - y * y);
            if (c == 0 && row0 + r < B) {
                p.dz3[(size_t)(row0 + r) * 64 + col] = v;
                p.neg_dq_da[(size_t)(row0 + r) * A + col] = gy[r * 16 + col];
            }
        }
        dz3[r * 16 + col] = v;
    }
    __syncthreads();
    // dz2a[r][k] = relu'(h2a[r][k]) * sum_j dz3[r][j] W3[k][j]   for the slice's k
    dense_bwdT_few_cols(dz3, 16, A, p.aw + am.o_w3 + (size_t)lo * A, wc, L.b, kPitch, L.c, kPitch, p.dh2a + lo, p.ld2a, row0, B);
    dense_bwdT(L.c, kPitch, wc, p.aw + am.o_w2 + lo, am.h2, am.h1, nullptr, 0, L.d, kPitch, L.wt,
               p.dh1ap + (size_t)c * p.Bp * p.ld1a, p.ld1a, row0, B);
    RLX_STAMP(65);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of up to kMaxJobs dense layers (dW = A^T G over the batch, db = column sums of G) as 32 x 32 MFMA
// tiles, one wave per tile, with the TF1 Adam step applied from the accumulators: the gradients never touch HBM.
constexpr int kMaxJobs = 16, kMaxNets = 3, kDwThreads = 256;

struct DwJob {
    const float *A; long long lda;
    const float *G; long long ldg;
    // g_parts > 1: the gradient operand is sum_q G[q * g_part_stride + ...] (a transposed product's slice partials) times
    // relu'(mask[b][n]) (mask = the layer's forward output)
    const float *mask; long long g_part_stride;
    long long o_w, o_b;
    int K, N, net, tile0, tiles_n, g_parts;
};
struct DwNet {
    float *w, *m, *v, *state, *target, *grads, *norm_out;
    float lr, beta1, beta2, eps, gscale, rate, omr;
    int tile_lo, tile_hi;
};
struct DwArgs {
    DwJob job[kMaxJobs];
    DwNet net[kMaxNets];
    int n_jobs, n_nets, B, tiles, write_grads;
    float *norm_part;                 // [tiles]
    unsigned *ticket;                 // one zeroed word
    // loss finalisation riding with the last arriver (TD3 / SAC critics): loss_out[s] = scale * sum(parts[s][..]) / B
    const float *loss_part; float *loss_out; int loss_streams, loss_parts; float loss_scale;
    long long *stamps;                // phase stamps of workgroup 0 / wave 0 (tools/ac_fused_phases.py) or null
};
#define RLX_DW_STAMP(slot) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[(slot)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// A workgroup owns a 64 x 64 block of one layer's dW (4 waves, a 32 x 32 MFMA tile each).  Both operands of the block —
// 64 feature columns of the layer's input and 64 columns of its output gradient, all batch rows — are staged in LDS by
// 16-byte coalesced loads first: fed straight from memory an MFMA step costs two 4-byte loads per lane, 200 load
// instructions per wave at B = 100 (x 6 where the gradient operand is a sum of slice partials under relu'), and the launch
// took 17 us of which 10 were those loads (profiles/r06_td3_dw_phases.txt).
constexpr int kDwBlock = 64, kDwRows = 128, kDwPitch = kDwBlock + 4;

// Operand contract: every A / G / mask matrix of a job is one of this library's workspace buffers, [roundup(B, 128)] rows
// of ld = roundup(cols, 64) floats, 16-byte aligned, ZERO outside [B][cols] (allocated zeroed, never written there): the
// staging is eight unconditional 16-byte loads per operand and thread — no clamps, no masks (the version with clamped
// addresses and masked stores spent 8 us in ~2700 instructions per thread here, profiles/r06_td3_dw_phases.txt).
constexpr int kDwPieces = kDwRows * (kDwBlock / 4) / kDwThreads;        // 8
__device__ __forceinline__ void dw_stage(f32x4 (&dst)[kDwPieces], const float *__restrict__ p, unsigned ld, int b0, int col0) {
    const unsigned t = threadIdx.x;
    const float *q = p + (unsigned)(b0 + (t >> 4)) * ld + (unsigned)col0 + 4u * (t & 15u);
#pragma unroll
    for (int i = 0; i < kDwPieces; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(q + (unsigned)(16 * i) * ld);
}

__global__ void __launch_bounds__(kDwThreads) mlp_dw_adam_kernel(const DwArgs a) {
    __shared__ __attribute__((aligned(16))) float As[kDwRows * kDwPitch];
    __shared__ __attribute__((aligned(16))) float Gs[kDwRows * kDwPitch];
    __shared__ float ss_s[kDwThreads / 64];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int tile = blockIdx.x;                        // block-uniform: scalar descriptor loads
    RLX_DW_STAMP(0);
    int j = 0;
    while (j + 1 < a.n_jobs && tile >= a.job[j + 1].tile0) ++j;
    const DwJob &jb = a.job[j];
    const DwNet &nt = a.net[jb.net];
    const int lt = tile - jb.tile0, ti = lt / jb.tiles_n, tj = lt - ti * jb.tiles_n;
    const int i0 = kDwBlock * ti + 32 * wi, n0 = kDwBlock * tj + 32 * wj, B = a.B;
    const bool gcol = n0 + l31 < jb.N;
    const bool mix = nt.target != nullptr && nt.rate >= 0.f;
    unsigned idx[16];                                    // element offsets in the network's flat buffers (< 2^31)
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + mfma_row(r, hi);
        ok[r] = gcol && row < jb.K;
        idx[r] = (unsigned)jb.o_w + (unsigned)min(row, jb.K - 1) * (unsigned)jb.N + (unsigned)min(n0 + l31, jb.N - 1);
    }
    const unsigned ib = (unsigned)jb.o_b + (unsigned)min(n0 + l31, jb.N - 1);
    float mi[17], vi[17], wgt[17], tg[17];
    const float *tsrc = mix ? nt.target : nt.w;          // (no soft update: the word is loaded and dropped)
    const float b1p = nt.state[0], b2p = nt.state[1];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    for (int b0 = 0; b0 < B; b0 += kDwRows) {
        if (b0 > 0) __syncthreads();
        // ---- stage rows b0 .. b0 + 127: thread t takes the 16-byte pieces t, t + 256, ... of each operand (16 per row)
        constexpr int kPieces = kDwPieces;
        RLX_DW_STAMP(7);
        f32x4 ap[kPieces], gp[kPieces];
        dw_stage(ap, jb.A, (unsigned)jb.lda, b0, kDwBlock * ti);
        dw_stage(gp, jb.G, (unsigned)jb.ldg, b0, kDwBlock * tj);
        if (jb.g_parts > 1) {
            // the gradient operand is the sum of the slices' partials (slice order) under relu' of the layer's output:
            // all requests of the phase first, then the arithmetic
            f32x4 part[kSplit - 1][kPieces], mk[kPieces];
#pragma unroll
            for (int z = 1; z < kSplit; ++z)
                dw_stage(part[z - 1], jb.G + (z < jb.g_parts ? z : 0) * jb.g_part_stride, (unsigned)jb.ldg, b0, kDwBlock * tj);
            dw_stage(mk, jb.mask, (unsigned)jb.ldg, b0, kDwBlock * tj);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
#pragma unroll
                for (int z = 1; z < kSplit; ++z)
                    if (z < jb.g_parts) gp[i] += part[z - 1][i];
#pragma unroll
                for (int e = 0; e < 4; ++e) gp[i][e] = mk[i][e] > 0.f ? gp[i][e] : 0.f;
            }
        }
        if (!a.write_grads && b0 == 0) {
            // ---- behind the first chunk's operand requests (vector-memory loads return in order: in front of them these
            //      cold lines — nothing has touched m / v since the last update — would hold the staging up): the optimiser
            //      state of this lane's 16 + 1 elements; unconditional loads from clamped addresses, masked at the store
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mi[r] = nt.m[idx[r]];
                vi[r] = nt.v[idx[r]];
                wgt[r] = nt.w[idx[r]];
                tg[r] = tsrc[idx[r]];
            }
            mi[16] = nt.m[ib]; vi[16] = nt.v[ib]; wgt[16] = nt.w[ib]; tg[16] = tsrc[ib];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int f = tid + i * kDwThreads, row = f >> 4, q = f & 15;
            *reinterpret_cast<f32x4 *>(As + row * kDwPitch + 4 * q) = ap[i];
            *reinterpret_cast<f32x4 *>(Gs + row * kDwPitch + 4 * q) = gp[i];
        }
        RLX_DW_STAMP(8);
        __syncthreads();
        RLX_DW_STAMP(9);
        const int rows = min(kDwRows, B - b0);
        const float *ar = As + hi * kDwPitch + 32 * wi + l31, *gr = Gs + hi * kDwPitch + 32 * wj + l31;
        for (int u = 0; 2 * u < rows; u += 8) {
            float x[8], y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                  // rows beyond `rows` were staged as zeros
                x[e] = ar[min(2 * (u + e), kDwRows - 2) * kDwPitch];
                y[e] = gr[min(2 * (u + e), kDwRows - 2) * kDwPitch];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (2 * (u + e) < kDwRows) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], y[e], acc, 0, 0, 0);
                    bsum += y[e];
                }
            }
        }
    }
    RLX_DW_STAMP(1);
    bsum += __shfl_xor(bsum, 32, 64);                  // the two lane halves hold the even / odd batch rows
    const bool bias_lane = ti == 0 && wi == 0 && gcol && hi == 0;
    if (a.write_grads) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) nt.grads[idx[r]] = acc[r];
        if (bias_lane) nt.grads[ib] = bsum;
        if (!a.loss_out) return;
    }
    // ---- tf.global_norm partial of this block, published BEFORE the Adam stores (an agent-scope store behind 50 stores
    //      would wait for all of them: vector-memory operations retire in order), then the ticket
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (ok[r]) ss += acc[r] * acc[r];
    if (bias_lane) ss += bsum * bsum;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d, 64);
    if (lane == 0) ss_s[wave] = ss;
    __syncthreads();
    if (tid == 0) {
        const float tot = ((ss_s[0] + ss_s[1]) + ss_s[2]) + ss_s[3];
        __hip_atomic_store(&a.norm_part[tile], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RLX_DW_STAMP(2);
        last_s = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        RLX_DW_STAMP(3);
    }
    // ---- Adam from the accumulators (every workgroup read the beta powers above, before it drew its ticket: the last
    //      arriver may advance them while the others are still storing)
    if (!a.write_grads) {
        const float alpha = nt.lr * sqrtf(1.f - b2p) / (1.f - b1p);
        const float omb1 = 1.f - nt.beta1, omb2 = 1.f - nt.beta2;
#pragma unroll
        for (int r = 0; r < 17; ++r) {
            const float g = r < 16 ? acc[r] : bsum;
            const float gr2 = g * nt.gscale;
            mi[r] += (gr2 - mi[r]) * omb1;
            vi[r] += (gr2 * gr2 - vi[r]) * omb2;
            wgt[r] -= (mi[r] * alpha) / (sqrtf(vi[r]) + nt.eps);
            tg[r] = nt.rate * wgt[r] + nt.omr * tg[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) {
                nt.m[idx[r]] = mi[r];
                nt.v[idx[r]] = vi[r];
                nt.w[idx[r]] = wgt[r];
                if (mix) nt.target[idx[r]] = tg[r];
            }
        if (bias_lane) {
            nt.m[ib] = mi[16];
            nt.v[ib] = vi[16];
            nt.w[ib] = wgt[16];
            if (mix) nt.target[ib] = tg[16];
        }
    }
    RLX_DW_STAMP(4);
    __syncthreads();
    RLX_DW_STAMP(5);
    if (!last_s) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (wave < a.n_nets && !a.write_grads) {
        // one wave per network: its blocks' sums of squares in block order (lanes stride, then a fixed butterfly)
        const DwNet &nn = a.net[wave];
        float s = 0.f;
        for (int t = nn.tile_lo + lane; t < nn.tile_hi; t += 64)
            s += __hip_atomic_load(&a.norm_part[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) {
            if (nn.norm_out) nn.norm_out[0] = sqrtf(s);
            const float p1 = nn.state[0], p2 = nn.state[1];
            nn.state[0] = p1 * nn.beta1;                // AdamOptimizer._finish: the beta powers advance
            nn.state[1] = p2 * nn.beta2;
        }
    }
    if (wave == 3 && a.loss_out) {
        float total = 0.f;
        for (int s = 0; s < a.loss_streams; ++s) {
            float v = 0.f;
            for (int i = lane; i < a.loss_parts; i += 64) v += a.loss_part[(size_t)s * a.loss_parts + i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
            v = a.loss_scale * v / (float)a.B;
            total += v;
            if (lane == 0) a.loss_out[s] = v;
        }
        if (lane == 0) a.loss_out[a.loss_streams] = total;
    }
    if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
inline Mlp3 to_dev(const rlx_mlp3 &m) {
    Mlp3 d;
    d.o_w1 = m.off_w1; d.o_b1 = m.off_b1; d.o_w2 = m.off_w2; d.o_b2 = m.off_b2; d.o_w3 = m.off_w3; d.o_b3 = m.off_b3;
    d.s1 = m.tower_stride1; d.s2 = m.tower_stride2; d.s3 = m.tower_stride3;
    d.d_in = m.d_in; d.h1 = m.h1; d.h2 = m.h2; d.d_out = m.d_out;
    return d;
}

inline int tiles_of(int K, int N) { return ((K + kDwBlock - 1) / kDwBlock) * ((N + kDwBlock - 1) / kDwBlock); }

struct DwBuilder {
    DwArgs a;
    DwBuilder() {
        a.n_jobs = a.n_nets = a.tiles = 0;
        a.loss_part = nullptr; a.loss_out = nullptr; a.loss_streams = a.loss_parts = 0; a.loss_scale = 1.f;
        a.stamps = nullptr;
    }
    int add_net(const rlx_fused_net &n) {
        DwNet &d = a.net[a.n_nets];
        d.w = n.weights; d.m = n.adam_m; d.v = n.adam_v; d.state = n.adam_state; d.target = n.target_weights; d.grads = n.grads;
        d.norm_out = n.norm_out;
        d.lr = n.learning_rate; d.beta1 = n.beta1; d.beta2 = n.beta2; d.eps = n.epsilon; d.gscale = n.grad_scale;
        d.rate = n.mix_rate; d.omr = (float)(1.0 - (double)n.mix_rate);
        d.tile_lo = a.tiles; d.tile_hi = a.tiles;
        return a.n_nets++;
    }
    void add_partial_job(int net, const float *A, long long lda, const float *G, long long ldg, int parts, long long part_stride,
                         const float *mask, long long o_w, long long o_b, int K, int N) {
        add_job(net, A, lda, G, ldg, o_w, o_b, K, N);
        DwJob &j = a.job[a.n_jobs - 1];
        j.g_parts = parts; j.g_part_stride = part_stride; j.mask = mask;
    }
    void add_job(int net, const float *A, long long lda, const float *G, long long ldg, long long o_w, long long o_b, int K, int N) {
        DwJob &j = a.job[a.n_jobs++];
        j.A = A; j.lda = lda; j.G = G; j.ldg = ldg; j.o_w = o_w; j.o_b = o_b; j.K = K; j.N = N; j.net = net;
        j.tile0 = a.tiles; j.tiles_n = (N + kDwBlock - 1) / kDwBlock; j.g_parts = 1; j.mask = nullptr; j.g_part_stride = 0;
        a.tiles += tiles_of(K, N);
        a.net[net].tile_hi = a.tiles;
    }
};

inline bool wide_ok(int n) { return n >= 64 && n <= kMaxWidth && (n % 4) == 0; }

template <typename K>
inline hipError_t set_lds(K kernel, size_t bytes = kChainLdsBytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct Td3Ws {
    long long zT, xm, h1c, h2c, qp, qTp, dq, dh2, dh1p, loss_part, xa, h1a, h2a, za, ya, h1q, dh1qp, dz3, dh2a, dh1ap, norm_part,
        stamps, total;
};
inline int pad64(int n) { return (n + 63) & ~63; }
inline int pad128(int n) { return (n + 127) & ~127; }
inline Td3Ws td3_layout(int B, int D, int A, const rlx_mlp3 &am, const rlx_mlp3 &cm) {
    Td3Ws w;
    long long o = 0;
    auto take = [&](long long n) { const long long at = o; o += (n + 3) & ~3LL; return at; };
    const int nrb = (B + R - 1) / R;
    const long long Bp = pad128(B), ldx = pad64(A + D), l1c = pad64(cm.h1), l2c = pad64(cm.h2), lxa = pad64(D), l1a = pad64(am.h1),
                    l2a = pad64(am.h2);
    // (the buffers the weight-gradient launch reads come first and padded; they rely on the workspace having been ZEROED
    // when it was allocated — nothing writes their padding)
    w.xm = take(Bp * ldx);
    w.h1c = take(2 * Bp * l1c); w.h2c = take(2 * Bp * l2c);
    w.dq = take(2 * Bp * 64); w.dh2 = take(2 * Bp * l2c); w.dh1p = take(2LL * kSplit * Bp * l1c);
    w.xa = take(Bp * lxa); w.h1a = take(Bp * l1a); w.h2a = take(Bp * l2a);
    w.dz3 = take(Bp * 64); w.dh2a = take(Bp * l2a); w.dh1ap = take((long long)kSplit * Bp * l1a);
    w.zT = take((long long)B * kSplit * A);
    w.qp = take(2LL * kSplit * B); w.qTp = take(2LL * kSplit * B);
    w.loss_part = take(2LL * nrb);
    w.za = take((long long)B * kSplit * A); w.ya = take((long long)B * A);
    w.h1q = take((long long)B * cm.h1); w.dh1qp = take((long long)kSplit * B * cm.h1);
    const long long tc = 2LL * (tiles_of(A + D, cm.h1) + tiles_of(cm.h1, cm.h2) + tiles_of(cm.h2, 1));
    const long long ta = tiles_of(D, am.h1) + tiles_of(am.h1, am.h2) + tiles_of(am.h2, A);
    w.norm_part = take(tc > ta ? tc : ta);
    w.stamps = take(2 * 96);                     // 96 int64
    w.total = o;
    return w;
}

inline int td3_check(const rlx_td3_fused_desc &d) {
    const rlx_mlp3 &am = d.actor_mlp, &cm = d.critic_mlp;
    if (d.batch < 1 || d.batch > 4096 || d.obs_dim < 1 || d.act_dim < 1 || d.act_dim > 16) return 0;
    if (d.obs_dim + d.act_dim > kMaxWidth - 8) return 0;
    if (am.d_in != d.obs_dim || am.d_out != d.act_dim || cm.d_in != d.obs_dim + d.act_dim || cm.d_out != 1) return 0;
    if (!wide_ok(am.h1) || !wide_ok(cm.h1)) return 0;
    if (!wide_ok(am.h2) || !wide_ok(cm.h2)) return 0;
    if (am.h2 < 32 * kSplit || cm.h2 < 32 * kSplit) return 0;       // every column slice of a middle layer: >= 32 columns
    return 1;
}

static bool g_stamps = false;            // rlx_fused_phase_stamps(1): workgroup 0 of every chain kernel records its phase boundaries

inline Td3Dev td3_dev(const rlx_td3_fused_desc &d) {
    Td3Dev p;
    const Td3Ws w = td3_layout(d.batch, d.obs_dim, d.act_dim, d.actor_mlp, d.critic_mlp);
    p.aw = d.actor.weights; p.awt = d.actor.target_weights; p.cw = d.critic.weights; p.cwt = d.critic.target_weights;
    p.am = to_dev(d.actor_mlp); p.cm = to_dev(d.critic_mlp);
    p.obs = d.obs; p.next_obs = d.next_obs; p.actions = d.actions; p.rewards = d.rewards; p.dones = d.game_overs;
    p.noise = d.noise; p.low = d.action_low; p.high = d.action_high;
    p.noise_clip = d.noise_clip; p.discount = d.discount; p.clip_lo = d.clip_low; p.clip_hi = d.clip_high;
    p.nonzero_terminal = d.use_non_zero_discount_for_terminal_states; p.has_clip = d.has_clip;
    p.actor_scale = d.actor_scale;
    p.B = d.batch; p.D = d.obs_dim; p.A = d.act_dim; p.nrb = (d.batch + R - 1) / R;
    p.Bp = pad128(d.batch); p.ldx = pad64(d.obs_dim + d.act_dim); p.ld1c = pad64(d.critic_mlp.h1); p.ld2c = pad64(d.critic_mlp.h2);
    p.ldxa = pad64(d.obs_dim); p.ld1a = pad64(d.actor_mlp.h1); p.ld2a = pad64(d.actor_mlp.h2);
    float *ws = d.workspace;
    p.zT = ws + w.zT; p.xm = ws + w.xm; p.h1c = ws + w.h1c; p.h2c = ws + w.h2c; p.qp = ws + w.qp; p.qTp = ws + w.qTp;
    p.dq = ws + w.dq; p.dh2 = ws + w.dh2; p.dh1p = ws + w.dh1p; p.loss_part = ws + w.loss_part;
    p.xa = ws + w.xa; p.h1a = ws + w.h1a; p.h2a = ws + w.h2a; p.za = ws + w.za; p.ya = ws + w.ya; p.h1q = ws + w.h1q; p.dh1qp = ws + w.dh1qp;
    p.dz3 = ws + w.dz3; p.dh2a = ws + w.dh2a; p.dh1ap = ws + w.dh1ap;
    p.td_targets = d.td_targets; p.q_min = d.q_min; p.neg_dq_da = d.neg_action_grad;
    p.stamps = g_stamps ? reinterpret_cast<long long *>(ws + w.stamps) : nullptr;
    return p;
}

}  // namespace

extern "C" {

int rlx_td3_fused_supported(const rlx_td3_fused_desc *d) { return d ? td3_check(*d) : 0; }

// Measurement switch: the chain kernels' workgroup 0 writes s_memtime at its phase boundaries into the last 96 int64 of
// the workspace (tools/ac_fused_phases.py reads them).  Off by default.
int rlx_fused_phase_stamps(int enable) {
    g_stamps = enable != 0;
    return RLX_OK;
}

int rlx_td3_fused_workspace_floats(const rlx_td3_fused_desc *d, long long *floats_host) {
    RLX_REQUIRE(d && floats_host, "rlx_td3_fused_workspace_floats: null pointer");
    RLX_REQUIRE(td3_check(*d), "rlx_td3_fused_workspace_floats: unsupported shape");
    *floats_host = td3_layout(d->batch, d->obs_dim, d->act_dim, d->actor_mlp, d->critic_mlp).total;
    return RLX_OK;
}

static int td3_common_checks(const rlx_td3_fused_desc &d, const char *who) {
    RLX_REQUIRE(td3_check(d), "%s: unsupported shape (batch=%d obs=%d act=%d actor %d-%d critic %d-%d)", who, d.batch,
                d.obs_dim, d.act_dim, d.actor_mlp.h1, d.actor_mlp.h2, d.critic_mlp.h1, d.critic_mlp.h2);
    RLX_REQUIRE(d.actor.weights && d.actor.target_weights && d.critic.weights && d.critic.target_weights && d.obs &&
                    d.next_obs && d.actions && d.rewards && d.game_overs && d.workspace && d.td_targets && d.q_min &&
                    d.neg_action_grad && d.loss,
                "%s: null pointer", who);
    RLX_REQUIRE(!d.noise || (d.action_low && d.action_high), "%s: smoothing needs the action bounds", who);
    const Td3Ws w = td3_layout(d.batch, d.obs_dim, d.act_dim, d.actor_mlp, d.critic_mlp);
    RLX_REQUIRE(d.workspace_floats >= w.total, "%s: workspace of %lld floats, need %lld", who, d.workspace_floats, w.total);
    return RLX_OK;
}

static int launch_dw(DwBuilder &b, int B, int write_grads, float *norm_part, unsigned *ticket, hipStream_t st) {
    b.a.B = B; b.a.write_grads = write_grads; b.a.norm_part = norm_part; b.a.ticket = ticket;
    RLX_LAUNCH((mlp_dw_adam_kernel), b.a.tiles, kDwThreads, 0, st, b.a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// TD3Agent.learn_from_batch's critic half (td3_agent.py:157-184): three launches.  write_grads != 0: the weight gradients
// go to critic.grads instead of through Adam (the caller all-reduces them and applies its own optimiser step).
int rlx_td3_fused_critic_update(const rlx_td3_fused_desc *d_host, int write_grads, void *stream) {
    RLX_REQUIRE(d_host != nullptr, "rlx_td3_fused_critic_update: null descriptor");
    const rlx_td3_fused_desc &d = *d_host;
    if (int rc = td3_common_checks(d, "rlx_td3_fused_critic_update")) return rc;
    RLX_REQUIRE(write_grads ? d.critic.grads != nullptr
                            : (d.critic.adam_m && d.critic.adam_v && d.critic.adam_state && d.critic.ticket),
                "rlx_td3_fused_critic_update: null optimiser pointer");
    static bool configured = false;
    if (!configured) {
        RLX_HIP(set_lds(td3_forward1_kernel, kFwdLdsBytes));
        RLX_HIP(set_lds(td3_forward2_kernel, kFwdLdsBytes));
        RLX_HIP(set_lds(td3_critic_backward_kernel));
        configured = true;
    }
    const Td3Dev p = td3_dev(d);
    hipStream_t st = rlx::as_stream(stream);
    RLX_LAUNCH((td3_forward1_kernel), 3 * kSplit * p.nrb, T, kFwdLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((td3_forward2_kernel), 2 * kSplit * p.nrb, T, kFwdLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((td3_critic_backward_kernel), 2 * kSplit * p.nrb, T, kChainLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    DwBuilder b;
    const int net = b.add_net(d.critic);
    const rlx_mlp3 &cm = d.critic_mlp;
    const int B = d.batch, M = d.obs_dim + d.act_dim;
    for (int s = 0; s < 2; ++s) {
        const size_t Bp = p.Bp;
        b.add_partial_job(net, p.xm, p.ldx, p.dh1p + (size_t)s * kSplit * Bp * p.ld1c, p.ld1c, kSplit, (long long)Bp * p.ld1c,
                          p.h1c + (size_t)s * Bp * p.ld1c, cm.off_w1 + s * cm.tower_stride1, cm.off_b1 + s * cm.tower_stride1, M,
                          cm.h1);
        b.add_job(net, p.h1c + (size_t)s * Bp * p.ld1c, p.ld1c, p.dh2 + (size_t)s * Bp * p.ld2c, p.ld2c,
                  cm.off_w2 + s * cm.tower_stride2, cm.off_b2 + s * cm.tower_stride2, cm.h1, cm.h2);
        b.add_job(net, p.h2c + (size_t)s * Bp * p.ld2c, p.ld2c, p.dq + (size_t)s * Bp * 64, 64, cm.off_w3 + s * cm.tower_stride3,
                  cm.off_b3 + s * cm.tower_stride3, cm.h2, 1);
    }
    b.a.loss_part = p.loss_part; b.a.loss_out = d.loss; b.a.loss_streams = 2; b.a.loss_parts = p.nrb; b.a.loss_scale = 1.f;
    const Td3Ws w = td3_layout(d.batch, d.obs_dim, d.act_dim, d.actor_mlp, d.critic_mlp);
    b.a.stamps = p.stamps ? p.stamps + 80 : nullptr;
    return launch_dw(b, B, write_grads, d.workspace + w.norm_part, d.critic.ticket, st);
}

// The actor half (td3_agent.py:186-207): two launches.
int rlx_td3_fused_actor_update(const rlx_td3_fused_desc *d_host, int write_grads, void *stream) {
    RLX_REQUIRE(d_host != nullptr, "rlx_td3_fused_actor_update: null descriptor");
    const rlx_td3_fused_desc &d = *d_host;
    if (int rc = td3_common_checks(d, "rlx_td3_fused_actor_update")) return rc;
    RLX_REQUIRE(write_grads ? d.actor.grads != nullptr
                            : (d.actor.adam_m && d.actor.adam_v && d.actor.adam_state && d.actor.ticket),
                "rlx_td3_fused_actor_update: null optimiser pointer");
    static bool configured = false;
    if (!configured) {
        RLX_HIP(set_lds(td3_actor_forward_kernel, kFwdLdsBytes));
        RLX_HIP(set_lds(td3_actor_q_kernel));
        RLX_HIP(set_lds(td3_actor_backward_kernel));
        configured = true;
    }
    const Td3Dev p = td3_dev(d);
    hipStream_t st = rlx::as_stream(stream);
    RLX_LAUNCH((td3_actor_forward_kernel), kSplit * p.nrb, T, kFwdLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((td3_actor_q_kernel), kSplit * p.nrb, T, kChainLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((td3_actor_backward_kernel), kSplit * p.nrb, T, kChainLdsBytes, st, p);
    RLX_LAUNCH_CHECK();
    DwBuilder b;
    const int net = b.add_net(d.actor);
    const rlx_mlp3 &am = d.actor_mlp;
    const int B = d.batch;
    b.add_partial_job(net, p.xa, p.ldxa, p.dh1ap, p.ld1a, kSplit, (long long)p.Bp * p.ld1a, p.h1a, am.off_w1, am.off_b1, d.obs_dim,
                      am.h1);
    b.add_job(net, p.h1a, p.ld1a, p.dh2a, p.ld2a, am.off_w2, am.off_b2, am.h1, am.h2);
    b.add_job(net, p.h2a, p.ld2a, p.dz3, 64, am.off_w3, am.off_b3, am.h2, d.act_dim);
    const Td3Ws w = td3_layout(d.batch, d.obs_dim, d.act_dim, d.actor_mlp, d.critic_mlp);
    return launch_dw(b, B, write_grads, d.workspace + w.norm_part, d.actor.ticket, st);
}

}  // extern "C"
