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
#define RLX_STAMP_WG(wg, slot) do { if (p.stamps && blockIdx.x == (wg) && threadIdx.x == 0) p.stamps[(slot)] = (long long)__builtin_readcyclecounter(); } while (0)

// the 4 rows' partial head sums of one column slice: out[r][j] = sum_{k in slice} h[r][k] W3[k][j]  (j < NO <= 16)
template <int RR = R>
__device__ __forceinline__ void head_partial(const float *hs, int hp, const float *__restrict__ W3, int n_lo, int n_hi, int NO,
                                             float *__restrict__ out, long long ld, int row0, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < RR * NO; o += T / 64) {
        const int r = o / NO, j = o - r * NO;
        float sum = 0.f;
        for (int k = n_lo + lane; k < n_hi; k += 64) sum = fmaf(hs[r * hp + (k - n_lo)], W3[(size_t)k * NO + j], sum);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane == 0 && row0 + r < B) out[(long long)(row0 + r) * ld + j] = sum;
    }
}

// the same sums for a WIDER head (SAC policy: 2 A outputs) whose slice rows ws [n_hi - n_lo][NO] already sit in LDS: a thread per
// (row, output), k ascending — no butterflies, no strided global reads (head_partial on 34 outputs: 34 rounds of a 136-byte
// stride gather + 6 shuffles per wave, ~24 us of sac_layer2_kernel's 33)
template <int RR = R>
__device__ __forceinline__ void head_partial_lds(const float *hs, int hp, const float *ws, int wc, int NO,
                                                 float *__restrict__ out, long long ld, int row0, int B) {
    for (int o = threadIdx.x; o < RR * NO; o += T) {
        const int r = o / NO, j = o - r * NO;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                  // wc % 4 == 0 (slice_lo)
        for (int k = 0; k < wc; k += 4) {
            s0 = fmaf(hs[r * hp + k], ws[k * NO + j], s0);
            s1 = fmaf(hs[r * hp + k + 1], ws[(k + 1) * NO + j], s1);
            s2 = fmaf(hs[r * hp + k + 2], ws[(k + 2) * NO + j], s2);
            s3 = fmaf(hs[r * hp + k + 3], ws[(k + 3) * NO + j], s3);
        }
        if (row0 + r < B) out[(long long)(row0 + r) * ld + j] = (s0 + s1) + (s2 + s3);
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
    const float *loss2_part; float *loss2_out; int loss2_parts;        // a second, single-stream loss (SAC: the V network's)
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
        if (a.loss2_out) {
            float v = 0.f;
            for (int i = lane; i < a.loss2_parts; i += 64) v += a.loss2_part[i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0) a.loss2_out[0] = v / (float)a.B;
        }
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
        a.stamps = nullptr; a.loss2_part = nullptr; a.loss2_out = nullptr; a.loss2_parts = 0;
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

// =================================================================================================================
// Soft Actor-Critic (rl_coach/agents/soft_actor_critic_agent.py:168-280) on the same building blocks: 8 batch rows per
// workgroup (B = 256: two 4-row MFMA block sets on the same weight registers), every wide layer in kSplit column slices.
// Here the FIRST layers are wide too (376 x 256), so a network's forward pass is two launches (a slice of layer 1 needs
// nothing, a slice of layer 2 needs all of layer 1) — and every launch carries the same level of ALL networks:
//   sac_layer1_kernel    h1 slices of policy(s), V(s), V_target(s') and relu(obs_fc_t(s)) of both Q towers
//   sac_layer2_kernel    h2 slices + partial head sums of policy / V / V_target; Q_t(s, a): act_fc, sum, fc1 slice, partial Q_t
//   sac_q_pi_kernel      policy head (first noise draw, :186-190) -> a ~ pi, log pi; Q_t(s, a ~ pi) up to partial Q_t (:198-200)
//   sac_q_grad_kernel    min(Q_1, Q_2), V targets (:244), d mean(min) / d Q_t -> this slice's share of d / d (fc1 input) -> its
//                        share of dQ/da (:216-217; linear in the hidden gradient, which therefore never goes to memory)
//   sac_backward_kernel  role P: sum of the dQ/da shares -> the policy head's gradient on the 2nd / 3rd noise draw (:210-227) -> dz2, partial
//                        dz1; role V: loss and backward of V (:250); role Q_t: y = r + (1 - done) gamma V_target(s')
//                        (:259-266), loss and backward of Q_t (:268)
//   mlp_dw_adam_kernel   all fourteen layers' weight gradients + the three Adam steps (+ V's soft target update)
// Every forward pass sees the weights as they were before this update (the reference applies the policy's step before it
// forms the V / Q targets, but no target reads the policy's weights: soft_actor_critic_agent.py:229-266).
constexpr int RS = 8, SP = 388, HP = 64;           // rows per workgroup, activation pitch (widths <= 384), narrow-array pitch
constexpr size_t kSacFwdLds = sizeof(float) * (3 * RS * SP + kPartFloats + 4 * RS * HP);
constexpr size_t kSacBwdLds = sizeof(float) * (4 * RS * SP + 4 * RS * HP + 2 * kTileFloats);
static_assert(2 * kSacFwdLds <= 160 * 1024 && kSacBwdLds <= 160 * 1024, "SAC chain kernels: LDS budget");
constexpr float kLogSigCapMin = -20.f, kLogSigCapMax = 2.f;       // sac_head.py:26-27
constexpr float kEpsF32 = 1.1920928955078125e-07f;                 // np.finfo(np.float32).eps (utils.py:38)
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

struct SacQ {                                       // SACQHead (heads/sac_q_head.py:46-96), tower t at + t * stride
    long long o_wo, o_bo, o_wa, o_ba, o_w1, o_b1, o_wq, o_bq;
    long long s_o, s_a, s_1, s_q;
};
struct SacDev {
    const float *pw, *vw, *vwt, *qw;
    Mlp3 pm, vm;
    SacQ qm;
    const float *obs, *next_obs, *actions, *rewards;
    const unsigned char *dones;
    const double *normals;                          // [3][B][A]
    double discount;
    int resample;
    int B, D, A, H, nrb, Bp, ldx, ldh;
    // padded (rows to 128, columns to 64, zero outside): operands of the weight-gradient launch
    float *xs, *ab, *h1P, *h1V, *ho, *h2P, *h2V, *haB, *hs, *h2Q;
    float *dyP, *dz2P, *dh1Pp, *dvV, *dz2V, *dh1Vp, *dqQ, *dfc1, *dhqp;
    // plain
    float *h1VT, *zP, *vp, *vTp, *qp, *logp0, *haP, *h2Qp, *qpp, *dapp, *v_loss_part, *q_loss_part;
    float *value_targets, *log_target, *td_targets, *dq_da;
    long long *stamps;
};

// mu_logsig[r][j] = bias[j] + the slices' partial head sums in slice order  (j < 2 A) -> zz [RS][HP]
__device__ __forceinline__ void sac_head_sums(const SacDev &p, int row0, float *zz) {
    const int A2 = 2 * p.A;
    for (int e = threadIdx.x; e < RS * HP; e += T) {
        const int r = e / HP, j = e - r * HP, i = row0 + r;
        float v = 0.f;
        if (i < p.B && j < A2) {
            const float *zp = p.zP + ((size_t)i * kSplit) * A2 + j;
            v = zp[0];
#pragma unroll
            for (int q = 1; q < kSplit; ++q) v += zp[(size_t)q * A2];
            v += p.pw[p.pm.o_b3 + j];
        }
        zz[e] = v;
    }
}

__global__ void __launch_bounds__(T, 4) sac_layer1_kernel(const SacDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *x = smem, *y = x + RS * SP, *parts = y + 2 * RS * SP;
    const int role = blockIdx.x % (5 * kSplit), rb = blockIdx.x / (5 * kSplit), row0 = rb * RS;
    const int net = role / kSplit, c = role % kSplit;
    const int B = p.B, D = p.D;
    RLX_STAMP(0);
    load_rows<RS>(x, SP, net == 2 ? p.next_obs : p.obs, D, D, row0, B);
    __syncthreads();
    RLX_STAMP(1);
    if (net == 0 && c == 0)
        for (int e = threadIdx.x; e < RS * D; e += T) {
            const int r = e / D, col = e - r * D;
            if (row0 + r < B) p.xs[(size_t)(row0 + r) * p.ldx + col] = x[r * SP + col];
        }
    const float *W, *bias;
    float *out;
    int N;
    long long gld = p.ldh;
    if (net == 0) { W = p.pw + p.pm.o_w1; bias = p.pw + p.pm.o_b1; N = p.pm.h1; out = p.h1P; }
    else if (net == 1) { W = p.vw + p.vm.o_w1; bias = p.vw + p.vm.o_b1; N = p.vm.h1; out = p.h1V; }
    else if (net == 2) { W = p.vwt + p.vm.o_w1; bias = p.vwt + p.vm.o_b1; N = p.vm.h1; out = p.h1VT; gld = N; }
    else {
        const int t = net - 3;
        W = p.qw + p.qm.o_wo + t * p.qm.s_o; bias = p.qw + p.qm.o_bo + t * p.qm.s_o; N = p.H;
        out = p.ho + (size_t)t * p.Bp * p.ldh;
    }
    const int lo = slice_lo(N, c), hi = slice_lo(N, c + 1);
    RLX_STAMP(2);
    dense_fwd<RS>(x, SP, D, W + lo, N, bias + lo, hi - lo, RLX_ACT_RELU, y, SP, parts, out + lo, gld, row0, B);
    RLX_STAMP(3);
}

__global__ void __launch_bounds__(T, 4) sac_layer2_kernel(const SacDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *x = smem, *y = x + RS * SP, *z = y + RS * SP, *parts = z + RS * SP, *small = parts + kPartFloats;
    const int tid = threadIdx.x, role = blockIdx.x % (5 * kSplit), rb = blockIdx.x / (5 * kSplit), row0 = rb * RS;
    const int net = role / kSplit, c = role % kSplit;
    const int B = p.B, A = p.A, H = p.H;
    if (net < 3) {
        const Mlp3 &m = net == 0 ? p.pm : p.vm;
        const float *w = net == 0 ? p.pw : (net == 1 ? p.vw : p.vwt);
        const float *h1 = net == 0 ? p.h1P : (net == 1 ? p.h1V : p.h1VT);
        const int lo = slice_lo(m.h2, c), hi = slice_lo(m.h2, c + 1);
        RLX_STAMP(8);
        // the policy head's rows of this slice ([hi - lo][2 A], contiguous) travel with the input rows: z holds them
        constexpr int kW3 = (RS * SP + T - 1) / T;
        float w3v[kW3];
        const int n3 = net == 0 ? (hi - lo) * 2 * A : 0;
        const float *w3g = w + m.o_w3 + (size_t)lo * 2 * A;
#pragma unroll
        for (int j = 0; j < kW3; ++j)
            if (j * T < n3) w3v[j] = w3g[min(tid + j * T, n3 - 1)];
        load_rows<RS>(x, SP, h1, net == 2 ? m.h1 : p.ldh, m.h1, row0, B);
#pragma unroll
        for (int j = 0; j < kW3; ++j)
            if (j * T < n3 && tid + j * T < n3) z[tid + j * T] = w3v[j];
        __syncthreads();
        RLX_STAMP(9);
        float *save = net == 0 ? p.h2P + lo : (net == 1 ? p.h2V + lo : nullptr);
        dense_fwd<RS>(x, SP, m.h1, w + m.o_w2 + lo, m.h2, w + m.o_b2 + lo, hi - lo, RLX_ACT_RELU, y, SP, parts, save, p.ldh, row0, B);
        RLX_STAMP(10);
        if (net == 0) head_partial_lds<RS>(y, SP, z, hi - lo, 2 * A, p.zP + (size_t)c * 2 * A, (long long)kSplit * 2 * A, row0, B);
        else head_partial<RS>(y, SP, w + m.o_w3, lo, hi, 1, (net == 1 ? p.vp : p.vTp) + (size_t)c * B, 1, row0, B);
        RLX_STAMP(11);
    } else {
        // Q_t(s, a): relu(obs_fc(s)) from the first launch + relu(act_fc(a)) -> fc1 slice -> the slice's share of Q_t
        const int t = net - 3;
        const SacQ &q = p.qm;
        const int lo = slice_lo(H, c), hi = slice_lo(H, c + 1);
        load_rows<RS>(x, SP, p.ho + (size_t)t * p.Bp * p.ldh, p.ldh, H, row0, B);
        load_rows<RS>(small, HP, p.actions, A, A, row0, B);
        __syncthreads();
        if (t == 0 && c == 0)
            for (int e = tid; e < RS * A; e += T) {
                const int r = e / A, col = e - r * A;
                if (row0 + r < B) p.ab[(size_t)(row0 + r) * 64 + col] = small[r * HP + col];
            }
        dense_fwd<RS>(small, HP, A, p.qw + q.o_wa + t * q.s_a, H, p.qw + q.o_ba + t * q.s_a, H, RLX_ACT_RELU, z, SP, parts,
                      c == 0 ? p.haB + (size_t)t * p.Bp * p.ldh : nullptr, p.ldh, row0, B);
        for (int e = tid; e < RS * pad16(H); e += T) {       // qi_obs_emb + qi_act_emb (sac_q_head.py:63-67)
            const int r = e / pad16(H), k = e - r * pad16(H);
            const float v = k < H ? x[r * SP + k] + z[r * SP + k] : 0.f;
            x[r * SP + k] = v;
            if (c == 0 && k < H && row0 + r < B) p.hs[((size_t)t * p.Bp + row0 + r) * p.ldh + k] = v;
        }
        __syncthreads();
        dense_fwd<RS>(x, SP, H, p.qw + q.o_w1 + t * q.s_1 + lo, H, p.qw + q.o_b1 + t * q.s_1 + lo, hi - lo, RLX_ACT_RELU, y, SP, parts,
                      p.h2Q + (size_t)t * p.Bp * p.ldh + lo, p.ldh, row0, B);
        head_partial<RS>(y, SP, p.qw + q.o_wq + t * q.s_q, lo, hi, 1, p.qp + ((size_t)t * kSplit + c) * B, 1, row0, B);
    }
}

// SACPolicyHead on the first noise draw (heads/sac_head.py:60-97): a = tanh(mu + sigma n), log pi; then Q_t(s, a)
__global__ void __launch_bounds__(T, 4) sac_q_pi_kernel(const SacDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *x = smem, *y = x + RS * SP, *z = y + RS * SP, *parts = z + RS * SP, *small = parts + kPartFloats;
    float *zz = small, *act0 = small + RS * HP, *t1 = small + 2 * RS * HP, *t2 = small + 3 * RS * HP;
    const int tid = threadIdx.x, role = blockIdx.x % (2 * kSplit), rb = blockIdx.x / (2 * kSplit), row0 = rb * RS;
    const int t = role / kSplit, c = role % kSplit;
    const int B = p.B, A = p.A, H = p.H;
    const SacQ &q = p.qm;
    const int lo = slice_lo(H, c), hi = slice_lo(H, c + 1);
    RLX_STAMP(16);
    sac_head_sums(p, row0, zz);
    load_rows<RS>(x, SP, p.ho + (size_t)t * p.Bp * p.ldh, p.ldh, H, row0, B);
    __syncthreads();
    RLX_STAMP(17);
    for (int e = tid; e < RS * HP; e += T) {
        const int r = e / HP, a = e - r * HP, i = row0 + r;
        float av = 0.f, lp = 0.f, cr = 0.f;
        if (i < B && a < A) {
            const float mu = zz[r * HP + a];
            const float ls = fminf(fmaxf(zz[r * HP + A + a], kLogSigCapMin), kLogSigCapMax);   // :65-66
            const float sd = expf(ls);
            const float nz = (float)p.normals[(size_t)i * A + a];
            const float raw = mu + sd * nz;                                             // sample() :79
            av = tanhf(raw);                                                            // :82
            const float zv = (raw - mu) / sd;
            lp = -0.5f * zv * zv - ls - kHalfLog2Pi;                                    // log_prob :90
            cr = logf(1.f - av * av + kEpsF32);                                         // :58
        }
        act0[e] = av; t1[e] = lp; t2[e] = cr;
    }
    __syncthreads();
    RLX_STAMP(18);
    if (t == 0 && c == 0 && tid < RS && row0 + tid < B) {      // the sums in action order, as the head's row loop
        float lp = 0.f, cr = 0.f;
        for (int a = 0; a < A; ++a) { lp += t1[tid * HP + a]; cr += t2[tid * HP + a]; }
        p.logp0[row0 + tid] = lp - cr;
    }
    dense_fwd<RS>(act0, HP, A, p.qw + q.o_wa + t * q.s_a, H, p.qw + q.o_ba + t * q.s_a, H, RLX_ACT_RELU, z, SP, parts,
                  c == 0 ? p.haP + (size_t)t * B * H : nullptr, H, row0, B);
    RLX_STAMP(19);
    for (int e = tid; e < RS * pad16(H); e += T) {
        const int r = e / pad16(H), k = e - r * pad16(H);
        x[r * SP + k] = k < H ? x[r * SP + k] + z[r * SP + k] : 0.f;
    }
    __syncthreads();
    dense_fwd<RS>(x, SP, H, p.qw + q.o_w1 + t * q.s_1 + lo, H, p.qw + q.o_b1 + t * q.s_1 + lo, hi - lo, RLX_ACT_RELU, y, SP, parts,
                  p.h2Qp + (size_t)t * B * H + lo, H, row0, B);
    RLX_STAMP(20);
    head_partial<RS>(y, SP, p.qw + q.o_wq + t * q.s_q, lo, hi, 1, p.qpp + ((size_t)t * kSplit + c) * B, 1, row0, B);
    RLX_STAMP(21);
}

// min(Q_1, Q_2)(s, a ~ pi), the V targets, d mean(min) / d Q_t and its way down to the input of fc1 (slice partial)
__global__ void __launch_bounds__(T) sac_q_grad_kernel(const SacDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *b = smem, *cb = b + RS * SP, *d = cb + RS * SP, *small = d + 2 * RS * SP, *wt = small + 4 * RS * HP;
    const int tid = threadIdx.x, role = blockIdx.x % (2 * kSplit), rb = blockIdx.x / (2 * kSplit), row0 = rb * RS;
    const int t = role / kSplit, c = role % kSplit;
    const int B = p.B, H = p.H;
    const SacQ &q = p.qm;
    const int lo = slice_lo(H, c), hi = slice_lo(H, c + 1), wc = hi - lo;
    const int A = p.A, lane = tid & 63, wave = tid >> 6;
    float *hm = d + RS * SP, *red = small;           // act_fc's output rows (relu mask); the waves' partial dQ/da tiles
    RLX_STAMP(24);
    // B operands of the dQ/da product at the end (act_fc^T of tower t, 16 x 16 x 4 MFMAs: lane l supplies
    // act_fc[a = 16 u + l % 16][n = 4 step + l / 16]; wave w owns steps [w NS, (w + 1) NS)): requested now, used last
    constexpr int kNS = (SP - 4 + 31) / 32;
    const int NS = (H + 31) >> 5;
    float wb[kNS][2];
#pragma unroll
    for (int st = 0; st < kNS; ++st)
        if (st < NS) {
            const int n = min((wave * NS + st) * 4 + (lane >> 4), H - 1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                wb[st][u] = p.qw[q.o_wa + t * q.s_a + (size_t)min(16 * u + (lane & 15), A - 1) * H + n];
        }
    load_rows<RS>(b, SP, p.h2Qp + (size_t)t * B * H + lo, H, wc, row0, B);
    load_rows<RS>(hm, SP, p.haP + (size_t)t * B * H, H, H, row0, B);
    if (tid < RS) {
        const int i = row0 + tid;
        float dq = 0.f;
        if (i < B) {
            float qv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float *pp = p.qpp + (size_t)u * kSplit * B + i;
                float v = pp[0];
#pragma unroll
                for (int z = 1; z < kSplit; ++z) v += pp[(size_t)z * B];
                qv[u] = v + p.qw[q.o_bq + u * q.s_q];
            }
            const bool first = qv[0] <= qv[1];               // tf.minimum: x <= y -> x gets the gradient
            const float m = first ? qv[0] : qv[1];
            if (t == 0 && c == 0) {
                p.log_target[i] = m;
                p.value_targets[i] = m - p.logp0[i];          // :244 (fp32 numpy arrays in the reference)
            }
            dq = (first == (t == 0)) ? 1.0f / (float)B : 0.f;
        }
        small[tid] = dq;
    }
    __syncthreads();
    RLX_STAMP(25);
    const float *wq = p.qw + q.o_wq + t * q.s_q + lo;
    for (int e = tid; e < RS * pad16(wc); e += T) {
        const int r = e / pad16(wc), k = e - r * pad16(wc);
        cb[r * SP + k] = (k < wc && b[r * SP + k] > 0.f) ? small[r] * wq[k] : 0.f;
    }
    __syncthreads();
    RLX_STAMP(26);
    dense_bwdT<RS>(cb, SP, wc, p.qw + q.o_w1 + t * q.s_1 + lo, H, H, nullptr, 0, d, SP, wt, nullptr, 0, row0, B);
    RLX_STAMP(27);
    // This slice's share of d mean(min Q) / d a (:216-217): (d under act_fc's relu) x act_fc^T.  The product is linear in d,
    // so the slices' shares are summed by the consumer (sac_backward_kernel) — the hidden gradient never goes to memory.
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int r = lane & 7;                              // A rows 8 .. 15 repeat rows 0 .. 7 (their results are dropped)
#pragma unroll
        for (int st = 0; st < kNS; ++st)
            if (st < NS) {
                const int n = (wave * NS + st) * 4 + (lane >> 4), nn = min(n, H - 1);
                const float av = (n < H && hm[r * SP + nn] > 0.f) ? d[r * SP + nn] : 0.f;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[st][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[st][1], acc[1], 0, 0, 0);
            }
        if (lane < 32) {                                     // D: row = 4 (lane / 16) + v, column = lane % 16
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[(wave * RS + 4 * (lane >> 4) + v) * 32 + 16 * u + (lane & 15)] = acc[u][v];
        }
    }
    __syncthreads();
    for (int o = tid; o < RS * A; o += T) {
        const int r = o / A, a = o - r * A;
        float v = red[r * 32 + a];
#pragma unroll
        for (int w = 1; w < T / 64; ++w) v += red[(w * RS + r) * 32 + a];
        if (row0 + r < B) p.dapp[(((size_t)t * kSplit + c) * B + row0 + r) * A + a] = v;
    }
    RLX_STAMP(28);
}

__global__ void __launch_bounds__(T) sac_backward_kernel(const SacDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *b = smem, *cb = b + RS * SP, *d = cb + RS * SP, *e2 = d + RS * SP, *small = e2 + RS * SP, *wt = small + 4 * RS * HP;
    const int tid = threadIdx.x, role = blockIdx.x % (4 * kSplit), rb = blockIdx.x / (4 * kSplit), row0 = rb * RS;
    const int kind = role / kSplit, c = role % kSplit;
    const int B = p.B, A = p.A, H = p.H;
    if (kind == 0) {
        // ---- policy: dQ/da at the sampled actions, the head's gradient, dz2 slice, partial dz1
        const SacQ &q = p.qm;
        const Mlp3 &m = p.pm;
        const int lo = slice_lo(m.h2, c), hi = slice_lo(m.h2, c + 1), wc = hi - lo;
        float *zz = small, *dyb = small + 3 * RS * HP;
        float *w3s = wt;                            // the head's rows of this slice [wc][2 A]
        RLX_STAMP(32);
        // ---- every operand of the head's gradient is requested before the first one is waited for (a dependent global
        // load costs ~1 us here; the first version of this branch paid ~40 of them in a row: profiles/r06_sac_backward_phases.txt)
        static_assert(RS * HP == T, "one element of the head's output per thread");
        const int A2 = 2 * A;
        const int zr = tid / HP, zj = tid - zr * HP, zi = row0 + zr;
        float zpart[kSplit], zbias, dap[2 * kSplit];
        double nz1, nz2;
        {
            const float *zp = p.zP + ((size_t)min(zi, B - 1) * kSplit) * A2 + min(zj, A2 - 1);
#pragma unroll
            for (int z = 0; z < kSplit; ++z) zpart[z] = zp[(size_t)z * A2];
            zbias = p.pw[m.o_b3 + min(zj, A2 - 1)];
            const size_t ne = (size_t)min(zi, B - 1) * A + min(zj, A - 1);
            nz1 = p.normals[(size_t)(p.resample ? 1 : 0) * B * A + ne];
            nz2 = p.normals[(size_t)(p.resample ? 2 : 0) * B * A + ne];
#pragma unroll
            for (int z = 0; z < 2 * kSplit; ++z) dap[z] = p.dapp[(size_t)z * B * A + ne];       // sac_q_grad_kernel's shares
        }
        float bv[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = min(tid + it * T, RS * wc - 1), r = e / wc, col = e - r * wc;
            bv[it] = p.h2P[(size_t)min(row0 + r, B - 1) * p.ldh + lo + col];
        }
        constexpr int kW3 = 9;                      // (h2 / kSplit + 4) 2 A <= 9 T (sac_check)
        float w3v[kW3];
        const int n3 = wc * A2;
        const float *w3g = p.pw + m.o_w3 + (size_t)lo * A2;
#pragma unroll
        for (int j = 0; j < kW3; ++j)
            if (j * T < n3) w3v[j] = w3g[min(tid + j * T, n3 - 1)];
        __builtin_amdgcn_sched_barrier(0);
        {
            float v = zpart[0];
#pragma unroll
            for (int z = 1; z < kSplit; ++z) v += zpart[z];
            zz[tid] = (zi < B && zj < A2) ? v + zbias : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + it * T, r = e / wc, col = e - r * wc;
            if (e < RS * wc) b[r * SP + col] = bv[it];
        }
#pragma unroll
        for (int j = 0; j < kW3; ++j)
            if (j * T < n3 && tid + j * T < n3) w3s[tid + j * T] = w3v[j];
        // d mean(min Q) / d a: tower 0's slices in order, then tower 1's (the shared action input sums the towers)
        float g_da = dap[0];
#pragma unroll
        for (int z = 1; z < 2 * kSplit; ++z) g_da += dap[z];
        __syncthreads();
        RLX_STAMP(35);
        {
            const int r = zr, a = zj, i = zi;
            float dmu = 0.f, dls = 0.f;
            if (i < B && a < A) {
                const float g = g_da;
                if (c == 0) p.dq_da[(size_t)i * A + a] = g;
                const float mu = zz[r * HP + a], ls_raw = zz[r * HP + A + a];
                const bool inside = ls_raw >= kLogSigCapMin && ls_raw <= kLogSigCapMax;      // clip_by_value's gradient
                const float ls = fminf(fmaxf(ls_raw, kLogSigCapMin), kLogSigCapMax);
                const float sd = expf(ls);
                const float w_lp = 1.0f / (float)B;
                // weighted_gradients[5] (mean log-prob, weight 1) and - weighted_gradients[3] (actions, weights dQ/da): on
                // the second and third noise draw (resample), or both on the first
                const float n1 = (float)nz1, n2 = (float)nz2;
                {
                    const float raw = mu + sd * n1, tt = tanhf(raw), one_m = 1.f - tt * tt;
                    const float dcorr = 2.f * tt * one_m / (one_m + kEpsF32);
                    const float g_raw = w_lp * dcorr;
                    if (p.resample) {
                        dmu = g_raw;
                        dls = inside ? g_raw * sd * n1 + w_lp * (-1.f) : 0.f;
                        const float raw2 = mu + sd * n2, t2 = tanhf(raw2), om2 = 1.f - t2 * t2;
                        float g2 = 0.f;
                        g2 += -1.f * g * om2;
                        dmu = dmu + g2;
                        dls = dls + (inside ? g2 * sd * n2 + 0.f * (-1.f) : 0.f);
                    } else {
                        float gr = g_raw;
                        gr += -1.f * g * one_m;
                        dmu = gr;
                        dls = inside ? gr * sd * n1 + w_lp * (-1.f) : 0.f;
                    }
                }
            }
            if (a < A) {
                dyb[r * HP + a] = dmu;
                dyb[r * HP + A + a] = dls;
                if (c == 0 && i < B) {
                    p.dyP[(size_t)i * 64 + a] = dmu;
                    p.dyP[(size_t)i * 64 + A + a] = dls;
                }
            } else if (a >= 2 * A) {
                dyb[r * HP + a] = 0.f;
            }
        }
        __syncthreads();
        RLX_STAMP(36);
        // dz2[:, slice] = relu'(h2) * (dy W3^T): the slice's rows of the head's weights are in LDS
        for (int e = tid; e < RS * pad16(wc); e += T) {
            const int r = e / pad16(wc), k = e - r * pad16(wc);
            float acc = 0.f;
            if (k < wc)
                for (int n = 0; n < A2; ++n) acc = fmaf(dyb[r * HP + n], w3s[k * A2 + n], acc);
            const float v = (k < wc && b[r * SP + k] > 0.f) ? acc : 0.f;
            cb[r * SP + k] = v;
            if (k < wc && row0 + r < B) p.dz2P[(size_t)(row0 + r) * p.ldh + lo + k] = v;
        }
        __syncthreads();
        RLX_STAMP(37);
        dense_bwdT<RS>(cb, SP, wc, p.pw + m.o_w2 + lo, m.h2, m.h1, nullptr, 0, d, SP, wt, p.dh1Pp + (size_t)c * p.Bp * p.ldh, p.ldh,
                       row0, B);
        RLX_STAMP(38);
    } else {
        // ---- V (kind 1) or Q tower kind - 2: the regression loss of a Dense(1) head and its backward pass
        const bool isv = kind == 1;
        const int t = kind - 2;
        const SacQ &q = p.qm;
        const Mlp3 &m = p.vm;
        const int NH = isv ? m.h2 : H;
        const int lo = slice_lo(NH, c), hi = slice_lo(NH, c + 1), wc = hi - lo;
        RLX_STAMP_WG(2 * kSplit, 48);
        load_rows<RS>(b, SP, (isv ? p.h2V : p.h2Q + (size_t)t * p.Bp * p.ldh) + lo, p.ldh, wc, row0, B);
        if (tid < RS) {
            const int i = row0 + tid;
            float dv = 0.f, term = 0.f;
            if (i < B) {
                const float *pp = (isv ? p.vp : p.qp + (size_t)t * kSplit * B) + i;
                float v = pp[0];
#pragma unroll
                for (int z = 1; z < kSplit; ++z) v += pp[(size_t)z * B];
                v += isv ? p.vw[m.o_b3] : p.qw[q.o_bq + t * q.s_q];
                float y;
                if (isv) {
                    y = p.value_targets[i];
                } else {
                    const float *tp = p.vTp + i;
                    float vt = tp[0];
#pragma unroll
                    for (int z = 1; z < kSplit; ++z) vt += tp[(size_t)z * B];
                    vt += p.vwt[m.o_b3];
                    const double yy = (double)p.rewards[i] + (1.0 - (p.dones[i] ? 1.0 : 0.0)) * p.discount * (double)vt;   // :259-266
                    y = (float)yy;
                    if (t == 0 && c == 0) p.td_targets[i] = y;
                }
                const float w = isv ? 1.f : 0.5f;                      // v_head.py MSE; sac_q_head.py:91-95 0.5 * mse each
                const float e = v - y;
                term = w * (e * e);
                dv = 1.f * w * (2.f * e) / (float)B;
                if (c == 0) (isv ? p.dvV : p.dqQ + (size_t)t * p.Bp * 64)[(size_t)i * 64] = dv;
            }
            small[tid] = dv;
            small[RS + tid] = term;
        }
        __syncthreads();
        RLX_STAMP_WG(2 * kSplit, 49);
        if (tid == 0 && c == 0) {
            float sacc = 0.f;
            for (int r = 0; r < RS; ++r) sacc += small[RS + r];
            (isv ? p.v_loss_part : p.q_loss_part + (size_t)t * p.nrb)[rb] = sacc;
        }
        const float *w3 = (isv ? p.vw + m.o_w3 : p.qw + q.o_wq + t * q.s_q) + lo;
        float *save = (isv ? p.dz2V : p.dfc1 + (size_t)t * p.Bp * p.ldh) + lo;
        for (int e = tid; e < RS * pad16(wc); e += T) {
            const int r = e / pad16(wc), k = e - r * pad16(wc);
            float v = 0.f;
            if (k < wc && b[r * SP + k] > 0.f) v = small[r] * w3[k];
            cb[r * SP + k] = v;
            if (k < wc && row0 + r < B) save[(size_t)(row0 + r) * p.ldh + k] = v;
        }
        __syncthreads();
        RLX_STAMP_WG(2 * kSplit, 50);
        const float *w2 = isv ? p.vw + m.o_w2 + lo : p.qw + q.o_w1 + t * q.s_1 + lo;
        float *part = isv ? p.dh1Vp + (size_t)c * p.Bp * p.ldh : p.dhqp + ((size_t)t * kSplit + c) * p.Bp * p.ldh;
        dense_bwdT<RS>(cb, SP, wc, w2, NH, isv ? m.h1 : H, nullptr, 0, d, SP, wt, part, p.ldh, row0, B);
        RLX_STAMP_WG(2 * kSplit, 51);
    }
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


struct SacWs {
    long long xs, ab, h1P, h1V, ho, h2P, h2V, haB, hs, h2Q, dyP, dz2P, dh1Pp, dvV, dz2V, dh1Vp, dqQ, dfc1, dhqp;
    long long h1VT, zP, vp, vTp, qp, logp0, haP, h2Qp, qpp, dapp, v_loss_part, q_loss_part, norm_part, stamps, total;
};
inline int sac_blocks(const rlx_sac_fused_desc &d) {
    const int D = d.obs_dim, A = d.act_dim, H = d.q_hidden;
    const rlx_mlp3 &pm = d.policy_mlp, &vm = d.v_mlp;
    return tiles_of(D, pm.h1) + tiles_of(pm.h1, pm.h2) + tiles_of(pm.h2, 2 * A) + tiles_of(D, vm.h1) + tiles_of(vm.h1, vm.h2) +
           tiles_of(vm.h2, 1) + 2 * (tiles_of(D, H) + tiles_of(A, H) + tiles_of(H, H) + tiles_of(H, 1));
}
inline SacWs sac_layout(const rlx_sac_fused_desc &d) {
    SacWs w;
    long long o = 0;
    auto take = [&](long long n) { const long long at = o; o += (n + 3) & ~3LL; return at; };
    const long long B = d.batch, Bp = pad128(d.batch), ldx = pad64(d.obs_dim), ldh = pad64(d.q_hidden), H = d.q_hidden, A = d.act_dim;
    const long long nrb = (B + RS - 1) / RS;
    // (the operands of the weight-gradient launch first, padded; they rely on the workspace having been ZEROED at allocation)
    w.xs = take(Bp * ldx); w.ab = take(Bp * 64);
    w.h1P = take(Bp * ldh); w.h1V = take(Bp * ldh); w.ho = take(2 * Bp * ldh); w.h2P = take(Bp * ldh); w.h2V = take(Bp * ldh);
    w.haB = take(2 * Bp * ldh); w.hs = take(2 * Bp * ldh); w.h2Q = take(2 * Bp * ldh);
    w.dyP = take(Bp * 64); w.dz2P = take(Bp * ldh); w.dh1Pp = take(kSplit * Bp * ldh);
    w.dvV = take(Bp * 64); w.dz2V = take(Bp * ldh); w.dh1Vp = take(kSplit * Bp * ldh);
    w.dqQ = take(2 * Bp * 64); w.dfc1 = take(2 * Bp * ldh); w.dhqp = take(2LL * kSplit * Bp * ldh);
    w.h1VT = take(B * H); w.zP = take(B * kSplit * 2 * A); w.vp = take(kSplit * B); w.vTp = take(kSplit * B); w.qp = take(2LL * kSplit * B);
    w.logp0 = take(B); w.haP = take(2 * B * H); w.h2Qp = take(2 * B * H); w.qpp = take(2LL * kSplit * B); w.dapp = take(2LL * kSplit * B * A);
    w.v_loss_part = take(nrb); w.q_loss_part = take(2 * nrb);
    w.norm_part = take(sac_blocks(d));
    w.stamps = take(2 * 96);
    w.total = o;
    return w;
}
inline int sac_check(const rlx_sac_fused_desc &d) {
    const rlx_mlp3 &pm = d.policy_mlp, &vm = d.v_mlp;
    const int H = d.q_hidden;
    if (d.batch < 1 || d.batch > 4096 || d.obs_dim < 1 || d.obs_dim > SP - 4 || d.act_dim < 1 || 2 * d.act_dim > HP - 16) return 0;
    if (pm.d_in != d.obs_dim || vm.d_in != d.obs_dim || pm.d_out != 2 * d.act_dim || vm.d_out != 1) return 0;
    const int widths[5] = {pm.h1, pm.h2, vm.h1, vm.h2, H};
    for (int i = 0; i < 5; ++i)
        if (widths[i] % 4 || widths[i] < 32 * kSplit || widths[i] > SP - 4) return 0;
    // a slice of the policy head's rows is staged in LDS (sac_layer2_kernel: one activation buffer; sac_backward_kernel: <= 9 per thread)
    if ((pm.h2 / kSplit + 4) * 2 * d.act_dim > 9 * T || RS * (pm.h2 / kSplit + 4) > 2 * T ||
        (pm.h2 / kSplit + 4) * 2 * d.act_dim > RS * SP) return 0;
    if ((d.q_off_act_w | d.q_stride_act | d.q_off_obs_w | d.q_stride_obs | d.q_off_fc_w | d.q_stride_fc) & 3) return 0;
    return 1;
}
inline SacDev sac_dev(const rlx_sac_fused_desc &d) {
    SacDev p;
    const SacWs w = sac_layout(d);
    p.pw = d.policy.weights; p.vw = d.v.weights; p.vwt = d.v.target_weights; p.qw = d.q.weights;
    p.pm = to_dev(d.policy_mlp); p.vm = to_dev(d.v_mlp);
    p.qm.o_wo = d.q_off_obs_w; p.qm.o_bo = d.q_off_obs_b; p.qm.o_wa = d.q_off_act_w; p.qm.o_ba = d.q_off_act_b;
    p.qm.o_w1 = d.q_off_fc_w; p.qm.o_b1 = d.q_off_fc_b; p.qm.o_wq = d.q_off_out_w; p.qm.o_bq = d.q_off_out_b;
    p.qm.s_o = d.q_stride_obs; p.qm.s_a = d.q_stride_act; p.qm.s_1 = d.q_stride_fc; p.qm.s_q = d.q_stride_out;
    p.obs = d.obs; p.next_obs = d.next_obs; p.actions = d.actions; p.rewards = d.rewards; p.dones = d.game_overs;
    p.normals = d.normals; p.discount = d.discount; p.resample = d.resample_noise_per_pass;
    p.B = d.batch; p.D = d.obs_dim; p.A = d.act_dim; p.H = d.q_hidden; p.nrb = (d.batch + RS - 1) / RS;
    p.Bp = pad128(d.batch); p.ldx = pad64(d.obs_dim); p.ldh = pad64(d.q_hidden);
    float *ws = d.workspace;
    p.xs = ws + w.xs; p.ab = ws + w.ab; p.h1P = ws + w.h1P; p.h1V = ws + w.h1V; p.ho = ws + w.ho; p.h2P = ws + w.h2P; p.h2V = ws + w.h2V;
    p.haB = ws + w.haB; p.hs = ws + w.hs; p.h2Q = ws + w.h2Q; p.dyP = ws + w.dyP; p.dz2P = ws + w.dz2P; p.dh1Pp = ws + w.dh1Pp;
    p.dvV = ws + w.dvV; p.dz2V = ws + w.dz2V; p.dh1Vp = ws + w.dh1Vp; p.dqQ = ws + w.dqQ; p.dfc1 = ws + w.dfc1; p.dhqp = ws + w.dhqp;
    p.h1VT = ws + w.h1VT; p.zP = ws + w.zP; p.vp = ws + w.vp; p.vTp = ws + w.vTp; p.qp = ws + w.qp; p.logp0 = ws + w.logp0;
    p.haP = ws + w.haP; p.h2Qp = ws + w.h2Qp; p.qpp = ws + w.qpp; p.dapp = ws + w.dapp; p.v_loss_part = ws + w.v_loss_part;
    p.q_loss_part = ws + w.q_loss_part;
    p.value_targets = d.value_targets; p.log_target = d.log_target; p.td_targets = d.td_targets; p.dq_da = d.dq_da;
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


int rlx_sac_fused_supported(const rlx_sac_fused_desc *d) { return d ? sac_check(*d) : 0; }

int rlx_sac_fused_workspace_floats(const rlx_sac_fused_desc *d, long long *floats_host) {
    RLX_REQUIRE(d && floats_host, "rlx_sac_fused_workspace_floats: null pointer");
    RLX_REQUIRE(sac_check(*d), "rlx_sac_fused_workspace_floats: unsupported shape");
    *floats_host = sac_layout(*d).total;
    return RLX_OK;
}

// SoftActorCriticAgent.learn_from_batch (soft_actor_critic_agent.py:168-280) as six launches.  write_grads != 0: the weight
// gradients of all three networks go to their grads buffers and no optimiser step is taken.
int rlx_sac_fused_update(const rlx_sac_fused_desc *d_host, int write_grads, void *stream) {
    RLX_REQUIRE(d_host != nullptr, "rlx_sac_fused_update: null descriptor");
    const rlx_sac_fused_desc &d = *d_host;
    RLX_REQUIRE(sac_check(d), "rlx_sac_fused_update: unsupported shape (batch=%d obs=%d act=%d hidden %d)", d.batch, d.obs_dim,
                d.act_dim, d.q_hidden);
    RLX_REQUIRE(d.policy.weights && d.q.weights && d.v.weights && d.v.target_weights && d.obs && d.next_obs && d.actions &&
                    d.rewards && d.game_overs && d.normals && d.workspace && d.value_targets && d.log_target && d.td_targets &&
                    d.dq_da && d.q_loss && d.v_loss,
                "rlx_sac_fused_update: null pointer");
    const rlx_fused_net *nets[3] = {&d.policy, &d.q, &d.v};
    for (int i = 0; i < 3; ++i)
        RLX_REQUIRE(write_grads ? nets[i]->grads != nullptr : (nets[i]->adam_m && nets[i]->adam_v && nets[i]->adam_state),
                    "rlx_sac_fused_update: null optimiser pointer (network %d)", i);
    RLX_REQUIRE(d.policy.ticket != nullptr, "rlx_sac_fused_update: null ticket");
    const SacWs w = sac_layout(d);
    RLX_REQUIRE(d.workspace_floats >= w.total, "rlx_sac_fused_update: workspace of %lld floats, need %lld", d.workspace_floats, w.total);
    static bool configured = false;
    if (!configured) {
        RLX_HIP(set_lds(sac_layer1_kernel, kSacFwdLds));
        RLX_HIP(set_lds(sac_layer2_kernel, kSacFwdLds));
        RLX_HIP(set_lds(sac_q_pi_kernel, kSacFwdLds));
        RLX_HIP(set_lds(sac_q_grad_kernel, kSacBwdLds));
        RLX_HIP(set_lds(sac_backward_kernel, kSacBwdLds));
        configured = true;
    }
    const SacDev p = sac_dev(d);
    hipStream_t st = rlx::as_stream(stream);
    RLX_LAUNCH((sac_layer1_kernel), 5 * kSplit * p.nrb, T, kSacFwdLds, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((sac_layer2_kernel), 5 * kSplit * p.nrb, T, kSacFwdLds, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((sac_q_pi_kernel), 2 * kSplit * p.nrb, T, kSacFwdLds, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((sac_q_grad_kernel), 2 * kSplit * p.nrb, T, kSacBwdLds, st, p);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((sac_backward_kernel), 4 * kSplit * p.nrb, T, kSacBwdLds, st, p);
    RLX_LAUNCH_CHECK();
    DwBuilder b;
    const int np = b.add_net(d.policy);
    const rlx_mlp3 &pm = d.policy_mlp, &vm = d.v_mlp;
    const long long Bp = p.Bp, ldh = p.ldh;
    const int D = d.obs_dim, A = d.act_dim, H = d.q_hidden;
    b.add_partial_job(np, p.xs, p.ldx, p.dh1Pp, ldh, kSplit, Bp * ldh, p.h1P, pm.off_w1, pm.off_b1, D, pm.h1);
    b.add_job(np, p.h1P, ldh, p.dz2P, ldh, pm.off_w2, pm.off_b2, pm.h1, pm.h2);
    b.add_job(np, p.h2P, ldh, p.dyP, 64, pm.off_w3, pm.off_b3, pm.h2, 2 * A);
    const int nq = b.add_net(d.q);
    for (int t = 0; t < 2; ++t) {
        b.add_partial_job(nq, p.xs, p.ldx, p.dhqp + (size_t)t * kSplit * Bp * ldh, ldh, kSplit, Bp * ldh, p.ho + (size_t)t * Bp * ldh,
                          d.q_off_obs_w + t * d.q_stride_obs, d.q_off_obs_b + t * d.q_stride_obs, D, H);
        b.add_partial_job(nq, p.ab, 64, p.dhqp + (size_t)t * kSplit * Bp * ldh, ldh, kSplit, Bp * ldh, p.haB + (size_t)t * Bp * ldh,
                          d.q_off_act_w + t * d.q_stride_act, d.q_off_act_b + t * d.q_stride_act, A, H);
        b.add_job(nq, p.hs + (size_t)t * Bp * ldh, ldh, p.dfc1 + (size_t)t * Bp * ldh, ldh, d.q_off_fc_w + t * d.q_stride_fc,
                  d.q_off_fc_b + t * d.q_stride_fc, H, H);
        b.add_job(nq, p.h2Q + (size_t)t * Bp * ldh, ldh, p.dqQ + (size_t)t * Bp * 64, 64, d.q_off_out_w + t * d.q_stride_out,
                  d.q_off_out_b + t * d.q_stride_out, H, 1);
    }
    const int nv = b.add_net(d.v);
    b.add_partial_job(nv, p.xs, p.ldx, p.dh1Vp, ldh, kSplit, Bp * ldh, p.h1V, vm.off_w1, vm.off_b1, D, vm.h1);
    b.add_job(nv, p.h1V, ldh, p.dz2V, ldh, vm.off_w2, vm.off_b2, vm.h1, vm.h2);
    b.add_job(nv, p.h2V, ldh, p.dvV, 64, vm.off_w3, vm.off_b3, vm.h2, 1);
    b.a.loss_part = p.q_loss_part; b.a.loss_out = d.q_loss; b.a.loss_streams = 2; b.a.loss_parts = p.nrb; b.a.loss_scale = 1.f;
    b.a.loss2_part = p.v_loss_part; b.a.loss2_out = d.v_loss; b.a.loss2_parts = p.nrb;
    return launch_dw(b, d.batch, write_grads, d.workspace + w.norm_part, d.policy.ticket, st);
}

}  // extern "C"
