// Clipped PPO, discrete heads: the last dense layer of BOTH towers (value / policy), the two heads' forward pass, both head
// losses and the heads' backward pass as ONE launch (rlx_ppo_fc_heads).
//
// Replaces, for the two-tower torso of ClippedPPONetworkParameters (rl_coach/agents/clipped_ppo_agent.py:41-58):
//   architectures/tensorflow_components/layers.py:168-185      the middleware's Dense(512) of each tower (forward)
//   heads/v_head.py:43-52, heads/ppo_head.py:52-116             VHead Dense(1) / PPOHead policy_fc + softmax, their losses
//   head.py:143-186, architecture.py:312-385                    tf.gradients from the losses down to the Dense(512) output
// i.e. three launches of the round-5 update — the tiled product with its 25 K splits (6.5 MB of partial sums for a 0.26 MB
// result), the row-finishing reduction that also ran the heads' forward, and the losses + heads' backward: 11.0 + 8.0 +
// 11.2 us (profiles/r06_call1_dispatch_hist.txt) for 0.41 GFLOP.
//
// Shape of the work: M = 64 rows per tower, N = 512, K = 3136 — a weight stream (12.8 MB) against a tiny left operand.
// Grid: (tower, 32-column tile, K eighth) = 2 x 16 x 8 = 256 workgroups, one per CU.  A workgroup requests its whole
// operand set at once — 64 x 392 of x, 392 x 32 of W: 150 KB, ONE memory round trip — stages it in LDS, and runs 98
// v_mfma_f32_32x32x2_f32 per wave (2 row tiles x 2 K halves).  Its 64 x 32 partial tile (8 KB) is published write-through
// (16-byte sc1 stores, no release fence: guide "publish-large"); the last of a tile's 8 workgroups adds them in split
// order, applies bias and activation, writes the tile of h and the tile's share of the heads' dot products (64 x A'
// numbers); the last of a tower's 16 tiles finishes the head outputs, the per-row loss terms and gradients, and then —
// the only part that needs all of h again (128 KB, from L2) — dz = act'(h) (dy W_head^T) for the layer's backward pass and
// the heads' weight gradients.  1 MB of partials instead of 6.5 MB; nothing is read that another launch wrote except x.
// Arithmetic of the losses: losses_body.hpp (the definitions the stand-alone loss kernels use).
#include "losses_body.hpp"

namespace {

using namespace rlx_losses;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256, kRows = 64, kTileN = 32, kSplits = 8, kMaxA = 16, kMaxKc = 392;
constexpr int kXP = kMaxKc + 1;                        // odd pitch: the MFMA A operand walks rows (lane = row)
constexpr size_t kLdsFloats = (size_t)kRows * kXP + (size_t)kMaxKc * kTileN;
static_assert(kLdsFloats * 4 <= 156 * 1024, "ppo_fc_fused: LDS budget");

struct FcHeadsDev {
    const float *x; long long x_ts;                    // [2][B][K]
    const float *w; long long w_ts;                    // [K][N] per tower
    const float *bias; long long b_ts;
    const float *wv, *bv, *wp, *bp;                    // heads: [N][1], [1], [N][A], [A]
    const float *v_target, *adv, *old_probs; long long ld_old;
    const int *actions;
    const float *clip_scale;
    float clip_eps, beta, grad_scale;
    int B, K, N, A, act, Kc, NT;
    float *h, *dzh;                                    // [2][B][N]
    float *v, *logits, *dv, *dlogits;                  // [B], [B][A]
    float *dwv, *dbv, *dwp, *dbp;                      // head gradients
    float *scalars, *ratio_out, *clipped_out;
    int *status;
    float *part;                                       // [2][NT][kSplits][kTileN][kRows]
    float *hpart;                                      // [2][NT][kRows][kMaxA]
    float *dyg;                                        // [2][kRows][kMaxA]   d loss / d head outputs, for the tower's tile owners
    unsigned *tick;                                    // [2 * NT] tile tickets, [2][4] tower words
    long long *stamps;
};
#define RLX_FC_STAMP_ANY(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define RLX_FC_STAMP(i) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
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
// write-through / L2-served accesses for data another workgroup of this launch wrote (guide G16: sc1 both sides)
__device__ __forceinline__ void store4_sc1(float *p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load4_sc1(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store1_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load1_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(kThreads) ppo_fc_heads_kernel(const FcHeadsDev p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int last_s;
    float *xs = smem;                                  // [kRows][kXP]
    float *ws = smem + (size_t)kRows * kXP;            // [Kc][kTileN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int ks = blockIdx.x % kSplits, g = blockIdx.x / kSplits, nt = g % p.NT, t = g / p.NT;
    const int n0 = nt * kTileN, k0 = ks * p.Kc, B = p.B, K = p.K, N = p.N, Kc = p.Kc;
    RLX_FC_STAMP(0);
    // ---- every operand byte of this workgroup is requested before anything is consumed; the slab is staged and consumed in
    //      two K chunks, so that the first chunk's products run while the second chunk's lines are still arriving
    constexpr int kXPieces = (kRows * (kMaxKc / 4) + kThreads - 1) / kThreads;      // 25 (13 + 12 per chunk)
    constexpr int kWPieces = (kMaxKc * (kTileN / 4) + kThreads - 1) / kThreads;     // 13 (7 + 6)
    constexpr int kXH = (kXPieces + 1) / 2, kWH = (kWPieces + 1) / 2;
    const int Kh = Kc >> 1, kqh = Kh >> 2;              // chunk length (a multiple of 4), its 16-byte pieces per x row
    const float *xb = p.x + (size_t)t * p.x_ts, *wb = p.w + (size_t)t * p.w_ts;
    f32x4 xv[2][kXH], wv[2][kWH];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int i = 0; i < kXH; ++i) {
            const int f = tid + i * kThreads, r = f / kqh, q = f - r * kqh;
            const int rr = min(r, B - 1), kk = min(k0 + ch * Kh + 4 * q, K - 4);
            xv[ch][i] = *reinterpret_cast<const f32x4 *>(xb + (size_t)rr * K + kk);
        }
#pragma unroll
        for (int i = 0; i < kWH; ++i) {
            const int f = tid + i * kThreads, kr = f >> 3, q = f & 7;
            const int kk = min(k0 + ch * Kh + kr, K - 1);
            wv[ch][i] = *reinterpret_cast<const f32x4 *>(wb + (size_t)kk * N + n0 + 4 * q);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int rt = wave & 1, kh = wave >> 1, kquart = Kh >> 1;       // wave = (row tile, half of the chunk)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
        for (int i = 0; i < kXH; ++i) {
            const int f = tid + i * kThreads, r = f / kqh, q = f - r * kqh;
            if (r < kRows) {
                const bool okr = r < B;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xs[r * kXP + ch * Kh + 4 * q + e] = (okr && k0 + ch * Kh + 4 * q + e < K) ? xv[ch][i][e] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < kWH; ++i) {
            const int f = tid + i * kThreads, kr = f >> 3, q = f & 7;
            if (kr < Kh) {
                const bool okk = k0 + ch * Kh + kr < K;
                f32x4 v = wv[ch][i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = okk ? v[e] : 0.f;
                *reinterpret_cast<f32x4 *>(ws + (ch * Kh + kr) * kTileN + 4 * q) = v;
            }
        }
        __syncthreads();
        if (ch == 0) RLX_FC_STAMP(1);
        // ---- 64 x 32 x Kh of this chunk
        const float *ar = xs + (32 * rt + l31) * kXP + ch * Kh + kh * kquart + hi;
        const float *br = ws + (ch * Kh + kh * kquart + hi) * kTileN + l31;
        for (int k = 0; k < kquart; k += 32) {
            float a[16], b[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kk = min(k + 2 * e, kquart - 2);
                a[e] = ar[kk];
                b[e] = br[kk * kTileN];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (k + 2 * e < kquart) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
    }
    __syncthreads();                                   // operands dead: the slab region becomes scratch
    float *red = smem;                                 // [2][16][64]
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(rt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(rt * 16 + r) * 64 + lane];
        // the partial tile, transposed ([column][row]: an accumulator quad is 4 consecutive rows), written through
        float *pt = p.part + ((size_t)g * kSplits + ks) * (kTileN * kRows) + l31 * kRows + 32 * rt + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) store4_sc1(pt + 8 * q, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RLX_FC_STAMP(2);
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&p.tick[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == kSplits - 1;
        if (last_s) __hip_atomic_store(&p.tick[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_s) return;
    if (g == 0) RLX_FC_STAMP_ANY(3);
    // ---- the tile's last arriver: partials in split order, bias, activation -> h tile; the tile's share of the heads
    const int AH = t == 0 ? 1 : p.A;                   // outputs of this tower's head
    const float *whead = t == 0 ? p.wv : p.wp;
    float *htile = smem;                               // [kRows][kTileN + 1]
    float *whs = htile + kRows * (kTileN + 1);         // [kTileN][kMaxA]   the head's weight rows of this tile (zero beyond AH)
    float *dy = whs + kTileN * kMaxA;                  // [kRows][kMaxA]    d loss / d head output (zero beyond B / AH)
    float *zs = dy + kRows * kMaxA;                    // [kRows][kMaxA]    head outputs before the bias (tower's last arriver)
    float *red3 = zs + kRows * kMaxA;                  // [3][kRows]
    {
        const int c = tid & 31, rg = tid >> 5;          // column, 8-row group
        f32x4 pv[kSplits][2];
        const float *pb = p.part + (size_t)g * kSplits * (kTileN * kRows) + c * kRows + 8 * rg;
#pragma unroll
        for (int s = 0; s < kSplits; ++s) {
            pv[s][0] = load4_sc1(pb + (size_t)s * (kTileN * kRows));
            pv[s][1] = load4_sc1(pb + (size_t)s * (kTileN * kRows) + 4);
        }
        const float bia = p.bias[(size_t)t * p.b_ts + n0 + c];
        for (int e = tid; e < kTileN * kMaxA; e += kThreads) {
            const int cc = e / kMaxA, j = e - cc * kMaxA;
            whs[e] = j < AH ? whead[(size_t)(n0 + cc) * AH + j] : 0.f;
        }
        // (the loads above are inline asm: the wait names their registers, so that nothing reads them ahead of it)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(pv[0][0]), "+v"(pv[0][1]), "+v"(pv[1][0]), "+v"(pv[1][1]), "+v"(pv[2][0]), "+v"(pv[2][1]), "+v"(pv[3][0]),
                       "+v"(pv[3][1]), "+v"(pv[4][0]), "+v"(pv[4][1]), "+v"(pv[5][0]), "+v"(pv[5][1]), "+v"(pv[6][0]), "+v"(pv[6][1]),
                       "+v"(pv[7][0]), "+v"(pv[7][1])
                     :
                     : "memory");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 s = pv[0][half];
#pragma unroll
            for (int q = 1; q < kSplits; ++q) s += pv[q][half];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 8 * rg + 4 * half + e;
                const float hv = row < B ? act_apply(s[e] + bia, p.act) : 0.f;
                htile[row * (kTileN + 1) + c] = hv;
                if (row < B) p.h[((size_t)t * B + row) * N + n0 + c] = hv;
            }
        }
    }
    __syncthreads();
    const int AH4 = (AH + 3) & ~3;                       // (the tower's last arriver reads 16 bytes at a time)
    for (int o = tid; o < kRows * AH4; o += kThreads) {
        const int row = o % kRows, j = o / kRows;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < kTileN; ++c) s = fmaf(htile[row * (kTileN + 1) + c], whs[c * kMaxA + j], s);
        store1_sc1(p.hpart + ((size_t)g * kRows + row) * kMaxA + j, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g == 0) RLX_FC_STAMP_ANY(12);
    __syncthreads();
    unsigned *tower_tick = p.tick + 2 * p.NT + 4 * t;  // [0] arrivals, [1] dy published, [2] readers done
    if (tid == 0)
        last_s = __hip_atomic_fetch_add(&tower_tick[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)p.NT - 1;
    __syncthreads();
    if (last_s) {
        // ---- the tower's last arriver: head outputs, per-row loss terms and gradients, scalars.  The tiles' shares in tile
        //      order, 16 bytes (4 outputs) per request, thread = (row, output quad): one round trip (a dependent chain of
        //      96 write-through reads was 77 us of this kernel's first version)
        RLX_FC_STAMP_ANY(4 + 4 * t);
        {
            const int r = tid & 63, j4 = tid >> 6;
            f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (4 * j4 < AH) {
                const float *hp = p.hpart + ((size_t)(t * p.NT) * kRows + r) * kMaxA + 4 * j4;
                for (int q0 = 0; q0 < p.NT; q0 += 16) {
                    f32x4 pv[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) pv[q] = load4_sc1(hp + (size_t)min(q0 + q, p.NT - 1) * kRows * kMaxA);
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                                   "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]),
                                   "+v"(pv[15])
                                 :
                                 : "memory");
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q0 + q < p.NT) acc4 += pv[q];
                }
            }
            *reinterpret_cast<f32x4 *>(zs + r * kMaxA + 4 * j4) = acc4;
            *reinterpret_cast<f32x4 *>(dy + r * kMaxA + 4 * j4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (tid < kRows) {
            const int r = tid;
            if (r < B) {
                float z[kMaxA];
#pragma unroll
                for (int j = 0; j < kMaxA; ++j) z[j] = j < AH ? zs[r * kMaxA + j] + (t == 0 ? p.bv[j] : p.bp[j]) : 0.f;
                if (t == 0) {                           // VHead: MSE(target, V), loss weight 1 (head.py:172-181)
                    const float e = z[0] - p.v_target[r];
                    float l, gq;
                    regression_terms(e, 0, l, gq);
                    const float d = regression_grad(p.grad_scale, 1.f, gq, B);
                    dy[r * kMaxA] = d;
                    p.v[r] = z[0];
                    p.dv[r] = d;
                    t0 = l;
                } else {
                    const float clip = p.clip_scale ? p.clip_eps * *p.clip_scale : p.clip_eps;
                    PpoRowTerms pt;
                    float dl[kMaxA];
                    if (ppo_discrete_row(z, p.old_probs + (size_t)r * p.ld_old, p.actions[r], p.A, p.adv[r], clip, p.beta,
                                         p.grad_scale, B, dl, p.ratio_out ? p.ratio_out + r : nullptr,
                                         p.clipped_out ? p.clipped_out + r : nullptr, pt)) {
                        t0 = pt.sur; t1 = pt.ent; t2 = pt.kl;
                        for (int j = 0; j < p.A; ++j) dy[r * kMaxA + j] = dl[j];
                    } else {
                        atomicOr(p.status, 1);
                    }
                    for (int j = 0; j < p.A; ++j) {
                        p.logits[(size_t)r * p.A + j] = z[j];
                        p.dlogits[(size_t)r * p.A + j] = dy[r * kMaxA + j];
                    }
                }
            }
            red3[r] = t0; red3[kRows + r] = t1; red3[2 * kRows + r] = t2;
        }
        __syncthreads();
        for (int d = kRows >> 1; d > 0; d >>= 1) {
            if (tid < d) {
                red3[tid] += red3[tid + d];
                red3[kRows + tid] += red3[kRows + tid + d];
                red3[2 * kRows + tid] += red3[2 * kRows + tid + d];
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (t == 0) p.scalars[4] = red3[0] / (float)B;
            else ppo_discrete_scalars(red3[0], red3[kRows], red3[2 * kRows], p.beta, B, p.scalars);
        }
        if (tid < AH) {                                 // the head's bias gradient
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dy[b * kMaxA + tid];
            (t == 0 ? p.dbv : p.dbp)[tid] = s;
        }
        // dy for the tower's other tiles: written through, then the flag
        store4_sc1(p.dyg + ((size_t)t * kRows * kMaxA) + 4 * tid, *reinterpret_cast<const f32x4 *>(dy + 4 * tid));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&tower_tick[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RLX_FC_STAMP_ANY(5 + 4 * t);
    } else {
        // ---- the tower's other tile owners wait for dy.  They are all resident (they are running), and the workgroup
        //      they wait for never waits: no ordering assumption, bounded spin (status bit 8 = gave up)
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(&tower_tick[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 20)) {
                    atomicOr(p.status, 8);
                    break;
                }
            }
        }
        __syncthreads();
        const f32x4 v = load4_sc1(p.dyg + ((size_t)t * kRows * kMaxA) + 4 * tid);
        f32x4 vv = v;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(vv) : : "memory");
        *reinterpret_cast<f32x4 *>(dy + 4 * tid) = vv;
    }
    __syncthreads();
    // ---- every tile owner: dz = act'(h) (dy W_head^T) of its 64 x 32 tile and the head's weight gradient rows of its units
    {
        const int c = tid & 31, rg = tid >> 5;
        f32x4 wq[kMaxA / 4];
#pragma unroll
        for (int j4 = 0; j4 < kMaxA / 4; ++j4) wq[j4] = *reinterpret_cast<const f32x4 *>(whs + c * kMaxA + 4 * j4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 8 * rg + e;
            float s = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < kMaxA / 4; ++j4) {
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(dy + row * kMaxA + 4 * j4);
#pragma unroll
                for (int q = 0; q < 4; ++q) s = fmaf(d4[q], wq[j4][q], s);
            }
            if (row < B) p.dzh[((size_t)t * B + row) * N + n0 + c] = act_deriv_out(htile[row * (kTileN + 1) + c], p.act) * s;
        }
        float *dwo = t == 0 ? p.dwv : p.dwp;
        for (int j = rg; j < AH; j += kThreads / 32) {
            float s = 0.f;
#pragma unroll 8
            for (int b = 0; b < kRows; ++b) s = fmaf(htile[b * (kTileN + 1) + c], dy[b * kMaxA + j], s);
            dwo[(size_t)(n0 + c) * AH + j] = s;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // the tower's last reader re-arms its three words
        if (__hip_atomic_fetch_add(&tower_tick[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)p.NT - 1) {
            __hip_atomic_store(&tower_tick[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tower_tick[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tower_tick[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (last_s) RLX_FC_STAMP_ANY(7 + 4 * t);
}

}  // namespace

extern "C" {

int rlx_ppo_fc_heads_supported(int batch, int in_features, int units, int n_actions) {
    if (batch < 1 || batch > kRows || n_actions < 2 || n_actions > kMaxA) return 0;
    if (units % kTileN || units < kTileN || in_features % 4 || in_features < 8 * kSplits) return 0;
    const int kc = (((in_features + kSplits - 1) / kSplits) + 7) & ~7;
    return kc <= kMaxKc ? 1 : 0;
}

int rlx_ppo_fc_heads_workspace(int units, long long *floats_host, long long *ticket_words_host) {
    RLX_REQUIRE(floats_host && ticket_words_host && units >= kTileN && units % kTileN == 0, "rlx_ppo_fc_heads_workspace: bad arguments");
    const long long NT = units / kTileN;
    *floats_host = 2 * NT * kSplits * kTileN * kRows + 2 * NT * kRows * kMaxA + 2 * kRows * kMaxA + 32;   /* + 16 int64 stamps */
    *ticket_words_host = 2 * NT + 8;
    return RLX_OK;
}

static bool g_fc_stamps = false;
int rlx_ppo_fc_heads_stamps(int enable) {
    g_fc_stamps = enable != 0;
    return RLX_OK;
}

int rlx_ppo_fc_heads(const rlx_ppo_fc_heads_desc *d_host, void *stream) {
    RLX_REQUIRE(d_host != nullptr, "rlx_ppo_fc_heads: null descriptor");
    const rlx_ppo_fc_heads_desc &d = *d_host;
    RLX_REQUIRE(rlx_ppo_fc_heads_supported(d.batch, d.in_features, d.units, d.n_actions),
                "rlx_ppo_fc_heads: unsupported shape (batch=%d <= 64, in=%d, units=%d (multiple of 32), actions=%d <= 16)", d.batch,
                d.in_features, d.units, d.n_actions);
    RLX_REQUIRE(d.x && d.weights && d.bias && d.value_w && d.value_b && d.policy_w && d.policy_b && d.value_targets &&
                    d.advantages && d.old_probs && d.actions && d.h && d.dz && d.values && d.logits && d.dvalues && d.dlogits &&
                    d.d_value_w && d.d_value_b && d.d_policy_w && d.d_policy_b && d.scalars && d.status && d.workspace && d.tickets,
                "rlx_ppo_fc_heads: null pointer");
    RLX_REQUIRE(d.activation >= 0 && d.activation <= 2, "rlx_ppo_fc_heads: unknown activation");
    RLX_REQUIRE((reinterpret_cast<uintptr_t>(d.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.weights) & 15) == 0 &&
                    (d.x_tower_stride & 3) == 0 && (d.weight_tower_stride & 3) == 0,
                "rlx_ppo_fc_heads: operands must be 16-byte aligned");
    long long need = 0, words = 0;
    rlx_ppo_fc_heads_workspace(d.units, &need, &words);
    RLX_REQUIRE(d.workspace_floats >= need, "rlx_ppo_fc_heads: workspace of %lld floats, need %lld", d.workspace_floats, need);
    FcHeadsDev p;
    p.x = d.x; p.x_ts = d.x_tower_stride; p.w = d.weights; p.w_ts = d.weight_tower_stride; p.bias = d.bias; p.b_ts = d.bias_tower_stride;
    p.wv = d.value_w; p.bv = d.value_b; p.wp = d.policy_w; p.bp = d.policy_b;
    p.v_target = d.value_targets; p.adv = d.advantages; p.old_probs = d.old_probs; p.ld_old = d.ld_old; p.actions = d.actions;
    p.clip_scale = d.clip_scale; p.clip_eps = d.clip_epsilon; p.beta = d.beta_entropy; p.grad_scale = d.grad_scale;
    p.B = d.batch; p.K = d.in_features; p.N = d.units; p.A = d.n_actions; p.act = d.activation;
    p.Kc = (((d.in_features + kSplits - 1) / kSplits) + 7) & ~7;
    p.NT = d.units / kTileN;
    p.h = d.h; p.dzh = d.dz; p.v = d.values; p.logits = d.logits; p.dv = d.dvalues; p.dlogits = d.dlogits;
    p.dwv = d.d_value_w; p.dbv = d.d_value_b; p.dwp = d.d_policy_w; p.dbp = d.d_policy_b;
    p.scalars = d.scalars; p.ratio_out = d.likelihood_ratio; p.clipped_out = d.clipped_likelihood_ratio; p.status = d.status;
    const long long NT = p.NT;
    p.part = d.workspace;
    p.hpart = d.workspace + 2 * NT * kSplits * kTileN * kRows;
    p.dyg = p.hpart + 2 * NT * kRows * kMaxA;
    p.stamps = g_fc_stamps ? reinterpret_cast<long long *>(p.dyg + 2 * kRows * kMaxA) : nullptr;
    p.tick = d.tickets;
    const size_t lds = kLdsFloats * sizeof(float);
    static bool configured = false;
    if (!configured) {
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ppo_fc_heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        configured = true;
    }
    RLX_LAUNCH((ppo_fc_heads_kernel), 2 * p.NT * kSplits, kThreads, lds, rlx::as_stream(stream), p);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
