// Row-local MLP chains for the continuous-control updates (TD3 / SAC) — device library of ac_fused.hip / sac_fused.hip.
//
// Why: at B = 100 .. 256 and widths 256 .. 400 one learn_from_batch of TD3 / SAC is 0.3 - 1.2 GFLOP behind 24 - 42 dependent
// launches of 4 - 9 us each (profiles/r05_final_c{4,5}_kernel_stats.csv): the update is bound by the launch CHAIN.  Every
// pass of these networks up to the weight gradients is ROW-LOCAL — a batch row's forward chain (actor -> smoothing ->
// critic -> TD target), its loss gradient and its input-gradient chain never read another row — so a workgroup that owns
// R = 4 batch rows walks a whole chain by itself: activations stay in LDS, the weights stream past once per workgroup
// from the XCD's L2.  Only the weight gradients sum over the batch; they run as ONE tile-parallel MFMA launch per update
// that applies Adam from the accumulators (mlp_dw_adam_kernel in ac_fused.hip).
//
// The products run on the matrix pipe as v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 outer products per instruction —
// the 4 rows of the workgroup (A operand, the same in every block) against 64 weight columns (B operand, one column per
// lane): a 4 x 64 slab of the layer's output per instruction and reduction index, 64 flop / clk / SIMD.  The first
// version did this arithmetic on the vector ALU; measured (tools/microbench/rowchain_probe.hip,
// profiles/r06_rowchain_valu_ablation.txt): v_pk_fma_f32 issues at 8 cycles per wave — 128 flop / clk / CU, half the
// matrix pipe's fp32 rate — and with the loads removed the 400 x 300 layer still took 11.8 k cycles (forward) / 31 k
// (transposed): these chains are bound by ISSUE, not by the weight stream.  Second lesson of the same probe: hipcc sinks
// every global load to its first use unless a sched_barrier stands between the loads and their consumers (one L2 round
// trip per weight row: 9 us per layer).
//
// Layer arithmetic: y[r][n] = act(sum_k x[r][k] W[k][n] + b[n]); the MFMA sums one k per instruction in ascending order
// inside a K slice (fp32 fmaf chain), the slices are added in slice order, then the bias (the order of the tiled kernels:
// products, then bias).  Tolerances against the oracle are those of tests/test_ac_nets.py (another fp32 summation order,
// not another algorithm).
#pragma once
#include "rlx_common.hpp"

namespace rlx_chain {

constexpr int R = 4;                       // batch rows per workgroup: one 4 x 4 MFMA block row set (the functions' default RR);
                                           // RR = 8 gives a workgroup two of them on the same weight registers (SAC: B = 256)
constexpr int T = 512;                     // threads per workgroup (8 waves)
constexpr int kPartFloats = 8192;          // K-split partial sums: S * R * N <= kPartFloats
constexpr int kMaxWidth = 512;             // widest activation row held in LDS
constexpr int kPitch = kMaxWidth + 4;      // row pitch of the activation buffers: the 4 rows of an A-operand read (one
                                           // ds_read_b128 per lane, row = lane % 4) land in 4 different bank quads
constexpr int kTileCols = 16;              // weight columns per LDS tile of a transposed product
constexpr int kTilePitch = kTileCols + 4;  // pitch / 4 odd: the 16-byte reads of 16 consecutive rows hit 64 distinct banks
constexpr int kTileFloats = kMaxWidth * kTilePitch;
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ int pad16(int n) { return (n + 15) & ~15; }

// Touch a weight span so that it is on its way into this XCD's L2 (and this CU's TLB) before the layer that streams it
// starts: the span divided over the workgroups of the XCD, one 128-byte line per thread and sweep.  The loaded words are
// dropped; the token only keeps their registers reserved until prefetch_retire has waited for them.
struct PrefetchToken {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f;
};
// this workgroup's index among the workgroups of its XCD and their number (block b runs on XCD b % 8: observed, used
// for speed only — a different placement costs time, not correctness)
__device__ __forceinline__ int xcd_part() { return blockIdx.x >> 3; }
__device__ __forceinline__ int xcd_parts() { return (gridDim.x - (blockIdx.x & 7) + 7) >> 3; }
#define RLX_PREFETCH(field, base, floats, sweep)                                                                 \
    {                                                                                                            \
        const long long ln_ = ((long long)(sweep) * xcd_parts() + xcd_part()) * T + threadIdx.x;                 \
        const float *p_ = (base) + (ln_ * 32 < (long long)(floats) ? ln_ * 32 : 0);                              \
        asm volatile("global_load_dword %0, %1, off" : "=v"(field) : "v"(p_) : "memory");                       \
    }
__device__ __forceinline__ void prefetch_retire(PrefetchToken &t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(t.v0), "v"(t.v1), "v"(t.v2), "v"(t.v3), "v"(t.v4), "v"(t.v5), "v"(t.v6), "v"(t.v7));
}
__device__ __forceinline__ long long layer_floats(int K, int N) { return (((long long)K * N + 3) & ~3LL) + ((N + 3) & ~3); }

// xs[r][c] = src[row0 + r][c] for c < cols, 0 up to `pitch` and for rows >= B
template <int RR = R>
__device__ __forceinline__ void load_rows(float *xs, int pitch, const float *__restrict__ src, long long ld, int cols,
                                          int row0, int B) {
    for (int e = threadIdx.x; e < RR * pitch; e += T) {
        const int r = e / pitch, c = e - r * pitch;
        xs[e] = (c < cols && row0 + r < B) ? src[(long long)(row0 + r) * ld + c] : 0.f;
    }
}

// ys[r][n] = act(bias[n] + sum_k xs[r][k] W[k][n]),  r < R.
//   xs: LDS, row pitch xp (multiple of 4; kPitch for conflict-free operand reads), ZERO from K up to the next multiple of 16.
//   W: global row-major [K][N].  ys: LDS, row pitch yp >= the next multiple of 16 of N (the tail is zeroed here).
//   gout != null: rows row0 + r < B also go to gout[(row0 + r) * gld + n].  bias == null: none (partial products).
//   parts: LDS scratch of kPartFloats.  Ends with a barrier: ys is readable by every thread.
//   ldw: row stride of W in floats (= N for a whole layer; a column SLICE of a layer passes the layer's width and W / bias /
//   gout already offset to the slice's first column).
template <int RR = R>
__device__ __forceinline__ void dense_fwd(const float *xs, int xp, int K, const float *__restrict__ W, int ldw,
                                          const float *__restrict__ bias, int N, int act, float *ys, int yp, float *parts,
                                          float *__restrict__ gout, long long gld, int row0, int B) {
    constexpr int RB = RR / 4;                       // 4-row MFMA block sets of the workgroup
    const int tid = threadIdx.x;
    int S;
    if ((N & 3) == 0 && N >= 32) {
        // a lane owns 4 adjacent columns (16-byte weight loads, a wave reads 1 KB of a weight row) and one K slice; the 4
        // lanes of an MFMA block share their slice (G4 is a multiple of 4), the blocks of a wave need not
        const int G = N >> 2, G4 = (G + 3) & ~3;
        S = T / G4;
        while (S > 1 && S * RR * N > kPartFloats) --S;
        if (S > 16) S = 16;
        const int Kc = (((K + S - 1) / S) + 15) & ~15;
        S = (K + Kc - 1) / Kc;
        const int g = tid % G4, s = tid / G4;
        if (s < S) {
            f32x4 acc[RB][4];                          // acc[rb][c][r] = out[4 rb + r][4 g + c]
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int K16 = (K + 15) & ~15;
            const int k1 = min(K16, (s + 1) * Kc);
            const unsigned nb = (unsigned)ldw * 4u;                        // bytes per weight row
            const char *wc = reinterpret_cast<const char *>(W + 4 * min(g, G - 1));
            const float *xrow = xs + (tid & 3) * xp;                       // A operand: this lane's row of the block (+ 4 rb rows)
#ifndef RLX_CHAIN_STEP
#define RLX_CHAIN_STEP 32
#endif
            // weight rows requested per step (a multiple of 16); 8-row workgroups keep it at 16: 128 registers, two workgroups per CU
            constexpr int kStep = RR > 4 ? 16 : RLX_CHAIN_STEP;
            for (int k = s * Kc; k < k1; k += kStep) {
                // all weight rows of the step are requested, THEN consumed (the sched_barriers keep the scheduler from
                // sinking each load to its first use); rows in [K, K16) re-read row K - 1 against zeros of xs, rows beyond
                // the slice's end are read and not used
                f32x4 w[kStep];
#ifndef RLX_CHAIN_ABLATE_LOADS
#pragma unroll
                for (int j = 0; j < kStep; ++j) w[j] = *reinterpret_cast<const f32x4 *>(wc + (unsigned)min(k + j, K - 1) * nb);
#else
#pragma unroll
                for (int j = 0; j < kStep; ++j) w[j] = f32x4{1.f + tid + j, 2.f, 3.f, 4.f + k};
#endif
                f32x4 xa[RB][kStep / 4];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int q = 0; q < kStep / 4; ++q)
                        xa[rb][q] = *reinterpret_cast<const f32x4 *>(xrow + 4 * rb * xp + min(k + 4 * q, K16 - 4));
                __builtin_amdgcn_sched_barrier(0);
#ifndef RLX_CHAIN_ABLATE_MATH
#pragma unroll
                for (int h = 0; h < kStep / 16; ++h) {
                    if (k + 16 * h < k1) {
#pragma unroll
                        for (int q = 4 * h; q < 4 * h + 4; ++q) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const f32x4 wv = w[4 * q + e];
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb) {
                                    acc[rb][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[rb][q][e], wv[0], acc[rb][0], 0, 0, 0);
                                    acc[rb][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[rb][q][e], wv[1], acc[rb][1], 0, 0, 0);
                                    acc[rb][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[rb][q][e], wv[2], acc[rb][2], 0, 0, 0);
                                    acc[rb][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[rb][q][e], wv[3], acc[rb][3], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
#else
#pragma unroll
                for (int j = 0; j < kStep; ++j) acc[0][j & 3][0] += w[j][0] + w[j][1] + w[j][2] + w[j][3] + xa[0][j & 3][0];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g < G) {
#pragma unroll
                for (int r = 0; r < RR; ++r)
                    *reinterpret_cast<float4 *>(parts + (size_t)(s * RR + r) * N + 4 * g) =
                        make_float4(acc[r >> 2][0][r & 3], acc[r >> 2][1][r & 3], acc[r >> 2][2][r & 3], acc[r >> 2][3][r & 3]);
            }
        }
    } else {
        // narrow layer (heads): a thread owns one column and one K slice, 4-byte loads coalesced along the columns
        S = T / N;
        if (S > 32) S = 32;
        if (S > K) S = K;
        if (S < 1) S = 1;
        const int Kc = (K + S - 1) / S;
        S = (K + Kc - 1) / Kc;
        const int n = tid % N, s = tid / N;
        if (s < S) {
            float acc[RR];
#pragma unroll
            for (int r = 0; r < RR; ++r) acc[r] = 0.f;
            const int k1 = min(K, (s + 1) * Kc);
            for (int k = s * Kc; k < k1; k += 8) {
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {          // eight loads in flight, none under a predicate
                    const float v = W[(size_t)min(k + j, k1 - 1) * ldw + n];
                    w[j] = k + j < k1 ? v : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = min(k + j, k1 - 1);
#pragma unroll
                    for (int r = 0; r < RR; ++r) acc[r] = fmaf(xs[r * xp + kk], w[j], acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RR; ++r) parts[(size_t)(s * RR + r) * N + n] = acc[r];
        }
    }
    __syncthreads();
    const int N16 = pad16(N);                        // the next layer reads its input in steps of 16: zero the tail
    for (int o = tid; o < RR * N16; o += T) {
        const int r = o / N16, n = o - r * N16;
        if (n >= N) {
            ys[r * yp + n] = 0.f;
            continue;
        }
        float v = parts[(size_t)r * N + n];
        for (int s = 1; s < S; ++s) v += parts[(size_t)(s * RR + r) * N + n];
        v = act_apply(bias ? v + bias[n] : v, act);
        ys[r * yp + n] = v;
        if (gout && row0 + r < B) gout[(long long)(row0 + r) * gld + n] = v;
    }
    __syncthreads();
}

// Transposed product of a wide layer: dxs[r][k] = mask(hs[r][k]) * sum_n dys[r][n] W[k][n],  k < K <= 512, for W [K][N]
// with N % 4 == 0 (the middle layers: 400 x 300, 256 x 256).  The B operand of the MFMA wants, per lane, the weight of ITS
// output k for the instruction's n — a column walk through W (4 bytes every N floats) from memory, so W passes through
// LDS: tiles of 16 weight COLUMNS (all K rows, 64 contiguous bytes per row) are loaded with 16-byte loads into registers
// two tiles ahead of the one being consumed, stored with row pitch 20, and lane l of wave w reads row k = 64 w + l as four
// 16-byte pieces (16 consecutive rows -> 64 distinct banks) against the same 16 entries of dy (A operand, row = lane % 4).
// Output k accumulates over n in four interleaved MFMA chains (n mod 4), added at the end: no cross-lane reduction.
//   dys: LDS [R][dp], ZERO from N up to the next multiple of 16; hs: LDS [R][hp] (the layer's OUTPUT activations of the
//   forward pass: relu mask) or null.  wt: LDS, 2 * kTileFloats.  Ends with a barrier.
//   ldw: row stride of W in floats; a column slice passes W offset to its first column, N = the slice's width and dys =
//   the slice's entries of dy.
template <int RR = R>
__device__ __forceinline__ void dense_bwdT(const float *dys, int dp, int N, const float *__restrict__ W, int ldw, int K,
                                           const float *hs, int hp, float *dxs, int xp, float *wt,
                                           float *__restrict__ gout, long long gld, int row0, int B) {
    constexpr int RB = RR / 4;
    const int tid = threadIdx.x;
    constexpr int kPre = kMaxWidth * (kTileCols / 4) / T;           // 4 pieces of 16 bytes per thread and tile
    static_assert(kPre == 4, "dense_bwdT: four pieces per thread and tile");
    f32x4 pre0[kPre], pre1[kPre];
    // piece i of this thread: weight row kk, columns 4 q .. 4 q + 3 of the tile (rows beyond K - 1 re-read row K - 1: what
    // they produce are outputs k >= K, never written)
    unsigned row_off[kPre], q4[kPre];
    int dst_off[kPre];
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
        const int f = tid + i * T, kk = f >> 2, q = f & 3;
        row_off[i] = (unsigned)min(kk, K - 1) * (unsigned)ldw;
        q4[i] = 4u * q;
        dst_off[i] = kk * kTilePitch + 4 * q;
    }
    const int tiles = (N + kTileCols - 1) / kTileCols;
    // (columns beyond N - 4 re-read the last 4: finite numbers against zeros of dy)
#define RLX_CHAIN_FETCH(pre, t)                                                                               \
    _Pragma("unroll") for (int i = 0; i < kPre; ++i)                                                          \
        pre[i] = *reinterpret_cast<const f32x4 *>(W + row_off[i] + min((unsigned)(t) * kTileCols + q4[i], (unsigned)N - 4u));
#define RLX_CHAIN_STASH(pre, buf)                                                                             \
    _Pragma("unroll") for (int i = 0; i < kPre; ++i) *reinterpret_cast<f32x4 *>((buf) + dst_off[i]) = pre[i];
    const int krow = min((tid >> 6) * 64 + (tid & 63), kMaxWidth - 1);       // this lane's output k
    const float *arow = dys + (tid & 3) * dp;
    f32x4 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#define RLX_CHAIN_TILE(cur, t)                                                                                \
    {                                                                                                         \
        float4 bq[4], aq[RB][4];                                                                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                       \
            bq[q] = *reinterpret_cast<const float4 *>((cur) + krow * kTilePitch + 4 * q);                     \
            _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                                 \
                aq[rb][q] = *reinterpret_cast<const float4 *>(arow + 4 * rb * dp + (t) * kTileCols + 4 * q);  \
        }                                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                       \
            _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) {                                               \
                acc[rb][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[rb][q].x, bq[q].x, acc[rb][0], 0, 0, 0);   \
                acc[rb][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[rb][q].y, bq[q].y, acc[rb][1], 0, 0, 0);   \
                acc[rb][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[rb][q].z, bq[q].z, acc[rb][2], 0, 0, 0);   \
                acc[rb][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[rb][q].w, bq[q].w, acc[rb][3], 0, 0, 0);   \
            }                                                                                                 \
        }                                                                                                     \
    }
    float *buf0 = wt, *buf1 = wt + kTileFloats;
    RLX_CHAIN_FETCH(pre0, 0)
    __builtin_amdgcn_sched_barrier(0);
    RLX_CHAIN_STASH(pre0, buf0)
    __builtin_amdgcn_sched_barrier(0);
    RLX_CHAIN_FETCH(pre0, 1)
    RLX_CHAIN_FETCH(pre1, 2)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < tiles; t += 2) {
        // in flight while tile t is consumed: tile t + 1 (pre0) and tile t + 2 (pre1)
#ifndef RLX_CHAIN_ABLATE_MATH
        RLX_CHAIN_TILE(buf0, t)
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef RLX_CHAIN_ABLATE_LOADS
        RLX_CHAIN_STASH(pre0, buf1)
        __builtin_amdgcn_sched_barrier(0);
        RLX_CHAIN_FETCH(pre0, t + 3)
#endif
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
#ifndef RLX_CHAIN_ABLATE_MATH
        if (t + 1 < tiles) {
            RLX_CHAIN_TILE(buf1, t + 1)
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef RLX_CHAIN_ABLATE_LOADS
        RLX_CHAIN_STASH(pre1, buf0)
        __builtin_amdgcn_sched_barrier(0);
        RLX_CHAIN_FETCH(pre1, t + 4)
#endif
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#undef RLX_CHAIN_FETCH
#undef RLX_CHAIN_STASH
#undef RLX_CHAIN_TILE
    const int k = (tid >> 6) * 64 + (tid & 63);
    if (k < K) {
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            float v = (acc[r >> 2][0][r & 3] + acc[r >> 2][1][r & 3]) + (acc[r >> 2][2][r & 3] + acc[r >> 2][3][r & 3]);
            if (hs && !(hs[r * hp + k] > 0.f)) v = 0.f;
            dxs[r * xp + k] = v;
            if (gout && row0 + r < B) gout[(long long)(row0 + r) * gld + k] = v;
        }
    }
    // the consumers read their input in steps of 16: zero the tail
    for (int e = tid; e < RR * (pad16(K) - K); e += T) {
        const int r = e / (pad16(K) - K), c = e - r * (pad16(K) - K);
        dxs[r * xp + K + c] = 0.f;
    }
    __syncthreads();
}

// Transposed product of a NARROW layer (heads, N <= 64): dxs[r][k] = mask * sum_n dys[r][n] W[k][n]; thread k reads its
// N contiguous weights (adjacent threads, adjacent rows: the wave covers one contiguous span).
template <int RR = R>
__device__ __forceinline__ void dense_bwdT_few_cols(const float *dys, int dp, int N, const float *__restrict__ W, int K,
                                                    const float *hs, int hp, float *dxs, int xp, float *__restrict__ gout,
                                                    long long gld, int row0, int B) {
    for (int k = threadIdx.x; k < pad16(K); k += T) {
        float acc[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) acc[r] = 0.f;
        if (k < K)
            for (int n = 0; n < N; ++n) {
                const float w = W[(size_t)k * N + n];
#pragma unroll
                for (int r = 0; r < RR; ++r) acc[r] = fmaf(dys[r * dp + n], w, acc[r]);
            }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            float v = acc[r];
            if (k >= K || (hs && !(hs[r * hp + k] > 0.f))) v = 0.f;
            dxs[r * xp + k] = v;
            if (gout && k < K && row0 + r < B) gout[(long long)(row0 + r) * gld + k] = v;
        }
    }
    __syncthreads();
}

// Transposed product onto FEW outputs (the action columns of a critic's first layer, K <= 64 rows of W): a wave per
// (row r, output k), lanes along n (coalesced), butterfly sum.  dxs[r][k] = scale * sum_n dys[r][n] W[k][n].
template <int RR = R>
__device__ __forceinline__ void dense_bwdT_few_rows(const float *dys, int dp, int N, const float *__restrict__ W, long long ldw,
                                                    int K, float scale, float *dxs, int xp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < RR * K; o += T / 64) {
        const int r = o / K, k = o - r * K;
        float s = 0.f;
        for (int n = lane; n < N; n += 64) s = fmaf(dys[r * dp + n], W[(size_t)k * ldw + n], s);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) dxs[r * xp + k] = scale * s;
    }
    __syncthreads();
}

}  // namespace rlx_chain
