// Device bodies of the convolution weight-gradient kernels that keep an image's (or image pair's) operands in LDS —
// conv_dw_u8.hip (first layer, uint8 frames), conv_dw_f32.hip (inner layers, fp32 activations) — shared with
// conv_dw_multi.hip, which runs all three layers of the Atari torso as ONE launch.  See those files for what each replaces.
#pragma once
#include "rlx_common.hpp"
#include <type_traits>

namespace rlx_convdw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256, kN = 64;

// one 16-byte global -> LDS request per lane; lds_dst: wave-uniform LDS byte address of lane 0's 16 bytes (gemm.hip dma16)
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}


// ------------------------------------------------------------------------------------------------ first layer, uint8 frames
struct DwU8 {
    const unsigned char *x;
    const float *dz;
    long long dz_ts;                  // tower stride of dz (floats)
    float *part, *cpart;              // [B][K][64], [B][64]
    float a_div;
    int B, H, W, C, KH, S, OH, OW, Co, K, P, rowf, NQ;
    long long *stamps;
};

#define RLX_DWU8_STAMP(i) do { if (a.stamps && bid == 0 && threadIdx.x == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

template <int OWT, int SCT, int ROWD>
__device__ __forceinline__ void conv_dw_u8_body(const DwU8 &a, const int bid, float *smem) {
    float *lut = smem;                                   // [256]; later the column-sum scratch
    float *xf = lut + 256;                               // [OH][2][rowf]
    float *dzl = xf + (size_t)a.OH * 2 * a.rowf;         // [P][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (image, kernel-row pair): x = bid % 8 is the XCD, the pairs of an image share it
    const int xcd = bid & 7, rest = bid >> 3;
    const int q = rest % a.NQ, b = (rest / a.NQ) * 8 + xcd;
    if (b >= a.B) return;
    RLX_DWU8_STAMP(0);
    lut[tid] = (float)tid / a.a_div;                     // kThreads == 256: gemm.hip's table, the same division
    // ---- frame bytes of the 2 OH rows (registers), then dz1 of the image (LDS DMA): all requested before any wait
    constexpr int kF = 16;                               // dwords per thread: 2 OH rowf / 4 <= 16 x 256 (checked by the host)
    const int rowd = ROWD > 0 ? ROWD : a.rowf >> 2, nd = 2 * a.OH * rowd;        // (a compile-time divisor in the specialised kernel)
    const unsigned char *img = a.x + (size_t)b * a.H * a.rowf;
    unsigned fb[kF];
#pragma unroll
    for (int j = 0; j < kF; ++j) {
        const int idx = min(tid + j * kThreads, nd - 1), r = idx / rowd, d = idx - r * rowd;
        const int src_row = a.S * (r >> 1) + 2 * q + (r & 1);
        fb[j] = *reinterpret_cast<const unsigned *>(img + (size_t)src_row * a.rowf + 4 * d);
    }
    {
        const int nblk = a.P >> 2;                       // 1 KB blocks of the dz tile: 4 positions x 64 channels
        const unsigned dz_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dzl));     // LDS byte address
        const int pos_l = lane >> 4, n4 = (lane & 15) * 4, tw = n4 / a.Co, ch = n4 - tw * a.Co;
        const float *src0 = a.dz + (size_t)tw * a.dz_ts + ((size_t)b * a.P + pos_l) * a.Co + ch;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        for (int blk = wave_u; blk < nblk; blk += kThreads / 64)
            dma16(src0 + (size_t)blk * 4 * a.Co, __builtin_amdgcn_readfirstlane(dz_base + (unsigned)blk * 1024u));
    }
    __syncthreads();                                     // the table (the DMA requests stay in flight: raw barrier is not needed, no wait is issued for them here)
    RLX_DWU8_STAMP(1);
#pragma unroll
    for (int j = 0; j < kF; ++j) {
        const int idx = tid + j * kThreads;
        if (j * kThreads < nd && idx < nd) {
            const unsigned w = fb[j];
            const f32x4 v = {lut[w & 255u], lut[(w >> 8) & 255u], lut[(w >> 16) & 255u], lut[w >> 24]};
            *reinterpret_cast<f32x4 *>(xf + 4 * idx) = v;          // [r][d]: rows are contiguous, r * rowf + 4 d == 4 idx
        }
    }
    if constexpr (OWT > 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // 13 of a wave's 25 requests: positions < 208
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RLX_DWU8_STAMP(2);
    // ---- 32 weight rows (kernel row 2 q + mh, all (kx, c)) x 32 channels per wave over the image's positions
    const int mh = wave & 1, nh = wave >> 1, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float *ap = xf + mh * a.rowf + l31;
    const float *bp = dzl + nh * 32 + l31;
    const int sc = a.S * a.C, row2 = 2 * a.rowf;
    if constexpr (OWT > 0) {
        // geometry known at compile time (OW even): a row of positions is OWT / 2 steps whose operand addresses are constant
        // offsets from two row pointers — no address arithmetic between the products; the next row's operands are read
        // while this row's products issue (one wave per SIMD: nothing else hides the LDS latency)
        constexpr int kS = OWT / 2;
        float av[2][kS], bv[2][kS];
        const float *ar = ap + hi * SCT, *br = bp + hi * kN;
#define RLX_DWU8_READ(buf)                                                  \
    _Pragma("unroll") for (int u = 0; u < kS; ++u) {                        \
        av[buf][u] = ar[2 * SCT * u];                                       \
        bv[buf][u] = br[2 * kN * u];                                        \
    }                                                                       \
    ar += row2; br += OWT * kN;
        // two accumulator chains (even / odd steps of a row, added at the end): a single dependent chain of 32x32x2 products
        // issues every ~85 cycles, two interleaved ones every 64 (tools/conv_dw_u8_phases.py)
        f32x16 acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
#define RLX_DWU8_MATH(buf)                                                                                       \
    _Pragma("unroll") for (int u = 0; u < kS; u += 2) {                                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u], bv[buf][u], acc, 0, 0, 0);                        \
        if (u + 1 < kS) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u + 1], bv[buf][u + 1], acc1, 0, 0, 0); \
    }
        // the image's dz arrives in position order (1 KB block b = positions 4 b .. 4 b + 3 is request b / 4 of wave b % 4):
        // the rows of the first half were waited for above, the second half lands under their products
        constexpr int kHalf = 10;
#define RLX_DWU8_ROWS(r0, r1)                                         \
    {                                                                 \
        RLX_DWU8_READ(0)                                              \
        for (int py = (r0); py < (r1); py += 2) {                     \
            RLX_DWU8_READ(1)                                          \
            RLX_DWU8_MATH(0)                                          \
            if (py + 2 < (r1)) { RLX_DWU8_READ(0) }                   \
            RLX_DWU8_MATH(1)                                          \
        }                                                             \
    }
        RLX_DWU8_ROWS(0, kHalf)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        RLX_DWU8_ROWS(kHalf, 2 * kHalf)
#undef RLX_DWU8_ROWS
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
#undef RLX_DWU8_READ
#undef RLX_DWU8_MATH
    } else {
        int px = hi, aoff = hi * sc;
        while (px >= a.OW) { px -= a.OW; aoff += row2 - a.OW * sc; }
        const int steps = a.P >> 1;                       // a multiple of 8 (checked by the host)
        const int wrap_add = row2 - a.OW * sc;
        for (int st0 = 0; st0 < steps; st0 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = ap[aoff];
                bv[u] = bp[(2 * (st0 + u) + hi) * kN];
                const bool wrap = px + 2 >= a.OW;
                px = wrap ? px + 2 - a.OW : px + 2;
                aoff += wrap ? 2 * sc + wrap_add : 2 * sc;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    RLX_DWU8_STAMP(3);
    // ---- the image's partial: rows (reg & 3) + 8 (reg >> 2) + 4 hi of the wave's 32, column = lane & 31
    float *out = a.part + ((size_t)b * a.K + q * 64 + mh * 32) * kN + nh * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * kN] = acc[r];
    if (q == 0) {
        // bias gradient: column sums of dz1 over the image's positions (4 interleaved chains per column, added in order)
        const int n = tid & 63, g = tid >> 6;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                    // P % 16 == 0
        for (int pos = g; pos < a.P; pos += 16) {
            s0 += dzl[pos * kN + n];
            s1 += dzl[(pos + 4) * kN + n];
            s2 += dzl[(pos + 8) * kN + n];
            s3 += dzl[(pos + 12) * kN + n];
        }
        lut[g * 64 + n] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < 64) a.cpart[(size_t)b * kN + tid] = ((lut[tid] + lut[64 + tid]) + lut[128 + tid]) + lut[192 + tid];
    }
    RLX_DWU8_STAMP(4);
}


// The same product in NP PASSES over the image's output rows (Atari geometry only: OH = OW = 20, rowf = 336): LDS holds the
// frame rows and the dz of 20 / NP output rows at a time — 79 KB (NP = 2) or 40 KB (NP = 4) instead of 157 KB, so that two /
// three workgroups share a CU and one workgroup's fills (its binding resource: ~30 GB/s per CU, 129 KB per workgroup) run
// under the others' products.  The frame bytes of all passes are requested at entry (registers); a pass's dz is requested
// when the previous pass's products are done.  Same sums: a wave's accumulators run through the positions in the same
// order, and the bias gradient's four chains per column (positions g + 16 k + {0, 4, 8, 12}) take the same positions in the
// same order, pass by pass.
template <int NP>
__device__ __forceinline__ void conv_dw_u8_body_passes(const DwU8 &a, const int bid, float *smem) {
    constexpr int OWT = 20, SCT = 16, ROWD = 84, kRows = 20 / NP, kPP = kRows * OWT;  // positions per pass
    constexpr int kRowF = 4 * ROWD;                                                    // 336
    static_assert(NP == 2 || NP == 4, "conv_dw_u8_body_passes: 2 or 4 passes");
    float *lut = smem;                                   // [256]; later the column-sum scratch
    float *xf = lut + 256;                               // [kRows][2][rowf]
    float *dzl = xf + kRows * 2 * kRowF;                 // [kPP][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = bid & 7, rest = bid >> 3;
    const int q = rest % a.NQ, b = (rest / a.NQ) * 8 + xcd;
    if (b >= a.B) return;
    RLX_DWU8_STAMP(0);
    lut[tid] = (float)tid / a.a_div;
    constexpr int kF = 14, kND = 2 * 20 * ROWD, kNDP = kND / NP;                     // 3360 dwords, per pass 1680 / 840
    const unsigned char *img = a.x + (size_t)b * a.H * kRowF;
    unsigned fb[kF];
#pragma unroll
    for (int j = 0; j < kF; ++j) {
        const int idx = min(tid + j * kThreads, kND - 1), r = idx / ROWD, d = idx - r * ROWD;
        const int src_row = a.S * (r >> 1) + 2 * q + (r & 1);
        fb[j] = *reinterpret_cast<const unsigned *>(img + (size_t)src_row * kRowF + 4 * d);
    }
    const unsigned dz_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dzl));
    const int pos_l = lane >> 4, n4 = (lane & 15) * 4, tw = n4 / a.Co, ch = n4 - tw * a.Co;
    const float *src0 = a.dz + (size_t)tw * a.dz_ts + ((size_t)b * a.P + pos_l) * a.Co + ch;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto request_dz = [&](const int pass) {              // kPP / 4 blocks of 1 KB: 4 positions x 64 channels each
        for (int blk = wave_u; blk < kPP / 4; blk += kThreads / 64)
            dma16(src0 + (size_t)(pass * (kPP / 4) + blk) * 4 * a.Co, __builtin_amdgcn_readfirstlane(dz_base + (unsigned)blk * 1024u));
    };
    auto convert = [&](const int pass) {                 // this pass's frame rows: dwords [pass * kNDP, (pass + 1) * kNDP)
#pragma unroll
        for (int j = 0; j < kF; ++j) {
            const int idx = tid + j * kThreads - pass * kNDP;
            if (idx >= 0 && idx < kNDP) {
                const unsigned w = fb[j];
                const f32x4 v = {lut[w & 255u], lut[(w >> 8) & 255u], lut[(w >> 16) & 255u], lut[w >> 24]};
                *reinterpret_cast<f32x4 *>(xf + 4 * idx) = v;
            }
        }
    };
    const int mh = wave & 1, nh = wave >> 1, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc1[i] = 0.f;
    const float *const ap = xf + mh * kRowF + l31;
    const float *const bp = dzl + nh * 32 + l31;
    constexpr int row2 = 2 * kRowF, kS = OWT / 2;
    float av[2][kS], bv[2][kS];
    const int cn = tid & 63, cg = tid >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // the bias gradient's chains of column cn, group cg (q == 0)
    auto rows = [&]() {
        const float *ar = ap + hi * SCT, *br = bp + hi * kN;
#define RLX_DWU8_READ(buf)                                                  \
    _Pragma("unroll") for (int u = 0; u < kS; ++u) {                        \
        av[buf][u] = ar[2 * SCT * u];                                       \
        bv[buf][u] = br[2 * kN * u];                                        \
    }                                                                       \
    ar += row2; br += OWT * kN;
#define RLX_DWU8_MATH(buf)                                                                                       \
    _Pragma("unroll") for (int u = 0; u < kS; u += 2) {                                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u], bv[buf][u], acc, 0, 0, 0);                        \
        if (u + 1 < kS) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u + 1], bv[buf][u + 1], acc1, 0, 0, 0); \
    }
        // (kRows is a compile-time count: the loop unrolls, every read is unconditional; sched_barrier keeps a row's reads in
        // one block in front of the previous row's products — the scheduler otherwise sinks each read to its use: two reads,
        // a full lgkmcnt(0) wait, two products)
        RLX_DWU8_READ(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int py = 0; py < kRows; py += 2) {
            if (py + 1 < kRows) { RLX_DWU8_READ(1) }
            __builtin_amdgcn_sched_barrier(0);
            RLX_DWU8_MATH(0)
            __builtin_amdgcn_sched_barrier(0);
            if (py + 1 < kRows) {
                if (py + 2 < kRows) { RLX_DWU8_READ(0) }
                __builtin_amdgcn_sched_barrier(0);
                RLX_DWU8_MATH(1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef RLX_DWU8_READ
#undef RLX_DWU8_MATH
    };
    auto column_sums = [&](const int pass) {             // positions [pass * kPP, (pass + 1) * kPP) of this thread's group:
        const int first = pass * kPP;                    // global position cg + 4 i belongs to chain i % 4
        for (int pos = cg + ((first - cg + 3) & ~3); pos < first + kPP; pos += 4) {
            const float v = dzl[(pos - first) * kN + cn];
            const int chain = ((pos - cg) >> 2) & 3;
            s0 += chain == 0 ? v : 0.f;
            s1 += chain == 1 ? v : 0.f;
            s2 += chain == 2 ? v : 0.f;
            s3 += chain == 3 ? v : 0.f;
        }
    };
    request_dz(0);
    __syncthreads();                                     // the table
    RLX_DWU8_STAMP(1);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass > 0) {
            __syncthreads();                             // the previous pass's operands are no longer read
            request_dz(pass);
        }
        convert(pass);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (pass == 0) { RLX_DWU8_STAMP(2); }
        rows();
        if (q == 0) column_sums(pass);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
    RLX_DWU8_STAMP(3);
    float *out = a.part + ((size_t)b * a.K + q * 64 + mh * 32) * kN + nh * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * kN] = acc[r];
    if (q == 0) {
        __syncthreads();                                 // (the table's words: every wave has converted its last pass)
        lut[cg * 64 + cn] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < 64) a.cpart[(size_t)b * kN + tid] = ((lut[tid] + lut[64 + tid]) + lut[128 + tid]) + lut[192 + tid];
    }
    RLX_DWU8_STAMP(4);
}
template <int NP>
constexpr size_t lds_u8_passes() { return sizeof(float) * (size_t)(256 + (20 / NP) * 2 * 336 + (20 / NP) * 20 * kN); }

// ONE tower of 32 filters (T * Co == 32: the DQN update's backward pass through the Atari torso, B = 32) — Atari geometry only.
// The staging is conv_dw_u8_body_passes<2>'s (frame rows of a pair of kernel rows as fp32 + the dz of 10 output rows at a
// time, 53 KB: three workgroups per CU); the dz tile is [positions][32], a 1 KB block = 8 positions.  With half the channels
// there is one 32-channel block per kernel row, so the four waves are (kernel row of the pair) x (HALF of the pass's output
// rows): a wave issues 100 instead of 200 products, and the workgroup leaves TWO splits per image (rows 0-4 + 10-14 and rows
// 5-9 + 15-19) of [K][32] in the deferred split-K workspace — 2 B splits in all, summed in split order by the reduction.
constexpr int kNH = 32;
__device__ __forceinline__ void conv_dw_u8_body_half(const DwU8 &a, const int bid, float *smem) {
    constexpr int NP = 2, OWT = 20, SCT = 16, ROWD = 84, kRows = 20 / NP, kPP = kRows * OWT, kRowF = 4 * ROWD;
    constexpr int kRW = kRows / 2;                       // output rows per wave and pass
    float *lut = smem;                                   // [256]; later the column-sum scratch
    float *xf = lut + 256;                               // [kRows][2][rowf]
    float *dzl = xf + kRows * 2 * kRowF;                 // [kPP][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = bid & 7, rest = bid >> 3;
    const int q = rest % a.NQ, b = (rest / a.NQ) * 8 + xcd;
    if (b >= a.B) return;
    RLX_DWU8_STAMP(0);
    lut[tid] = (float)tid / a.a_div;
    constexpr int kF = 14, kND = 2 * 20 * ROWD, kNDP = kND / NP;
    const unsigned char *img = a.x + (size_t)b * a.H * kRowF;
    unsigned fb[kF];
#pragma unroll
    for (int j = 0; j < kF; ++j) {
        const int idx = min(tid + j * kThreads, kND - 1), r = idx / ROWD, d = idx - r * ROWD;
        const int src_row = a.S * (r >> 1) + 2 * q + (r & 1);
        fb[j] = *reinterpret_cast<const unsigned *>(img + (size_t)src_row * kRowF + 4 * d);
    }
    const unsigned dz_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dzl));
    const float *src0 = a.dz + (size_t)b * a.P * kNH + lane * 4;          // the image's dz: [P][32] contiguous
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int kBlk = kPP * kNH / 256;                // 25 blocks of 1 KB per pass
    auto request_dz = [&](const int pass) {
        for (int blk = wave_u; blk < kBlk; blk += kThreads / 64)
            dma16(src0 + (size_t)(pass * kBlk + blk) * 256, __builtin_amdgcn_readfirstlane(dz_base + (unsigned)blk * 1024u));
    };
    auto convert = [&](const int pass) {
#pragma unroll
        for (int j = 0; j < kF; ++j) {
            const int idx = tid + j * kThreads - pass * kNDP;
            if (idx >= 0 && idx < kNDP) {
                const unsigned w = fb[j];
                const f32x4 v = {lut[w & 255u], lut[(w >> 8) & 255u], lut[(w >> 16) & 255u], lut[w >> 24]};
                *reinterpret_cast<f32x4 *>(xf + 4 * idx) = v;
            }
        }
    };
    const int mh = wave & 1, rh = wave >> 1, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc1[i] = 0.f;
    constexpr int row2 = 2 * kRowF, kS = OWT / 2;
    const float *const ap = xf + mh * kRowF + l31 + hi * SCT + rh * kRW * row2;
    const float *const bp = dzl + l31 + hi * kNH + rh * kRW * OWT * kNH;
    float av[2][kS], bv[2][kS];
    auto rows = [&]() {
        const float *ar = ap, *br = bp;
#define RLX_DWU8_READ(buf)                                                  \
    _Pragma("unroll") for (int u = 0; u < kS; ++u) {                        \
        av[buf][u] = ar[2 * SCT * u];                                       \
        bv[buf][u] = br[2 * kNH * u];                                       \
    }                                                                       \
    ar += row2; br += OWT * kNH;
#define RLX_DWU8_MATH(buf)                                                                                       \
    _Pragma("unroll") for (int u = 0; u < kS; u += 2) {                                                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u], bv[buf][u], acc, 0, 0, 0);                        \
        if (u + 1 < kS) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u + 1], bv[buf][u + 1], acc1, 0, 0, 0); \
    }
        RLX_DWU8_READ(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int py = 0; py < kRW; py += 2) {
            if (py + 1 < kRW) { RLX_DWU8_READ(1) }
            __builtin_amdgcn_sched_barrier(0);
            RLX_DWU8_MATH(0)
            __builtin_amdgcn_sched_barrier(0);
            if (py + 1 < kRW) {
                if (py + 2 < kRW) { RLX_DWU8_READ(0) }
                __builtin_amdgcn_sched_barrier(0);
                RLX_DWU8_MATH(1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef RLX_DWU8_READ
#undef RLX_DWU8_MATH
    };
    const int cn = tid & 31, cg = tid >> 5;              // bias gradient (q == 0): column cn, positions cg + 8 k
    float s0 = 0.f, s1 = 0.f;
    request_dz(0);
    __syncthreads();                                     // the table
    RLX_DWU8_STAMP(1);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass > 0) {
            __syncthreads();                             // the previous pass's operands are no longer read
            request_dz(pass);
        }
        convert(pass);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (pass == 0) { RLX_DWU8_STAMP(2); }
        rows();
        if (q == 0) {
            int pos = cg;
            for (; pos + 8 < kPP; pos += 16) {           // two chains, added at the end
                s0 += dzl[pos * kNH + cn];
                s1 += dzl[(pos + 8) * kNH + cn];
            }
            if (pos < kPP) s0 += dzl[pos * kNH + cn];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
    RLX_DWU8_STAMP(3);
    float *out = a.part + ((size_t)(2 * b + rh) * a.K + q * 64 + mh * 32) * kNH + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * kNH] = acc[r];
    if (q == 0) {
        __syncthreads();                                 // (the table's words: every wave has converted its last pass)
        lut[cg * kNH + cn] = s0 + s1;
        __syncthreads();
        if (tid < kNH) {
            float s = lut[tid];
#pragma unroll
            for (int g = 1; g < 8; ++g) s += lut[g * kNH + tid];
            a.cpart[(size_t)(2 * b) * kNH + tid] = s;
            a.cpart[(size_t)(2 * b + 1) * kNH + tid] = 0.f;
        }
    }
    RLX_DWU8_STAMP(4);
}
constexpr size_t lds_u8_half() { return sizeof(float) * (size_t)(256 + 10 * 2 * 336 + 10 * 20 * kNH); }

struct GeometryU8 {
    int OH, OW, K, P, rowf, NQ;
    size_t lds;
    int nch, splits;                  // output columns of a split (64, or 32: conv_dw_u8_body_half) and splits (B, or 2 B)
};
inline bool geometry_u8(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, GeometryU8 *g) {
    if (B < 2 || B > 128 || H < KH || W < KW || S < 1 || KH < 2 || (KH & 1)) return false;
    const bool half = T == 1 && Co == kNH;                                           // one tower of 32 filters: conv_dw_u8_body_half
    if (KW * C != 32 || (T * Co != kN && !half) || (Co & 3) || (kN % Co)) return false;   // a patch row = 32 floats, 64 folded channels
    g->OH = (H - KH) / S + 1;
    g->OW = (W - KW) / S + 1;
    g->K = KH * KW * C;
    g->P = g->OH * g->OW;
    g->rowf = W * C;
    g->NQ = KH / 2;
    g->nch = half ? kNH : kN;
    g->splits = half ? 2 * B : B;
    if (half) {                                                                      // the Atari geometry only, 2 B <= 128 splits
        if (g->OH != 20 || g->OW != 20 || S * C != 16 || g->rowf != 336 || KH != 8 || B > 64) return false;
        g->lds = lds_u8_half();
        return true;
    }
    if ((g->rowf & 3) || (g->P & 15) || g->OW < 2) return false;
    if (2 * g->OH * (g->rowf / 4) > 16 * kThreads) return false;
    g->lds = sizeof(float) * (256 + (size_t)g->OH * 2 * g->rowf + (size_t)g->P * kN);
    return g->lds <= 160 * 1024;
}


// ------------------------------------------------------------------------------------------- inner layers, fp32 activations
struct DwF32 {
    const float *x; long long x_ts;   // [T][B][H][W][C]
    const float *dz; long long dz_ts; // [T][B * P][64]
    float *part, *cpart;              // [T][splits][K][64], [T][splits][64]
    int B, H, OH, KH, splits, units;  // units = towers * splits
    long long *stamps;
    int ppw;                          // image pairs per workgroup (two-pass body; 1: a split per pair)
};

#define RLX_DWF_STAMP(i) do { if (a.stamps && bid == 0 && threadIdx.x == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

// C input channels, KW kernel columns, S stride, W input columns, OW output columns
template <int C, int KW, int S, int W, int OW>
__device__ __forceinline__ void conv_dw_f32_body(const DwF32 &a, const int bid, float *smem) {
    constexpr int kRowF = W * C;                          // floats of an input row
    constexpr int kMT = KW * C;                           // weight rows of a kernel row
    constexpr int kSub = kMT / 32 * 2 / 4;                // 32 x 32 blocks per wave
    static_assert(kMT % 64 == 0 && (OW - 1) * S + KW <= W, "conv_dw_f32: geometry");
    const int OH = a.OH, P = OH * OW;
    const int xf_floats = (2 * OH * kRowF + 255) & ~255, dz_floats = (2 * P * kN + 255) & ~255;
    float *xf = smem;                                     // [2 images][OH][kRowF]
    float *dzl = xf + xf_floats;                          // [2 images][P][64]
    float *red = dzl + dz_floats;                         // [256] column-sum scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (tower, image pair, kernel row); the KH blocks of a pair are 8 apart in block order (same XCD: the pair's
    // operands come through one L2)
    const int xcd = bid & 7, rest = bid >> 3;
    const int ky = rest % a.KH, pt = (rest / a.KH) * 8 + xcd;          // pt = tower * splits + pair
    if (pt >= a.units) return;
    const int t = pt / a.splits, g = pt - t * a.splits;
    const int img0 = 2 * g, n_img = min(2, a.B - img0);
    RLX_DWF_STAMP(0);
    {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // input rows S py + ky of the pair's images -> xf (an image that does not exist repeats the first one: its dz is zeroed)
        const unsigned xf_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xf));
        const float *xsrc = a.x + (size_t)t * a.x_ts;
        const int per_img = OH * kRowF;
        for (int blk = wave_u; blk < xf_floats / 256; blk += kThreads / 64) {
            const int f = min(blk * 256 + lane * 4, 2 * per_img - 4);
            const int im = f / per_img, r0 = f - im * per_img, py = r0 / kRowF, col = r0 - py * kRowF;
            const int img = img0 + min(im, n_img - 1);
            dma16(xsrc + ((size_t)img * a.H + (S * py + ky)) * kRowF + col,
                  __builtin_amdgcn_readfirstlane(xf_base + (unsigned)blk * 1024u));
        }
        const unsigned dz_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dzl));
        const float *dsrc = a.dz + (size_t)t * a.dz_ts + (size_t)img0 * P * kN;
        const int have = n_img * P * kN;
        for (int blk = wave_u; blk < dz_floats / 256; blk += kThreads / 64) {
            const int f = min(blk * 256 + lane * 4, have - 4);
            dma16(dsrc + f, __builtin_amdgcn_readfirstlane(dz_base + (unsigned)blk * 1024u));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_img < 2) {                                      // the second image of the last pair of an odd batch: no contribution
        for (int e = tid; e < P * kN; e += kThreads) dzl[P * kN + e] = 0.f;
        __syncthreads();
    }
    RLX_DWF_STAMP(1);
    // ---- wave: filter half nh, weight-row blocks mb = (wave >> 1) + 2 j; lane half hi = image of the pair
    const int nh = wave & 1, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[kSub];
#pragma unroll
    for (int j = 0; j < kSub; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const float *ar = xf + hi * OH * kRowF + 32 * (wave >> 1) + l31;
    const float *br = dzl + hi * P * kN + nh * 32 + l31;
    float av[2][OW][kSub], bv[2][OW];
#define RLX_DWF_READ(buf)                                                          \
    _Pragma("unroll") for (int u = 0; u < OW; ++u) {                               \
        _Pragma("unroll") for (int j = 0; j < kSub; ++j) av[buf][u][j] = ar[S * C * u + 64 * j]; \
        bv[buf][u] = br[kN * u];                                                   \
    }                                                                              \
    ar += kRowF; br += OW * kN;
#define RLX_DWF_MATH(buf)                                                          \
    _Pragma("unroll") for (int u = 0; u < OW; ++u)                                 \
        _Pragma("unroll") for (int j = 0; j < kSub; ++j)                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][j], bv[buf][u], acc[j], 0, 0, 0);
    RLX_DWF_READ(0)
    for (int py = 0; py < OH; py += 2) {
        if (py + 1 < OH) { RLX_DWF_READ(1) }
        RLX_DWF_MATH(0)
        if (py + 1 < OH) {
            if (py + 2 < OH) { RLX_DWF_READ(0) }
            RLX_DWF_MATH(1)
        }
    }
#undef RLX_DWF_READ
#undef RLX_DWF_MATH
    RLX_DWF_STAMP(2);
    // ---- the pair's partial: block rows (reg & 3) + 8 (reg >> 2) + 4 hi, column = lane & 31
    const int K = a.KH * kMT;
    float *out = a.part + (((size_t)t * a.splits + g) * K + (size_t)ky * kMT + 32 * (wave >> 1)) * kN + nh * 32 + l31;
#pragma unroll
    for (int j = 0; j < kSub; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)(64 * j + (r & 3) + 8 * (r >> 2) + 4 * hi) * kN] = acc[j][r];
    if (ky == 0 && a.cpart) {
        // bias gradient: column sums of dz over the pair's positions (4 interleaved chains per column, added in order)
        const int n = tid & 63, q = tid >> 6, rows = 2 * P;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int pos = q;
        for (; pos + 12 < rows; pos += 16) {                  // four independent chains: the reads of a step overlap
            s0 += dzl[pos * kN + n];
            s1 += dzl[(pos + 4) * kN + n];
            s2 += dzl[(pos + 8) * kN + n];
            s3 += dzl[(pos + 12) * kN + n];
        }
        for (; pos < rows; pos += 4) s0 += dzl[pos * kN + n];
        red[q * 64 + n] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < 64) a.cpart[((size_t)t * a.splits + g) * kN + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    }
    RLX_DWF_STAMP(3);
}


// conv_dw_f32_body in TWO PASSES over the output rows ([0, R0) and [R0, OH)): the pair's input rows and dz of one pass at a
// time in LDS (conv2 of the Atari torso: 50 KB instead of 88 KB), for the same reason as conv_dw_u8_body_two_pass.  The
// accumulators run through the positions in the same order; the bias gradient's column sums are taken pass by pass
// (another grouping of that sum than the one-pass body's).
template <int C, int KW, int S, int W, int OW, int R0, int OHT>
__device__ __forceinline__ void conv_dw_f32_body_two_pass(const DwF32 &a, const int bid, float *smem) {
    constexpr int kRowF = W * C;
    constexpr int kMT = KW * C;
    constexpr int kSub = kMT / 32 * 2 / 4;
    static_assert(kMT % 64 == 0 && (OW - 1) * S + KW <= W, "conv_dw_f32: geometry");
    const int P = OHT * OW;
    constexpr int xf_floats = (2 * R0 * kRowF + 255) & ~255, dz_floats = (2 * R0 * OW * kN + 255) & ~255;
    float *xf = smem;                                     // [2 images][rows of the pass][kRowF]
    float *dzl = xf + xf_floats;                          // [2 images][positions of the pass][64]
    float *red = dzl + dz_floats;                         // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = bid & 7, rest = bid >> 3;
    const int ky = rest % a.KH, pt = (rest / a.KH) * 8 + xcd;
    if (pt >= a.units) return;
    const int t = pt / a.splits, g = pt - t * a.splits;
    // a.ppw image pairs per workgroup, one after the other into the same accumulators (ppw = 2: half the splits — half the
    // partial sums written here and read by the deferred reduction)
    const int ppw = a.ppw > 1 ? a.ppw : 1, n_pairs = (a.B + 1) >> 1;
    int img0 = 2 * g * ppw, n_img = min(2, a.B - img0);
    RLX_DWF_STAMP(0);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned xf_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xf));
    const unsigned dz_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(dzl));
    const float *xsrc = a.x + (size_t)t * a.x_ts;
    const float *dsrc = a.dz + (size_t)t * a.dz_ts;
    auto stage = [&](const int r0, const int nr) {        // output rows [r0, r0 + nr)
        const int per_img = nr * kRowF, per_dz = nr * OW * kN;
        for (int blk = wave_u; blk * 256 < 2 * per_img; blk += kThreads / 64) {
            const int f = min(blk * 256 + lane * 4, 2 * per_img - 4);
            const int im = f / per_img, q0 = f - im * per_img, py = q0 / kRowF, col = q0 - py * kRowF;
            const int img = img0 + min(im, n_img - 1);
            dma16(xsrc + ((size_t)img * a.H + (S * (r0 + py) + ky)) * kRowF + col,
                  __builtin_amdgcn_readfirstlane(xf_base + (unsigned)blk * 1024u));
        }
        for (int blk = wave_u; blk * 256 < 2 * per_dz; blk += kThreads / 64) {
            const int f = min(blk * 256 + lane * 4, 2 * per_dz - 4);
            const int im = f / per_dz, off = f - im * per_dz;
            const int img = img0 + min(im, n_img - 1);
            dma16(dsrc + ((size_t)img * P + (size_t)r0 * OW) * kN + off,
                  __builtin_amdgcn_readfirstlane(dz_base + (unsigned)blk * 1024u));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (n_img < 2) {                                  // the second image of the last pair of an odd batch: no contribution
            for (int e = tid; e < per_dz; e += kThreads) dzl[per_dz + e] = 0.f;
            __syncthreads();
        }
    };
    const int nh = wave & 1, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[kSub];
#pragma unroll
    for (int j = 0; j < kSub; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float av[2][OW][kSub], bv[2][OW];
    const int cn = tid & 63, cq = tid >> 6;
    float csum = 0.f;
    // NR is a compile-time row count: the row loop unrolls completely and every LDS read is unconditional — with the next
    // row's reads under a run-time "is there a next row" the compiler waits for ALL outstanding LDS reads (lgkmcnt(0)) in
    // front of a row's products, the double buffer hides nothing (profiles/r06_conv32_tail16.txt, the same mechanism)
    auto rows = [&](auto nr_c) {
        constexpr int nr = decltype(nr_c)::value;
        const float *ar = xf + hi * nr * kRowF + 32 * (wave >> 1) + l31;
        const float *br = dzl + hi * nr * OW * kN + nh * 32 + l31;
#define RLX_DWF_READ(buf)                                                          \
    _Pragma("unroll") for (int u = 0; u < OW; ++u) {                               \
        _Pragma("unroll") for (int j = 0; j < kSub; ++j) av[buf][u][j] = ar[S * C * u + 64 * j]; \
        bv[buf][u] = br[kN * u];                                                   \
    }                                                                              \
    ar += kRowF; br += OW * kN;
#define RLX_DWF_MATH(buf)                                                          \
    _Pragma("unroll") for (int u = 0; u < OW; ++u)                                 \
        _Pragma("unroll") for (int j = 0; j < kSub; ++j)                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][j], bv[buf][u], acc[j], 0, 0, 0);
        // (sched_barrier: the scheduler otherwise sinks every read to its use — two reads, a full wait, two products)
        RLX_DWF_READ(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int py = 0; py < nr; py += 2) {
            if (py + 1 < nr) { RLX_DWF_READ(1) }
            __builtin_amdgcn_sched_barrier(0);
            RLX_DWF_MATH(0)
            __builtin_amdgcn_sched_barrier(0);
            if (py + 1 < nr) {
                if (py + 2 < nr) { RLX_DWF_READ(0) }
                __builtin_amdgcn_sched_barrier(0);
                RLX_DWF_MATH(1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef RLX_DWF_READ
#undef RLX_DWF_MATH
        if (ky == 0 && a.cpart)                           // bias gradient: this thread's share of column cn over the pass
            for (int pos = cq; pos < 2 * nr * OW; pos += 4) csum += dzl[pos * kN + cn];
    };
    for (int pp = 0; pp < ppw && g * ppw + pp < n_pairs; ++pp) {
        if (pp > 0) {
            __syncthreads();                              // the previous pair's operands are no longer read
            img0 = 2 * (g * ppw + pp);
            n_img = min(2, a.B - img0);
        }
        stage(0, R0);
        if (pp == 0) { RLX_DWF_STAMP(1); }
        rows(std::integral_constant<int, R0>());
        __syncthreads();                                  // pass 0's operands are no longer read
        stage(R0, OHT - R0);
        rows(std::integral_constant<int, OHT - R0>());
    }
    RLX_DWF_STAMP(2);
    const int K = a.KH * kMT;
    float *out = a.part + (((size_t)t * a.splits + g) * K + (size_t)ky * kMT + 32 * (wave >> 1)) * kN + nh * 32 + l31;
#pragma unroll
    for (int j = 0; j < kSub; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)(64 * j + (r & 3) + 8 * (r >> 2) + 4 * hi) * kN] = acc[j][r];
    if (ky == 0 && a.cpart) {
        red[cq * 64 + cn] = csum;
        __syncthreads();
        if (tid < 64) a.cpart[((size_t)t * a.splits + g) * kN + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    }
    RLX_DWF_STAMP(3);
}
template <int C, int W, int OW, int R0>
constexpr size_t lds_f32_two_pass() {
    return sizeof(float) * (size_t)(((2 * R0 * W * C + 255) & ~255) + ((2 * R0 * OW * kN + 255) & ~255) + 256);
}

struct GeometryF32 {
    int OH, OW, K, P, splits, kind;   // kind 1: 3 x 3 x 64 stride 1 on 9 x 9; 2: 4 x 4 x 32 stride 2 on 20 x 20
    size_t lds;
};
inline bool geometry_f32(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, GeometryF32 *g, int ppw = 1) {
    if (B < 3 || Co != kN || T < 1) return false;
    g->kind = 0;
    if (H == 9 && W == 9 && C == 64 && KH == 3 && KW == 3 && S == 1) g->kind = 1;
    if (H == 20 && W == 20 && C == 32 && KH == 4 && KW == 4 && S == 2) g->kind = 2;
    if (!g->kind) return false;
    g->OH = (H - KH) / S + 1;
    g->OW = (W - KW) / S + 1;
    g->K = KH * KW * C;
    g->P = g->OH * g->OW;
    g->splits = ((B + 1) / 2 + ppw - 1) / ppw;
    if (g->splits > 128) return false;
    g->lds = sizeof(float) * ((size_t)((2 * g->OH * W * C + 255) & ~255) + (size_t)((2 * g->P * kN + 255) & ~255) + 256);
    return g->lds <= 160 * 1024;
}


}  // namespace rlx_convdw
