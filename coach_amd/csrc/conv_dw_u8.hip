// Weight gradient of the FIRST convolution of an image torso, straight from the uint8 frames (rlx_conv_dw_u8).
//
// Replaces, for the Atari torso's conv1 (8 x 8 x 4 stride 4 on 84 x 84 x 4 uint8 frames, 32 filters per tower;
// rl_coach/architectures/tensorflow_components/embedders/image_embedder.py:33-40, layers.py:108-121, and the
// tf.gradients pass of architecture.py:187-220) the implicit-im2col product dW1 = cols(frames)^T dz1 that rlx_gemm ran
// on its register-staged uint8 loop (gemm_fast_kernel: 18.6 us per Clipped-PPO minibatch update, 0.28 of the fp32 MFMA
// peak — every workgroup re-gathered and re-converted patches byte by byte from memory).
//
// Shape of the work: M = K1 = KH * KW * C = 256 weight rows, N = towers * filters = 64 (both towers read the SAME frames),
// reduction over batch x OH x OW = 64 x 400 positions.  A workgroup owns ONE image and ONE pair of kernel rows (ky = 2 q,
// 2 q + 1 -> 64 weight rows): everything it needs is in LDS once —
//   * the 2 x OH frame rows those kernel rows touch (40 rows x 336 bytes), converted to fp32 once per workgroup through
//     the 256-entry table of byte / a_div (the division the forward pass uses): a position's patch row is 32 CONTIGUOUS
//     floats there (kx, c), i.e. the MFMA A operand of 32 weight rows is one conflict-free ds_read_b32;
//   * the image's dz1 [400 positions][64 channels] (100 KB), fetched by global_load_lds_dwordx4 — no staging registers;
// then 4 waves x 200 v_mfma_f32_32x32x2_f32 (wave = 32 weight rows x 32 channels, two positions per instruction).
// The image's 64 x 64 partial goes to the deferred split-K workspace (rlx_splitk_job: one split per image), where the
// update's closing reduction launch adds the 64 images in image order; the bias gradient (column sums of dz1) rides along.
// Grid: images x KH / 2 = 256 workgroups for the 64-image minibatch; the four workgroups of an image are 8 apart in
// block order (same XCD: the image's dz1 comes through ONE L2).
// Bound: MFMA issue — 200 x 64 cycles per wave = 6.1 us at 2.1 GHz — behind one round trip for 113 KB per workgroup.
#include "rlx_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256, kN = 64;

struct DwU8 {
    const unsigned char *x;
    const float *dz;
    long long dz_ts;                  // tower stride of dz (floats)
    float *part, *cpart;              // [B][K][64], [B][64]
    float a_div;
    int B, H, W, C, KH, S, OH, OW, Co, K, P, rowf, NQ;
    long long *stamps;
};

// one 16-byte global -> LDS request per lane; lds_dst: wave-uniform LDS byte address of lane 0's 16 bytes (gemm.hip dma16)
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

#define RLX_DWU8_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

template <int OWT, int SCT, int ROWD>
__global__ void __launch_bounds__(kThreads) conv_dw_u8_kernel(const DwU8 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lut = smem;                                   // [256]; later the column-sum scratch
    float *xf = lut + 256;                               // [OH][2][rowf]
    float *dzl = xf + (size_t)a.OH * 2 * a.rowf;         // [P][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (image, kernel-row pair): x = bid % 8 is the XCD, the pairs of an image share it
    const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3;
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

struct Geometry {
    int OH, OW, K, P, rowf, NQ;
    size_t lds;
};
inline bool geometry(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, Geometry *g) {
    if (B < 2 || B > 128 || H < KH || W < KW || S < 1 || KH < 2 || (KH & 1)) return false;
    if (KW * C != 32 || T * Co != kN || (Co & 3) || (kN % Co)) return false;       // a patch row = 32 floats, 64 folded channels
    g->OH = (H - KH) / S + 1;
    g->OW = (W - KW) / S + 1;
    g->K = KH * KW * C;
    g->P = g->OH * g->OW;
    g->rowf = W * C;
    g->NQ = KH / 2;
    if ((g->rowf & 3) || (g->P & 15) || g->OW < 2) return false;
    if (2 * g->OH * (g->rowf / 4) > 16 * kThreads) return false;
    g->lds = sizeof(float) * (256 + (size_t)g->OH * 2 * g->rowf + (size_t)g->P * kN);
    return g->lds <= 160 * 1024;
}

static long long *g_stamps = nullptr;

}  // namespace

extern "C" {

int rlx_conv_dw_u8_supported(int B, int H, int W, int C, int KH, int KW, int S, int Co, int towers) {
    Geometry g;
    return geometry(B, H, W, C, KH, KW, S, Co, towers, &g) ? 1 : 0;
}

int rlx_conv_dw_u8_workspace_floats(int B, int H, int W, int C, int KH, int KW, int S, int Co, int towers, long long *floats_host) {
    Geometry g;
    RLX_REQUIRE(floats_host != nullptr, "rlx_conv_dw_u8_workspace_floats: null pointer");
    RLX_REQUIRE(geometry(B, H, W, C, KH, KW, S, Co, towers, &g), "rlx_conv_dw_u8_workspace_floats: unsupported shape");
    *floats_host = (long long)B * ((long long)g.K * kN + kN);
    return RLX_OK;
}

int rlx_conv_dw_u8_stamps(long long *device_words5) {
    g_stamps = device_words5;
    return RLX_OK;
}

int rlx_conv_dw_u8(const unsigned char *frames, float a_div, const float *dz, long long dz_tower_stride, int B, int H, int W, int C,
                   int KH, int KW, int S, int Co, int towers, float *dw, long long dw_tower_stride, float *db,
                   long long db_tower_stride, float *workspace, long long workspace_floats, rlx_splitk_job *job_host,
                   void *stream) {
    Geometry g;
    RLX_REQUIRE(frames && dz && dw && workspace && job_host, "rlx_conv_dw_u8: null pointer");
    RLX_REQUIRE(a_div != 0.f, "rlx_conv_dw_u8: a_div must be non-zero");
    RLX_REQUIRE(geometry(B, H, W, C, KH, KW, S, Co, towers, &g),
                "rlx_conv_dw_u8: unsupported shape (B=%d %dx%dx%d kernel %dx%d stride %d filters %d towers %d)", B, H, W, C, KH, KW,
                S, Co, towers);
    const long long need = (long long)B * ((long long)g.K * kN + kN);
    RLX_REQUIRE(workspace_floats >= need, "rlx_conv_dw_u8: workspace of %lld floats, need %lld", workspace_floats, need);
    RLX_REQUIRE((((uintptr_t)frames) & 3) == 0 && (((uintptr_t)dz) & 15) == 0 && (dz_tower_stride & 3) == 0 &&
                    (((uintptr_t)workspace) & 15) == 0,
                "rlx_conv_dw_u8: misaligned operand");
    static bool configured = false;
    if (!configured) {
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_u8_kernel<20, 16, 84>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_u8_kernel<0, 0, 0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    DwU8 a;
    a.x = frames; a.dz = dz; a.dz_ts = dz_tower_stride;
    a.part = workspace; a.cpart = workspace + (size_t)B * g.K * kN;
    a.a_div = a_div;
    a.B = B; a.H = H; a.W = W; a.C = C; a.KH = KH; a.S = S; a.OH = g.OH; a.OW = g.OW; a.Co = Co; a.K = g.K; a.P = g.P;
    a.rowf = g.rowf; a.NQ = g.NQ;
    a.stamps = g_stamps;
    const int grid = ((B + 7) / 8) * 8 * g.NQ;
    if (g.OW == 20 && g.OH == 20 && S * C == 16 && g.rowf == 336)          // the Atari torso: 84 x 84 x 4 frames, 8 x 8 stride 4
        RLX_LAUNCH((conv_dw_u8_kernel<20, 16, 84>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    else
        RLX_LAUNCH((conv_dw_u8_kernel<0, 0, 0>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    rlx_splitk_job &j = *job_host;
    j.partials = a.part; j.colsum_partials = db ? a.cpart : nullptr;
    j.C = dw; j.colsum_out = db;
    j.ldc = Co; j.c_batch_stride = dw_tower_stride; j.colsum_batch_stride = db_tower_stride;
    j.M = g.K; j.N = kN; j.batch = 1; j.splits = B; j.n_fold = Co;
    return RLX_OK;
}

}  // extern "C"
