// Weight gradient of an inner convolution layer from fp32 activations held in LDS (rlx_conv_dw_f32).
//
// Replaces, for the second and third convolution of the Atari torso (4 x 4 x 32 stride 2 on 20 x 20 and 3 x 3 x 64
// stride 1 on 9 x 9, 64 filters; rl_coach/architectures/tensorflow_components/embedders/image_embedder.py:33-40,
// layers.py:108-121, and the tf.gradients pass of architecture.py:187-220) the implicit-im2col product
// dW = cols(x)^T dz of rlx_gemm WHEN IT RUNS ALONE (the input gradients taken by rlx_conv32_input_grad): 14.0 us each in
// the Clipped-PPO minibatch update for 0.46 / 0.34 GFLOP — the tiled kernel gathers every patch element through offset
// tables from memory, workgroup by workgroup.
//
// Shape of the work: M = KH * KW * C weight rows (576 / 512), N = 64 filters, reduction over batch x OH x OW positions
// (64 x 49 / 64 x 81), per tower.  A workgroup owns (tower, PAIR of images, kernel row ky): the OH input rows that kernel
// row touches (S py + ky) of both images and the two images' dz [P][64] sit in LDS once (66 / 88 KB, LDS DMA, no staging
// registers).  Its KW * C = 192 / 128 weight rows x 64 filters are 12 / 8 blocks of 32 x 32, three / two per wave, all of a
// wave's blocks on the same filter half (one B operand read per step).  v_mfma_f32_32x32x2_f32 sums two reduction
// indices per instruction: index 0 = position p of the pair's first image, index 1 = position p of its second — both
// halves of the wave walk the SAME raster, so a row of positions is OW steps at constant operand offsets from two row
// pointers (no address arithmetic between products); the next row's operands are read while this row's products issue.
// The pair's partial goes to the deferred split-K workspace (rlx_splitk_job: one split per image pair, summed in pair
// order by the update's closing reduction launch); the bias gradient (column sums of dz) rides along with ky = 0.
// Grid: towers x pairs x KH = 192 / 256 workgroups for the 64-image, two-tower minibatch.
// Bound: MFMA issue — 147 / 162 products per wave = 4.3 / 4.7 us — behind one round trip for the operands.
#include "rlx_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256, kN = 64;

struct DwF32 {
    const float *x; long long x_ts;   // [T][B][H][W][C]
    const float *dz; long long dz_ts; // [T][B * P][64]
    float *part, *cpart;              // [T][splits][K][64], [T][splits][64]
    int B, H, OH, KH, splits, units;  // units = towers * splits
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

#define RLX_DWF_STAMP(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)

// C input channels, KW kernel columns, S stride, W input columns, OW output columns
template <int C, int KW, int S, int W, int OW>
__global__ void __launch_bounds__(kThreads) conv_dw_f32_kernel(const DwF32 a) {
    constexpr int kRowF = W * C;                          // floats of an input row
    constexpr int kMT = KW * C;                           // weight rows of a kernel row
    constexpr int kSub = kMT / 32 * 2 / 4;                // 32 x 32 blocks per wave
    static_assert(kMT % 64 == 0 && (OW - 1) * S + KW <= W, "conv_dw_f32: geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int OH = a.OH, P = OH * OW;
    const int xf_floats = (2 * OH * kRowF + 255) & ~255, dz_floats = (2 * P * kN + 255) & ~255;
    float *xf = smem;                                     // [2 images][OH][kRowF]
    float *dzl = xf + xf_floats;                          // [2 images][P][64]
    float *red = dzl + dz_floats;                         // [256] column-sum scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (tower, image pair, kernel row); the KH blocks of a pair are 8 apart in block order (same XCD: the pair's
    // operands come through one L2)
    const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3;
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

struct Geometry {
    int OH, OW, K, P, splits, kind;   // kind 1: 3 x 3 x 64 stride 1 on 9 x 9; 2: 4 x 4 x 32 stride 2 on 20 x 20
    size_t lds;
};
inline bool geometry(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, Geometry *g) {
    if (B < 3 || Co != kN || T < 1) return false;
    g->kind = 0;
    if (H == 9 && W == 9 && C == 64 && KH == 3 && KW == 3 && S == 1) g->kind = 1;
    if (H == 20 && W == 20 && C == 32 && KH == 4 && KW == 4 && S == 2) g->kind = 2;
    if (!g->kind) return false;
    g->OH = (H - KH) / S + 1;
    g->OW = (W - KW) / S + 1;
    g->K = KH * KW * C;
    g->P = g->OH * g->OW;
    g->splits = (B + 1) / 2;
    if (g->splits > 128) return false;
    g->lds = sizeof(float) * ((size_t)((2 * g->OH * W * C + 255) & ~255) + (size_t)((2 * g->P * kN + 255) & ~255) + 256);
    return g->lds <= 160 * 1024;
}

static long long *g_stamps = nullptr;

}  // namespace

extern "C" {

int rlx_conv_dw_f32_supported(int B, int H, int W, int C, int KH, int KW, int S, int Co, int towers) {
    Geometry g;
    return geometry(B, H, W, C, KH, KW, S, Co, towers, &g) ? 1 : 0;
}

int rlx_conv_dw_f32_workspace_floats(int B, int H, int W, int C, int KH, int KW, int S, int Co, int towers, long long *floats_host) {
    Geometry g;
    RLX_REQUIRE(floats_host != nullptr, "rlx_conv_dw_f32_workspace_floats: null pointer");
    RLX_REQUIRE(geometry(B, H, W, C, KH, KW, S, Co, towers, &g), "rlx_conv_dw_f32_workspace_floats: unsupported shape");
    *floats_host = (long long)towers * g.splits * ((long long)g.K * kN + kN);
    return RLX_OK;
}

int rlx_conv_dw_f32_stamps(long long *device_words4) {
    g_stamps = device_words4;
    return RLX_OK;
}

int rlx_conv_dw_f32(const float *x, long long x_tower_stride, const float *dz, long long dz_tower_stride, int B, int H, int W, int C,
                    int KH, int KW, int S, int Co, int towers, float *dw, long long dw_tower_stride, float *db,
                    long long db_tower_stride, float *workspace, long long workspace_floats, rlx_splitk_job *job_host,
                    void *stream) {
    Geometry g;
    RLX_REQUIRE(x && dz && dw && workspace && job_host, "rlx_conv_dw_f32: null pointer");
    RLX_REQUIRE(geometry(B, H, W, C, KH, KW, S, Co, towers, &g),
                "rlx_conv_dw_f32: unsupported shape (B=%d %dx%dx%d kernel %dx%d stride %d filters %d towers %d)", B, H, W, C, KH, KW,
                S, Co, towers);
    const long long need = (long long)towers * g.splits * ((long long)g.K * kN + kN);
    RLX_REQUIRE(workspace_floats >= need, "rlx_conv_dw_f32: workspace of %lld floats, need %lld", workspace_floats, need);
    RLX_REQUIRE((((uintptr_t)x | (uintptr_t)dz | (uintptr_t)workspace) & 15) == 0 && (x_tower_stride & 3) == 0 &&
                    (dz_tower_stride & 3) == 0,
                "rlx_conv_dw_f32: misaligned operand");
    static bool configured = false;
    if (!configured) {
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_f32_kernel<64, 3, 1, 9, 7>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_f32_kernel<32, 4, 2, 20, 9>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    DwF32 a;
    a.x = x; a.x_ts = x_tower_stride; a.dz = dz; a.dz_ts = dz_tower_stride;
    a.part = workspace; a.cpart = db ? workspace + (size_t)towers * g.splits * g.K * kN : nullptr;
    a.B = B; a.H = H; a.OH = g.OH; a.KH = KH; a.splits = g.splits;
    a.stamps = g_stamps;
    a.units = towers * g.splits;                                  // (tower, pair)
    const int grid = ((a.units + 7) / 8) * 8 * KH;
    if (g.kind == 1)
        RLX_LAUNCH((conv_dw_f32_kernel<64, 3, 1, 9, 7>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    else
        RLX_LAUNCH((conv_dw_f32_kernel<32, 4, 2, 20, 9>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    rlx_splitk_job &j = *job_host;
    j.partials = a.part; j.colsum_partials = a.cpart;
    j.C = dw; j.colsum_out = db;
    j.ldc = Co; j.c_batch_stride = dw_tower_stride; j.colsum_batch_stride = db_tower_stride;
    j.M = g.K; j.N = kN; j.batch = towers; j.splits = g.splits; j.n_fold = 0;
    return RLX_OK;
}

}  // extern "C"
