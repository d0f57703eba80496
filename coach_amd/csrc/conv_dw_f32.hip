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
#include "conv_dw_body.hpp"

namespace {

using namespace rlx_convdw;
typedef GeometryF32 Geometry;
inline bool geometry(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, Geometry *g) {
    return geometry_f32(B, H, W, C, KH, KW, S, Co, T, g);
}

template <int C, int KW, int S, int W, int OW>
__global__ void __launch_bounds__(kThreads) conv_dw_f32_kernel(const DwF32 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_dw_f32_body<C, KW, S, W, OW>(a, blockIdx.x, smem);
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
    a.B = B; a.H = H; a.OH = g.OH; a.KH = KH; a.splits = g.splits; a.ppw = 1;
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
