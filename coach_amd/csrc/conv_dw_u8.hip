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
#include "conv_dw_body.hpp"

namespace {

using namespace rlx_convdw;
typedef GeometryU8 Geometry;
inline bool geometry(int B, int H, int W, int C, int KH, int KW, int S, int Co, int T, Geometry *g) {
    return geometry_u8(B, H, W, C, KH, KW, S, Co, T, g);
}

template <int OWT, int SCT, int ROWD>
__global__ void __launch_bounds__(kThreads) conv_dw_u8_kernel(const DwU8 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_dw_u8_body<OWT, SCT, ROWD>(a, blockIdx.x, smem);
}
// one tower of 32 filters (the DQN update, B = 32): waves = kernel row x half of the output rows, two splits per image
__global__ void __launch_bounds__(kThreads) conv_dw_u8_half_kernel(const DwU8 a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_dw_u8_body_half(a, blockIdx.x, smem);
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
    if (g.nch == kNH)
        RLX_LAUNCH((conv_dw_u8_half_kernel), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    else if (g.OW == 20 && g.OH == 20 && S * C == 16 && g.rowf == 336)     // the Atari torso: 84 x 84 x 4 frames, 8 x 8 stride 4
        RLX_LAUNCH((conv_dw_u8_kernel<20, 16, 84>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    else
        RLX_LAUNCH((conv_dw_u8_kernel<0, 0, 0>), grid, kThreads, g.lds, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    rlx_splitk_job &j = *job_host;
    j.partials = a.part; j.colsum_partials = db ? a.cpart : nullptr;
    j.C = dw; j.colsum_out = db;
    j.ldc = Co; j.c_batch_stride = dw_tower_stride; j.colsum_batch_stride = db_tower_stride;
    j.M = g.K; j.N = g.nch; j.batch = 1; j.splits = g.splits; j.n_fold = Co;
    return RLX_OK;
}

}  // extern "C"
