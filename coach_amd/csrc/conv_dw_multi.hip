// The weight gradients of ALL convolution layers of an image torso as ONE launch (rlx_conv_dw_multi).
//
// Replaces, in the backward pass of the Atari torso (rl_coach/architectures/tensorflow_components/embedders/
// image_embedder.py:33-40; tf.gradients of three tf.layers.conv2d, architecture.py:187-220), the three launches
// rlx_conv_dw_u8 (conv1) + 2 x rlx_conv_dw_f32 (conv2, conv3): once the input-gradient chain has passed, the three products
// are independent of each other, and every workgroup of them is self-contained — its operands sit in ITS LDS, its partial
// goes to ITS slice of the deferred split-K workspace.  One grid of 256 + 256 + 192 workgroups (longest first) lets a CU
// start the next workgroup the moment one finishes: no launch ramp and no tail between the layers (14.7 + 12.9 + 11.6 us
// as three launches in the Clipped-PPO update).  Same device code (conv_dw_body.hpp), same sums, same partials.
#include "conv_dw_body.hpp"

namespace {

using namespace rlx_convdw;

struct MultiArgs {
    DwU8 u8;
    DwF32 f[2];
    int nb_u8, nb_f[2], kind[2];
    int u8_half;               // the uint8 item is one tower of 32 filters: conv_dw_u8_body_half
};

// PASSES > 1: the bodies take their operands in passes over the output rows (conv_dw_u8_body_passes,
// conv_dw_f32_body_two_pass) so that several workgroups share a CU — one's fills under the others' products (every
// workgroup of the launch is allocated the largest body's LDS: at 157 KB even the 59 KB conv3 workgroups ran one per CU).
//   2: conv1 in two passes (79 KB), conv2 in two (50 KB), conv3 in two (34 KB)      -> two workgroups per CU
//   4: conv1 in four passes (40 KB), conv2 in two (50 KB), conv3 in two (34 KB)     -> three workgroups per CU
template <int PASSES>
__global__ void __launch_bounds__(kThreads) conv_dw_multi_kernel(const MultiArgs m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    if (bid < m.nb_u8) {
        if (m.u8_half) conv_dw_u8_body_half(m.u8, bid, smem);
        else if constexpr (PASSES > 1) conv_dw_u8_body_passes<PASSES>(m.u8, bid, smem);
        else conv_dw_u8_body<20, 16, 84>(m.u8, bid, smem);
        return;
    }
    bid -= m.nb_u8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (bid < m.nb_f[i]) {
            if (m.kind[i] == 1) {
                if constexpr (PASSES > 1) conv_dw_f32_body_two_pass<64, 3, 1, 9, 7, 4, 7>(m.f[i], bid, smem);
                else conv_dw_f32_body<64, 3, 1, 9, 7>(m.f[i], bid, smem);
            } else {
                if constexpr (PASSES > 1) conv_dw_f32_body_two_pass<32, 4, 2, 20, 9, 5, 9>(m.f[i], bid, smem);
                else conv_dw_f32_body<32, 4, 2, 20, 9>(m.f[i], bid, smem);
            }
            return;
        }
        bid -= m.nb_f[i];
    }
}

int g_passes = 2;          // rlx_conv_dw_passes
int g_pairs_per_wg = 0;    // rlx_conv_dw_pairs_per_workgroup (0: by the number of workgroups)
// image pairs per workgroup of an fp32 item: two where (tower, pair) units are plentiful (the Clipped-PPO minibatch: 2 x 32 —
// same launch time, half the partial sums: 8.9 MB less written and read back per update, profiles/r06_ab_conv_dw_pairs.txt),
// one where that would leave CUs without work (the DQN update's 16 pairs: 165.8 against 156.7 ms per C3 step)
inline int pairs_per_wg(int B, int towers) {
    if (g_passes <= 1) return 1;
    if (g_pairs_per_wg > 0) return g_pairs_per_wg;
    return towers * ((B + 1) / 2) >= 64 ? 2 : 1;
}

}  // namespace

extern "C" {

int rlx_conv_dw_multi(const rlx_conv_dw_item *items, rlx_splitk_job *jobs, int n_items, void *stream) {
    RLX_REQUIRE(items && jobs && n_items >= 1 && n_items <= 3, "rlx_conv_dw_multi: 1 .. 3 items");
    // the one-launch form: at most one uint8 item of the Atari geometry and up to two fp32 items; anything else goes out item by item
    int n_u8 = 0, n_f = 0;
    bool one = n_items >= 2;
    GeometryU8 gu;
    GeometryF32 gf[2];
    const rlx_conv_dw_item *iu = nullptr, *fi[2] = {nullptr, nullptr};
    int ju = -1, jf[2] = {-1, -1};
    for (int i = 0; i < n_items && one; ++i) {
        const rlx_conv_dw_item &it = items[i];
        if (it.x_is_u8) {
            one = n_u8 == 0 && geometry_u8(it.B, it.H, it.W, it.C, it.KH, it.KW, it.S, it.filters, it.towers, &gu) && gu.OW == 20 &&
                  gu.OH == 20 && it.S * it.C == 16 && gu.rowf == 336;
            iu = &it; ju = i; ++n_u8;
        } else {
            one = n_f < 2 && geometry_f32(it.B, it.H, it.W, it.C, it.KH, it.KW, it.S, it.filters, it.towers, &gf[n_f < 2 ? n_f : 1],
                                          pairs_per_wg(it.B, it.towers));
            if (n_f < 2) { fi[n_f] = &it; jf[n_f] = i; }
            ++n_f;
        }
    }
    if (!one) {
        for (int i = 0; i < n_items; ++i) {
            const rlx_conv_dw_item &it = items[i];
            const int rc = it.x_is_u8
                ? rlx_conv_dw_u8(static_cast<const unsigned char *>(it.x), it.a_div, it.dz, it.dz_tower_stride, it.B, it.H, it.W, it.C,
                                 it.KH, it.KW, it.S, it.filters, it.towers, it.dw, it.dw_tower_stride, it.db, it.db_tower_stride,
                                 it.workspace, it.workspace_floats, &jobs[i], stream)
                : rlx_conv_dw_f32(static_cast<const float *>(it.x), it.x_tower_stride, it.dz, it.dz_tower_stride, it.B, it.H, it.W,
                                  it.C, it.KH, it.KW, it.S, it.filters, it.towers, it.dw, it.dw_tower_stride, it.db,
                                  it.db_tower_stride, it.workspace, it.workspace_floats, &jobs[i], stream);
            if (rc != RLX_OK) return rc;
        }
        return RLX_OK;
    }
    MultiArgs m;
    m.nb_u8 = m.nb_f[0] = m.nb_f[1] = 0;
    m.kind[0] = m.kind[1] = 0;
    m.u8_half = 0;
    size_t lds = 0;
    if (iu) {
        const rlx_conv_dw_item &it = *iu;
        RLX_REQUIRE(it.x && it.dz && it.dw && it.workspace && it.a_div != 0.f, "rlx_conv_dw_multi: null pointer (uint8 item)");
        const long long need = (long long)it.B * ((long long)gu.K * kN + kN);
        RLX_REQUIRE(it.workspace_floats >= need, "rlx_conv_dw_multi: workspace of %lld floats, need %lld", it.workspace_floats, need);
        RLX_REQUIRE((((uintptr_t)it.x) & 3) == 0 && ((((uintptr_t)it.dz) | ((uintptr_t)it.workspace)) & 15) == 0 &&
                        (it.dz_tower_stride & 3) == 0, "rlx_conv_dw_multi: misaligned operand (uint8 item)");
        DwU8 &a = m.u8;
        a.x = static_cast<const unsigned char *>(it.x); a.dz = it.dz; a.dz_ts = it.dz_tower_stride;
        a.part = it.workspace; a.cpart = it.workspace + (size_t)it.B * gu.K * kN;
        a.a_div = it.a_div;
        a.B = it.B; a.H = it.H; a.W = it.W; a.C = it.C; a.KH = it.KH; a.S = it.S; a.OH = gu.OH; a.OW = gu.OW; a.Co = it.filters;
        a.K = gu.K; a.P = gu.P; a.rowf = gu.rowf; a.NQ = gu.NQ; a.stamps = nullptr;
        m.nb_u8 = ((it.B + 7) / 8) * 8 * gu.NQ;
        m.u8_half = gu.nch == kNH;
        lds = gu.lds;
        rlx_splitk_job &j = jobs[ju];
        j.partials = a.part; j.colsum_partials = it.db ? a.cpart : nullptr;
        j.C = it.dw; j.colsum_out = it.db;
        j.ldc = it.filters; j.c_batch_stride = it.dw_tower_stride; j.colsum_batch_stride = it.db_tower_stride;
        j.M = gu.K; j.N = gu.nch; j.batch = 1; j.splits = gu.splits; j.n_fold = it.filters;
    }
    for (int k = 0; k < 2; ++k) {
        if (!fi[k]) continue;
        const rlx_conv_dw_item &it = *fi[k];
        const GeometryF32 &g = gf[k];
        RLX_REQUIRE(it.x && it.dz && it.dw && it.workspace, "rlx_conv_dw_multi: null pointer (fp32 item)");
        const long long need = (long long)it.towers * g.splits * ((long long)g.K * kN + kN);
        RLX_REQUIRE(it.workspace_floats >= need, "rlx_conv_dw_multi: workspace of %lld floats, need %lld", it.workspace_floats, need);
        RLX_REQUIRE(((((uintptr_t)it.x) | ((uintptr_t)it.dz) | ((uintptr_t)it.workspace)) & 15) == 0 && (it.x_tower_stride & 3) == 0 &&
                        (it.dz_tower_stride & 3) == 0, "rlx_conv_dw_multi: misaligned operand (fp32 item)");
        DwF32 &a = m.f[k];
        a.x = static_cast<const float *>(it.x); a.x_ts = it.x_tower_stride; a.dz = it.dz; a.dz_ts = it.dz_tower_stride;
        a.part = it.workspace; a.cpart = it.db ? it.workspace + (size_t)it.towers * g.splits * g.K * kN : nullptr;
        a.B = it.B; a.H = it.H; a.OH = g.OH; a.KH = it.KH; a.splits = g.splits; a.units = it.towers * g.splits;
        a.ppw = pairs_per_wg(it.B, it.towers);
        a.stamps = nullptr;
        m.nb_f[k] = ((a.units + 7) / 8) * 8 * it.KH;
        m.kind[k] = g.kind;
        lds = g.lds > lds ? g.lds : lds;
        rlx_splitk_job &j = jobs[jf[k]];
        j.partials = a.part; j.colsum_partials = a.cpart;
        j.C = it.dw; j.colsum_out = it.db;
        j.ldc = it.filters; j.c_batch_stride = it.dw_tower_stride; j.colsum_batch_stride = it.db_tower_stride;
        j.M = g.K; j.N = kN; j.batch = it.towers; j.splits = g.splits; j.n_fold = 0;
    }
    static bool configured = false;
    if (!configured) {
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_multi_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_multi_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RLX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw_multi_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    const unsigned grid = m.nb_u8 + m.nb_f[0] + m.nb_f[1];
    if (g_passes > 1) {
        // per-body LDS of the multi-pass forms
        size_t lds2 = iu ? (m.u8_half ? lds_u8_half() : (g_passes == 4 ? lds_u8_passes<4>() : lds_u8_passes<2>())) : 0;
        for (int k = 0; k < 2; ++k) {
            if (!fi[k]) continue;
            const size_t l = m.kind[k] == 2 ? lds_f32_two_pass<32, 20, 9, 5>() : lds_f32_two_pass<64, 9, 7, 4>();
            lds2 = l > lds2 ? l : lds2;
        }
        if (g_passes == 4) RLX_LAUNCH((conv_dw_multi_kernel<4>), grid, kThreads, lds2, rlx::as_stream(stream), m);
        else RLX_LAUNCH((conv_dw_multi_kernel<2>), grid, kThreads, lds2, rlx::as_stream(stream), m);
    } else {
        RLX_LAUNCH((conv_dw_multi_kernel<1>), grid, kThreads, lds, rlx::as_stream(stream), m);
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv_dw_pairs_per_workgroup(int pairs) {
    RLX_REQUIRE(pairs >= 0 && pairs <= 2, "rlx_conv_dw_pairs_per_workgroup: 0 (by the number of workgroups), 1 or 2");
    g_pairs_per_wg = pairs;
    return RLX_OK;
}

int rlx_conv_dw_passes(int passes) {
    RLX_REQUIRE(passes == 1 || passes == 2 || passes == 4, "rlx_conv_dw_passes: 1, 2 or 4");
    g_passes = passes;
    return RLX_OK;
}

}  // extern "C"
