// fp32 MFMA GEMM family for the policy / value / Q networks on gfx950.
//
// Replaces the TensorFlow ops behind rl_coach/architectures/tensorflow_components/layers.py:
//   Dense  -> tf.layers.dense  (:168-185)       y = act(x W + b),           W stored [in, out]
//   Conv2d -> tf.layers.conv2d (:108-121)       VALID padding, NHWC, kernel stored [KH,KW,Cin,Cout]
// and their gradients (tf.gradients in tensorflow_components/architecture.py:187-220).
//
// One kernel template covers every product the learner needs:
//     C[M,N] = epilogue( A[M,K] * B[K,N] )
//   forward      (NN): A = activations (row-major, or an implicit-im2col gather), B = W[K,N]
//   weight grad  (TN): A = X^T (X row-major or gathered), B = dY[M,N], reduction over the batch
//   input grad   (NT): A = dY, B = W^T (W row-major [K,N] read with swapped strides)
// Operand elements are addressed as  base[ offO(outer) + offR(red) ]  where each offset is either
// index*stride or a lookup in a small int32 table: a convolution's im2col matrix is separable,
//   addr(m,k) = rowbase[m] + koff[k],  rowbase = ((b*H + oy*s)*W + ox*s)*C,  koff = (ky*W + kx)*C + c
// so the conv never materialises its patches (the uint8 frame stack is converted on the fly:
// value = byte / a_div, ObservationEmbedder input_rescaling, embedders/embedder.py:107-108).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 products, fp32 accumulate, bitwise a k-ordered
// fmaf chain (cdna_hip_programming.md §3); peak 157.3 TFLOP/s.  A workgroup is 4 waves, each owning
// one 32x32 accumulator tile (64x64 or 128x32 per workgroup); K is staged through LDS in slabs of
// 32 as As[k][m], Bs[k][n] (row pitch +1 word -> the transposing stores are conflict-free, the
// MFMA operand reads are 32 consecutive words per half-wave).  Small M*N with long K (the FC layer,
// every weight gradient) is split along K over blockIdx.z into a workspace and reduced
// deterministically by splitk_reduce_kernel, which also applies bias / activation.
// blockIdx.z also carries a batch index (the two separate Clipped-PPO towers run as one launch).
#include "rlx_common.hpp"
#include "dense_small_body.hpp"
#include "losses_body.hpp"
#include "splitk_reduce_body.hpp"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int kThreads = 256;
constexpr int kTabChunk = 2048;     // longest K chunk of a table-addressed (implicit im2col) launch of the fast kernel
constexpr int kWinChunk = 1024;     // longest reduction of a windowed-gather launch (three LDS tables of this length)

struct OperandDev {
    const void *base;
    const int *tab_o;   // outer-index offset table (elements) or null
    const int *tab_r;   // reduction-index offset table or null
    long long stride_o, stride_r, batch_stride, batch_stride2;
    int vec_ok;         // 16-byte (4-byte for u8) vector loads are legal for this launch
};

struct GemmDev {
    int xcd_mode;               // workgroup id -> tile through xcd_tile_position (process-wide, rlx_gemm_tuning)
    OperandDev a, b;
    float *c;
    long long ldc, c_batch_stride;
    const float *bias;
    long long bias_batch_stride, bias_batch_stride2;
    int inner;                  // batch index b = bo * inner + bi: operand offset = bo * stride2 + bi * stride
    const float *aux;           // epilogue multiplies by act'(aux[m][n])
    long long aux_ld, aux_batch_stride;
    float *ws;                  // split-K partials [batch][split][M][N]
    float *colsum;              // optional: colsum[n] = sum_k B[k][n] (bias gradient), per batch
    float *ws_colsum;           // split-K partials of the column sums [batch][split][N]
    long long colsum_batch_stride;
    int M, N, K;
    int splits, kchunk;
    int act, deriv, accumulate;
    int vec_epi;                // fast kernel: 16-byte epilogue accesses are legal for this launch
    int fold;                   // > 0: column n belongs to tower n / fold (its B, C, bias, colsum live at
                                // tower * batch_stride + n % fold): towers that share A run as ONE GEMM
    float a_div;
    unsigned long long *stamps; // diagnostics (rlx_gemm_debug_stamps): [workgroup][4] wall-clock ticks, or null
    // windowed gather (A_WIN kernels: the input gradient of a convolution computed directly, rlx_conv_input_grad):
    // A(m, k) = a.base[a.tab_o[m] + a.tab_r[k]] where the window test passes, else 0
    const int *win_row_yx;      // [M]  (Y << 16) | X of row m; rows that do not exist carry Y = 0x7fff
    const int *win_k_jyx;       // [K]  (jy << 16) | jx of reduction index k
    const int *win_b_koff;      // [phases][K] offset of B's reduction index k (the weights' tap layout)
    const int *win_c_row;       // [M]  element offset of C's (and the derivative operand's) row m, or -1: no such row
    int win_oh, win_ow;         // the window: valid iff 0 <= Y - jy < win_oh and 0 <= X - jx < win_ow
    int win_rows_per_phase;     // rows [ph * rpp, (ph + 1) * rpp) use B table ph (rpp is a multiple of the tile height)
};

// two-level batch offset (paired online/target passes over multi-stream layers): b = bo * inner + bi
__device__ __forceinline__ long long batch_off(int batch, int inner, long long stride, long long stride2) {
    const int bo = batch / inner;
    return (long long)bo * stride2 + (long long)(batch - bo * inner) * stride;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == RLX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RLX_ACT_TANH) return tanhf(v);
    return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_deriv(float y, int kind) {
    if (kind == RLX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (kind == RLX_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// Loads 4 logically consecutive elements (along the vector dimension) of an operand tile.
//   VEC_RED = true : the 4 elements run along the reduction index r (o fixed)
//   VEC_RED = false: along the outer index o (r fixed)
template <bool VEC_RED, bool U8>
__device__ __forceinline__ float4 load4(const OperandDev &op, const unsigned char *base, int o,
                                        int r, int o_lim, int r_lim, float div) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o >= o_lim || r >= r_lim) return v;
    const long long off_o = op.tab_o ? (long long)op.tab_o[o] : (long long)o * op.stride_o;
    const long long off_r = op.tab_r ? (long long)op.tab_r[r] : (long long)r * op.stride_r;
    const int remaining = VEC_RED ? (r_lim - r) : (o_lim - o);
    if (op.vec_ok && remaining >= 4) {
        if (U8) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(base + off_o + off_r);
            v.x = (float)(w & 0xffu) / div;
            v.y = (float)((w >> 8) & 0xffu) / div;
            v.z = (float)((w >> 16) & 0xffu) / div;
            v.w = (float)(w >> 24) / div;
        } else {
            v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + off_o + off_r);
        }
        return v;
    }
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < remaining) {
            long long off;
            if (VEC_RED)
                off = off_o + (op.tab_r ? (long long)op.tab_r[r + i] : (long long)(r + i) * op.stride_r);
            else
                off = (op.tab_o ? (long long)op.tab_o[o + i] : (long long)(o + i) * op.stride_o) + off_r;
            t[i] = U8 ? (float)base[off] / div : reinterpret_cast<const float *>(base)[off];
        }
    }
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <int BM, int BN, bool A_VEC_RED, bool A_U8, bool B_VEC_RED>
__global__ void __launch_bounds__(kThreads) gemm_kernel(const GemmDev g) {
    constexpr int WN = BN / 32;                      // waves along N
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int NA = BM * BK / 4 / kThreads;       // float4 loads per thread for A
    constexpr int NB = BN * BK / 4 / kThreads;
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bz = blockIdx.z;
    const int batch = bz / g.splits, split = bz - batch * g.splits;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    const unsigned char *abase = static_cast<const unsigned char *>(g.a.base) +
                                 batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2) * (A_U8 ? 1 : 4);
    const unsigned char *bbase = static_cast<const unsigned char *>(g.b.base) +
                                 batch_off(batch, g.inner, g.b.batch_stride, g.b.batch_stride2) * 4;

    float4 ra[NA], rb[NB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int v = tid + p * kThreads;
            if (A_VEC_RED) {
                const int row = v / (BK / 4), kq = (v % (BK / 4)) * 4;
                ra[p] = load4<true, A_U8>(g.a, abase, m0 + row, k0 + kq, g.M, kend, g.a_div);
            } else {
                const int kr = v / (BM / 4), mq = (v % (BM / 4)) * 4;
                ra[p] = load4<false, A_U8>(g.a, abase, m0 + mq, k0 + kr, g.M, kend, g.a_div);
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int v = tid + p * kThreads;
            if (B_VEC_RED) {
                const int col = v / (BK / 4), kq = (v % (BK / 4)) * 4;
                rb[p] = load4<true, false>(g.b, bbase, n0 + col, k0 + kq, g.N, kend, 1.f);
            } else {
                const int kr = v / (BN / 4), nq = (v % (BN / 4)) * 4;
                rb[p] = load4<false, false>(g.b, bbase, n0 + nq, k0 + kr, g.N, kend, 1.f);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int v = tid + p * kThreads;
            if (A_VEC_RED) {
                const int row = v / (BK / 4), kq = (v % (BK / 4)) * 4;
                As[(kq + 0) * LDA_S + row] = ra[p].x;
                As[(kq + 1) * LDA_S + row] = ra[p].y;
                As[(kq + 2) * LDA_S + row] = ra[p].z;
                As[(kq + 3) * LDA_S + row] = ra[p].w;
            } else {
                const int kr = v / (BM / 4), mq = (v % (BM / 4)) * 4;
                float *d = &As[kr * LDA_S + mq];
                d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w;
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int v = tid + p * kThreads;
            if (B_VEC_RED) {
                const int col = v / (BK / 4), kq = (v % (BK / 4)) * 4;
                Bs[(kq + 0) * LDB_S + col] = rb[p].x;
                Bs[(kq + 1) * LDB_S + col] = rb[p].y;
                Bs[(kq + 2) * LDB_S + col] = rb[p].z;
                Bs[(kq + 3) * LDB_S + col] = rb[p].w;
            } else {
                const int kr = v / (BN / 4), nq = (v % (BN / 4)) * 4;
                float *d = &Bs[kr * LDB_S + nq];
                d[0] = rb[p].x; d[1] = rb[p].y; d[2] = rb[p].z; d[3] = rb[p].w;
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // bias gradient rides along: the first row of workgroups also sums the columns of every B slab
    // it stages (db = 1^T dz costs no extra HBM traffic and no extra launch)
    const bool do_colsum = g.colsum != nullptr && blockIdx.y == 0 && tid < BN;
    float csum = 0.f;

    if (kbeg < kend) {
        load_tiles(kbeg);
        store_tiles();
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = k0 + BK < kend;
            if (more) load_tiles(k0 + BK);          // global loads fly while the MFMAs run
            const float *ap = &As[hi * LDA_S + wm * 32 + l31];
            const float *bp = &Bs[hi * LDB_S + wn * 32 + l31];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a = ap[kk * LDA_S];
                const float b = bp[kk * LDB_S];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (do_colsum) {
                float sc = 0.f;
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) sc += Bs[kk * LDB_S + tid];
                csum += sc;
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    if (do_colsum && n0 + tid < g.N) {
        if (g.splits > 1)
            g.ws_colsum[((size_t)batch * g.splits + split) * g.N + n0 + tid] = csum;
        else
            g.colsum[(size_t)batch * g.colsum_batch_stride + n0 + tid] = csum;
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + l31;
    if (col >= g.N) return;
    if (g.splits > 1) {
        float *ws = g.ws + ((size_t)batch * g.splits + split) * (size_t)g.M * g.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < g.M) ws[(size_t)row * g.N + col] = acc[r];
        }
        return;
    }
    float *c = g.c + (size_t)batch * g.c_batch_stride;
    const float bias = g.bias ? g.bias[batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + col] : 0.f;
    const float *aux = g.aux ? g.aux + (size_t)batch * g.aux_batch_stride : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < g.M) {
            float v = apply_act(acc[r] + bias, g.act);
            if (aux) v *= act_deriv(aux[(size_t)row * g.aux_ld + col], g.deriv);
            float *dst = &c[(size_t)row * g.ldc + col];
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fast path.  Same tiling and arithmetic as gemm_kernel, restructured around what the profile of
// the generic kernel showed (profiles/r01_*): its bounds-checked loader compiles to one
// exec-masked region per load, so every global load waited for the previous one (~3.8 us per
// K-slab against 0.43 us of MFMA work).  Here, under launch-time guarantees checked by rlx_gemm
// (vector groups full and aligned: extents along the vector dimension are multiples of 4):
//   * all addresses are branch-free: out-of-range outer indices are clamped (their products land in
//     rows/columns the epilogue never stores), out-of-range reduction indices are clamped and the
//     loaded value multiplied by 0;
//   * offsets of the outer index (and im2col row/patch tables) are computed once before the K loop;
//     reduction-index table entries are prefetched one slab ahead of the data they address;
//   * the NA + NB 16-byte loads of slab k+2 are issued back to back before the MFMAs of slab k
//     (two register sets), stored to the other LDS buffer after them: one barrier per slab;
//   * uint8 frames are converted through a 256-entry LDS table of byte / a_div (exact fp32
//     quotients, no per-element division).
struct Raw4 { uint32_t x, y, z, w; };
__device__ __forceinline__ void load_raw(Raw4 &r, const unsigned char *base, long long off) {
    const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(base) + off);
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
}
struct Raw1 { uint32_t x; };
__device__ __forceinline__ void load_raw(Raw1 &r, const unsigned char *base, long long off) {
    r.x = *reinterpret_cast<const uint32_t *>(base + off);
}
__device__ __forceinline__ float4 raw_to_float4(const Raw4 &r, const float *) {
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z),
                       __uint_as_float(r.w));
}
__device__ __forceinline__ float4 raw_to_float4(const Raw1 &r, const float *lut) {
    const uint32_t w = r.x;
    return make_float4(lut[w & 0xffu], lut[(w >> 8) & 0xffu], lut[(w >> 16) & 0xffu], lut[w >> 24]);
}
template <bool U8> struct ARaw { typedef Raw4 type; };
template <> struct ARaw<true> { typedef Raw1 type; };

// TM x TN accumulator tiles of 32x32 per wave (wave tile 32*TM x 32*TN).  Every instantiation uses ONE tile per wave
// (at most 168 registers, so that three workgroups share a CU).
// The body of one workgroup; (bx, by, bz) of a grid (gdx, gdy, .) are passed in so that ONE launch can carry the
// workgroups of two independent problems (gemm_fast_pair_kernel).  smem / lut / tab_s: the caller's LDS.
template <int BM, int BN, int TM, int TN>
struct FastTile {
    static constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    static constexpr int A_BUF = BK * LDA_S, B_BUF = BK * LDB_S;
    static constexpr int kSmemFloats = 2 * A_BUF + 2 * B_BUF;
};

// Everything after the main loop of a fast-kernel workgroup: the in-workgroup K-split sum (KW > 1), the bias-gradient
// column sums, and the bias / activation / derivative / accumulate (or split-K partial) stores of the accumulator tiles.
// Shared by the register-staged main loop (gemm_fast_body) and the LDS-DMA ring main loop (gemm_dma_body): the C/D layout
// of the accumulators does not depend on how the operands reached the MFMAs.  `smem` must no longer be read by anyone
// (KW == 1 callers pass through the __syncthreads() of the staging patches below).
// The epilogue's own operands — the tile's bias and the four activation-derivative float4s of a lane — when the caller
// requested them BEFORE its main loop (gemm_dma_body): inside the update they are cold lines, and fetched here their
// round trip sits exposed between the last MFMA and the first store (conv1 / conv2 forward on a slow-class box:
// 8.4-8.9 us of epilogue against 2.0-3.7 us warm, profiles/r04_ab_gemm_pipeline.txt).
struct EpiPre {
    float4 bv, av[4];
};
template <int BM, int BN, int TM, int TN, int KW, bool A_WIN, bool PRE = false>
__device__ __forceinline__ void fast_epilogue(const GemmDev &g, f32x16 (&acc)[TM][TN], float *const smem, const int m0,
                                              const int n0, const int batch, const int split, const bool do_colsum,
                                              const float csum, unsigned long long *const stamp,
                                              const EpiPre *const pre = nullptr) {
    static_assert(!PRE || (TM == 1 && TN == 1 && !A_WIN), "prefetched epilogue operands: one tile per wave");
    constexpr int WN = BN / (32 * TN);
    constexpr int WMN = (BM / (32 * TM)) * WN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid / WMN, wmn = wid - wk * WMN;
    const int wm = wmn / WN, wn = wmn % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    if (stamp) stamp[2] = wall_clock64();
    const bool epi = wk == 0;                 // the waves that hold the complete tile
    if (KW > 1) {
        // partial tiles of wave groups 1 .. KW-1 -> LDS (a wave-private 32 x 33 patch each, C/D layout of the MFMA
        // undone), added by the wave group 0 that owns the same tile, in ascending wk
        __syncthreads();                      // every wave is done reading the operand buffers
        float *patch = smem + wid * (32 * 33);
        if (wk > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[0][0][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int q = 1; q < KW; ++q) {
                const float *src = smem + (q * WMN + wmn) * (32 * 33);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += src[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31];
            }
        }
    }
    if (do_colsum && n0 + tid < g.N) {
        const int n = n0 + tid;
        if (g.splits > 1)
            g.ws_colsum[((size_t)batch * g.splits + split) * g.N + n] = csum;
        else if (g.fold)
            g.colsum[(size_t)(n / g.fold) * g.colsum_batch_stride + n % g.fold] = csum;
        else
            g.colsum[(size_t)batch * g.colsum_batch_stride + n] = csum;
    }

    float *c = g.c + (size_t)batch * g.c_batch_stride;
    const float *aux = g.aux ? g.aux + (size_t)batch * g.aux_batch_stride : nullptr;
    float *ws = g.splits > 1 ? g.ws + ((size_t)batch * g.splits + split) * (size_t)g.M * g.N : nullptr;
    if (g.vec_epi) {
        // 16-byte epilogue: every accumulator tile goes through a wave-private 32x33 LDS patch so
        // that a lane owns 4 consecutive columns of a row (float4 loads of the derivative operand,
        // float4 stores; a wave instruction covers 8 rows x 128 B) instead of 16 scalar accesses.
        float *stage = smem + wid * (32 * 33);
        float *dst_base = ws ? ws : c;
        const long long ld = ws ? (long long)g.N : g.ldc;
        int crow[4] = {0, 0, 0, 0};      // A_WIN: where the lane's four rows of the tile live in C (and in aux), or -1
        if (A_WIN) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int erow = m0 + wm * 32 + ((it * 64 + lane) >> 3);
                crow[it] = epi && erow < g.M ? g.win_c_row[erow] : -1;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // the tile's bias (one float4 per lane: the column group does not depend on `it`) and its four
                // activation-derivative float4s are requested BEFORE the staging barriers: one round trip that the
                // staging covers, instead of one per `it` exposed between an LDS read and a store the compiler may not
                // move a load across
                const int ecol = n0 + wn * (32 * TN) + 32 * j + (lane & 7) * 4;
                const bool ecol_ok = epi && ecol < g.N;
                const int etw = (!ws && g.fold) ? ecol / g.fold : 0, ecl = (!ws && g.fold) ? ecol % g.fold : ecol;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 av4[4];
                if (PRE)
                    bv = pre->bv;
                else if (!ws && g.bias && ecol_ok)
                    bv = *reinterpret_cast<const float4 *>(
                        g.fold ? g.bias + (size_t)etw * g.bias_batch_stride + ecl :
                        g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + ecol);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int erow = m0 + wm * (32 * TM) + 32 * i + ((it * 64 + lane) >> 3);
                    av4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (PRE) {
                        av4[it] = pre->av[it];
                    } else if (A_WIN) {
                        if (aux && ecol_ok && crow[it] >= 0)
                            av4[it] = *reinterpret_cast<const float4 *>(aux + crow[it] + ecol);
                    } else if (!ws && aux && ecol_ok && erow < g.M)
                        av4[it] = *reinterpret_cast<const float4 *>(aux + (size_t)erow * g.aux_ld + ecol);
                }
                __syncthreads();
                if (epi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[i][j][r];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane;
                    const int rl = idx >> 3, c4 = (idx & 7) * 4;
                    const int row = m0 + wm * (32 * TM) + 32 * i + rl;
                    const int col = n0 + wn * (32 * TN) + 32 * j + c4;
                    if (!epi || row >= g.M || col >= g.N) continue;
                    if (A_WIN && crow[it] < 0) continue;
                    const float *sp = stage + rl * 33 + c4;
                    float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                    float *dst = A_WIN ? c + crow[it] + col : dst_base + (size_t)row * ld + col;
                    const int tw = etw, cl = ecl;
                    if (!ws && g.fold) dst = g.c + (size_t)tw * g.c_batch_stride + (size_t)row * g.ldc + cl;
                    if (!ws) {
                        if (g.bias) {
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                        }
                        v.x = apply_act(v.x, g.act); v.y = apply_act(v.y, g.act);
                        v.z = apply_act(v.z, g.act); v.w = apply_act(v.w, g.act);
                        if (aux) {
                            const float4 av = av4[it];
                            v.x *= act_deriv(av.x, g.deriv); v.y *= act_deriv(av.y, g.deriv);
                            v.z *= act_deriv(av.z, g.deriv); v.w *= act_deriv(av.w, g.deriv);
                        }
                        if (g.accumulate) {
                            const float4 ov = *reinterpret_cast<const float4 *>(dst);
                            v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
                        }
                    }
                    *reinterpret_cast<float4 *>(dst) = v;
                }
            }
        }
        if (stamp) { __syncthreads(); stamp[3] = wall_clock64(); }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (32 * TN) + 32 * j + l31;
        if (!epi || col >= g.N) continue;
        const float bias = (!ws && g.bias) ? g.bias[batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row >= g.M) continue;
                if (ws) {
                    ws[(size_t)row * g.N + col] = acc[i][j][r];
                } else {
                    float v = apply_act(acc[i][j][r] + bias, g.act);
                    if (aux) v *= act_deriv(aux[(size_t)row * g.aux_ld + col], g.deriv);
                    float *dst = &c[(size_t)row * g.ldc + col];
                    *dst = g.accumulate ? *dst + v : v;
                }
            }
        }
    }
    if (stamp) { __syncthreads(); stamp[3] = wall_clock64(); }
}

// KW > 1: the 4 waves also split every K slab — wave group wk multiplies rows [wk*BK/KW, (wk+1)*BK/KW) of the staged
// slab for the same output tile, and the KW partial tiles are summed through LDS in the fixed order wk = 0, 1, ...
// before the epilogue.  A 32 x 64 (KW = 2) or 32 x 32 (KW = 4) tile gives a mid-sized problem 2-4x the workgroups of
// the 64 x 64 tiling WITHOUT partial sums in memory and a second (reduce) launch: what splitting K over workgroups
// costs on this chip is the launch boundary behind several MB of freshly written partials, not the additions.
// A_WIN: windowed gather — see GemmDev::win_*; needs A_TAB, both operands vectorised along the reduction index and
// one accumulator tile per wave.  tab_s then holds three tables of kWinChunk entries: A offsets, (jy, jx), B offsets.
template <int BM, int BN, int TM, int TN, int KW, bool A_VEC_RED, bool A_U8, bool B_VEC_RED, bool A_TAB,
          bool A_WIN = false>
__device__ __forceinline__ void gemm_fast_body(const GemmDev &g, const int bx, const int by, const int bz,
                                               const int gdx, const int gdy, float *const smem, float *const lut,
                                               int *const tab_s) {
    static_assert(!A_WIN || (A_TAB && A_VEC_RED && B_VEC_RED && !A_U8 && TM == 1 && TN == 1),
                  "the windowed gather is a table-addressed NT product with one accumulator tile per wave");
    constexpr int WN = BN / (32 * TN);
    constexpr int WMN = (BM / (32 * TM)) * WN;
    static_assert(WMN * KW == 4, "a workgroup is 4 waves");
    static_assert(KW == 1 || (TM == 1 && TN == 1), "the in-workgroup K split keeps one accumulator tile per wave");
    static_assert(BK % (2 * KW) == 0, "every wave group owns whole MFMA k-steps of a slab");
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int NA = BM * BK / 4 / kThreads;
    constexpr int NB = BN * BK / 4 / kThreads;
    constexpr int A_BUF = BK * LDA_S, B_BUF = BK * LDB_S;
    float *const As = smem;
    float *const Bs = smem + 2 * A_BUF;
    static_assert(4 * 32 * 33 <= 2 * A_BUF + 2 * B_BUF, "epilogue staging fits in the operand buffers");

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid / WMN, wmn = wid - wk * WMN;
    const int wm = wmn / WN, wn = wmn % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int batch = bz / g.splits, split = bz - batch * g.splits;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    // phase stamps of every workgroup (entry, first slab staged, main loop done, exit) — tools/gemm_timeline.py
    unsigned long long *const stamp = g.stamps && tid == 0
        ? g.stamps + 4 * ((size_t)(bz * gdy + by) * gdx + bx) : nullptr;
    if (stamp) stamp[0] = wall_clock64();
    if (A_U8) {
        lut[tid] = (float)tid / g.a_div;        // kThreads == 256
        __syncthreads();
    }

    const unsigned char *abase = static_cast<const unsigned char *>(g.a.base) +
                                 batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2) * (A_U8 ? 1 : 4);
    const float *bbase = static_cast<const float *>(g.b.base) + batch_off(batch, g.inner, g.b.batch_stride, g.b.batch_stride2);

    // ---- loop-invariant parts of the operand addresses
    // A, vector along the reduction index: thread owns k-group kq of rows (row_p); else vector along
    // the outer index: thread owns outer group mq of reduction rows (kr_p).
    long long a_off_o[A_VEC_RED ? NA : 1];
    int a_kq = 0, a_kr[NA];
    int a_yx[A_WIN ? NA : 1];
    int *const tab2_s = A_WIN ? tab_s + kWinChunk : tab_s, *const tabb_s = A_WIN ? tab_s + 2 * kWinChunk : tab_s;
    const int *const btab = A_WIN ? g.win_b_koff + (size_t)(m0 / g.win_rows_per_phase) * g.K : nullptr;
    if (A_VEC_RED) {
        a_kq = (tid % (BK / 4)) * 4;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int row = min(m0 + (tid + p * kThreads) / (BK / 4), g.M - 1);
            a_off_o[p] = A_TAB ? (long long)g.a.tab_o[row] : (long long)row * g.a.stride_o;
            if (A_WIN) a_yx[p] = g.win_row_yx[row];
        }
    } else {
        const int mq = min(m0 + (tid % (BM / 4)) * 4, g.M - 4);
        a_off_o[0] = A_TAB ? (long long)g.a.tab_o[mq] : (long long)mq * g.a.stride_o;
#pragma unroll
        for (int p = 0; p < NA; ++p) a_kr[p] = (tid + p * kThreads) / (BM / 4);
    }
    long long b_off_o[B_VEC_RED ? NB : 1];
    int b_kq = 0, b_kr[NB];
    if (B_VEC_RED) {
        b_kq = (tid % (BK / 4)) * 4;
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int col = min(n0 + (tid + p * kThreads) / (BK / 4), g.N - 1);
            b_off_o[p] = (long long)col * g.b.stride_o;
        }
    } else {
        const int nq = min(n0 + (tid % (BN / 4)) * 4, g.N - 4);
        b_off_o[0] = g.fold ? (long long)(nq / g.fold) * g.b.batch_stride + (long long)(nq % g.fold) * g.b.stride_o
                            : (long long)nq * g.b.stride_o;
#pragma unroll
        for (int p = 0; p < NB; ++p) b_kr[p] = (tid + p * kThreads) / (BN / 4);
    }

    // reduction-index offsets of A for one slab.  Table lookups would be DEPENDENT global loads, and waiting for
    // one drains the whole in-order load queue (s_waitcnt vmcnt) — the operand loads in flight included — so the
    // main loop reads the chunk's offsets from LDS (staged once per workgroup; LDS has its own counter); only the
    // two slabs the prologue loads look their offsets up in global memory, all at once.
    // (windowed gather: whether an element exists depends on the row AND the reduction index, so a register set
    // carries one absolute offset and one mask per row it loads)
    constexpr int NR = A_WIN ? NA : (A_VEC_RED ? 1 : NA);
    auto a_red_offsets = [&](int k0, long long (&off)[NR], float (&msk)[NR], bool from_lds) {
        if (A_WIN) {
            const int r = k0 + a_kq;
            const int rc = min(r, kend - 4);
            const int ko = from_lds ? tab_s[rc - kbeg] : g.a.tab_r[rc];
            const int jyx = from_lds ? tab2_s[rc - kbeg] : g.win_k_jyx[rc];
            const int jy = jyx >> 16, jx = jyx & 0xffff;
#pragma unroll
            for (int p = 0; p < NA; ++p) {
                const int Y = a_yx[p] >> 16, X = a_yx[p] & 0xffff;
                const bool ok = r < kend && (unsigned)(Y - jy) < (unsigned)g.win_oh &&
                                (unsigned)(X - jx) < (unsigned)g.win_ow;
                msk[p] = ok ? 1.f : 0.f;
                off[p] = ok ? a_off_o[p] + (long long)ko : 0;
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int r = k0 + (A_VEC_RED ? a_kq : a_kr[p]);
            const int rc = min(r, kend - (A_VEC_RED ? 4 : 1));
            msk[p] = r < kend ? 1.f : 0.f;
            off[p] = A_TAB ? (long long)(from_lds ? tab_s[rc - kbeg] : g.a.tab_r[rc]) : (long long)rc * g.a.stride_r;
        }
    };
    typedef typename ARaw<A_U8>::type a_raw_t;
    auto load_a = [&](const long long (&off)[NR], a_raw_t (&ra)[NA]) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const long long o = A_WIN ? off[p] : (A_VEC_RED ? a_off_o[p] : a_off_o[0]) + off[A_VEC_RED ? 0 : p];
            load_raw(ra[p], abase, o);
        }
    };
    auto load_b = [&](int k0, float4 (&rb)[NB], float (&msk)[NB], bool from_lds) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int r = k0 + (B_VEC_RED ? b_kq : b_kr[p]);
            const int rc = min(r, kend - (B_VEC_RED ? 4 : 1));
            msk[p] = r < kend ? 1.f : 0.f;
            const long long ro = A_WIN ? (long long)(from_lds ? tabb_s[rc - kbeg] : btab[rc])
                                       : (long long)rc * g.b.stride_r;
            const long long o = (B_VEC_RED ? b_off_o[p] : b_off_o[0]) + ro;
            rb[p] = *reinterpret_cast<const float4 *>(bbase + o);
        }
    };
    auto store = [&](int buf, const a_raw_t (&ra)[NA], const float (&ma)[NR], const float4 (&rb)[NB],
                     const float (&mb)[NB]) {
        float *as = As + buf * A_BUF, *bs = Bs + buf * B_BUF;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            float4 v;
            const float mk = ma[(A_VEC_RED && !A_WIN) ? 0 : p];
            v = raw_to_float4(ra[p], lut);
            v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
            if (A_VEC_RED) {
                const int row = (tid + p * kThreads) / (BK / 4);
                as[(a_kq + 0) * LDA_S + row] = v.x;
                as[(a_kq + 1) * LDA_S + row] = v.y;
                as[(a_kq + 2) * LDA_S + row] = v.z;
                as[(a_kq + 3) * LDA_S + row] = v.w;
            } else {
                float *d = &as[a_kr[p] * LDA_S + (tid % (BM / 4)) * 4];
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            float4 v = rb[p];
            const float mk = mb[p];
            v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
            if (B_VEC_RED) {
                const int col = (tid + p * kThreads) / (BK / 4);
                bs[(b_kq + 0) * LDB_S + col] = v.x;
                bs[(b_kq + 1) * LDB_S + col] = v.y;
                bs[(b_kq + 2) * LDB_S + col] = v.z;
                bs[(b_kq + 3) * LDB_S + col] = v.w;
            } else {
                float *d = &bs[b_kr[p] * LDB_S + (tid % (BN / 4)) * 4];
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_colsum = g.colsum != nullptr && by == 0 && tid < BN;
    float csum = 0.f;

    if (kbeg < kend) {
        // Two register sets, each = the raw operand data + masks of one slab.  Loads are issued two slabs ahead of
        // their use and UNCONDITIONALLY inside the steady-state loop, so that the compiler knows how many are in
        // flight and waits with vmcnt(loads of one slab) — not vmcnt(0) — before a set is staged into LDS
        // (measured with rlx_gemm_debug_stamps: with conditional loads / table lookups in the loop every slab step
        // exposed a full load latency, 1.1-1.5 us per step against 0.43 us of MFMA issue).
        a_raw_t ra0[NA], ra1[NA];
        float4 rb0[NB], rb1[NB];
        long long ao0[NR], ao1[NR];
        float ma0[NR], ma1[NR], mb0[NB], mb1[NB];
        const int nslab = (kend - kbeg + BK - 1) / BK;
        // prologue: slabs 0 and 1 requested back to back; slab 0 -> LDS buffer 0
        a_red_offsets(kbeg, ao0, ma0, false);
        a_red_offsets(kbeg + BK, ao1, ma1, false);
        if (A_TAB) {
            for (int i = tid; i < kend - kbeg; i += kThreads) tab_s[i] = g.a.tab_r[kbeg + i];
        }
        if (A_WIN) {
            for (int i = tid; i < kend - kbeg; i += kThreads) {
                tab2_s[i] = g.win_k_jyx[kbeg + i];
                tabb_s[i] = btab[kbeg + i];
            }
        }
        load_a(ao0, ra0);
        load_b(kbeg, rb0, mb0, false);
        load_a(ao1, ra1);
        load_b(kbeg + BK, rb1, mb1, false);
        store(0, ra0, ma0, rb0, mb0);
        __syncthreads();
        if (stamp) stamp[1] = wall_clock64();
        // step s: LDS[s&1] = slab s; set N holds slab s+1 (in flight); set F is free and receives slab s+2
        auto mfma_slab = [&](int cur) {
            constexpr int KSPAN = BK / KW;                       // this wave group's rows of the slab
            const float *ap = As + cur * A_BUF + (wk * KSPAN + hi) * LDA_S + wm * (32 * TM) + l31;
            const float *bp = Bs + cur * B_BUF + (wk * KSPAN + hi) * LDB_S + wn * (32 * TN) + l31;
#pragma unroll
            for (int kk = 0; kk < KSPAN; kk += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = ap[kk * LDA_S + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bp[kk * LDB_S + 32 * j];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (do_colsum) {
                const float *bc = Bs + cur * B_BUF;
                float sc = 0.f;
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) sc += bc[kk * LDB_S + tid];
                csum += sc;
            }
        };
        // full step: request slab s+2, multiply slab s, stage slab s+1
        auto full = [&](int s, a_raw_t (&ra_n)[NA], float (&ma_n)[NR], float4 (&rb_n)[NB], float (&mb_n)[NB],
                        a_raw_t (&ra_f)[NA], long long (&ao_f)[NR], float (&ma_f)[NR], float4 (&rb_f)[NB],
                        float (&mb_f)[NB]) {
            a_red_offsets(kbeg + (s + 2) * BK, ao_f, ma_f, true);
            load_a(ao_f, ra_f);
            load_b(kbeg + (s + 2) * BK, rb_f, mb_f, true);
            __builtin_amdgcn_sched_barrier(0);        // keep the requests AHEAD of this slab's MFMAs
            mfma_slab(s & 1);
            store((s & 1) ^ 1, ra_n, ma_n, rb_n, mb_n);
            __syncthreads();
        };
        // tail step: nothing left to request
        auto tail = [&](int s, bool stage_next, a_raw_t (&ra_n)[NA], float (&ma_n)[NR], float4 (&rb_n)[NB],
                        float (&mb_n)[NB]) {
            mfma_slab(s & 1);
            if (stage_next) {
                store((s & 1) ^ 1, ra_n, ma_n, rb_n, mb_n);
                __syncthreads();
            }
        };
        int s = 0;
        for (; s + 3 < nslab; s += 2) {              // slabs s+2 and s+3 exist
            full(s, ra1, ma1, rb1, mb1, ra0, ao0, ma0, rb0, mb0);
            full(s + 1, ra0, ma0, rb0, mb0, ra1, ao1, ma1, rb1, mb1);
        }
        const int left = nslab - s;                  // 1, 2 or 3 slabs; s is even
        if (left == 3) {
            full(s, ra1, ma1, rb1, mb1, ra0, ao0, ma0, rb0, mb0);
            tail(s + 1, true, ra0, ma0, rb0, mb0);
            tail(s + 2, false, ra1, ma1, rb1, mb1);
        } else if (left == 2) {
            tail(s, true, ra1, ma1, rb1, mb1);
            tail(s + 1, false, ra0, ma0, rb0, mb0);
        } else {
            tail(s, false, ra1, ma1, rb1, mb1);
        }
    }

    fast_epilogue<BM, BN, TM, TN, KW, A_WIN>(g, acc, smem, m0, n0, batch, split, do_colsum, csum, stamp);
}

// Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, Workgroup dispatch) and every XCD has its own L2:
// with the plain id -> tile order, neighbouring tiles — which share an operand panel, the overlapping rows of an im2col
// gather, or the A slice of one split-K chunk — sit on eight different L2s and each fetches the shared bytes again
// (FC forward: A was fetched 8 x, profiles/r03_pmc_calibration.json).  The remap hands the tile order out in GROUPS:
// mode G > 0 gives XCD x the groups x, x + 8, x + 16, ... of G consecutive tiles (neighbours share an L2, and the eight
// XCDs still sweep the same neighbourhood of the operands at the same time, so what one fetched from HBM the next finds
// in the Infinity Cache); mode -1 gives each XCD one contiguous share of the whole order.  Bijective for any count (the
// last, incomplete round of groups keeps the plain order).  Placement is a speed matter only — every tile is computed
// by exactly one workgroup either way.  `local` counts from the first workgroup of the problem: in a pair launch the
// halves are remapped separately (all ids with equal local % 8 still share an XCD).
__device__ __forceinline__ int xcd_tile_position(int mode, int local, int count) {
    if (mode > 0) {                                    // G = mode, a power of two
        const int round = 8 * mode;
        if (local >= count / round * round) return local;
        const int x = local & 7, slot = local >> 3;
        return ((slot / mode) * 8 + x) * mode + (slot & (mode - 1));
    }
    if (mode < 0) {
        const int x = local & 7, q = count >> 3, r = count & 7;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (local >> 3);
    }
    return local;
}
// (bx, by, bz) of a 3-D grid after the remap; bx runs fastest, as in the hardware's own linear order
__device__ __forceinline__ void xcd_tile_block(int mode, int &bx, int &by, int &bz) {
    if (mode == 0) return;
    const int gx = gridDim.x, gy = gridDim.y;
    const int v = xcd_tile_position(mode, bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
    bx = v % gx;
    by = (v / gx) % gy;
    bz = v / (gx * gy);
}

template <int BM, int BN, int TM, int TN, int KW, bool A_VEC_RED, bool A_U8, bool B_VEC_RED, bool A_TAB>
__global__ void __launch_bounds__(kThreads, (TM * TN == 1 ? 3 : 1)) gemm_fast_kernel(const GemmDev g) {
    __shared__ __attribute__((aligned(16))) float smem[FastTile<BM, BN, TM, TN>::kSmemFloats];
    __shared__ float lut[A_U8 ? 256 : 1];
    __shared__ int tab_s[A_TAB ? kTabChunk : 1];     // reduction-index offsets of this workgroup's K chunk
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    xcd_tile_block(g.xcd_mode, bx, by, bz);
    gemm_fast_body<BM, BN, TM, TN, KW, A_VEC_RED, A_U8, B_VEC_RED, A_TAB>(g, bx, by, bz, gridDim.x, gridDim.y, smem, lut,
                                                                           tab_s);
}

// The input gradient of a convolution as ONE product (rlx_conv_input_grad): a windowed gather of dY against the
// weights' taps — see GemmDev::win_*.
template <int BM, int BN, int KW>
__global__ void __launch_bounds__(kThreads, 3) gemm_win_kernel(const GemmDev g) {
    __shared__ __attribute__((aligned(16))) float smem[FastTile<BM, BN, 1, 1>::kSmemFloats];
    __shared__ float lut[1];
    __shared__ int tab_s[3 * kWinChunk];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    xcd_tile_block(g.xcd_mode, bx, by, bz);
    gemm_fast_body<BM, BN, 1, 1, KW, true, false, true, true, true>(g, bx, by, bz, gridDim.x, gridDim.y, smem, lut, tab_s);
}

// Two INDEPENDENT problems in one launch — a layer's weight gradient (dW = X^T dY) and its input gradient
// (dX = dY W^T) read the same dY and feed different consumers: as two launches they run one after the other, each
// too small to fill the chip (profiles/r02_gemm_timeline_*.txt), as one they share it and one launch boundary
// disappears.  Workgroups [0, n0) belong to problem 0 (grid g0x x g0y x g0z), the rest to problem 1.
struct GemmPairDev {
    GemmDev g[2];
    int gx[2], gy[2];
    int n0;
    int t16[2];                 // thin pair: the half runs on 16 x 16 tiles
};
// DX_KW = 2 / 4: the input-gradient half runs on 32 x 64 / 32 x 32 tiles with the K slab split over the wave groups (a
// mid-sized dX such as the FC layer's 64 x 3136 x 512: no partial sums in memory, no reduce launch)
template <bool A_TAB0, int DX_KW>
__global__ void __launch_bounds__(kThreads, 3) gemm_fast_pair_kernel(const GemmPairDev p) {
    __shared__ __attribute__((aligned(16))) float smem[FastTile<64, 64, 1, 1>::kSmemFloats];
    __shared__ float lut[1];
    __shared__ int tab_s[A_TAB0 ? kTabChunk : 1];
    int flat = blockIdx.x;
    const int which = flat >= p.n0;
    // each half keeps its own share of every XCD (the halves differ in cost per tile: the chip stays balanced)
    if (which) flat -= p.n0;
    flat = xcd_tile_position(p.g[0].xcd_mode, flat, which ? (int)gridDim.x - p.n0 : p.n0);
    const int gx = p.gx[which], gy = p.gy[which];
    const int bx = flat % gx, by = (flat / gx) % gy, bz = flat / (gx * gy);
    if (!which)      // weight gradient: A = X^T (vector along the outer index; im2col tables for a convolution)
        gemm_fast_body<64, 64, 1, 1, 1, false, false, false, A_TAB0>(p.g[0], bx, by, bz, gx, gy, smem, lut, tab_s);
    else if (DX_KW == 2) // input gradient: A = dY (vector along k), B = W^T
        gemm_fast_body<32, 64, 1, 1, 2, true, false, true, false>(p.g[1], bx, by, bz, gx, gy, smem, lut, tab_s);
    else if (DX_KW == 4)
        gemm_fast_body<32, 32, 1, 1, 4, true, false, true, false>(p.g[1], bx, by, bz, gx, gy, smem, lut, tab_s);
    else
        gemm_fast_body<64, 64, 1, 1, 1, true, false, true, false>(p.g[1], bx, by, bz, gx, gy, smem, lut, tab_s);
}


// ---------------------------------------------------------------------------------------------
// LDS-DMA ring main loop (round 4).  What the in-update kernel timer showed for the register-staged loop above: on the C2
// shapes a workgroup spends 1.2-1.5 us per 32-deep K slab against 0.2-0.4 us of MFMA issue — one or two workgroups per
// CU, each wave alone on its SIMD, operands cold (written by the previous kernel, possibly on another XCD), and only two
// slabs of loads in flight because every slab in flight costs a register set.  Here the operand slabs go from global
// memory STRAIGHT into a ring of kDmaDepth LDS buffers (global_load_lds_dwordx4: no staging registers — 70 VGPRs instead
// of 140-168 — and no ds_write pass), kDmaDepth - 1 slabs are in flight at any time, and a slab step is
// wait(counted vmcnt) -> ONE barrier -> request slab s + D - 1 -> LDS operand reads + MFMAs of slab s.
//
// LDS-DMA writes lane-linearly (LDS byte = wave-uniform base + lane * 16), so the LDS image of an operand slab IS the
// order in which the lanes address global memory (cdna_hip_programming.md §5.4 rule 21):
//   * operand contiguous along its OUTER index (W[K][N] in a forward pass, X^T and dY of a weight gradient):
//     image [k = 32][outer], a lane moves 4 consecutive outer elements of one k; MFMA operands are 32 consecutive
//     words per half-wave (ds_read_b32, conflict-free) — as in the register-staged kernel, without its transposing
//     stores;
//   * operand contiguous along the REDUCTION index (activations of a forward pass, dY and W^T of an input gradient):
//     image [outer][8 groups of 4 k], the lane that fills (row r, slot j) fetches k-group j ^ ((r >> 1) & 7) — the
//     swizzle sits on the SOURCE address — and a lane reads its MFMA operands as one ds_read_b128 per four k-steps
//     (conflict-free for the 16-lane groups of ds_read_b128, MI355X_MICROARCH.md LDS table).
// Both reads agree on ONE order of the reduction inside a slab: MFMA step t of half-wave h multiplies
// k = 8 * (t / 4) + 4 * h + t % 4 (the plain kernel: 2 t + h).  fp32 sums are taken in that order — deterministic, the
// same for every launch of a shape, not bit-identical to the register-staged kernel.
// Out-of-range reduction indices read a 16-byte zero block (their slab rows hold zeros); out-of-range outer indices are
// clamped as in the register-staged kernel (their products land in rows / columns the epilogue never stores).
// The DMA is issued from inline asm, so hipcc neither counts it nor drains it (no vmcnt(0) in front of the operand
// reads, cdna_hip_programming.md §5.7): completion is counted by hand — every slab is L requests per lane, requests of
// non-existent slabs are still issued (zero block) so that "all but the newest (D - 2) * L" is the same immediate
// in every step.  Ordinary loads may not appear inside the loop: table lookups come from LDS.
// Ring depth and occupancy, measured on the C2 update (same-box A/Bs of builds, profiles/r04_ab_gemm_pipeline.txt): what
// pays is NOT a deeper ring — three slabs in flight (depth 4: 64 KB of LDS, two workgroups per CU) were 6 % SLOWER than the
// register-staged kernel, whose 42 KB let three workgroups share a CU — but the ring's small footprint: depth 2 (one slab
// in flight behind the one being multiplied; 32 KB + 8 KB of tables) at three workgroups per CU is 3.7 % faster than the
// register-staged kernel (266 vs 276 us per update), four per CU 2.5 %.  The waves of the neighbouring workgroups hide
// what a deeper ring would.
#ifndef RLX_DMA_DEPTH
#define RLX_DMA_DEPTH 2
#endif
#ifndef RLX_DMA_WGS
#define RLX_DMA_WGS 3          // workgroups per CU the DMA kernels are compiled for (LDS: depth x 16 KB + 8 KB of tables)
#endif
#ifndef RLX_DMA_TAB
#define RLX_DMA_TAB 2048       // im2col offsets of one K chunk staged in LDS by the DMA kernels (ints); longer chunks take the register-staged kernel
#endif
constexpr int kDmaDepth = RLX_DMA_DEPTH;
#ifndef RLX_DMA_DEPTH_U8
#define RLX_DMA_DEPTH_U8 4       // a uint8 A slab is 2 KB (fp32: 8 KB): four slabs deep still leaves three workgroups per CU
#endif
constexpr int kDmaDepthU8 = RLX_DMA_DEPTH_U8;
constexpr int kDmaTabChunk = RLX_DMA_TAB;
static_assert(kDmaDepth >= 2 && kDmaTabChunk <= kTabChunk, "ring depth / table size");
__device__ __attribute__((aligned(16))) float g_dma_zero[4] = {0.f, 0.f, 0.f, 0.f};

template <int BM, int BN, bool A_U8 = false>
struct DmaTile {
    static constexpr int kASlabFloats = A_U8 ? BM * BK / 4 : BM * BK;       // uint8 A: the slab image holds raw bytes
    static constexpr int kSlabFloats = kASlabFloats + BN * BK;
    static constexpr int kDepth = A_U8 ? kDmaDepthU8 : kDmaDepth;
    static constexpr int kRingFloats = kDepth * kSlabFloats;
    static constexpr int kSmemFloats = kRingFloats > 4 * 32 * 33 ? kRingFloats : 4 * 32 * 33;   // >= the epilogue's staging
};

// one 16-byte global -> LDS request per lane; lds_dst: wave-uniform LDS byte address of lane 0's 16 bytes
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the 4-byte request (uint8 operands: four consecutive bytes per lane; LDS byte = base + lane * 4)
__device__ __forceinline__ void dma4(const unsigned char *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// A_U8 (the first convolution: uint8 frames through im2col tables): the A slab image holds RAW BYTES — k-vector: [row][32
// bytes], a lane requests the 4 bytes of k-group slot ^ ((row >> 2) & 7) (32 banks of 4 bytes: 4 rows per bank row, the
// 8 rows of a bank class get 8 different slots: conflict-free ds_read_b32); outer-vector: [k][BM bytes], a lane requests 4
// consecutive outer elements of one k — and the conversion byte / a_div (exact quotients from a 256-entry LDS table, as in
// the register-staged kernel) happens when the MFMA operands are read.
// TM x TN: 32 x 32 accumulator tiles per wave (1 x 1 everywhere in the update's shapes; 2 x 2 = a 64 x 64 wave tile for the
// large products — half the LDS operand reads and half the slab bytes per MFMA; the order of every element's sum does not
// depend on it).
template <int BM, int BN, int KW, bool A_VEC_RED, bool B_VEC_RED, bool A_TAB, bool A_U8 = false, int TM = 1, int TN = 1>
__device__ __forceinline__ void gemm_dma_body(const GemmDev &g, const int bx, const int by, const int bz,
                                              const int gdx, const int gdy, float *const smem, int *const tab_s,
                                              const float *const lut = nullptr) {
    static_assert(!A_U8 || A_TAB, "uint8 operands come through im2col tables");
    static_assert(TM * TN == 1 || (KW == 1 && !A_U8), "multi-tile waves: fp32 operands, no in-workgroup K split");
    constexpr int D = DmaTile<BM, BN, A_U8>::kDepth;
    constexpr int WN = BN / (32 * TN);
    constexpr int WMN = (BM / (32 * TM)) * WN;
    static_assert(WMN * KW == 4, "a workgroup is 4 waves");
    static_assert(BK == 32, "slab images are 32 k deep");
    constexpr int A_SLAB = DmaTile<BM, BN, A_U8>::kASlabFloats, SLAB = DmaTile<BM, BN, A_U8>::kSlabFloats;
    constexpr int NA_I = BM / 32, NB_I = BN / 32;          // requests per lane per slab (16 bytes each; uint8 A: 4 bytes)
    constexpr int L = NA_I + NB_I;
    constexpr unsigned A_REQ = A_U8 ? 1024u : 4096u;       // bytes one workgroup-wide A request fills
    static_assert(L * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(4 * 32 * 33 <= DmaTile<BM, BN, A_U8>::kSmemFloats, "epilogue staging fits in the ring");

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid / WMN, wmn = wid - wk * WMN;
    const int wm = wmn / WN, wn = wmn % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int batch = bz / g.splits, split = bz - batch * g.splits;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    unsigned long long *const stamp = g.stamps && tid == 0
        ? g.stamps + 4 * ((size_t)(bz * gdy + by) * gdx + bx) : nullptr;
    if (stamp) stamp[0] = wall_clock64();

    // (uint8 A: `abase`, the offset tables and every A pointer below count BYTES; typed float * only to share the code)
    const float *const abase = A_U8
        ? reinterpret_cast<const float *>(static_cast<const unsigned char *>(g.a.base) +
                                          batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2))
        : static_cast<const float *>(g.a.base) + batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2);
    const float *const bbase = static_cast<const float *>(g.b.base) +
                               batch_off(batch, g.inner, g.b.batch_stride, g.b.batch_stride2);
    const float *const zero = g_dma_zero;
    auto a_at = [&](const float *base, long long elems) {   // base advanced by `elems` A elements
        return A_U8 ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(base) + elems) : base + elems;
    };
    // branch-free "valid ? p : zero block" (a C++ ?: here became an exec-masked branch around the table lookup)
    auto pick = [&](const float *p, bool valid) {
        const uintptr_t m = (uintptr_t)0 - (uintptr_t)valid;
        return reinterpret_cast<const float *>((reinterpret_cast<uintptr_t>(p) & m) |
                                               (reinterpret_cast<uintptr_t>(zero) & ~m));
    };

    // ---- per-lane invariants of the request addresses
    //   k-vector operand:     lane = (row tid / 8 [+ 32 p], slot tid % 8)          -> k-group slot ^ ((row >> 1) & 7)
    //   outer-vector operand: lane = (k-row tid / LPR [+ 1024 / extent * p], outer group tid % LPR), LPR = extent / 4
    long long a_off_o[A_VEC_RED ? NA_I : 1];
    int a_k = 0;                                            // k-vector: first k of the lane's group within a slab
    if (A_VEC_RED) {
        a_k = A_U8 ? ((tid & 7) ^ ((tid >> 5) & 7)) * 4     // uint8: slot ^ ((row >> 2) & 7), row = tid >> 3 (+ 32 p)
                   : ((tid & 7) ^ ((tid >> 4) & 7)) * 4;    // row = tid >> 3 (+ 32 p): (row >> 1) & 7 = (tid >> 4) & 7
#pragma unroll
        for (int p = 0; p < NA_I; ++p) {
            const int row = min(m0 + (tid >> 3) + 32 * p, g.M - 1);
            a_off_o[p] = A_TAB ? (long long)g.a.tab_o[row] : (long long)row * g.a.stride_o;
        }
    } else {
        const int mq = min(m0 + (tid % (BM / 4)) * 4, g.M - 4);
        a_off_o[0] = A_TAB ? (long long)g.a.tab_o[mq] : (long long)mq * g.a.stride_o;
    }
    long long b_off_o[B_VEC_RED ? NB_I : 1];
    int b_k = 0;
    if (B_VEC_RED) {
        b_k = ((tid & 7) ^ ((tid >> 4) & 7)) * 4;
#pragma unroll
        for (int p = 0; p < NB_I; ++p) {
            const int col = min(n0 + (tid >> 3) + 32 * p, g.N - 1);
            b_off_o[p] = (long long)col * g.b.stride_o;
        }
    } else {
        const int nq = min(n0 + (tid % (BN / 4)) * 4, g.N - 4);
        b_off_o[0] = g.fold ? (long long)(nq / g.fold) * g.b.batch_stride + (long long)(nq % g.fold) * g.b.stride_o
                            : (long long)nq * g.b.stride_o;
    }
    // the epilogue's operands, requested now (see EpiPre): same addresses and conditions as fast_epilogue's own loads
    EpiPre pre;
    pre.bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) pre.av[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool pre_ok = TM * TN == 1 && g.vec_epi && g.splits == 1;        // (split-K partials have no epilogue operands)
    if (pre_ok) {
        const int ecol = n0 + wn * (32 * TN) + (lane & 7) * 4;
        const bool ecol_ok = wk == 0 && ecol < g.N;
        if (g.bias && ecol_ok)
            pre.bv = *reinterpret_cast<const float4 *>(
                g.fold ? g.bias + (size_t)(ecol / g.fold) * g.bias_batch_stride + ecol % g.fold
                       : g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + ecol);
        if (g.aux && ecol_ok) {
            const float *aux = g.aux + (size_t)batch * g.aux_batch_stride;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int erow = m0 + wm * (32 * TM) + ((it * 64 + lane) >> 3);
                if (erow < g.M) pre.av[it] = *reinterpret_cast<const float4 *>(aux + (size_t)erow * g.aux_ld + ecol);
            }
        }
    }
    if (A_U8) const_cast<float *>(lut)[tid] = (float)tid / g.a_div;       // kThreads == 256; the barrier below publishes it
    if (A_TAB) {
        // the chunk's reduction-index offsets, requested together with the row offsets above (one round trip)
        for (int i = tid; i < kend - kbeg; i += kThreads) tab_s[i] = g.a.tab_r[kbeg + i];
        __syncthreads();
    }

    // ---- the requests of one slab.  Everything that does not change from slab to slab is computed once: the LDS
    // destination (wave-uniform, an SGPR plus compile-time offsets), the k of the lane inside a slab, the part of the
    // source address that belongs to the outer index; what changes is ONE pointer per request, advanced by a constant
    // (plain strides) or re-derived from one LDS table word (im2col).  `FULL` slabs lie inside the K chunk for every
    // lane: no validity test, no zero-block select — the tail (a partial slab, and the requests of non-existent slabs
    // that keep the vmcnt arithmetic uniform) takes the general form.  (The first version recomputed every address from
    // scratch: 9-11 VALU instructions per MFMA, profiles/r04_pmc_wave_states.json.)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + (unsigned)wid * 1024u);
    const unsigned lds0a = A_U8 ? __builtin_amdgcn_readfirstlane(           // a wave's 4-byte requests fill 256 bytes
        static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + (unsigned)wid * 256u) : lds0;
    const int len = kend - kbeg;
    int ak[NA_I], bk[NB_I];
    const float *ap[NA_I], *bp[NB_I];                       // A non-TAB / B: the source of the NEXT slab to request
    const float *arow[A_TAB ? NA_I : 1];                    // A im2col: base + outer-index offset (the k offset comes from tab_s)
#pragma unroll
    for (int p = 0; p < NA_I; ++p) {
        ak[p] = A_VEC_RED ? a_k : tid / (BM / 4) + (1024 / BM) * p;
        const float *base = a_at(abase, a_off_o[A_VEC_RED ? p : 0]);
        if (A_TAB) arow[p] = base;
        ap[p] = A_TAB ? base : base + (long long)(kbeg + ak[p]) * g.a.stride_r;
    }
#pragma unroll
    for (int p = 0; p < NB_I; ++p) {
        bk[p] = B_VEC_RED ? b_k : tid / (BN / 4) + (1024 / BN) * p;
        bp[p] = bbase + b_off_o[B_VEC_RED ? p : 0] + (long long)(kbeg + bk[p]) * g.b.stride_r;
    }
    const long long a_step = (long long)BK * g.a.stride_r, b_step = (long long)BK * g.b.stride_r;
    const int nfull = len / BK;                             // slabs [0, nfull) are inside the chunk for every lane
    auto issue = [&](const int s_idx, const int buf) {      // buf is a compile-time constant at every call site
#ifdef RLX_DBG_NO_DMA       // timing experiment only (wrong results): what the loop costs without the operand traffic
        return;
#endif
        const unsigned dst = lds0 + (unsigned)(buf * SLAB) * 4u;
        const bool full = s_idx < nfull;                    // wave-uniform
#pragma unroll
        for (int p = 0; p < NA_I; ++p) {
            const float *src;
            if (A_TAB) {
                const int ti = s_idx * BK + ak[p];
                const int tc = full ? ti : min(ti, len - (A_VEC_RED ? 4 : 1));
                src = a_at(arow[p], (long long)tab_s[tc]);
                if (!full) src = pick(src, ti < len);
            } else {
                src = full ? ap[p] : pick(ap[p], s_idx * BK + ak[p] < len);
                ap[p] += a_step;
            }
            if (A_U8) dma4(reinterpret_cast<const unsigned char *>(src), lds0a + (unsigned)(buf * SLAB) * 4u + A_REQ * p);
            else dma16(src, dst + A_REQ * p);
        }
#pragma unroll
        for (int p = 0; p < NB_I; ++p) {
            const float *src = full ? bp[p] : pick(bp[p], s_idx * BK + bk[p] < len);
            bp[p] += b_step;
            dma16(src, dst + (unsigned)A_SLAB * 4u + 4096u * p);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_colsum = g.colsum != nullptr && by == 0 && tid < BN;
    float csum = 0.f;

    // this wave group's MFMA steps of a slab: k-quads q = wk * QS .. + QS - 1, each 8 k (4 per half-wave).  The LDS
    // read addresses are lane constants (computed here) plus compile-time offsets (buffer, quad, step).
    constexpr int QS = 4 / KW;
    int a_rd[TM][QS], b_rd[TN][QS];                         // float index of the lane's operand(s) of quad qq inside a slab image
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) {
        const int q = wk * QS + qq;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int a_row = (wm * TM + i) * 32 + l31;
            a_rd[i][qq] = A_U8 ? (A_VEC_RED ? a_row * 8 + ((2 * q + hi) ^ ((a_row >> 2) & 7))      // dword of the lane's 4 bytes
                                            : (8 * q + 4 * hi) * BM + a_row)                         // byte of its first k
                               : A_VEC_RED ? a_row * 32 + (((2 * q + hi) ^ ((a_row >> 1) & 7)) << 2) : (8 * q + 4 * hi) * BM + a_row;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int b_col = (wn * TN + j) * 32 + l31;
            b_rd[j][qq] = B_VEC_RED ? b_col * 32 + (((2 * q + hi) ^ ((b_col >> 1) & 7)) << 2) : (8 * q + 4 * hi) * BN + b_col;
        }
    }
    // The slab step is software-pipelined through REGISTERS: while the MFMA chain of slab s runs, the operands of slab
    // s + 1 are already being read from LDS into the other register set, behind the wait / barrier / DMA issue that made
    // them available — those sit in the MIDDLE of the chain (an MFMA is issued and executes on its own; the wave goes on
    // to the independent instructions behind it).  With one wave per SIMD (the forward products: 196-400 workgroups on
    // 256 CUs) nothing else can fill the matrix pipe while a wave waits, reads and synchronises: the step took ~1 275
    // cycles for 512 cycles of MFMAs at KW = 2, whatever the ring depth (profiles/r04_timeline_ring_depth.txt).
    auto load_ops = [&](const int buf, float (&av)[TM][QS][4], float (&bv)[TN][QS][4]) {   // buf: compile-time constant at every call site
        const float *as = smem + buf * SLAB, *bs = as + A_SLAB;
#pragma unroll
        for (int qq = 0; qq < QS; ++qq) {
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                if (A_U8 && A_VEC_RED) {
                    const uint32_t w = reinterpret_cast<const uint32_t *>(as)[a_rd[ti][qq]];
                    av[ti][qq][0] = lut[w & 0xffu]; av[ti][qq][1] = lut[(w >> 8) & 0xffu];
                    av[ti][qq][2] = lut[(w >> 16) & 0xffu]; av[ti][qq][3] = lut[w >> 24];
                } else if (A_U8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[ti][qq][i] = lut[reinterpret_cast<const unsigned char *>(as)[a_rd[ti][qq] + i * BM]];
                } else if (A_VEC_RED) {
                    const float4 v = *reinterpret_cast<const float4 *>(as + a_rd[ti][qq]);
                    av[ti][qq][0] = v.x; av[ti][qq][1] = v.y; av[ti][qq][2] = v.z; av[ti][qq][3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[ti][qq][i] = as[a_rd[ti][qq] + i * BM];
                }
            }
#pragma unroll
            for (int tj = 0; tj < TN; ++tj) {
                if (B_VEC_RED) {
                    const float4 v = *reinterpret_cast<const float4 *>(bs + b_rd[tj][qq]);
                    bv[tj][qq][0] = v.x; bv[tj][qq][1] = v.y; bv[tj][qq][2] = v.z; bv[tj][qq][3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bv[tj][qq][i] = bs[b_rd[tj][qq] + i * BN];
                }
            }
        }
        if (do_colsum) {
            float sc = 0.f;
            if (B_VEC_RED) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = *reinterpret_cast<const float4 *>(bs + tid * 32 + (((j + (tid >> 1)) & 7) << 2));
                    sc += (v.x + v.y) + (v.z + v.w);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) sc += bs[kk * BN + tid];
            }
            csum += sc;
        }
    };
    // MFMA steps [t0, t1) of a slab's 4 * QS (the order of the sum inside a slab is unchanged: quad by quad, i = 0..3)
    auto mfma_steps = [&](const float (&av)[TM][QS][4], const float (&bv)[TN][QS][4], const int t0, const int t1) {
#ifdef RLX_DBG_NO_MFMA      // timing experiment only (wrong results): what the loop costs without the matrix work
#pragma unroll
        for (int t = t0; t < t1; ++t) acc[0][0][t & 15] += av[0][t >> 2][t & 3] * bv[0][t >> 2][t & 3];
#else
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t >> 2][t & 3], bv[j][t >> 2][t & 3], acc[i][j], 0, 0, 0);
#endif
    };

    if (kbeg < kend) {
        const int nslab = (len + BK - 1) / BK;
        float ra[2][TM][QS][4], rb[2][TN][QS][4];            // operand registers of the slab in the chain / the next one
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(d, d);
        // slab 0: landed once all but the (D - 2) * L newest requests of this lane are complete, for every wave
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (D - 2)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        issue(D - 1, D - 1);
        load_ops(0, ra[0], rb[0]);
        if (stamp) stamp[1] = wall_clock64();
        // U slab steps per trip so that buffer index (s % D) and register set (s % 2) are compile-time constants
        constexpr int U = D % 2 == 0 ? D : 2 * D;
        constexpr int T = 4 * QS, TH = T / 2;
        for (int s = 0; s < nslab; s += U) {
#pragma unroll
            for (int d = 0; d < U; ++d) {
                if (s + d < nslab) {                         // wave-uniform
                    mfma_steps(ra[d & 1], rb[d & 1], 0, TH);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + d + 1 < nslab) {
                        // slab s + d + 1 has landed (this lane's requests) ...
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L * (D - 2)) : "memory");
                        // ... this wave's LDS reads of slab s + d are complete (they fed the MFMAs above) ...
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        // ... for every wave: slab s + d + 1 may be read, the buffer of slab s + d refilled
                        asm volatile("s_barrier" ::: "memory");
                        issue(s + d + D, d % D);
                        load_ops((d + 1) % D, ra[(d + 1) & 1], rb[(d + 1) & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_steps(ra[d & 1], rb[d & 1], TH, T);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // requests of non-existent slabs still target the ring
    }
    __syncthreads();                                          // the ring becomes the epilogue's staging area
    if (stamp) stamp[2] = wall_clock64();
    if constexpr (TM * TN == 1) {
        if (pre_ok)
            fast_epilogue<BM, BN, 1, 1, KW, false, true>(g, acc, smem, m0, n0, batch, split, do_colsum, csum, stamp, &pre);
        else
            fast_epilogue<BM, BN, 1, 1, KW, false>(g, acc, smem, m0, n0, batch, split, do_colsum, csum, stamp);
    } else {
        fast_epilogue<BM, BN, TM, TN, KW, false>(g, acc, smem, m0, n0, batch, split, do_colsum, csum, stamp);
    }
}

template <int BM, int BN, int KW, bool A_VEC_RED, bool B_VEC_RED, bool A_TAB, bool A_U8 = false>
__global__ void __launch_bounds__(kThreads, RLX_DMA_WGS) gemm_dma_kernel(const GemmDev g) {
    __shared__ __attribute__((aligned(1024))) float smem[DmaTile<BM, BN, A_U8>::kSmemFloats];
    __shared__ int tab_s[A_TAB ? kDmaTabChunk : 1];
    __shared__ float lut[A_U8 ? 256 : 1];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    xcd_tile_block(g.xcd_mode, bx, by, bz);
    gemm_dma_body<BM, BN, KW, A_VEC_RED, B_VEC_RED, A_TAB, A_U8>(g, bx, by, bz, gridDim.x, gridDim.y, smem, tab_s, lut);
}

// Large products (thousands of tiles: the whole-dataset passes of Clipped PPO, acting on big vectors, the box calibration):
// several accumulator tiles per wave on the same ring — 128 x 128 per workgroup with 64 x 64 per wave (half the slab bytes and
// half the LDS operand reads per MFMA of the 64 x 64 workgroup tile; 64 KB of ring: two workgroups per CU), or, when N is 64,
// 128 x 64 with 64 x 32 per wave (three quarters; 48 KB: three per CU).
// Every output element is still ONE chain over the slabs in k order: bit-identical to the 64 x 64 tiling.
template <int BM, int BN, int TM, int TN, bool A_VEC_RED, bool B_VEC_RED, bool A_TAB>
__global__ void __launch_bounds__(kThreads, 2) gemm_dma_big_kernel(const GemmDev g) {
    __shared__ __attribute__((aligned(1024))) float smem[DmaTile<BM, BN>::kSmemFloats];
    __shared__ int tab_s[A_TAB ? kDmaTabChunk : 1];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    xcd_tile_block(g.xcd_mode, bx, by, bz);
    gemm_dma_body<BM, BN, 1, A_VEC_RED, B_VEC_RED, A_TAB, false, TM, TN>(g, bx, by, bz, gridDim.x, gridDim.y, smem, tab_s);
}

// a layer's weight gradient and input gradient as one launch (see gemm_fast_pair_kernel), both on the LDS-DMA ring
template <bool A_TAB0, int DX_KW>
__global__ void __launch_bounds__(kThreads, RLX_DMA_WGS) gemm_dma_pair_kernel(const GemmPairDev p) {
    __shared__ __attribute__((aligned(1024))) float smem[DmaTile<64, 64>::kSmemFloats];
    __shared__ int tab_s[A_TAB0 ? kDmaTabChunk : 1];
    int flat = blockIdx.x;
    const int which = flat >= p.n0;
    if (which) flat -= p.n0;
    flat = xcd_tile_position(p.g[0].xcd_mode, flat, which ? (int)gridDim.x - p.n0 : p.n0);
    const int gx = p.gx[which], gy = p.gy[which];
    const int bx = flat % gx, by = (flat / gx) % gy, bz = flat / (gx * gy);
    if (!which)
        gemm_dma_body<64, 64, 1, false, false, A_TAB0>(p.g[0], bx, by, bz, gx, gy, smem, tab_s);
    else if (DX_KW == 2)
        gemm_dma_body<32, 64, 2, true, true, false>(p.g[1], bx, by, bz, gx, gy, smem, tab_s);
    else if (DX_KW == 4)
        gemm_dma_body<32, 32, 4, true, true, false>(p.g[1], bx, by, bz, gx, gy, smem, tab_s);
    else
        gemm_dma_body<64, 64, 1, true, true, false>(p.g[1], bx, by, bz, gx, gy, smem, tab_s);
}


// Up to three WEIGHT-GRADIENT products of a backward pass as one launch (rlx_gemm_multi_defer): the convolution layers'
// dW = cols^T dz are independent of each other once the input-gradient chain has produced every layer's dz, each is a
// split-K product of a few hundred workgroups that starts cold, and issued back to back (inside their layers' dW + dX
// pairs) they ran one after the other.  Workgroups [start[i], start[i + 1]) belong to product i; kind 0: fp32 activations
// through im2col tables on the LDS-DMA ring, kind 1: uint8 frames through the register-staged loop.  Each product keeps
// the tiling and the K split rlx_gemm would give it alone: the sums are bit-identical to the per-layer launches.
struct GemmMultiDev {
    GemmDev g[3];
    int gx[3], gy[3];
    int start[4];
    int kind[3];
    int n;
};
__global__ void __launch_bounds__(kThreads, 3) gemm_multi_dw_kernel(const GemmMultiDev p) {
    constexpr int kFloats = DmaTile<64, 64>::kSmemFloats > FastTile<64, 64, 1, 1>::kSmemFloats
                                ? DmaTile<64, 64>::kSmemFloats : FastTile<64, 64, 1, 1>::kSmemFloats;
    __shared__ __attribute__((aligned(1024))) float smem[kFloats];
    __shared__ int tab_s[kTabChunk];
    __shared__ float lut[256];
    int flat = blockIdx.x;
    const int which = (p.n > 2 && flat >= p.start[2]) ? 2 : (flat >= p.start[1] ? 1 : 0);
    flat = xcd_tile_position(p.g[0].xcd_mode, flat - p.start[which], p.start[which + 1] - p.start[which]);
    const int gx = p.gx[which], gy = p.gy[which];
    const int bx = flat % gx, by = (flat / gx) % gy, bz = flat / (gx * gy);
    if (p.kind[which] == 1)
        gemm_fast_body<64, 64, 1, 1, 1, false, true, false, true>(p.g[which], bx, by, bz, gx, gy, smem, lut, tab_s);
    else
        gemm_dma_body<64, 64, 1, false, false, true>(p.g[which], bx, by, bz, gx, gy, smem, tab_s);
}

// ---------------------------------------------------------------------------------------------
// Thin GEMM: small M*N with a short reduction (the MLP layers of DQN / TD3 / SAC at B = 32..256, the
// FC input gradient of the conv nets).  The tiled kernels above cover such a problem with a handful of
// 64x64 workgroups and recover parallelism by splitting K over workgroups — which costs a second
// (reduce) launch, and a launch is ~5 us here.  This kernel splits K over the 4 WAVES of a workgroup
// instead: a workgroup owns one 32x32 output tile, every 128-wide K slab is staged once in LDS, wave w
// multiplies sub-slab [32w, 32w+32), and the four partial tiles are summed through LDS in a fixed
// order (reproducible) before the bias / activation / derivative epilogue.  One launch, 4x the
// workgroups of the 64x64 tiling.  Scalar (4-byte) operand loads with clamped indices: no alignment
// or divisibility requirements.
constexpr int kThinKS = 128, kThinLD = 33;
template <bool A_CONTIG_K, bool B_CONTIG_N>
__device__ __forceinline__ void gemm_thin_body(const GemmDev &g, const int bx, const int by, const int bz,
                                               float *const As, float *const Bs) {
    constexpr int KS = kThinKS, LD = kThinLD, NE = 32 * KS / kThreads;      // 16 elements per thread per operand
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int batch = bz;
    const int m0 = by * 32, n0 = bx * 32;
    const float *abase = static_cast<const float *>(g.a.base) + batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2);
    const float *bbase = static_cast<const float *>(g.b.base) + batch_off(batch, g.inner, g.b.batch_stride, g.b.batch_stride2);
    // epilogue operands (bias, activation-derivative input) requested FIRST: their round trip hides behind the
    // reduction instead of sitting, exposed, between the last MFMA and the stores
    const int rl = tid >> 3, c4 = (tid & 7) * 4;
    const int row = m0 + rl;
    float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_aux[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float *bias = g.bias ? g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) : nullptr;
        const float *aux = g.aux && row < g.M ? g.aux + (size_t)batch * g.aux_batch_stride + (size_t)row * g.aux_ld : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + c4 + j;
            if (col < g.N) {
                if (bias) e_bias[j] = bias[col];
                if (aux) e_aux[j] = aux[col];
            }
        }
    }

    // element e = tid + p*256 of a 32 x 128 operand tile: (outer, k) with the contiguous index fastest
    int a_o[NE], a_k[NE], b_o[NE], b_k[NE];
    long long a_oo[NE], b_oo[NE];
#pragma unroll
    for (int p = 0; p < NE; ++p) {
        const int e = tid + p * kThreads;
        a_k[p] = A_CONTIG_K ? e % KS : e / 32;
        a_o[p] = A_CONTIG_K ? e / KS : e % 32;
        b_k[p] = B_CONTIG_N ? e / 32 : e % KS;
        b_o[p] = B_CONTIG_N ? e % 32 : e / KS;
        a_oo[p] = (long long)min(m0 + a_o[p], g.M - 1) * g.a.stride_o;
        b_oo[p] = (long long)min(n0 + b_o[p], g.N - 1) * g.b.stride_o;
    }
    float ra[NE], rb[NE];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NE; ++p) {
            const int ka = k0 + a_k[p], kb = k0 + b_k[p];
            const float va = abase[a_oo[p] + (long long)min(ka, g.K - 1) * g.a.stride_r];
            const float vb = bbase[b_oo[p] + (long long)min(kb, g.K - 1) * g.b.stride_r];
            ra[p] = ka < g.K ? va : 0.f;
            rb[p] = kb < g.K ? vb : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int p = 0; p < NE; ++p) {
            As[a_k[p] * LD + a_o[p]] = ra[p];
            Bs[b_k[p] * LD + b_o[p]] = rb[p];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool do_colsum = g.colsum != nullptr && by == 0 && tid < 32;
    float csum = 0.f;
    load(0);
    store();
    __syncthreads();
    for (int k0 = 0; k0 < g.K; k0 += KS) {
        const bool more = k0 + KS < g.K;
        if (more) load(k0 + KS);
        const float *ap = As + (w * 32 + hi) * LD + l31;
        const float *bp = Bs + (w * 32 + hi) * LD + l31;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * LD], bp[kk * LD], acc, 0, 0, 0);
        if (do_colsum) {
            float sc = 0.f;
            for (int kk = 0; kk < KS; ++kk) sc += Bs[kk * LD + tid];
            csum += sc;
        }
        __syncthreads();
        if (more) {
            store();
            __syncthreads();
        }
    }
    if (do_colsum && n0 + tid < g.N) g.colsum[(size_t)batch * g.colsum_batch_stride + n0 + tid] = csum;

    // ---- the four K-quarter partials -> LDS (As is exactly 4 x 32 x 33 floats), fixed-order sum
    float *stage = As + w * (32 * LD);
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hi) * LD + l31] = acc[r];
    __syncthreads();
    if (row >= g.M) return;
    float *c = g.c + (size_t)batch * g.c_batch_stride + (size_t)row * g.ldc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + c4 + j;
        if (col >= g.N) continue;
        const int o = rl * LD + c4 + j;
        float v = ((As[o] + As[32 * LD + o]) + As[2 * 32 * LD + o]) + As[3 * 32 * LD + o];
        v = apply_act(v + e_bias[j], g.act);
        if (g.aux) v *= act_deriv(e_aux[j], g.deriv);
        c[col] = g.accumulate ? c[col] + v : v;
    }
}

// ---- the same kernel on 16 x 16 output tiles (v_mfma_f32_16x16x4_f32): 4x the workgroups.  A thin launch is not
// bound by work but by what ONE CU does in series for its tile (profiles/r02_ab_thin_prefetch_all.txt: operand loads
// through one L2 port, LDS staging, 50 dependent MFMAs, and nothing overlaps with one wave per SIMD) while most of the
// chip idles — 100 x 300 is 40 tiles of 32 x 32 on 256 CUs.  Quartering the tile halves the operand bytes and the staging
// per workgroup and quarters its MFMA passes.  The reduction index keeps its assignment (slabs of 128, wave w owns
// [32w, 32w + 32), ascending k inside a wave, the four wave partials summed in the same fixed order).
constexpr int kThin16LD = 17;
template <bool A_CONTIG_K, bool B_CONTIG_N>
__device__ __forceinline__ void gemm_thin16_body(const GemmDev &g, const int bx, const int by, const int bz,
                                                 float *const As, float *const Bs) {
    constexpr int KS = kThinKS, LD = kThin16LD, NE = 16 * KS / kThreads;    // 8 elements per thread per operand
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int batch = bz;
    const int m0 = by * 16, n0 = bx * 16;
    const float *abase = static_cast<const float *>(g.a.base) + batch_off(batch, g.inner, g.a.batch_stride, g.a.batch_stride2);
    const float *bbase = static_cast<const float *>(g.b.base) + batch_off(batch, g.inner, g.b.batch_stride, g.b.batch_stride2);
    // epilogue operands first (one output element per thread)
    const int rl = tid >> 4, cl = tid & 15;
    const int row = m0 + rl, col = n0 + cl;
    const bool live = row < g.M && col < g.N;
    float e_bias = 0.f, e_aux = 0.f;
    if (live) {
        if (g.bias) e_bias = g.bias[batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + col];
        if (g.aux) e_aux = g.aux[(size_t)batch * g.aux_batch_stride + (size_t)row * g.aux_ld + col];
    }
    int a_o[NE], a_k[NE], b_o[NE], b_k[NE];
    long long a_oo[NE], b_oo[NE];
#pragma unroll
    for (int p = 0; p < NE; ++p) {
        const int e = tid + p * kThreads;
        a_k[p] = A_CONTIG_K ? e % KS : e / 16;
        a_o[p] = A_CONTIG_K ? e / KS : e % 16;
        b_k[p] = B_CONTIG_N ? e / 16 : e % KS;
        b_o[p] = B_CONTIG_N ? e % 16 : e / KS;
        a_oo[p] = (long long)min(m0 + a_o[p], g.M - 1) * g.a.stride_o;
        b_oo[p] = (long long)min(n0 + b_o[p], g.N - 1) * g.b.stride_o;
    }
    float ra[NE], rb[NE];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NE; ++p) {
            const int ka = k0 + a_k[p], kb = k0 + b_k[p];
            const float va = abase[a_oo[p] + (long long)min(ka, g.K - 1) * g.a.stride_r];
            const float vb = bbase[b_oo[p] + (long long)min(kb, g.K - 1) * g.b.stride_r];
            ra[p] = ka < g.K ? va : 0.f;
            rb[p] = kb < g.K ? vb : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int p = 0; p < NE; ++p) {
            As[a_k[p] * LD + a_o[p]] = ra[p];
            Bs[b_k[p] * LD + b_o[p]] = rb[p];
        }
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool do_colsum = g.colsum != nullptr && by == 0 && tid < 16;
    float csum = 0.f;
    load(0);
    store();
    __syncthreads();
    for (int k0 = 0; k0 < g.K; k0 += KS) {
        const bool more = k0 + KS < g.K;
        if (more) load(k0 + KS);
        const float *ap = As + (w * 32 + kq) * LD + l15;
        const float *bp = Bs + (w * 32 + kq) * LD + l15;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * LD], bp[kk * LD], acc, 0, 0, 0);
        if (do_colsum) {
            float sc = 0.f;
            for (int kk = 0; kk < KS; ++kk) sc += Bs[kk * LD + tid];
            csum += sc;
        }
        __syncthreads();
        if (more) {
            store();
            __syncthreads();
        }
    }
    if (do_colsum && n0 + tid < g.N) g.colsum[(size_t)batch * g.colsum_batch_stride + n0 + tid] = csum;
    // the four K-quarter partials -> LDS (4 x 16 x 17 floats of As), fixed-order sum.  C/D layout of the 16x16 MFMA:
    // col = lane & 15, row = 4 * (lane >> 4) + r
    float *stage = As + w * (16 * LD);
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[(4 * kq + r) * LD + l15] = acc[r];
    __syncthreads();
    if (!live) return;
    const int o = rl * LD + cl;
    float v = ((As[o] + As[16 * LD + o]) + As[2 * 16 * LD + o]) + As[3 * 16 * LD + o];
    v = apply_act(v + e_bias, g.act);
    if (g.aux) v *= act_deriv(e_aux, g.deriv);
    float *c = g.c + (size_t)batch * g.c_batch_stride + (size_t)row * g.ldc + col;
    *c = g.accumulate ? *c + v : v;
}

template <bool A_CONTIG_K, bool B_CONTIG_N>
__global__ void __launch_bounds__(kThreads) gemm_thin16_kernel(const GemmDev g) {
    __shared__ float As[kThinKS * kThin16LD];
    __shared__ float Bs[kThinKS * kThin16LD];
    gemm_thin16_body<A_CONTIG_K, B_CONTIG_N>(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

template <bool A_CONTIG_K, bool B_CONTIG_N>
__global__ void __launch_bounds__(kThreads) gemm_thin_kernel(const GemmDev g) {
    __shared__ float As[kThinKS * kThinLD];
    __shared__ float Bs[kThinKS * kThinLD];
    gemm_thin_body<A_CONTIG_K, B_CONTIG_N>(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// dW = X^T dY (A strided along k, B contiguous along n) and dX = dY W^T (A contiguous along k, B strided along n)
// of an MLP layer as one grid (see gemm_fast_pair_kernel)
__global__ void __launch_bounds__(kThreads) gemm_thin_pair_kernel(const GemmPairDev p) {
    __shared__ float As[kThinKS * kThinLD];
    __shared__ float Bs[kThinKS * kThinLD];
    int flat = blockIdx.x;
    const int which = flat >= p.n0;
    if (which) flat -= p.n0;
    const int gx = p.gx[which], gy = p.gy[which];
    const int bx = flat % gx, by = (flat / gx) % gy, bz = flat / (gx * gy);
    if (!which) {
        if (p.t16[0]) gemm_thin16_body<false, true>(p.g[0], bx, by, bz, As, Bs);
        else gemm_thin_body<false, true>(p.g[0], bx, by, bz, As, Bs);
    } else {
        if (p.t16[1]) gemm_thin16_body<true, false>(p.g[1], bx, by, bz, As, Bs);
        else gemm_thin_body<true, false>(p.g[1], bx, by, bz, As, Bs);
    }
}

__global__ void splitk_reduce_kernel(const GemmDev g) {
    const long long mn = (long long)g.M * g.N;
    const int batch = blockIdx.y;
    const float *ws = g.ws + (size_t)batch * g.splits * mn;
    float *c = g.c + (size_t)batch * g.c_batch_stride;
    const float *bias = g.bias ? g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) : nullptr;
    const float *aux = g.aux ? g.aux + (size_t)batch * g.aux_batch_stride : nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < mn;
         i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < g.splits; ++k) s += ws[(size_t)k * mn + i];   // fixed order
        const int row = (int)(i / g.N), col = (int)(i - (long long)row * g.N);
        float v = apply_act(s + (bias ? bias[col] : 0.f), g.act);
        if (aux) v *= act_deriv(aux[(size_t)row * g.aux_ld + col], g.deriv);
        float *dst = &c[(size_t)row * g.ldc + col];
        *dst = g.accumulate ? *dst + v : v;
    }
    if (g.colsum) {
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < g.N; n += gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int k = 0; k < g.splits; ++k) s += g.ws_colsum[((size_t)batch * g.splits + k) * g.N + n];
            g.colsum[(size_t)batch * g.colsum_batch_stride + n] = s;
        }
    }
}

// Vectorised split-K reduction (N % 4 == 0, M*N < 2^31): a workgroup = 64 float4 output groups x SG
// split groups; split group q sums partials q, q+SG, q+2SG, ... in increasing order, the group sums
// are combined in the fixed order ((s0 + s1) + s2) + ... — the result is reproducible.  SG = 4 for
// few splits, 16 for the long-K weight gradients (29-62 splits of a small M*N: without the wider
// split parallelism 64 workgroups chased 16 dependent loads each, 18 us for 4 MB).
// 32-bit index arithmetic (the generic kernel's 64-bit divisions cost more than its memory traffic).
template <int SG>
__global__ void __launch_bounds__(64 * SG) splitk_reduce4_kernel(const GemmDev g) {
    __shared__ float4 part[SG][64];
    const int mn4 = (g.M * g.N) >> 2;
    const int batch = blockIdx.y;
    const int ox = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int gid = blockIdx.x * 64 + ox;
    const size_t mn = (size_t)g.M * g.N;
    const float4 *ws = reinterpret_cast<const float4 *>(g.ws + (size_t)batch * g.splits * mn);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // the epilogue operands of the finishing threads (bias, activation-derivative input, the accumulate target) are
    // requested before the partials: one more load in flight instead of a round trip after the sum
    float eb[4] = {0.f, 0.f, 0.f, 0.f}, ea[4] = {0.f, 0.f, 0.f, 0.f}, ec[4] = {0.f, 0.f, 0.f, 0.f};
    float *c = nullptr;
    if (q == 0 && gid < mn4) {
        const int i = gid << 2;
        const int row = i / g.N, col = i - row * g.N;
        const int tw = g.fold ? col / g.fold : batch, cl = g.fold ? col % g.fold : col;
        c = g.c + (size_t)tw * g.c_batch_stride + (size_t)row * g.ldc + cl;
        const float *bias = !g.bias ? nullptr
                            : g.fold ? g.bias + (size_t)tw * g.bias_batch_stride + cl
                                     : g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + col;
        const float *aux = g.aux ? g.aux + (size_t)batch * g.aux_batch_stride + (size_t)row * g.aux_ld + col
                                 : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bias) eb[j] = bias[j];
            if (aux) ea[j] = aux[j];
            if (g.accumulate) ec[j] = c[j];
        }
    }
    if (gid < mn4) {
        int k = q;
        for (; k + 3 * SG < g.splits; k += 4 * SG) {    // 4 independent loads in flight per thread
            const float4 a = ws[(size_t)k * mn4 + gid];
            const float4 b = ws[(size_t)(k + SG) * mn4 + gid];
            const float4 c = ws[(size_t)(k + 2 * SG) * mn4 + gid];
            const float4 d = ws[(size_t)(k + 3 * SG) * mn4 + gid];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        }
        for (; k < g.splits; k += SG) {
            const float4 a = ws[(size_t)k * mn4 + gid];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    part[q][ox] = s;
    __syncthreads();
    if (q == 0 && gid < mn4) {
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int t = 1; t < SG; ++t) {
            const float4 p = part[t][ox];
            v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = apply_act(v[j] + eb[j], g.act);
            if (g.aux) t *= act_deriv(ea[j], g.deriv);
            v[j] = g.accumulate ? ec[j] + t : t;
        }
        if ((g.ldc & 3) == 0 && (((uintptr_t)c) & 15) == 0) {
            *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
        }
    }
    if (g.colsum && blockIdx.x == 0) {
        // bias-gradient partials [split][N]: same split-parallel scheme, 64 columns at a time
        float *cpart = reinterpret_cast<float *>(&part[0][0]);      // [SG][64] floats
        for (int n0 = 0; n0 < g.N; n0 += 64) {
            __syncthreads();
            const int n = n0 + ox;
            float t = 0.f;
            if (n < g.N)
                for (int k = q; k < g.splits; k += SG)
                    t += g.ws_colsum[((size_t)batch * g.splits + k) * g.N + n];
            cpart[q * 64 + ox] = t;
            __syncthreads();
            if (q == 0 && n < g.N) {
#pragma unroll
                for (int u = 1; u < SG; ++u) t += cpart[u * 64 + ox];
                if (g.fold)
                    g.colsum[(size_t)(n / g.fold) * g.colsum_batch_stride + n % g.fold] = t;
                else
                    g.colsum[(size_t)batch * g.colsum_batch_stride + n] = t;
            }
        }
    }
}

// The outstanding reductions of several deferred products in ONE launch (rlx_splitk_reduce_jobs): blockIdx.z = job,
// splitk_reduce4_kernel<16> for ONE ROW per workgroup (blockIdx.x = row, blockIdx.y = batch entry; 64 float4 groups per
// pass, N / 256 passes: every output goes through exactly that kernel's additions) + the narrow layers that read the row
// (rlx_gemm_desc.row_heads): the finished row — bias and activation applied — is kept in LDS, and the first four waves
// then run dense_small_fwd_row's arithmetic on it (a quarter of K per wave, lanes stride over k, wave butterfly, the four
// partials combined in order): the head outputs are bit-identical to rlx_dense_small_forward_multi's, one launch earlier.
constexpr int kRowHeadsMaxN = 2048;
// P = passes of 64 float4 groups that cover a row (N <= 256 P).  Everything the workgroup reads is requested before
// the first barrier — the partials of ALL passes, the epilogue operands, the head's weight rows — so that the row costs
// one exposed memory round trip, like the reduction alone did.
// PPO (rlx_ppo_fc_rows): the row also gets everything of the discrete Clipped-PPO update that is LOCAL to it — the head's loss
// terms, the gradient at the head's outputs (losses_body.hpp: the arithmetic of the stand-alone loss kernels) and
// dz = act'(h) (dy W_head^T), the gradient at this dense layer's pre-activation output (dense_small_bwd_body's chain: n
// ascending from 0.f, then the activation derivative) — from the row in LDS and the head's weight rows in registers.  What
// needs all rows (the heads' weight gradients, the loss scalars) is left to rlx_ppo_heads_tail / rlx_splitk_reduce_jobs_ppo_tail.
struct PpoRowsDev {
    int kind[rlx_small::kMaxProblems];        // per batch entry: 0 none, 1 value head (MSE), 2 discrete policy head
    float *dy[rlx_small::kMaxProblems];       // [M][N] gradient at the head's outputs
    float *dx[rlx_small::kMaxProblems];       // [M][K] gradient at the dense layer's pre-activation output (this tower)
    const float *v_target, *adv, *old_probs; long long ld_old;
    const int *actions;
    const float *clip_scale;
    float *row_terms;                         // [M][4]: value loss term, surrogate, entropy, KL
    float *ratio_out, *clipped_out;
    int *status;
    float clip_eps, beta, grad_scale;
    int lower_act;
};

template <int NN, int P, bool PPO = false>
__global__ void __launch_bounds__(1024) splitk_reduce_rows_kernel(const GemmDev g, const rlx_small::MultiFwd heads,
                                                                  const PpoRowsDev r) {
    constexpr int SG = 16;
    __shared__ float4 part[SG][64];
    __shared__ float hrow[256 * P];
    __shared__ float hpart[4][NN];
    __shared__ float yrow[NN], dzrow[NN], porow[NN], prow[2];
    __shared__ int prow_act;
    const int row = blockIdx.x, batch = blockIdx.y;
    const int n4 = g.N >> 2, mn4 = (g.M * g.N) >> 2;
    const int ox = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int lane = ox, wave = q;
    const size_t mn = (size_t)g.M * g.N;
    const float4 *ws = reinterpret_cast<const float4 *>(g.ws + (size_t)batch * g.splits * mn);
    const rlx_small::SmallDense &hp = heads.p[batch];
    const bool has_head = hp.w != nullptr;                    // workgroup-uniform
    // the head's weight rows of this lane (waves 0-3: a quarter of K each, lanes stride over k): P rows of N floats
    const int kq = has_head ? (hp.K + 3) / 4 : 0;
    const int k1 = has_head && wave < 4 ? min(hp.K, (wave + 1) * kq) : 0;
    float wreg[P][NN];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = wave * kq + lane + 64 * j;
#pragma unroll
        for (int n = 0; n < NN; ++n) wreg[j][n] = (has_head && wave < 4 && k < k1 && n < hp.N) ? hp.w[(size_t)k * hp.N + n] : 0.f;
    }
    // PPO: what thread 0 needs of this row's sample, requested with everything else — one value per thread (threads
    // 0 .. N-1: the old policy's probabilities; 16, 17, 18: advantage, action, clip range; value tower: thread 0 the target)
    float pre = 0.f;
    int pre_i = 0;
    if (PPO) {
        const int tid = threadIdx.x;
        if (r.kind[batch] == 2) {
            if (tid < hp.N) pre = r.old_probs[(size_t)row * r.ld_old + tid];
            else if (tid == 16) pre = r.adv[row];
            else if (tid == 17) pre_i = r.actions[row];
            else if (tid == 18) pre = r.clip_scale ? r.clip_eps * *r.clip_scale : r.clip_eps;
        } else if (r.kind[batch] == 1 && tid == 0) {
            pre = r.v_target[row];
        }
    }
    float4 sp[P];
    float eb[P][4], ea[P][4], ec[P][4];
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        const bool live = ps * 64 + ox < n4;
        const int gid = row * n4 + ps * 64 + ox, col = (ps * 64 + ox) << 2;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) eb[ps][jj] = ea[ps][jj] = ec[ps][jj] = 0.f;
        if (q == 0 && live) {
            const float *c = g.c + (size_t)batch * g.c_batch_stride + (size_t)row * g.ldc + col;
            const float *bias = !g.bias ? nullptr
                                        : g.bias + batch_off(batch, g.inner, g.bias_batch_stride, g.bias_batch_stride2) + col;
            const float *aux = g.aux ? g.aux + (size_t)batch * g.aux_batch_stride + (size_t)row * g.aux_ld + col : nullptr;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (bias) eb[ps][jj] = bias[jj];
                if (aux) ea[ps][jj] = aux[jj];
                if (g.accumulate) ec[ps][jj] = c[jj];
            }
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {                                           // splitk_reduce4_kernel<16>'s additions, in its order
            int k = q;
            for (; k + 3 * SG < g.splits; k += 4 * SG) {
                const float4 a = ws[(size_t)k * mn4 + gid];
                const float4 b = ws[(size_t)(k + SG) * mn4 + gid];
                const float4 cc = ws[(size_t)(k + 2 * SG) * mn4 + gid];
                const float4 d = ws[(size_t)(k + 3 * SG) * mn4 + gid];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
                s.x += cc.x; s.y += cc.y; s.z += cc.z; s.w += cc.w;
                s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
            }
            for (; k < g.splits; k += SG) {
                const float4 a = ws[(size_t)k * mn4 + gid];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
        }
        sp[ps] = s;
    }
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        const bool live = ps * 64 + ox < n4;
        const int col = (ps * 64 + ox) << 2;
        if (ps > 0) __syncthreads();                          // part is reused
        part[q][ox] = sp[ps];
        __syncthreads();
        if (q == 0 && live) {
            float *c = g.c + (size_t)batch * g.c_batch_stride + (size_t)row * g.ldc + col;
            float v[4] = {sp[ps].x, sp[ps].y, sp[ps].z, sp[ps].w};
#pragma unroll
            for (int t = 1; t < SG; ++t) {
                const float4 pp = part[t][ox];
                v[0] += pp.x; v[1] += pp.y; v[2] += pp.z; v[3] += pp.w;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float t = apply_act(v[jj] + eb[ps][jj], g.act);
                if (g.aux) t *= act_deriv(ea[ps][jj], g.deriv);
                v[jj] = g.accumulate ? ec[ps][jj] + t : t;
                hrow[col + jj] = v[jj];
            }
            if ((g.ldc & 3) == 0 && (((uintptr_t)c) & 15) == 0) {
                *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
            }
        }
    }
    if (!has_head) return;
    __syncthreads();                                          // the row is complete in LDS
    // ---- dense_small_fwd_row on it: lane's k = wave * kq + lane + 64 j, ascending, fmaf per output
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = 0.f;
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int k = wave * kq + lane + 64 * j;
            if (k < k1) {
                const float xv = hrow[k];
#pragma unroll
                for (int n = 0; n < NN; ++n)
                    if (n < hp.N) acc[n] = fmaf(xv, wreg[j][n], acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            float v = acc[n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) hpart[wave][n] = v;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < hp.N) {
        const int n = threadIdx.x;
        float v = ((hpart[0][n] + hpart[1][n]) + hpart[2][n]) + hpart[3][n];
        v += hp.b ? hp.b[n] : 0.f;
        v = rlx_small::act_apply(v, hp.act);
        hp.y[(size_t)row * hp.N + n] = v;
        if (PPO) yrow[n] = v;
    }
    if (!PPO) return;
    const int kind = r.kind[batch];                           // workgroup-uniform
    if (kind == 0) return;
    if (kind == 2) {
        const int tid = threadIdx.x;
        if (tid < hp.N) porow[tid] = pre;
        else if (tid == 16) prow[0] = pre;
        else if (tid == 17) prow_act = pre_i;
        else if (tid == 18) prow[1] = pre;
    }
    __syncthreads();
    if (wave == 0) {
        using namespace rlx_losses;
        const int B = g.M;
        if (kind == 1) {                                      // VHead: MSE(target, V), loss weight 1 (head.py:172-181)
            if (lane == 0) {
                const float w = 1.f;
                const float e = yrow[0] - pre;
                float l, gg;
                regression_terms(e, 0, l, gg);
                dzrow[0] = regression_grad(r.grad_scale, w, gg, B);
                r.row_terms[(size_t)row * 4] = w * l;
                if (r.dy[batch]) r.dy[batch][row] = dzrow[0];
            }
        } else {                                              // discrete PPOHead (ppo_head.py:52-116), one action per lane
            PpoRowTerms t;
            float *dyr = r.dy[batch] ? r.dy[batch] + (size_t)row * hp.N : nullptr;
            if (!ppo_discrete_row_wave(yrow, porow, prow_act, hp.N, prow[0], prow[1], r.beta, r.grad_scale, B, dzrow,
                                       r.ratio_out ? r.ratio_out + row : nullptr, r.clipped_out ? r.clipped_out + row : nullptr, t)) {
                if (lane == 0) atomicOr(r.status, 1);
                if (lane < hp.N) dzrow[lane] = 0.f;
                t.sur = t.ent = t.kl = 0.f;
            }
            if (lane == 0) {
                r.row_terms[(size_t)row * 4 + 1] = t.sur;
                r.row_terms[(size_t)row * 4 + 2] = t.ent;
                r.row_terms[(size_t)row * 4 + 3] = t.kl;
            }
            if (dyr && lane < hp.N) dyr[lane] = dzrow[lane];
        }
    }
    __syncthreads();
    if (wave < 4 && r.dx[batch]) {                            // dense_small_bwd_body's dx of this row
        float *dx = r.dx[batch] + (size_t)row * hp.K;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int k = wave * kq + lane + 64 * j;
            if (k < k1) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < NN; ++n)
                    if (n < hp.N) s = fmaf(dzrow[n], wreg[j][n], s);
                dx[k] = r.lower_act ? s * rlx_small::act_deriv_out(hrow[k], r.lower_act) : s;
            }
        }
    }
}

// The part of the discrete Clipped-PPO heads' backward pass that needs ALL rows, behind splitk_reduce_rows_kernel<.., true>:
// dW and db of both heads from the rows' head-output gradients (dense_small_bwd_body: the sums of rlx_ppo_heads_loss_backward,
// bit for bit) and the loss scalars from the rows' terms (the same reduction trees).  blockIdx-free: kblock = 32-feature
// block, is_policy = head; 256 threads.
struct PpoTailDev {
    rlx_small::SmallDenseBwd value, policy;   // dy = the head-output gradients the rows kernel wrote; dx = nullptr
    const float *row_terms;
    float *scalars;
    int batch, red_threads, nn;
    float beta;
};
template <int NN>
__device__ __forceinline__ void ppo_heads_tail_body(const PpoTailDev &a, int kblock, bool is_policy, float *smem,
                                                    float (*red)[256]) {
    using namespace rlx_losses;
    const rlx_small::SmallDenseBwd &p = is_policy ? a.policy : a.value;
    if (kblock * rlx_small::kKL >= p.K) return;
    if (kblock == 0) {
        const int b = threadIdx.x, B = a.batch;
        if (is_policy) {
            float l_sur = 0.f, l_ent = 0.f, l_kl = 0.f;
            if (b < B) {
                l_sur = a.row_terms[(size_t)b * 4 + 1];
                l_ent = a.row_terms[(size_t)b * 4 + 2];
                l_kl = a.row_terms[(size_t)b * 4 + 3];
            }
            block_sum3_first(l_sur, l_ent, l_kl, red, a.red_threads);
            if (threadIdx.x == 0) ppo_discrete_scalars(l_sur, l_ent, l_kl, a.beta, B, a.scalars);
        } else {
            const float local = b < B ? a.row_terms[(size_t)b * 4] : 0.f;
            const float s = block_sum_first(local, red[0], a.red_threads);
            if (threadIdx.x == 0) a.scalars[4] = s / (float)B;
        }
    }
    rlx_small::dense_small_bwd_body<NN, rlx_small::kKL, rlx_small::kRG, false>(p, kblock, 0, smem);
}
constexpr int kTailSmemFloats = 256 * rlx_small::kMaxN + rlx_small::kRG * rlx_small::kKL * rlx_small::kMaxN;

template <int NN>
__global__ void __launch_bounds__(256) ppo_heads_tail_kernel(const PpoTailDev a) {
    __shared__ float smem[kTailSmemFloats];
    __shared__ float red[3][256];
    ppo_heads_tail_body<NN>(a, blockIdx.x, blockIdx.z == 1, smem, red);
}

// (struct ReduceJobs, splitk_reduce_job_body: splitk_reduce_body.hpp — shared with sumtree.hip)
using rlx_reduce::ReduceJobs;
using rlx_reduce::splitk_reduce_job_body;
__global__ void __launch_bounds__(1024) splitk_reduce_jobs_kernel(const ReduceJobs jobs) {
    __shared__ float4 part[16][64];
    splitk_reduce_job_body(jobs.job[blockIdx.z], part);
}
// ... and, in the same launch, the all-rows part of the discrete Clipped-PPO heads' backward pass (ppo_heads_tail_body) as
// the workgroups of one more z slice (z = 0): blockIdx.x = 32-feature block, blockIdx.y = head; their first four waves work, the
// other twelve leave at once (s_barrier counts the surviving waves of a workgroup).  The heads' gradients and the loss
// scalars are not read before the optimizer step, which is behind this launch anyway: one launch less in the chain.
template <int NN>
__global__ void __launch_bounds__(1024) splitk_reduce_jobs_tail_kernel(const ReduceJobs jobs, int n_jobs, const PpoTailDev tail) {
    __shared__ float4 part[16][64];
    __shared__ float smem[kTailSmemFloats];
    __shared__ float red[3][256];
    // z = 0: the tail (dispatched first — its workgroups are few and each is a chain of dependent round trips; behind the
    // 800 reduction workgroups they started 3-4 us late and the launch took 11.1 instead of 6.5 us)
    if (blockIdx.z > 0) {
        splitk_reduce_job_body(jobs.job[blockIdx.z - 1], part);
        return;
    }
    if (blockIdx.y > 1 || threadIdx.x >= 256) return;
    ppo_heads_tail_body<NN>(tail, blockIdx.x, blockIdx.y == 1, smem, red);
}

// Column sums for bias gradients: out[n] = sum_m x[m][n]  (deterministic two-stage reduction).
__global__ void colsum_partial_kernel(const float *__restrict__ x, int M, int N, long long ld,
                                      int rows_per_block, float *__restrict__ part) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += x[(size_t)r * ld + n];
    part[(size_t)blockIdx.y * N + n] = s;
}
__global__ void colsum_final_kernel(const float *__restrict__ part, int nparts, int N,
                                    float *__restrict__ out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * N + n];
    out[n] = accumulate ? out[n] + s : s;
}

// dz = dy * act'(y) in place (the activation derivative written through its output).
__global__ void act_backward_kernel(float *__restrict__ dy, const float *__restrict__ y, long long n,
                                    int kind) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        dy[i] *= act_deriv(y[i], kind);
}

// im2col offset tables (see header comment).
__global__ void conv_tables_kernel(int *__restrict__ rowbase, int *__restrict__ koff, int batch,
                                   int H, int W, int C, int KH, int KW, int stride, int OH, int OW) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = batch * OH * OW, K = KH * KW * C;
    if (t < M) {
        const int ox = t % OW, oy = (t / OW) % OH, b = t / (OW * OH);
        rowbase[t] = ((b * H + oy * stride) * W + ox * stride) * C;
    }
    if (t < K) {
        const int c = t % C, kx = (t / C) % KW, ky = t / (C * KW);
        koff[t] = (ky * W + kx) * C + c;
    }
}

// Tables of the direct input gradient of a convolution (rlx_conv_input_grad).  With stride s the input positions
// split into s*s phases (py, px) = (iy % s, ix % s); inside a phase, row (b, Y, X) = input position (s*Y + py, s*X + px)
// receives  sum over taps (jy, jx), ky = py + s*jy, kx = px + s*jx, and channels co of
//     dY[b, Y - jy, X - jx, co] * W[ky, kx, c, co]        where 0 <= Y - jy < OH and 0 <= X - jx < OW,
// a stride-1 correlation of dY with the phase's sub-kernel: ONE product with M = phases * Mp rows (Mp = rows of a
// phase, padded to a multiple of 128 so that no tile straddles two phases), N = C, K = (KH/s)*(KW/s)*Co.
//   rowbase[m] = ((b*OH + Y)*OW + X)*Co, yx[m] = Y<<16 | X (0x7fff: the row does not exist), crow[m] = offset of the
//   input position in dX / x, or -1;   koff_a[k] = -(jy*OW + jx)*Co + co, jyx[k] = jy<<16 | jx,
//   koff_b[phase][k] = (ky*KW + kx)*C*Co + co   (the weights are stored [KH][KW][C][Co]).
__global__ void conv_dx_tables_kernel(int *__restrict__ rowbase, int *__restrict__ yx, int *__restrict__ crow,
                                      int *__restrict__ koff_a, int *__restrict__ jyx, int *__restrict__ koff_b,
                                      int batch, int H, int W, int C, int KH, int KW, int s, int Co, int OH, int OW,
                                      int Mp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = s * s, Hq = (H + s - 1) / s, Wq = (W + s - 1) / s, Jx = KW / s, K = (KH / s) * Jx * Co;
    if (t < P * Mp) {
        const int ph = t / Mp, r = t - ph * Mp, py = ph / s, px = ph - py * s;
        const int X = r % Wq, Y = (r / Wq) % Hq, b = r / (Wq * Hq);
        const int iy = s * Y + py, ix = s * X + px;
        const bool real = r < batch * Hq * Wq && iy < H && ix < W;
        rowbase[t] = real ? ((b * OH + Y) * OW + X) * Co : 0;
        yx[t] = real ? (Y << 16) | X : (0x7fff << 16);
        crow[t] = real ? ((b * H + iy) * W + ix) * C : -1;
    }
    if (t < K) {
        const int tap = t / Co, co = t - tap * Co, jy = tap / Jx, jx = tap - jy * Jx;
        koff_a[t] = -(jy * OW + jx) * Co + co;
        jyx[t] = (jy << 16) | jx;
        for (int ph = 0; ph < P; ++ph) {
            const int py = ph / s, px = ph - py * s;
            koff_b[ph * K + t] = ((py + s * jy) * KW + (px + s * jx)) * C * Co + co;
        }
    }
}

// col2im as a gather: dX[b,iy,ix,c] = sum over the (ky,kx) whose output position exists of
// dcol[(b,oy,ox)][(ky,kx,c)], multiplied by act'(x) of the layer that produced x.
// VEC = 4: one thread per 4 channels (C % 4 == 0), 16-byte loads; 32-bit index arithmetic.
template <int VEC>
__global__ void col2im_kernel(const float *__restrict__ dcol, float *__restrict__ dx,
                              const float *__restrict__ x_out, int deriv, int batch, int H, int W,
                              int C, int KH, int KW, int stride, int OH, int OW) {
    const int CV = C / VEC;
    const int total = batch * H * W * CV;
    const int K = KH * KW * C;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int c = (t % CV) * VEC;
        const int p = t / CV;
        const int ix = p % W;
        const int q = p / W;
        const int iy = q % H;
        const int b = q / H;
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        for (int ky = iy % stride; ky < KH; ky += stride) {
            const int oy = (iy - ky) / stride;
            if (iy < ky || oy >= OH) continue;
            for (int kx = ix % stride; kx < KW; kx += stride) {
                const int ox = (ix - kx) / stride;
                if (ix < kx || ox >= OW) continue;
                const float *src = dcol + ((size_t)(b * OH + oy) * OW + ox) * K + (ky * KW + kx) * C + c;
                if (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(src);
                    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                } else {
                    s[0] += src[0];
                }
            }
        }
        const size_t o = (size_t)p * C + c;
        if (x_out) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] *= act_deriv(x_out[o + j], deriv);
        }
        if (VEC == 4) {
            *reinterpret_cast<float4 *>(dx + o) = make_float4(s[0], s[1], s[2], s[3]);
        } else {
            dx[o] = s[0];
        }
    }
}

template <int BM, int BN>
int launch_variant(const GemmDev &g, bool a_vec_red, bool a_u8, bool b_vec_red, dim3 grid,
                   hipStream_t s) {
#define RLX_GEMM_CASE(AV, AU, BV)                                                        \
    if (a_vec_red == AV && a_u8 == AU && b_vec_red == BV) {                              \
        RLX_LAUNCH((gemm_kernel<BM, BN, AV, AU, BV>), grid, kThreads, 0, s, g);                    \
        return 0;                                                                        \
    }
    RLX_GEMM_CASE(true, false, false)
    RLX_GEMM_CASE(true, false, true)
    RLX_GEMM_CASE(false, false, false)
    RLX_GEMM_CASE(false, false, true)
    RLX_GEMM_CASE(true, true, false)
    RLX_GEMM_CASE(false, true, false)
#undef RLX_GEMM_CASE
    return -1;
}

template <int BM, int BN, int TM, int TN, int KW = 1>
int launch_fast(const GemmDev &g, bool a_vec_red, bool a_u8, bool b_vec_red, bool a_tab, dim3 grid,
                hipStream_t s) {
#define RLX_FAST_CASE(AV, AU, BV, AT)                                                        \
    if (a_vec_red == AV && a_u8 == AU && b_vec_red == BV && a_tab == AT) {                   \
        RLX_LAUNCH((gemm_fast_kernel<BM, BN, TM, TN, KW, AV, AU, BV, AT>), grid, kThreads, 0, s, g);   \
        return 0;                                                                            \
    }
    RLX_FAST_CASE(true, false, false, false)
    RLX_FAST_CASE(true, false, true, false)
    RLX_FAST_CASE(false, false, false, false)
    RLX_FAST_CASE(false, false, true, false)
    RLX_FAST_CASE(true, false, false, true)
    RLX_FAST_CASE(false, false, false, true)
    RLX_FAST_CASE(true, true, false, true)
    RLX_FAST_CASE(false, true, false, true)
#undef RLX_FAST_CASE
    return -1;
}

template <int BM, int BN, int KW>
int launch_dma(const GemmDev &g, bool a_vec_red, bool b_vec_red, bool a_tab, bool a_u8, dim3 grid, hipStream_t s) {
    if (a_u8) {                  // uint8 frames: through im2col tables, B along N (what fast_combo admits)
        if (!a_tab || b_vec_red) return -1;
        if (a_vec_red) RLX_LAUNCH((gemm_dma_kernel<BM, BN, KW, true, false, true, true>), grid, kThreads, 0, s, g);
        else RLX_LAUNCH((gemm_dma_kernel<BM, BN, KW, false, false, true, true>), grid, kThreads, 0, s, g);
        return 0;
    }
#define RLX_DMA_CASE(AV, BV, AT)                                                                      \
    if (a_vec_red == AV && b_vec_red == BV && a_tab == AT) {                                          \
        RLX_LAUNCH((gemm_dma_kernel<BM, BN, KW, AV, BV, AT>), grid, kThreads, 0, s, g);               \
        return 0;                                                                                     \
    }
    RLX_DMA_CASE(true, false, false)
    RLX_DMA_CASE(true, true, false)
    RLX_DMA_CASE(false, false, false)
    RLX_DMA_CASE(false, true, false)
    RLX_DMA_CASE(true, false, true)
    RLX_DMA_CASE(false, false, true)
#undef RLX_DMA_CASE
    return -1;
}

template <int BM, int BN, int TM, int TN>
int launch_dma_big(const GemmDev &g, bool a_vec_red, bool b_vec_red, bool a_tab, dim3 grid, hipStream_t s) {
#define RLX_BIG_CASE(AV, BV, AT)                                                                      \
    if (a_vec_red == AV && b_vec_red == BV && a_tab == AT) {                                          \
        RLX_LAUNCH((gemm_dma_big_kernel<BM, BN, TM, TN, AV, BV, AT>), grid, kThreads, 0, s, g);               \
        return 0;                                                                                     \
    }
    RLX_BIG_CASE(true, false, false)
    RLX_BIG_CASE(true, false, true)
    RLX_BIG_CASE(true, true, false)
#undef RLX_BIG_CASE
    return -1;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// Tuning constants of the path selection (each was measured on the BASELINE shapes; the A/B records are under
// profiles/: r02_ab_gemm_split_heuristic.txt, r02_ab_thin_16x16_tiles.txt, r01_thin_gemm_ab.txt).
constexpr int kSplitWgsPerCu = 2;          // split K until ~2 workgroups per CU exist ...
constexpr int kSplitMaxTiles = rlx::kCUs;  // ... unless the launch already covers every CU
constexpr int kBigMinTiles = rlx::kCUs;    // 128 x 128 workgroup tiles: at least one per CU (128 x 64: two)
constexpr int kThinMaxTiles = 96;          // thin kernel: at most this many 64 x 64 tiles
constexpr int kThin16MaxTiles = 128;       // 16 x 16 tiles inside a dW + dX pair grid up to this many 32 x 32 tiles ...
constexpr int kThin16SingleMaxTiles = 256; // ... and up to this many for a launch of its own
// in-workgroup K split: taken when the 64 x 64 tiling has fewer than g_kw_below_tiles tiles and the smaller tiling at
// least g_kw_min_tiles (rlx_gemm_tuning: an explicit knob for same-process A/Bs, tools/ab_c2.py)
int g_kw_below_tiles = 192, g_kw_min_tiles = 192, g_xcd_mode = -1;
int g_split_cap = 64;      // most K splits one product is cut into (rlx_gemm_split_cap)
// main loop of the fast tiled kernels: 1 = operands through the LDS-DMA ring (gemm_dma_body), 0 = register-staged
// (gemm_fast_body); rlx_gemm_pipeline, for same-process A/Bs (profiles/r04_ab_gemm_pipeline.txt).  128 x 32 tiles always take the latter;
// uint8 operands take the ring only in mode 2 (4-byte requests: measured equal to the register path, profiles/r04_ab_u8_dma.txt).
int g_dma = 1;
int g_big_tiles = 1;      // 64 x 64 per wave for products of >= kBigMinTiles such tiles (rlx_gemm_big_tiles)

// diagnostics: per-workgroup phase stamps of the fast kernel, one region per rlx_gemm call
struct StampCall { int M, N, K, batch, splits, gx, gy, gz; long long offset; };
struct StampState {
    unsigned long long *buf = nullptr;
    long long capacity = 0, cursor = 0;
    int ncalls = 0;
    StampCall calls[512];
};
StampState g_stamps;

}  // namespace

extern "C" {

int rlx_gemm_debug_stamps(void *buffer, long long capacity_u64) {
    g_stamps.buf = static_cast<unsigned long long *>(buffer);
    g_stamps.capacity = buffer ? capacity_u64 : 0;
    g_stamps.cursor = 0;
    g_stamps.ncalls = 0;
    return RLX_OK;
}

int rlx_gemm_debug_calls(long long *out_host, int max_calls, int *n_calls_host) {
    RLX_REQUIRE(out_host && n_calls_host && max_calls >= 0, "rlx_gemm_debug_calls: bad arguments");
    const int n = g_stamps.ncalls < max_calls ? g_stamps.ncalls : max_calls;
    for (int i = 0; i < n; ++i) {
        const StampCall &c = g_stamps.calls[i];
        long long *o = out_host + 9 * i;
        o[0] = c.M; o[1] = c.N; o[2] = c.K; o[3] = c.batch; o[4] = c.splits;
        o[5] = c.gx; o[6] = c.gy; o[7] = c.gz; o[8] = c.offset;
    }
    *n_calls_host = n;
    return RLX_OK;
}

int rlx_gemm_tuning(int kw_below_tiles, int kw_min_tiles, int xcd_mode) {
    RLX_REQUIRE(kw_below_tiles >= 0 && kw_min_tiles >= 1, "rlx_gemm_tuning: bad thresholds");
    RLX_REQUIRE(xcd_mode == -1 || (xcd_mode >= 0 && xcd_mode <= 1024 && (xcd_mode & (xcd_mode - 1)) == 0),
                "rlx_gemm_tuning: xcd_mode %d is neither -1, 0 nor a power of two", xcd_mode);
    g_kw_below_tiles = kw_below_tiles;
    g_kw_min_tiles = kw_min_tiles;
    g_xcd_mode = xcd_mode;
    return RLX_OK;
}

int rlx_gemm_tuning_get(int *kw_below_tiles_host, int *kw_min_tiles_host, int *xcd_mode_host) {
    if (kw_below_tiles_host) *kw_below_tiles_host = g_kw_below_tiles;
    if (kw_min_tiles_host) *kw_min_tiles_host = g_kw_min_tiles;
    if (xcd_mode_host) *xcd_mode_host = g_xcd_mode;
    return RLX_OK;
}

int rlx_gemm_split_cap(int max_splits) {
    RLX_REQUIRE(max_splits >= 1 && max_splits <= 256, "rlx_gemm_split_cap: 1 <= max_splits <= 256");
    g_split_cap = max_splits;
    return RLX_OK;
}

int rlx_gemm_big_tiles(int on) {
    g_big_tiles = on < 0 ? 0 : (on > 2 ? 2 : on);      // 2: also the 128 x 64 variant for N <= 64
    return RLX_OK;
}

int rlx_gemm_pipeline(int lds_dma_ring) {
    RLX_REQUIRE(lds_dma_ring >= 0 && lds_dma_ring <= 2,
                "rlx_gemm_pipeline: 0 (register-staged), 1 (LDS-DMA ring) or 2 (ring for uint8 operands too)");
    g_dma = lds_dma_ring;
    return RLX_OK;
}

int rlx_gemm_workspace_floats(int M, int N, int K, int batch, long long *floats_host) {
    RLX_REQUIRE(floats_host && M > 0 && N > 0 && K > 0 && batch > 0,
                "rlx_gemm_workspace_floats: bad arguments");
    // upper bound used by rlx_gemm's split heuristic (at most rlx_gemm_split_cap splits, 64 by default)
    *floats_host = ((long long)M * N + N) * batch * g_split_cap;   /* + column-sum partials */
    return RLX_OK;
}

}  // extern "C"

namespace {
// what rlx_gemm decided for one descriptor: everything a launch needs (gemm_impl in planning mode fills it instead
// of launching; rlx_gemm_pair launches two plans as one grid)
struct GemmPlan {
    bool tiled_fast;            // the fast tiled kernel with 64x64 tiles would run (else: thin / generic / folded paths)
    bool thin, a_ck, b_cn;      // the thin kernel would run, with these operand layouts
    bool t16;                   // ... on 16 x 16 output tiles (grid sized accordingly)
    int kw;                     // 2 / 4: the fast kernel on 32 x 64 / 32 x 32 tiles with the slab split over the wave groups would run
    GemmDev g;
    dim3 grid;
    bool a_vec_red, u8, b_vec_red, a_tab;
    int splits, M, N, batch;
    bool allow_fold = false;    // in: towers folded into N may count as tiled_fast (the caller's kernel runs the fold epilogue)
    // what a launch of this descriptor would be (rlx_gemm_describe): the vector-load tiled kernel or not, its tile, wave
    // groups per K slab, K chunks over workgroups, chunk length, operands through the LDS-DMA ring or staged by registers
    bool q_fast = false, q_ring = false;
    int q_bm = 0, q_bn = 0, q_kw = 0, q_splits = 0, q_kchunk = 0;
};

int launch_splitk_reduce(const GemmDev &g, int M, int N, int batch, int splits, hipStream_t s) {
    const long long mn = (long long)M * N;
    if (N % 4 == 0 && mn < (1LL << 31) && aligned16(g.ws)) {
        dim3 rgrid((unsigned)((mn / 4 + 63) / 64), batch);
        if (splits > 16)
            RLX_LAUNCH((splitk_reduce4_kernel<16>), rgrid, 1024, 0, s, g);
        else
            RLX_LAUNCH((splitk_reduce4_kernel<4>), rgrid, 256, 0, s, g);
    } else {
        dim3 rgrid(rlx::grid_for(mn, 256, 1024), batch);
        RLX_LAUNCH((splitk_reduce_kernel), rgrid, 256, 0, s, g);
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// rlx_gemm_desc.row_heads: the reduction pass of a product split over K finishes whole rows and runs the narrow layers that
// read them (splitk_reduce_rows_kernel).  Returns false when the combination is not the one that kernel reproduces bit
// for bit (the caller then reduces as usual and rlx_gemm launches the heads behind it).
thread_local bool tl_row_heads_done = false;
thread_local const rlx_ppo_rows_desc *tl_ppo_rows = nullptr;      // rlx_ppo_fc_rows: the rows also get the Clipped-PPO epilogue
bool launch_reduce_with_row_heads(const GemmDev &g, const rlx_gemm_desc &d, int splits, hipStream_t s) {
    const long long mn = (long long)d.M * d.N;
    if (splits <= 16 || d.N % 4 != 0 || mn >= (1LL << 31) || !aligned16(g.ws) || g.fold || g.colsum ||
        d.N > kRowHeadsMaxN || d.batch > rlx_small::kMaxProblems || d.n_row_heads > d.batch)
        return false;
    rlx_small::MultiFwd m;
    m.n = d.batch;
    for (int t = 0; t < rlx_small::kMaxProblems; ++t) {
        m.p[t] = rlx_small::SmallDense{nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, 0, 0};
        m.towers[t] = 0;
    }
    int nn = 1;
    for (int i = 0; i < d.n_row_heads; ++i) {
        const rlx_small_dense_problem &q = d.row_heads[i];
        if (!q.x || !q.w || !q.y || q.towers != 1 || q.M != d.M || q.K != d.N || q.N < 1 || q.N > rlx_small::kMaxN ||
            q.activation < 0 || q.activation > 2)
            return false;
        int t = -1;
        for (int b = 0; b < d.batch; ++b)
            if (q.x == d.C + (size_t)b * d.c_batch_stride) t = b;
        if (t < 0 || m.p[t].w || d.ldc != d.N) return false;
        m.p[t] = rlx_small::SmallDense{q.x, 0, q.w, 0, q.bias, 0, q.y, 0, q.M, q.K, q.N, q.activation};
        m.towers[t] = 1;
        const int c = q.N == 1 ? 1 : (q.N <= 4 ? 4 : (q.N <= 8 ? 8 : 16));
        if (c > nn) nn = c;
    }
    dim3 rgrid(d.M, d.batch);
    const int passes = (d.N / 4 + 63) / 64;                 // 1 .. 8
    PpoRowsDev r{};
    if (tl_ppo_rows) {
        const rlx_ppo_rows_desc &q = *tl_ppo_rows;
        if (passes > 4 || nn < 4) return false;
        const rlx_small_dense_problem *hq[2] = {&q.value_head, &q.policy_head};
        for (int i = 0; i < 2; ++i) {
            int t = -1;
            for (int b = 0; b < d.batch; ++b)
                if (hq[i]->x == d.C + (size_t)b * d.c_batch_stride) t = b;
            if (t < 0) return false;
            r.kind[t] = i + 1;
            r.dy[t] = const_cast<float *>(hq[i]->dy);
            r.dx[t] = hq[i]->dx;
        }
        r.v_target = q.value_targets; r.adv = q.advantages; r.old_probs = q.old_probs; r.ld_old = q.ld_old;
        r.actions = q.actions; r.clip_scale = q.clip_scale; r.row_terms = q.row_terms;
        r.ratio_out = q.likelihood_ratio; r.clipped_out = q.clipped_likelihood_ratio; r.status = q.status;
        r.clip_eps = q.clip_epsilon; r.beta = q.beta_entropy; r.grad_scale = q.grad_scale;
        r.lower_act = q.value_head.lower_activation;
#define RLX_ROWS_PPO(NNV)                                                                                  \
    if (passes <= 1) RLX_LAUNCH((splitk_reduce_rows_kernel<NNV, 1, true>), rgrid, 1024, 0, s, g, m, r);     \
    else if (passes <= 2) RLX_LAUNCH((splitk_reduce_rows_kernel<NNV, 2, true>), rgrid, 1024, 0, s, g, m, r); \
    else RLX_LAUNCH((splitk_reduce_rows_kernel<NNV, 4, true>), rgrid, 1024, 0, s, g, m, r)
        if (nn == 4) { RLX_ROWS_PPO(4); }
        else if (nn == 8) { RLX_ROWS_PPO(8); }
        else { RLX_ROWS_PPO(16); }
#undef RLX_ROWS_PPO
        return true;
    }
#define RLX_ROWS(NNV, PV) RLX_LAUNCH((splitk_reduce_rows_kernel<NNV, PV>), rgrid, 1024, 0, s, g, m, r)
#define RLX_ROWS_P(NNV)                                    \
    if (passes <= 1) RLX_ROWS(NNV, 1);                     \
    else if (passes <= 2) RLX_ROWS(NNV, 2);                \
    else if (passes <= 4) RLX_ROWS(NNV, 4);                \
    else RLX_ROWS(NNV, 8)
    if (nn == 1) { RLX_ROWS_P(1); }
    else if (nn == 4) { RLX_ROWS_P(4); }
    else if (nn == 8) { RLX_ROWS_P(8); }
    else { RLX_ROWS_P(16); }
#undef RLX_ROWS_P
#undef RLX_ROWS
    return true;
}

// what a deferred reduction needs (rlx_gemm_defer): only a product with the plain-store epilogue and partials the
// float4 reduce scheme can read qualifies; anything else is reduced at once (job->splits = 0)
bool deferrable(const GemmDev &g, int M, int N) {
    return !g.bias && !g.aux && !g.accumulate && g.act == RLX_ACT_NONE && N % 4 == 0 &&
           (long long)M * N < (1LL << 31) && aligned16(g.ws);
}
void fill_job(rlx_splitk_job *job, const GemmDev &g, int M, int N, int batch, int splits) {
    job->partials = g.ws; job->colsum_partials = g.colsum ? g.ws_colsum : nullptr;
    job->C = g.c; job->colsum_out = g.colsum;
    job->ldc = g.ldc; job->c_batch_stride = g.c_batch_stride; job->colsum_batch_stride = g.colsum_batch_stride;
    job->M = M; job->N = N; job->batch = batch; job->splits = splits; job->n_fold = g.fold;
}

int gemm_impl(const rlx_gemm_desc *d_host, void *stream, GemmPlan *plan, rlx_splitk_job *defer = nullptr) {
    if (plan) { plan->tiled_fast = plan->thin = plan->t16 = false; plan->kw = 1; }
    if (defer) defer->splits = 0;
    RLX_REQUIRE(d_host != nullptr, "rlx_gemm: null descriptor");
    const rlx_gemm_desc &d = *d_host;
    RLX_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0,
                "rlx_gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
    RLX_REQUIRE(d.A && d.B && d.C, "rlx_gemm: null operand");
    RLX_REQUIRE(d.a_row_tab || d.a_k_tab || d.a_row_stride == 1 || d.a_k_stride == 1,
                "rlx_gemm: A must be contiguous along one index");
    RLX_REQUIRE(d.b_k_stride == 1 || d.b_n_stride == 1, "rlx_gemm: B must be contiguous along one index");
    RLX_REQUIRE(!d.a_is_u8 || d.a_div != 0.f, "rlx_gemm: a_div must be non-zero for uint8 input");
    RLX_REQUIRE(d.activation >= 0 && d.activation <= 2 && d.deriv_kind >= 0 && d.deriv_kind <= 2,
                "rlx_gemm: unknown activation");

    GemmDev g;
    g.xcd_mode = g_xcd_mode;
    g.M = d.M; g.N = d.N; g.K = d.K;
    g.a.base = d.A; g.a.tab_o = d.a_row_tab; g.a.tab_r = d.a_k_tab;
    g.a.stride_o = d.a_row_stride; g.a.stride_r = d.a_k_stride; g.a.batch_stride = d.a_batch_stride;
    g.b.base = d.B; g.b.tab_o = nullptr; g.b.tab_r = nullptr;
    g.b.stride_o = d.b_n_stride; g.b.stride_r = d.b_k_stride; g.b.batch_stride = d.b_batch_stride;
    g.c = d.C; g.ldc = d.ldc; g.c_batch_stride = d.c_batch_stride;
    g.bias = d.bias; g.bias_batch_stride = d.bias_batch_stride;
    g.fold = d.n_fold > 0 ? d.n_fold : 0;
    g.inner = d.batch_inner > 0 ? d.batch_inner : (1 << 30);
    g.a.batch_stride2 = d.a_batch_stride2; g.b.batch_stride2 = d.b_batch_stride2;
    g.bias_batch_stride2 = d.bias_batch_stride2;
    RLX_REQUIRE(d.batch_inner <= 0 || d.batch % d.batch_inner == 0,
                "rlx_gemm: batch %d is not a multiple of batch_inner %d", d.batch, d.batch_inner);
    g.aux = d.deriv_aux; g.aux_ld = d.aux_ld; g.aux_batch_stride = d.aux_batch_stride;
    g.act = d.activation; g.deriv = d.deriv_kind; g.accumulate = d.accumulate;
    g.a_div = d.a_is_u8 ? d.a_div : 1.f;
    g.colsum = d.colsum_out; g.colsum_batch_stride = d.colsum_batch_stride; g.ws_colsum = nullptr;
    g.stamps = nullptr;

    // vector dimension of each operand: the index whose stride is 1 (tables: declared by caller)
    const bool a_vec_red = d.a_row_tab || d.a_k_tab ? (d.a_vec_along_k != 0) : (d.a_k_stride == 1);
    const bool b_vec_red = d.b_k_stride == 1 && d.b_n_stride != 1;
    const int a_elem = d.a_is_u8 ? 1 : 4;
    if (d.a_row_tab || d.a_k_tab) {
        g.a.vec_ok = d.a_tab_vec_ok != 0;
    } else {
        const long long other = a_vec_red ? d.a_row_stride : d.a_k_stride;
        g.a.vec_ok = (((uintptr_t)d.A) % (4 * a_elem) == 0) && (other % 4 == 0) &&
                     ((d.a_batch_stride % 4) == 0) && ((d.a_batch_stride2 % 4) == 0);
    }
    {
        const long long other = b_vec_red ? d.b_n_stride : d.b_k_stride;
        g.b.vec_ok = aligned16(d.B) && (other % 4 == 0) && ((d.b_batch_stride % 4) == 0) &&
                     ((d.b_batch_stride2 % 4) == 0);
    }

    // thin path: few output tiles and a short reduction -> one launch with K split over the waves
    if (!d.a_row_tab && !d.a_k_tab && !d.a_is_u8 && d.n_fold <= 0 && d.K <= 1024) {
        const long long t64 = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.batch;
        if (t64 <= kThinMaxTiles && (long long)d.N * d.K <= (1 << 18)) {      // MLP-sized weights only: the scalar loads lose on wide B
            const bool a_ck = d.a_k_stride == 1, b_cn = d.b_n_stride == 1;
            // 32 x 32 tiles that leave most CUs idle -> 16 x 16 tiles (4x the workgroups, same reduction order)
            // (a launch of its own profits up to 256 tiles — dW 400x300x100, 130 tiles: 5.22 -> 3.81 us; as one half of a
            // dW + dX pair grid the same problem is better left on 32 x 32 tiles: 5.96 vs 6.6 us for the pair)
            const long long t32 = (long long)((d.M + 31) / 32) * ((d.N + 31) / 32) * d.batch;
            const bool t16 = t32 <= (plan ? kThin16MaxTiles : kThin16SingleMaxTiles);
            const int tile = t16 ? 16 : 32;
            dim3 tgrid((d.N + tile - 1) / tile, (d.M + tile - 1) / tile, d.batch);
            g.splits = 1; g.kchunk = d.K; g.ws = nullptr; g.vec_epi = 0; g.fold = 0;
            if (plan) {
                plan->thin = true; plan->a_ck = a_ck; plan->b_cn = b_cn; plan->t16 = t16;
                plan->g = g; plan->grid = tgrid; plan->splits = 1;
                return RLX_OK;
            }
            hipStream_t ts = rlx::as_stream(stream);
            if (t16) {
                if (a_ck && b_cn) RLX_LAUNCH((gemm_thin16_kernel<true, true>), tgrid, kThreads, 0, ts, g);
                else if (a_ck) RLX_LAUNCH((gemm_thin16_kernel<true, false>), tgrid, kThreads, 0, ts, g);
                else if (b_cn) RLX_LAUNCH((gemm_thin16_kernel<false, true>), tgrid, kThreads, 0, ts, g);
                else RLX_LAUNCH((gemm_thin16_kernel<false, false>), tgrid, kThreads, 0, ts, g);
            }
            else if (a_ck && b_cn) RLX_LAUNCH((gemm_thin_kernel<true, true>), tgrid, kThreads, 0, ts, g);
            else if (a_ck) RLX_LAUNCH((gemm_thin_kernel<true, false>), tgrid, kThreads, 0, ts, g);
            else if (b_cn) RLX_LAUNCH((gemm_thin_kernel<false, true>), tgrid, kThreads, 0, ts, g);
            else RLX_LAUNCH((gemm_thin_kernel<false, false>), tgrid, kThreads, 0, ts, g);
            RLX_LAUNCH_CHECK();
            return RLX_OK;
        }
    }

    // fast path: every 4-element vector group of both operands is full, in range and aligned
    const bool a_tab = d.a_row_tab && d.a_k_tab;
    bool fast = g.a.vec_ok && g.b.vec_ok &&
                      (a_tab || (!d.a_row_tab && !d.a_k_tab)) &&
                      (a_vec_red ? d.K % 4 == 0 : (d.M % 4 == 0 && d.M >= 4)) &&
                      (b_vec_red ? d.K % 4 == 0 : (d.N % 4 == 0 && d.N >= 4)) && d.K >= 4;
    // tile shape: narrow-N problems use 128x32 workgroup tiles, everything else 64x64 (2x2 / 2x1 accumulator tiles
    // per wave measured 15-40 % slower at the C2 shapes: epilogue-bound, profiles/r01_gemm_microbench.txt)
    const bool narrow = d.N <= 32;
    int BM = narrow ? 128 : 64, BN = narrow ? 32 : 64;
    auto tiles_of = [&](int bm, int bn) {
        return (long long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.batch;
    };
    // Mid-sized problems (too few 64 x 64 tiles for the chip, enough 32 x 64 or 32 x 32 ones): smaller tiles with the K
    // slab split over the waves of the workgroup instead of K split over workgroups — no partials, no reduce launch.
    int KW = 1;
    const bool u8_in = d.a_is_u8 != 0;
    // (operand combinations launch_fast instantiates; anything else falls back to the bounds-checked 64 x 64 kernel)
    const bool fast_combo = (!u8_in && !a_tab) || (a_tab && !b_vec_red);
    if (fast && fast_combo && !narrow && BM == 64 && BN == 64 && tiles_of(64, 64) < g_kw_below_tiles) {   // default: 3/4 of the CUs
        const int kw_min = d.kw_min_tiles > 0 ? d.kw_min_tiles : g_kw_min_tiles;       // (a per-product hint: rlx.h)
        if (tiles_of(32, 64) >= kw_min) { BM = 32; BN = 64; KW = 2; }
        else if (tiles_of(32, 32) >= kw_min) { BM = 32; BN = 32; KW = 4; }
    }
    const int tiles = (int)tiles_of(BM, BN);
    // split K until ~2 workgroups per CU exist, keeping >= 2 slabs of 32 per split; a launch that
    // already covers every CU is left alone (its reduce pass would cost more than the imbalance)
    // (measured on C2 / C3, splitting every under-filled launch is fastest: 118 vs 129 ms and 325 vs 383 ms per
    // bench step against a minimum K of 1024.)
    int splits = 1;
    if (KW == 1 && d.workspace && tiles <= kSplitMaxTiles) {
        const int want = (kSplitWgsPerCu * rlx::kCUs + tiles - 1) / tiles;
        const int max_by_k = d.K / (2 * BK);
        splits = want < max_by_k ? want : max_by_k;
        if (splits > g_split_cap) splits = g_split_cap;
        if (splits < 1) splits = 1;
        while (splits > 1 && ((long long)d.M * d.N + d.N) * d.batch * splits > d.workspace_floats) --splits;
    }
    int kchunk = ((d.K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (d.K + kchunk - 1) / kchunk;
    if (fast && a_tab && kchunk > kTabChunk) {
        // the fast kernel stages a chunk's im2col offsets in LDS; a longer chunk (no workspace to split K into) takes
        // the bounds-checked kernel
        fast = false;
        BM = narrow ? 128 : 64;
        BN = narrow ? 32 : 64;
        KW = 1;
    }
    // large products: 2 x 2 accumulator tiles per wave (gemm_dma_big_kernel) — fp32 operands on the ring, no K split
    int big = 0;            // 1: 128 x 128 (64 x 64 per wave), 2: 128 x 64 (64 x 32 per wave)
    if (fast && g_dma && g_big_tiles && !plan && !u8_in && !narrow && KW == 1 && splits == 1 && !g.fold &&
        BM == 64 && BN == 64 && (!a_tab || kchunk <= kDmaTabChunk) && !(!a_vec_red) && !(a_tab && b_vec_red)) {
        if (d.N >= 128 && tiles_of(128, 128) >= kBigMinTiles) { big = 1; BM = 128; BN = 128; }
        // (128 x 64 with 64 x 32 per wave for the N = 64 convolutions of the whole-dataset passes: neutral to slightly worse,
        // profiles/r05_ab_big_wave_tiles.txt — g_big_tiles == 2 keeps it reachable for A/Bs)
        else if (g_big_tiles == 2 && d.N <= 64 && tiles_of(128, 64) >= 2 * kBigMinTiles) { big = 2; BM = 128; BN = 64; }
    }
    g.splits = splits;
    g.kchunk = kchunk;
    g.ws = d.workspace;
    if (splits > 1) g.ws_colsum = d.workspace + (size_t)d.M * d.N * d.batch * splits;

    g.vec_epi = d.N % 4 == 0 &&
                (splits > 1 ? aligned16(d.workspace) && ((long long)d.M * d.N) % 4 == 0
                            : (aligned16(d.C) && d.ldc % 4 == 0 && d.c_batch_stride % 4 == 0 &&
                               (!d.bias || (aligned16(d.bias) && d.bias_batch_stride % 4 == 0 && d.bias_batch_stride2 % 4 == 0)) &&
                               (!d.deriv_aux || (aligned16(d.deriv_aux) && d.aux_ld % 4 == 0 &&
                                                 d.aux_batch_stride % 4 == 0))));
    hipStream_t s = rlx::as_stream(stream);
    dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch * splits);
    if (g.fold) {
        // towers folded into N: only the fast kernel with 16-byte epilogue and the float4 reduce know the
        // column -> (tower, column) mapping; anything else runs the equivalent batched problem
        // (one tower per batch index, A shared through a zero batch stride)
        const bool fold_ok = fast && g.vec_epi && !b_vec_red && d.batch == 1 && d.N % g.fold == 0 &&
                             g.fold % 4 == 0 && !d.deriv_aux && !d.accumulate && d.batch_inner <= 0 &&
                             d.b_batch_stride % 4 == 0 && d.c_batch_stride % 4 == 0 &&
                             d.bias_batch_stride % 4 == 0 && (long long)d.M * d.N < (1LL << 31);
        if (!fold_ok) {
            RLX_REQUIRE(d.batch == 1 && d.N % g.fold == 0, "rlx_gemm: n_fold=%d needs batch 1 and N %% n_fold == 0",
                        g.fold);
            rlx_gemm_desc t = d;
            t.batch = d.N / g.fold;
            t.N = g.fold;
            t.n_fold = 0;
            t.a_batch_stride = 0;
            if (plan) return RLX_OK;
            return gemm_impl(&t, stream, nullptr, defer);
        }
    }
    if (plan) {
        plan->q_fast = fast; plan->q_bm = BM; plan->q_bn = BN; plan->q_kw = KW; plan->q_splits = splits; plan->q_kchunk = kchunk;
        plan->q_ring = fast && g_dma && (g_dma >= 2 || !d.a_is_u8) && !narrow && (!a_tab || kchunk <= kDmaTabChunk);
        if (fast && ((BM == 64 && BN == 64) || KW > 1) && (!g.fold || plan->allow_fold)) {
            plan->tiled_fast = KW == 1;
            plan->kw = KW;
            plan->g = g;
            plan->grid = grid;
            plan->a_vec_red = a_vec_red; plan->u8 = d.a_is_u8 != 0; plan->b_vec_red = b_vec_red; plan->a_tab = a_tab;
            plan->splits = splits; plan->M = d.M; plan->N = d.N; plan->batch = d.batch;
        }
        return RLX_OK;
    }
    if (g_stamps.buf && fast) {
        const long long need = 4LL * grid.x * grid.y * grid.z;
        if (g_stamps.cursor + need <= g_stamps.capacity && g_stamps.ncalls < 512) {
            g.stamps = g_stamps.buf + g_stamps.cursor;
            g_stamps.calls[g_stamps.ncalls++] = {d.M, d.N, d.K, d.batch, splits, (int)grid.x, (int)grid.y,
                                                 (int)grid.z, g_stamps.cursor};
            g_stamps.cursor += need;
        }
    }
    int rc = -1;
    if (big == 1) rc = launch_dma_big<128, 128, 2, 2>(g, a_vec_red, b_vec_red, a_tab, grid, s);
    else if (big == 2) rc = launch_dma_big<128, 64, 2, 1>(g, a_vec_red, b_vec_red, a_tab, grid, s);
    if (rc != 0 && big) {                       // (no instance for this operand combination: back to 64 x 64)
        big = 0; BM = BN = 64;
        grid = dim3((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch * splits);
    }
    if (rc != 0 && fast && g_dma && (g_dma >= 2 || !d.a_is_u8) && !narrow && (!a_tab || kchunk <= kDmaTabChunk)) {
        const bool u8 = d.a_is_u8 != 0;
        if (KW == 2) rc = launch_dma<32, 64, 2>(g, a_vec_red, b_vec_red, a_tab, u8, grid, s);
        else if (KW == 4) rc = launch_dma<32, 32, 4>(g, a_vec_red, b_vec_red, a_tab, u8, grid, s);
        else rc = launch_dma<64, 64, 1>(g, a_vec_red, b_vec_red, a_tab, u8, grid, s);
    }
    if (fast && rc != 0) {
        const bool u8 = d.a_is_u8 != 0;
        if (narrow) rc = launch_fast<128, 32, 1, 1>(g, a_vec_red, u8, b_vec_red, a_tab, grid, s);
        else if (KW == 2) rc = launch_fast<32, 64, 1, 1, 2>(g, a_vec_red, u8, b_vec_red, a_tab, grid, s);
        else if (KW == 4) rc = launch_fast<32, 32, 1, 1, 4>(g, a_vec_red, u8, b_vec_red, a_tab, grid, s);
        else rc = launch_fast<64, 64, 1, 1>(g, a_vec_red, u8, b_vec_red, a_tab, grid, s);
    }
    if (rc != 0) {
        RLX_REQUIRE(BM * BN == 4096, "rlx_gemm: internal tile selection error");
        rc = narrow ? launch_variant<128, 32>(g, a_vec_red, d.a_is_u8 != 0, b_vec_red, grid, s)
                    : launch_variant<64, 64>(g, a_vec_red, d.a_is_u8 != 0, b_vec_red, grid, s);
    }
    RLX_REQUIRE(rc == 0, "rlx_gemm: unsupported operand combination (uint8 A with transposed B)");
    RLX_LAUNCH_CHECK();
    if (splits > 1) {
        if (defer && deferrable(g, d.M, d.N)) {
            fill_job(defer, g, d.M, d.N, d.batch, splits);
            return RLX_OK;
        }
        if (d.row_heads && d.n_row_heads > 0 && launch_reduce_with_row_heads(g, d, splits, s)) {
            tl_row_heads_done = true;
            RLX_LAUNCH_CHECK();
            return RLX_OK;
        }
        return launch_splitk_reduce(g, d.M, d.N, d.batch, splits, s);
    }
    return RLX_OK;
}

}  // namespace

extern "C" {

int rlx_gemm_describe(const rlx_gemm_desc *desc, int *out8_host) {
    RLX_REQUIRE(desc && out8_host, "rlx_gemm_describe: null pointer");
    GemmPlan plan{};
    const int rc = gemm_impl(desc, nullptr, &plan, nullptr);
    if (rc != RLX_OK) return rc;
    out8_host[0] = plan.q_fast; out8_host[1] = plan.q_bm; out8_host[2] = plan.q_bn; out8_host[3] = plan.q_kw;
    out8_host[4] = plan.q_splits; out8_host[5] = plan.q_kchunk; out8_host[6] = plan.q_ring; out8_host[7] = plan.thin;
    return RLX_OK;
}

int rlx_gemm(const rlx_gemm_desc *d_host, void *stream) {
    tl_row_heads_done = false;
    const int rc = gemm_impl(d_host, stream, nullptr);
    if (rc != RLX_OK || !d_host->row_heads || d_host->n_row_heads <= 0 || tl_row_heads_done) return rc;
    return rlx_dense_small_forward_multi(d_host->row_heads, d_host->n_row_heads, stream);   // not fused: behind the product
}

static int gemm_pair_impl(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad, void *stream,
                          rlx_splitk_job *defer);

int rlx_gemm_pair(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad, void *stream) {
    return gemm_pair_impl(weight_grad, input_grad, stream, nullptr);
}

int rlx_gemm_defer(const rlx_gemm_desc *desc_host, rlx_splitk_job *job_host, void *stream) {
    RLX_REQUIRE(job_host != nullptr, "rlx_gemm_defer: null job");
    return gemm_impl(desc_host, stream, nullptr, job_host);
}

int rlx_gemm_pair_defer(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad,
                        rlx_splitk_job *weight_grad_job_host, void *stream) {
    RLX_REQUIRE(weight_grad_job_host != nullptr, "rlx_gemm_pair_defer: null job");
    return gemm_pair_impl(weight_grad, input_grad, stream, weight_grad_job_host);
}

int rlx_splitk_reduce_jobs(const rlx_splitk_job *jobs_host, int n_jobs, void *stream) {
    RLX_REQUIRE(jobs_host && n_jobs >= 0 && n_jobs <= RLX_MAX_SPLITK_JOBS,
                "rlx_splitk_reduce_jobs: 0..%d jobs (got %d)", RLX_MAX_SPLITK_JOBS, n_jobs);
    ReduceJobs jobs;
    int n = 0, gx = 1, gy = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const rlx_splitk_job &j = jobs_host[i];
        if (j.splits <= 1) continue;
        RLX_REQUIRE(j.partials && j.C && j.M > 0 && j.N > 0 && j.N % 4 == 0 && j.batch > 0 &&
                    (long long)j.M * j.N < (1LL << 31) && aligned16(j.partials),
                    "rlx_splitk_reduce_jobs: job %d is not a float4-reducible product", i);
        jobs.job[n++] = j;
        const int bx = (int)(((long long)j.M * j.N / 4 + 63) / 64);
        gx = bx > gx ? bx : gx;
        gy = j.batch > gy ? j.batch : gy;
    }
    if (n == 0) return RLX_OK;
    RLX_LAUNCH((splitk_reduce_jobs_kernel), dim3(gx, gy, n), 1024, 0, rlx::as_stream(stream), jobs);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// ---- discrete Clipped PPO: the last dense layer's K-split reduction finishes the rows AND what of the heads' losses /
// backward pass is local to a row (rlx_ppo_fc_rows); the rest rides on the backward pass's deferred reductions
namespace {
int ppo_rows_check(const rlx_ppo_rows_desc *q, const char *who) {
    RLX_REQUIRE(q && q->value_targets && q->actions && q->advantages && q->old_probs && q->row_terms && q->scalars &&
                q->status, "%s: null pointer", who);
    const rlx_small_dense_problem *h[2] = {&q->value_head, &q->policy_head};
    for (int i = 0; i < 2; ++i) {
        RLX_REQUIRE(h[i]->x && h[i]->w && h[i]->y && h[i]->dy && h[i]->dw && h[i]->dx, "%s: null pointer in head %d", who, i);
        RLX_REQUIRE(h[i]->towers == 1 && h[i]->M == q->batch && h[i]->K > 0 && h[i]->activation == 0,
                    "%s: head %d must be one linear tower over the %d rows of the batch", who, i, q->batch);
        RLX_REQUIRE(h[i]->lower_activation >= 0 && h[i]->lower_activation <= 2, "%s: unknown activation", who);
    }
    RLX_REQUIRE(q->value_head.N == 1 && q->policy_head.N >= 2 && q->policy_head.N <= rlx_small::kMaxN &&
                q->value_head.K == q->policy_head.K && q->value_head.lower_activation == q->policy_head.lower_activation,
                "%s: value head N = 1, policy head 2 <= N <= %d, both on the same dense layer", who, rlx_small::kMaxN);
    RLX_REQUIRE(q->batch >= 1 && q->batch <= 256, "%s: 1 <= batch <= 256 (one row per thread in the scalar sums)", who);
    return RLX_OK;
}
PpoTailDev ppo_tail_dev(const rlx_ppo_rows_desc &q) {
    PpoTailDev t{};
    const rlx_small_dense_problem *h[2] = {&q.value_head, &q.policy_head};
    rlx_small::SmallDenseBwd *d[2] = {&t.value, &t.policy};
    for (int i = 0; i < 2; ++i)
        *d[i] = rlx_small::SmallDenseBwd{h[i]->x, 0, h[i]->w, 0, h[i]->dy, 0, nullptr, 0, h[i]->dw, 0, h[i]->db, 0, nullptr, 0,
                                         h[i]->M, h[i]->K, h[i]->N, 0, 0};
    t.row_terms = q.row_terms; t.scalars = q.scalars; t.batch = q.batch; t.beta = q.beta_entropy;
    t.red_threads = 64;
    while (t.red_threads < q.batch) t.red_threads <<= 1;     // block_for() of the stand-alone loss kernels
    t.nn = q.policy_head.N <= 4 ? 4 : (q.policy_head.N <= 8 ? 8 : 16);
    return t;
}
}  // namespace

int rlx_ppo_fc_rows_supported(const rlx_gemm_desc *fc, int n_actions) {
    if (!fc || fc->row_heads || fc->batch != 2 || fc->N % 4 != 0 || fc->N > 1024 || fc->M > 256 || n_actions < 2 ||
        n_actions > rlx_small::kMaxN || fc->ldc != fc->N || fc->n_fold > 0 || fc->colsum_out)
        return 0;
    GemmPlan plan{};
    if (gemm_impl(fc, nullptr, &plan, nullptr) != RLX_OK) return 0;
    return plan.q_splits > 16 ? 1 : 0;
}

int rlx_ppo_fc_rows(const rlx_gemm_desc *fc, const rlx_ppo_rows_desc *rows, void *stream) {
    const int rc0 = ppo_rows_check(rows, "rlx_ppo_fc_rows");
    if (rc0 != RLX_OK) return rc0;
    RLX_REQUIRE(fc && !fc->row_heads && rlx_ppo_fc_rows_supported(fc, rows->policy_head.N) && fc->M == rows->batch &&
                fc->N == rows->value_head.K,
                "rlx_ppo_fc_rows: not a product rlx_ppo_fc_rows_supported accepts, or its shape does not match the heads");
    rlx_small_dense_problem heads[2] = {rows->value_head, rows->policy_head};
    rlx_gemm_desc d = *fc;
    d.row_heads = heads;
    d.n_row_heads = 2;
    tl_row_heads_done = false;
    tl_ppo_rows = rows;
    const int rc = gemm_impl(&d, stream, nullptr);
    tl_ppo_rows = nullptr;
    if (rc != RLX_OK) return rc;
    RLX_REQUIRE(tl_row_heads_done, "rlx_ppo_fc_rows: the product did not take the row-finishing reduction");
    return RLX_OK;
}

int rlx_ppo_heads_tail(const rlx_ppo_rows_desc *rows, void *stream) {
    const int rc0 = ppo_rows_check(rows, "rlx_ppo_heads_tail");
    if (rc0 != RLX_OK) return rc0;
    const PpoTailDev t = ppo_tail_dev(*rows);
    const int kb = (rows->value_head.K + rlx_small::kKL - 1) / rlx_small::kKL;
    dim3 grid(kb, 1, 2);
    hipStream_t s = rlx::as_stream(stream);
    if (t.nn == 4) RLX_LAUNCH((ppo_heads_tail_kernel<4>), grid, 256, 0, s, t);
    else if (t.nn == 8) RLX_LAUNCH((ppo_heads_tail_kernel<8>), grid, 256, 0, s, t);
    else RLX_LAUNCH((ppo_heads_tail_kernel<16>), grid, 256, 0, s, t);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_splitk_reduce_jobs_ppo_tail(const rlx_splitk_job *jobs_host, int n_jobs, const rlx_ppo_rows_desc *rows, void *stream) {
    RLX_REQUIRE(jobs_host && n_jobs >= 0 && n_jobs <= RLX_MAX_SPLITK_JOBS,
                "rlx_splitk_reduce_jobs_ppo_tail: 0..%d jobs (got %d)", RLX_MAX_SPLITK_JOBS, n_jobs);
    const int rc0 = ppo_rows_check(rows, "rlx_splitk_reduce_jobs_ppo_tail");
    if (rc0 != RLX_OK) return rc0;
    ReduceJobs jobs;
    int n = 0, gx = 1, gy = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const rlx_splitk_job &j = jobs_host[i];
        if (j.splits <= 1) continue;
        RLX_REQUIRE(j.partials && j.C && j.M > 0 && j.N > 0 && j.N % 4 == 0 && j.batch > 0 &&
                    (long long)j.M * j.N < (1LL << 31) && aligned16(j.partials),
                    "rlx_splitk_reduce_jobs_ppo_tail: job %d is not a float4-reducible product", i);
        jobs.job[n++] = j;
        const int bx = (int)(((long long)j.M * j.N / 4 + 63) / 64);
        gx = bx > gx ? bx : gx;
        gy = j.batch > gy ? j.batch : gy;
    }
    if (n == 0) return rlx_ppo_heads_tail(rows, stream);
    const PpoTailDev t = ppo_tail_dev(*rows);
    const int kb = (rows->value_head.K + rlx_small::kKL - 1) / rlx_small::kKL;
    gx = kb > gx ? kb : gx;
    gy = gy < 2 ? 2 : gy;
    dim3 grid(gx, gy, n + 1);
    hipStream_t s = rlx::as_stream(stream);
    if (t.nn == 4) RLX_LAUNCH((splitk_reduce_jobs_tail_kernel<4>), grid, 1024, 0, s, jobs, n, t);
    else if (t.nn == 8) RLX_LAUNCH((splitk_reduce_jobs_tail_kernel<8>), grid, 1024, 0, s, jobs, n, t);
    else RLX_LAUNCH((splitk_reduce_jobs_tail_kernel<16>), grid, 1024, 0, s, jobs, n, t);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int gemm_pair_impl(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad, void *stream,
                          rlx_splitk_job *defer) {
    if (defer) defer->splits = 0;
    GemmPlan pw, px;
    int rc = gemm_impl(weight_grad, stream, &pw);
    if (rc != RLX_OK) return rc;
    rc = gemm_impl(input_grad, stream, &px);
    if (rc != RLX_OK) return rc;
    const bool pairable = pw.tiled_fast && (px.tiled_fast || px.kw > 1) &&
                          !pw.a_vec_red && !pw.u8 && !pw.b_vec_red &&                 // X^T dY
                          px.a_vec_red && !px.u8 && px.b_vec_red && !px.a_tab &&       // dY W^T
                          true;
    if (pw.thin && px.thin && !pw.a_ck && pw.b_cn && px.a_ck && !px.b_cn) {
        GemmPairDev p;
        p.g[0] = pw.g; p.g[1] = px.g;
        p.gx[0] = pw.grid.x; p.gy[0] = pw.grid.y; p.gx[1] = px.grid.x; p.gy[1] = px.grid.y;
        p.n0 = (int)(pw.grid.x * pw.grid.y * pw.grid.z);
        p.t16[0] = pw.t16; p.t16[1] = px.t16;
        const unsigned total = (unsigned)p.n0 + px.grid.x * px.grid.y * px.grid.z;
        RLX_LAUNCH((gemm_thin_pair_kernel), total, kThreads, 0, rlx::as_stream(stream), p);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    // both problems may split K: their partials must not share workspace memory
    const bool ws_clash = pw.splits > 1 && px.splits > 1 && pw.g.ws == px.g.ws;
    if (!pairable || ws_clash) {
        rc = gemm_impl(weight_grad, stream, nullptr, defer);
        if (rc != RLX_OK) return rc;
        return gemm_impl(input_grad, stream, nullptr);
    }
    GemmPairDev p;
    p.g[0] = pw.g; p.g[1] = px.g;
    p.gx[0] = pw.grid.x; p.gy[0] = pw.grid.y; p.gx[1] = px.grid.x; p.gy[1] = px.grid.y;
    p.n0 = (int)(pw.grid.x * pw.grid.y * pw.grid.z);
    p.t16[0] = p.t16[1] = 0;
    const unsigned total = (unsigned)p.n0 + px.grid.x * px.grid.y * px.grid.z;
    hipStream_t s = rlx::as_stream(stream);
#define RLX_PAIR_CASE(AT, KWX)                                                                   \
    if (pw.a_tab == AT && px.kw == KWX) {                                                        \
        if (g_dma && (!AT || pw.g.kchunk <= kDmaTabChunk))                                       \
            RLX_LAUNCH((gemm_dma_pair_kernel<AT, KWX>), total, kThreads, 0, s, p);               \
        else RLX_LAUNCH((gemm_fast_pair_kernel<AT, KWX>), total, kThreads, 0, s, p);             \
    }
    RLX_PAIR_CASE(true, 1) RLX_PAIR_CASE(true, 2) RLX_PAIR_CASE(true, 4)
    RLX_PAIR_CASE(false, 1) RLX_PAIR_CASE(false, 2) RLX_PAIR_CASE(false, 4)
#undef RLX_PAIR_CASE
    RLX_LAUNCH_CHECK();
    if (pw.splits > 1) {
        if (defer && deferrable(pw.g, pw.M, pw.N)) {
            fill_job(defer, pw.g, pw.M, pw.N, pw.batch, pw.splits);
        } else {
            rc = launch_splitk_reduce(pw.g, pw.M, pw.N, pw.batch, pw.splits, s);
            if (rc != RLX_OK) return rc;
        }
    }
    if (px.splits > 1) return launch_splitk_reduce(px.g, px.M, px.N, px.batch, px.splits, s);
    return RLX_OK;
}

int rlx_gemm_multi_defer(const rlx_gemm_desc *descs_host, int n, rlx_splitk_job *jobs_host, void *stream) {
    RLX_REQUIRE(descs_host && jobs_host && n >= 1 && n <= 3, "rlx_gemm_multi_defer: 1 .. 3 descriptors and their jobs");
    GemmPlan pl[3];
    bool multi = n >= 2 && g_dma != 0;
    for (int i = 0; i < n; ++i) {
        jobs_host[i].splits = 0;
        pl[i].allow_fold = true;
        const int rc = gemm_impl(&descs_host[i], stream, &pl[i]);
        if (rc != RLX_OK) return rc;
        // what the one-launch form takes: 64 x 64 tiles, A^T gathered through im2col tables (vector along the outer
        // index), B along N; fp32 on the ring (its table chunk bound) or uint8 on the register-staged loop
        multi = multi && pl[i].tiled_fast && !pl[i].a_vec_red && !pl[i].b_vec_red && pl[i].a_tab &&
                (pl[i].u8 ? pl[i].g.kchunk <= kTabChunk : pl[i].g.kchunk <= kDmaTabChunk);
    }
    for (int i = 0; multi && i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (pl[i].splits > 1 && pl[j].splits > 1 && pl[i].g.ws == pl[j].g.ws) multi = false;   // partials would collide
    if (!multi) {
        for (int i = 0; i < n; ++i) {
            const int rc = gemm_impl(&descs_host[i], stream, nullptr, &jobs_host[i]);
            if (rc != RLX_OK) return rc;
        }
        return RLX_OK;
    }
    GemmMultiDev p;
    p.n = n;
    p.start[0] = 0;
    for (int i = 0; i < 3; ++i) {
        const int k = i < n ? i : n - 1;
        p.g[i] = pl[k].g;
        p.gx[i] = pl[k].grid.x; p.gy[i] = pl[k].grid.y;
        p.kind[i] = pl[k].u8 ? 1 : 0;
        p.start[i + 1] = p.start[i] + (i < n ? (int)(pl[k].grid.x * pl[k].grid.y * pl[k].grid.z) : 0);
    }
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((gemm_multi_dw_kernel), (unsigned)p.start[n], kThreads, 0, s, p);
    RLX_LAUNCH_CHECK();
    for (int i = 0; i < n; ++i) {
        if (pl[i].splits <= 1) continue;
        if (deferrable(pl[i].g, pl[i].M, pl[i].N)) {
            fill_job(&jobs_host[i], pl[i].g, pl[i].M, pl[i].N, pl[i].batch, pl[i].splits);
        } else {
            const int rc = launch_splitk_reduce(pl[i].g, pl[i].M, pl[i].N, pl[i].batch, pl[i].splits, s);
            if (rc != RLX_OK) return rc;
        }
    }
    return RLX_OK;
}

int rlx_colsum(const float *x, int M, int N, long long ld, float *out, int accumulate,
               float *workspace, long long workspace_floats, void *stream) {
    RLX_REQUIRE(x && out && workspace, "rlx_colsum: null pointer");
    RLX_REQUIRE(M > 0 && N > 0 && ld >= N, "rlx_colsum: bad shape");
    int parts = (M + 255) / 256;
    if (parts > 256) parts = 256;
    while (parts > 1 && (long long)parts * N > workspace_floats) --parts;
    RLX_REQUIRE((long long)parts * N <= workspace_floats, "rlx_colsum: workspace too small");
    const int rows_per_block = (M + parts - 1) / parts;
    parts = (M + rows_per_block - 1) / rows_per_block;
    hipStream_t s = rlx::as_stream(stream);
    dim3 grid((N + 63) / 64, parts);
    RLX_LAUNCH((colsum_partial_kernel), grid, 64, 0, s, x, M, N, ld, rows_per_block, workspace);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((colsum_final_kernel), (N + 63) / 64, 64, 0, s, workspace, parts, N, out, accumulate);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_act_backward(float *dy, const float *y, long long n, int kind, void *stream) {
    RLX_REQUIRE(dy && y && n > 0, "rlx_act_backward: bad arguments");
    RLX_REQUIRE(kind >= 0 && kind <= 2, "rlx_act_backward: unknown activation %d", kind);
    if (kind == RLX_ACT_NONE) return RLX_OK;
    RLX_LAUNCH((act_backward_kernel), rlx::grid_for(n, 256), 256, 0, rlx::as_stream(stream), dy, y, n, kind);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv_tables(int *rowbase, int *koff, int batch, int H, int W, int C, int KH, int KW,
                    int stride, void *stream) {
    RLX_REQUIRE(rowbase && koff, "rlx_conv_tables: null pointer");
    RLX_REQUIRE(batch > 0 && H >= KH && W >= KW && C > 0 && KH > 0 && KW > 0 && stride > 0,
                "rlx_conv_tables: bad convolution geometry");
    const int OH = (H - KH) / stride + 1, OW = (W - KW) / stride + 1;   // VALID padding
    const int M = batch * OH * OW, K = KH * KW * C;
    RLX_REQUIRE((long long)batch * H * W * C < (1LL << 31), "rlx_conv_tables: input too large for int32 offsets");
    const int n = M > K ? M : K;
    RLX_LAUNCH((conv_tables_kernel), (n + 255) / 256, 256, 0, rlx::as_stream(stream), rowbase, koff, batch, H, W, C, KH, KW, stride, OH, OW);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int conv_dx_geometry(int batch, int H, int W, int C, int KH, int KW, int stride, int Co, int *OH, int *OW,
                            int *Mp, int *K, long long *ints) {
    RLX_REQUIRE(batch > 0 && H >= KH && W >= KW && C > 0 && Co > 0 && KH > 0 && KW > 0 && stride > 0,
                "rlx_conv_input_grad: bad convolution geometry");
    RLX_REQUIRE(KH % stride == 0 && KW % stride == 0,
                "rlx_conv_input_grad: the kernel (%d x %d) must be a multiple of the stride (%d)", KH, KW, stride);
    RLX_REQUIRE(C % 4 == 0 && Co % 4 == 0, "rlx_conv_input_grad: channel counts must be multiples of 4 (C %d, Co %d)", C, Co);
    *OH = (H - KH) / stride + 1;
    *OW = (W - KW) / stride + 1;
    const int Hq = (H + stride - 1) / stride, Wq = (W + stride - 1) / stride;
    const long long rows = (long long)batch * Hq * Wq;
    *Mp = (int)((rows + 127) / 128 * 128);
    *K = (KH / stride) * (KW / stride) * Co;
    RLX_REQUIRE(*K <= kWinChunk, "rlx_conv_input_grad: reduction of %d exceeds %d", *K, kWinChunk);
    const long long P = (long long)stride * stride;
    RLX_REQUIRE(P * *Mp < (1LL << 30) && (long long)batch * H * W * C < (1LL << 31) && H < 0x7fff && W < 0x7fff,
                "rlx_conv_input_grad: tensor too large for 32-bit tables");
    *ints = 3 * P * *Mp + 2LL * *K + P * *K;
    return RLX_OK;
}

int rlx_conv_input_grad_tables_ints(int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                                    long long *ints_host) {
    RLX_REQUIRE(ints_host, "rlx_conv_input_grad_tables_ints: null pointer");
    int OH, OW, Mp, K;
    return conv_dx_geometry(batch, H, W, C, KH, KW, stride, Co, &OH, &OW, &Mp, &K, ints_host);
}

int rlx_conv_input_grad_tables(int *tables, int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                               void *stream) {
    RLX_REQUIRE(tables, "rlx_conv_input_grad_tables: null pointer");
    int OH, OW, Mp, K;
    long long ints;
    const int rc = conv_dx_geometry(batch, H, W, C, KH, KW, stride, Co, &OH, &OW, &Mp, &K, &ints);
    if (rc != RLX_OK) return rc;
    const int P = stride * stride, M = P * Mp;
    int *rowbase = tables, *yx = tables + M, *crow = tables + 2 * M, *koff_a = tables + 3 * M, *jyx = koff_a + K,
        *koff_b = jyx + K;
    const int n = M > K ? M : K;
    RLX_LAUNCH((conv_dx_tables_kernel), (n + 255) / 256, 256, 0, rlx::as_stream(stream), rowbase, yx, crow, koff_a, jyx, koff_b, batch, H, W, C, KH, KW, stride, Co, OH, OW, Mp);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv_input_grad(const float *dy, const float *weights, float *dx, const float *x_out, int deriv_kind,
                        const int *tables, int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                        int towers, long long dy_tower_stride, long long w_tower_stride, long long dx_tower_stride,
                        void *stream) {
    RLX_REQUIRE(dy && weights && dx && tables, "rlx_conv_input_grad: null pointer");
    RLX_REQUIRE(towers > 0 && deriv_kind >= 0 && deriv_kind <= 2, "rlx_conv_input_grad: bad arguments");
    int OH, OW, Mp, K;
    long long ints;
    const int rc = conv_dx_geometry(batch, H, W, C, KH, KW, stride, Co, &OH, &OW, &Mp, &K, &ints);
    if (rc != RLX_OK) return rc;
    RLX_REQUIRE(aligned16(dy) && aligned16(weights) && aligned16(dx) && (!x_out || aligned16(x_out)) &&
                dy_tower_stride % 4 == 0 && w_tower_stride % 4 == 0 && dx_tower_stride % 4 == 0,
                "rlx_conv_input_grad: operands must be 16-byte aligned");
    const int P = stride * stride, M = P * Mp;
    GemmDev g = {};
    g.xcd_mode = g_xcd_mode;
    g.M = M; g.N = C; g.K = K;
    g.a.base = dy; g.a.tab_o = tables; g.a.tab_r = tables + 3 * M; g.a.batch_stride = dy_tower_stride; g.a.vec_ok = 1;
    g.b.base = weights; g.b.stride_o = Co; g.b.stride_r = 1; g.b.batch_stride = w_tower_stride; g.b.vec_ok = 1;
    g.win_row_yx = tables + M; g.win_c_row = tables + 2 * M;
    g.win_k_jyx = tables + 3 * M + K; g.win_b_koff = tables + 3 * M + 2 * K;
    g.win_oh = OH; g.win_ow = OW; g.win_rows_per_phase = Mp;
    g.c = dx; g.ldc = C; g.c_batch_stride = dx_tower_stride;
    g.aux = deriv_kind != RLX_ACT_NONE ? x_out : nullptr; g.aux_ld = C; g.aux_batch_stride = dx_tower_stride;
    RLX_REQUIRE(deriv_kind == RLX_ACT_NONE || x_out, "rlx_conv_input_grad: the activation derivative needs x_out");
    g.deriv = deriv_kind; g.act = RLX_ACT_NONE;
    g.inner = 1 << 30; g.splits = 1; g.kchunk = K; g.vec_epi = 1; g.a_div = 1.f;
    hipStream_t s = rlx::as_stream(stream);
    if (C <= 32) {
        RLX_LAUNCH((gemm_win_kernel<128, 32, 1>), dim3((C + 31) / 32, M / 128, towers), kThreads, 0, s, g);
    } else if ((long long)(M / 64) * ((C + 63) / 64) * towers >= 192) {
        RLX_LAUNCH((gemm_win_kernel<64, 64, 1>), dim3((C + 63) / 64, M / 64, towers), kThreads, 0, s, g);
    } else {
        RLX_LAUNCH((gemm_win_kernel<32, 64, 2>), dim3((C + 63) / 64, M / 32, towers), kThreads, 0, s, g);
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_col2im(const float *dcol, float *dx, const float *x_out, int deriv_kind, int batch, int H,
               int W, int C, int KH, int KW, int stride, void *stream) {
    RLX_REQUIRE(dcol && dx, "rlx_col2im: null pointer");
    RLX_REQUIRE(batch > 0 && H >= KH && W >= KW && C > 0 && stride > 0, "rlx_col2im: bad geometry");
    const int OH = (H - KH) / stride + 1, OW = (W - KW) / stride + 1;
    const long long total = (long long)batch * H * W * C;
    RLX_REQUIRE(total < (1LL << 31) && (long long)batch * OH * OW * KH * KW * C < (1LL << 40),
                "rlx_col2im: tensor too large for 32-bit indexing");
    const bool vec = C % 4 == 0 && (((uintptr_t)dcol | (uintptr_t)dx | (uintptr_t)x_out) & 15) == 0;
    if (vec)
        RLX_LAUNCH((col2im_kernel<4>), rlx::grid_for(total / 4, 256, 8192), 256, 0, rlx::as_stream(stream), dcol, dx, x_out, deriv_kind, batch, H, W, C, KH, KW, stride, OH, OW);
    else
        RLX_LAUNCH((col2im_kernel<1>), rlx::grid_for(total, 256, 8192), 256, 0, rlx::as_stream(stream), dcol, dx, x_out, deriv_kind, batch, H, W, C, KH, KW, stride, OH, OW);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
