// The input-gradient chain of the Atari torso's third and second convolution as ONE launch: both column matrices stay in LDS.
//
// Replaces, in the backward pass of the image embedder (architectures/embedder_parameters.py "Medium": conv 64 x 3 x 3 / 1
// on 9 x 9 x 64 behind conv 64 x 4 x 4 / 2 on 20 x 20 x 32; tf.gradients through tf.layers.conv2d,
// tensorflow_components/layers.py:108-121 + architecture.py:312-385), for each of the two layers
//     dcol = dz W^T                      (rlx_gemm, one half of the layer's dW + dcol pair launch)
//     dx   = col2im(dcol) * act'(x)      (rlx_col2im)
// — 14.5 + 21 MB of column matrix written by one launch and read back by the next, per minibatch update.  Here one
// workgroup owns HALF AN IMAGE of one tower, cut by rows of the conv1 activation whose gradient it produces:
//
//   dz3 rows [0, 5) / [2, 7) (35 positions x 64)   -> LDS
//   dcol3 = dz3 W3^T (35 x 576, K = 64)            -> LDS
//   dz2 rows [0, 5) / [4, 9) = gather(dcol3) * act'(y2) (45 x 64) -> LDS, and the rows the half owns ([0, 4) / [4, 9)) -> memory
//   dcol2 = dz2 W2^T (45 x 512, K = 64)            -> the same LDS
//   dz1 rows [0, 10) / [10, 20) = gather(dcol2) * act'(y1) (200 x 32) -> memory
//
// (two of seven dz3 rows and one of nine dz2 rows are used by both halves: 43 % / 11 % of the two products are computed
// twice.)  The weight-gradient products of the two layers then run as launches of their own: they read dz3 / dz2 from memory.
//
// ARITHMETIC: bit-identical to the launches it replaces where rlx_gemm runs the dcol products on the LDS-DMA ring without a
// K split (rlx_gemm_describe: what coach_amd/nn/graph.py asks before taking this launch).  Every dcol element is one fp32
// v_mfma_f32_32x32x2_f32 chain over K = 64: two slabs of 32, inside a slab step t multiplying k = 8 (t / 4) + t % 4
// (half-wave 0) and that + 4 (half-wave 1) — the same instruction in the same order here.  The gathers add the taps of an
// input position in col2im_kernel's order ((ky, kx) ascending, from 0.f) and multiply by act_deriv of the stored
// activation as it does.
//
// Work split: 32 x 32 tiles of dcol, job j = (column tile j / 2, row tile j % 2) on wave j % 8 — a wave keeps ONE row tile,
// whose A operands (its dz rows, 32 values per lane) live in registers for the whole product; the B operands of a job are
// 8 float4 per lane read STRAIGHT from the weight matrix (a dcol column is a weight row: 64 contiguous k), requested one job
// ahead; no LDS staging of the weights, no barrier inside a product.  (A first version on 16 x 16 blocks with
// v_mfma_f32_16x16x4_f32 was bit-identical too and twice as slow — it ran ONE dependent chain per wave: 44 cycles per
// instruction instead of the 32 the instruction issues at with two independent chains (box.mfma_cycles_per_instruction,
// rlx_probe_mfma: the same flop rate as 32 x 32 x 2 then).  Two chains per wave in THIS kernel measured slower again —
// 35.9 against 32.8 us, profiles/r06_ab_conv32_pairs.txt: a SIMD already interleaves the chains of its two waves, and the B
// operands come straight from memory, so a second chain only adds registers.)
// LDS 115 KB: column matrix 93 KB, dz3 9.5 KB, dz2 12 KB.
// Bound: MFMA issue of one CU — 36 + 32 jobs of 32 MFMAs on 4 SIMDs = 34.8 k cycles = 16 us — plus two gather passes.
#include "rlx_common.hpp"
#include <type_traits>
// the prioritized replay's priority update as a rider (see conv32_input_grad_kernel's RIDER): glibc's pow with every fusion
// explicit — no contraction inside these two headers, whatever this file is compiled with
#pragma clang fp contract(off)
#include "libm_pow.hpp"
#include "per_update_body.hpp"
#pragma clang fp contract(fast)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int H1 = 20, W1 = 20, C1 = 32;             // conv1 activation = conv2 input
constexpr int K2 = 4, S2 = 2, C2 = 64, O2 = 9;       // conv2
constexpr int K3 = 3, C3 = 64, O3 = 7;               // conv3
constexpr int N3 = K3 * K3 * C2, N2 = K2 * K2 * C1;  // columns of the two column matrices: 576, 512
constexpr int LD3 = N3 + 4, LD2 = N2 + 4, PZ = 64 + 4;
constexpr int R3 = 5, R2 = 5, R1 = 10;               // rows of dz3 / dz2 / dz1 a half works on
constexpr int NP3 = R3 * O3, NP2 = R2 * O2, NP1 = R1 * W1;      // 35, 45, 200 positions
constexpr int kDcolFloats = NP2 * LD2 > NP3 * LD3 ? NP2 * LD2 : NP3 * LD3;
constexpr int kSmemFloats = kDcolFloats + NP3 * PZ + NP2 * PZ;
static_assert(C3 == 64 && C2 == 64, "K = 64 in both products");
static_assert(NP3 <= 64 && NP2 <= 64, "two row tiles of 32");
static_assert(kSmemFloats * 4 <= 160 * 1024, "LDS");
static_assert((kDcolFloats % 4) == 0 && ((NP3 * PZ) % 4) == 0, "16-byte alignment of the LDS regions");

struct ConvBwdArgs {
    const float *dz3; long long dz3_ts;              // [T][B * 49][64]  gradient at conv3's pre-activation
    const float *w3; long long w3_ts;                // [T][576][64]
    const float *y2; long long y2_ts;                // [T][B * 81][64]  conv2's activation
    float *dz2; long long dz2_ts;                    // [T][B * 81][64]  out
    const float *w2; long long w2_ts;                // [T][512][64]
    const float *y1; long long y1_ts;                // [T][B * 400][32] conv1's activation
    float *dz1; long long dz1_ts;                    // [T][B * 400][32] out
    int B, T, act;
    unsigned long long *stamps;
};
struct PerRider {                                    // rlx_per_update's arguments (n <= 64 leaves)
    double *sum, *mn, *mx, *max_priority;
    const int *idx;
    const double *err;
    int *status;
    double alpha, eps;
    int cap, levels, n;
};

__device__ __forceinline__ float act_deriv(float y, int kind) {      // gemm.hip's: through the activation's OUTPUT
    if (kind == RLX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (kind == RLX_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// TAIL16: the second row tile of both products (positions 32 .. 34 of 35, 32 .. 44 of 45) as a 16-row tile on
// v_mfma_f32_16x16x4_f32 — a tail job is 16 rows x 32 columns = TWO independent 16 x 16 accumulator chains (the two column
// halves), which is what lets that instruction issue every 32 cycles: 32 of them = 1024 cycles per job against 2048 for the
// padded 32 x 32 tile.  Waves 0 .. 3 then take the full row tile, waves 4 .. 7 the tail (one of each per SIMD), column
// tiles w % 4 + 4 i.  A tail element sums its k in the order 16 g + j (lane group g of step j): rows >= 32 of a column
// matrix differ from the 32 x 32 x 2 chain in the last bits.
// PF: how many jobs ahead a wave requests its B operands (PF + 1 register sets of 8 float4 in a ring): 1 = the original; a
// tail job is half as long as a full one, so one job ahead no longer covers the load (rlx_conv32_prefetch).
// RIDER (the DQN update: B = 32, one tower -> 64 workgroups on 256 CUs): one MORE workgroup, the last of the grid, runs
// PrioritizedExperienceReplay.update_priorities (prioritized_experience_replay.py:203-217) of the batch's <= 64 leaves on its
// first wave — a 10 us chain of dependent round trips that was a launch of its own behind every update.  Its inputs (the TD
// errors) exist since the head's loss; nothing reads the trees before the next sample(); a CU that this launch leaves idle
// takes it.  An instantiation of its own: the kernel the Clipped-PPO update runs is compiled without it.
template <bool TAIL16, int PF, bool RIDER = false>
__global__ void __launch_bounds__(kThreads, 2) conv32_input_grad_kernel(const ConvBwdArgs a, const PerRider rider) {
    if (RIDER && (int)blockIdx.x == 2 * a.B * a.T) {
        if (threadIdx.x < 64)
            rlx_per::per_update_paths_body<true, false>(rider.sum, rider.mn, rider.mx, rider.cap, rider.levels, rider.idx,
                                                        rider.err, nullptr, nullptr, rider.n, 0, rider.alpha, rider.eps,
                                                        rider.max_priority, 0, rider.status);
        return;
    }
    __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];
    float *const dcol = smem;
    float *const dz3s = dcol + kDcolFloats;
    float *const dz2s = dz3s + NP3 * PZ;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int half = blockIdx.x & 1;
    const int img = (blockIdx.x >> 1) % a.B, t = (blockIdx.x >> 1) / a.B;
    const int z0 = half ? 2 : 0;                     // first dz3 row of this half
    const int x20 = half ? 4 : 0;                    // first dz2 row
    const int own2_lo = half ? 4 : 0, own2_hi = half ? 9 : 4;
    const int x10 = half ? 10 : 0;                   // first dz1 row

    unsigned long long *const stamp = a.stamps && tid == 0 ? a.stamps + 8 * (size_t)blockIdx.x : nullptr;
    if (stamp) stamp[0] = wall_clock64();
    const float *const w3 = a.w3 + (size_t)t * a.w3_ts, *const w2 = a.w2 + (size_t)t * a.w2_ts;
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    };
    // B operands of one job: column n = 32 ct + lane % 32 of dcol = row n of the weight matrix, k = 8 q + 4 hi + i of slab
    // s -> float4 number 8 s + 2 q + hi of that row.  A tail job (TAIL16, rt == 1): columns 32 ct + 16 h + lane % 16 for
    // the two chains h, k = 16 g + 4 c + 0 .. 3 -> float4 number 4 g + c of the row, kept as bq[4 h + c].
    const int rt = TAIL16 ? w >> 2 : w & 1;          // this wave's row tile
    const int c0 = TAIL16 ? w & 3 : w >> 1;          // its column tiles: c0 + 4 i
    const int l15 = lane & 15, g4 = lane >> 4;
    auto load_b = [&](const float *wmat, const int ct, float4 (&bq)[8]) {
        if (TAIL16 && rt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 *row = reinterpret_cast<const float4 *>(wmat + (size_t)(ct * 32 + 16 * h + l15) * 64) + 4 * g4;
#pragma unroll
                for (int c = 0; c < 4; ++c) bq[4 * h + c] = row[c];
            }
            return;
        }
        const float4 *row = reinterpret_cast<const float4 *>(wmat + (size_t)(ct * 32 + l31) * 64) + hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) bq[c] = row[2 * c];
    };
    constexpr int kSets = PF + 1;
    float4 bq[kSets][8];
    float4 y2v[2], y1v[4];
    const float *const y2 = a.y2 + (size_t)t * a.y2_ts + ((size_t)img * (O2 * O2) + (size_t)x20 * O2) * C2;
    const float *const y1 = a.y1 + (size_t)t * a.y1_ts + ((size_t)img * (H1 * W1) + (size_t)x10 * W1) * C1;
    const float4 *const src3 = reinterpret_cast<const float4 *>(a.dz3 + (size_t)t * a.dz3_ts +
                                                                ((size_t)img * (O3 * O3) + (size_t)z0 * O3) * C3);
    if constexpr (TAIL16) {
        // request order = need order, every request unconditional (clamped): dz3's rows first — the first product waits for
        // nothing else —, then the first job's weight rows, then the activations the gathers multiply by (needed 10 us
        // later).  Loads return in order: with the activations in front, the staging of dz3 waited for 48 KB of cold
        // activation lines, and its wait was a vmcnt(0).
        static_assert(NP3 * 16 > kThreads && NP3 * 16 <= 2 * kThreads, "two dz3 requests per lane");
        const float4 d0 = src3[tid], d1 = src3[min(tid + kThreads, NP3 * 16 - 1)];
#pragma unroll
        for (int p = 0; p < PF; ++p) load_b(w3, c0 + 4 * p, bq[p]);
#pragma unroll
        for (int it = 0; it < 2; ++it) y2v[it] = *reinterpret_cast<const float4 *>(y2 + (size_t)min(it * kThreads + tid, NP2 * 16 - 1) * 4);
#pragma unroll
        for (int it = 0; it < 4; ++it) y1v[it] = *reinterpret_cast<const float4 *>(y1 + (size_t)min(it * kThreads + tid, NP1 * 8 - 1) * 4);
        *reinterpret_cast<float4 *>(dz3s + (tid >> 4) * PZ + (tid & 15) * 4) = d0;
        if (tid + kThreads < NP3 * 16)
            *reinterpret_cast<float4 *>(dz3s + ((tid + kThreads) >> 4) * PZ + (tid & 15) * 4) = d1;
    } else {
#pragma unroll
        for (int p = 0; p < PF; ++p) load_b(w3, c0 + 4 * p, bq[p]);      // (the first PF jobs of this wave: 18 column tiles >= c0 + 4 PF)
        // the activations both gathers multiply by, requested now: this lane's items of gather 3 (2) and of gather 2 (4)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * kThreads + tid;
            y2v[it] = item < NP2 * 16 ? *reinterpret_cast<const float4 *>(y2 + (size_t)item * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * kThreads + tid;
            y1v[it] = item < NP1 * 8 ? *reinterpret_cast<const float4 *>(y1 + (size_t)item * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // dz3 rows [z0, z0 + 5) of this image: 35 x 64 contiguous floats -> dz3s[position][PZ]
        for (int f = tid; f < NP3 * 16; f += kThreads)
            *reinterpret_cast<float4 *>(dz3s + (f >> 4) * PZ + (f & 15) * 4) = src3[f];
    }

    // one product: dcol[np x 32 ncols] = dz (LDS, np x 64) wmat^T; bq[0] holds the B operands of this wave's first job;
    // next_w / next_first: where to prefetch from after the last job (the next product's first job), or null
    auto product = [&](const float *dzs, const int np, const int ld, const float *wmat, const int nct,
                       const float *next_w, const int stamp_i) {
        lds_barrier();                               // dzs is complete (and the previous gather's reads of dcol are done)
        if (stamp) stamp[stamp_i] = wall_clock64();
        float av[32];
        if (TAIL16 && rt) {                          // rows 32 + lane % 16, k = 16 g + 0 .. 15
            const float *ar = dzs + min(32 + l15, np - 1) * PZ + 16 * g4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 v = *reinterpret_cast<const float4 *>(ar + 4 * c);
                av[4 * c] = v.x; av[4 * c + 1] = v.y; av[4 * c + 2] = v.z; av[4 * c + 3] = v.w;
            }
        } else {
            const float *ar = dzs + min(rt * 32 + l31, np - 1) * PZ + 4 * hi;
#pragma unroll
            for (int c = 0; c < 8; ++c) {            // float4 number 2 c + hi of the row: k = 8 c + 4 hi + 0 .. 3
                const float4 v = *reinterpret_cast<const float4 *>(ar + 8 * c);
                av[4 * c] = v.x; av[4 * c + 1] = v.y; av[4 * c + 2] = v.z; av[4 * c + 3] = v.w;
            }
        }
        auto comp = [](const float4 &b4, const int i) { return i == 0 ? b4.x : i == 1 ? b4.y : i == 2 ? b4.z : b4.w; };
        const int nj = (nct - c0 + 3) >> 2;          // jobs of this wave in this product
        auto do_job = [&](auto cur, const int ct, const int idx) {
            constexpr int c = decltype(cur)::value;
            constexpr int cn = (c + PF) % kSets;     // the set job idx + PF goes to
            if constexpr (TAIL16) {
                // ALWAYS issued (past the last job of the last product: the current rows once more, never used).  With the
                // request under a branch the compiler's s_waitcnt insertion has to assume it may not have been issued and
                // waits for all but the newest two loads by the end of a job (vmcnt(9) .. vmcnt(2) in front of the eight MFMA
                // groups): the prefetch then hides under a quarter of a job instead of a whole one and a K loop takes MFMA
                // time PLUS load time — measured by taking either out: 11.4 = 8.3 + 5.2 - 2 us (profiles/r06_conv32_tail16.txt).
                // Unconditional: vmcnt(16) .. vmcnt(10), the loads have a whole job to land.
                const bool here = idx + PF < nj;
                const float *const src = here || !next_w ? wmat : next_w;
                const int sct = here ? ct + 4 * PF : (next_w ? c0 + 4 * (idx + PF - nj) : ct);
                load_b(src, sct, bq[cn]);
            } else {                                 // (the bit-identical form as it was measured in rounds 5 / 6)
                if (idx + PF < nj) load_b(wmat, ct + 4 * PF, bq[cn]);
                else if (next_w) load_b(next_w, c0 + 4 * (idx + PF - nj), bq[cn]);
            }
            if (TAIL16 && rt) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], comp(bq[c][i >> 2], i & 3), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], comp(bq[c][4 + (i >> 2)], i & 3), acc1, 0, 0, 0);
                }
                float *dst = dcol + ct * 32 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 32 + 4 * g4 + r;
                    if (row < np) {
                        dst[row * ld] = acc0[r];
                        dst[row * ld + 16] = acc1[r];
                    }
                }
                return;
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], comp(bq[c][i >> 2], i & 3), acc, 0, 0, 0);
            float *dst = dcol + ct * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < np) dst[row * ld] = acc[r];
            }
        };
        // the wave's jobs in turn through the register sets 0, 1 (, 2), ...: job idx in set idx % kSets (compile-time indices)
        for (int idx = 0, ct = c0; ct < nct;) {
            do_job(std::integral_constant<int, 0>(), ct, idx);
            ct += 4; ++idx;
            if (ct >= nct) break;
            do_job(std::integral_constant<int, 1>(), ct, idx);
            ct += 4; ++idx;
            if constexpr (kSets > 2) {
                if (ct >= nct) break;
                do_job(std::integral_constant<int, kSets - 1>(), ct, idx);
                ct += 4; ++idx;
            }
        }
        if (next_w) {                                // the next product's first PF jobs sit in sets nj % kSets, ...: to 0, 1
            const int r = nj % kSets;
            if (kSets == 2 && r == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) bq[0][c] = bq[1][c];
            }
            if (kSets == 3 && r == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { bq[0][c] = bq[1][c]; bq[1][c] = bq[kSets - 1][c]; }
            }
            if (kSets == 3 && r == 2) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float4 t = bq[0][c]; bq[0][c] = bq[kSets - 1][c]; bq[1][c] = t; }
            }
        }
        if (stamp) stamp[stamp_i + 1] = wall_clock64();
        lds_barrier();                               // dcol complete
    };

    // ---- dcol3 = dz3 W3^T
    product(dz3s, NP3, LD3, w3, N3 / 32, w2, 1);
    // ---- dz2 = col2im(dcol3) * act'(y2): item = (position of dz2 rows [x20, x20 + 5), 4 channels)
    {
        float *const dz2 = a.dz2 + (size_t)t * a.dz2_ts + ((size_t)img * (O2 * O2) + (size_t)x20 * O2) * C2;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * kThreads + tid;
            if (item >= NP2 * 16) continue;
            const int pl = item >> 4, c4 = (item & 15) * 4;
            const int iy = x20 + pl / O2, ix = pl % O2;
            // all nine taps requested at once (a tap that does not exist: a valid address, its value not added), then added in
            // col2im's order — a chain of dependent LDS reads under branches before (the gather pass was latency, not work)
            float4 tv[K3 * K3];
            bool tok[K3 * K3];
#pragma unroll
            for (int ky = 0; ky < K3; ++ky)
#pragma unroll
                for (int kx = 0; kx < K3; ++kx) {
                    const int oy = iy - ky, ox = ix - kx;
                    const bool ok = iy >= ky && oy < O3 && ix >= kx && ox < O3;
                    tok[ky * K3 + kx] = ok;
                    tv[ky * K3 + kx] = *reinterpret_cast<const float4 *>(dcol + (ok ? (oy - z0) * O3 + ox : 0) * LD3 + (ky * K3 + kx) * C2 + c4);
                }
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < K3 * K3; ++q)
                if (tok[q]) { s.x += tv[q].x; s.y += tv[q].y; s.z += tv[q].z; s.w += tv[q].w; }
            s.x *= act_deriv(y2v[it].x, a.act); s.y *= act_deriv(y2v[it].y, a.act);
            s.z *= act_deriv(y2v[it].z, a.act); s.w *= act_deriv(y2v[it].w, a.act);
            *reinterpret_cast<float4 *>(dz2s + pl * PZ + c4) = s;
            if (iy >= own2_lo && iy < own2_hi) *reinterpret_cast<float4 *>(dz2 + (size_t)item * 4) = s;
        }
    }
    // ---- dcol2 = dz2 W2^T (its first barrier publishes dz2s and closes the gather's reads of dcol)
    product(dz2s, NP2, LD2, w2, N2 / 32, nullptr, 3);
    // ---- dz1 = col2im(dcol2) * act'(y1): item = (position of dz1 rows [x10, x10 + 10), 4 channels)
    {
        float *const dz1 = a.dz1 + (size_t)t * a.dz1_ts + ((size_t)img * (H1 * W1) + (size_t)x10 * W1) * C1;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * kThreads + tid;
            if (item >= NP1 * 8) continue;
            const int pl = item >> 3, c4 = (item & 7) * 4;
            const int iy = x10 + pl / W1, ix = pl % W1;
            // the (K2 / S2)^2 = 4 taps of this position's phase, requested at once, added in col2im's order
            constexpr int kT = K2 / S2;
            float4 tv[kT * kT];
            bool tok[kT * kT];
#pragma unroll
            for (int a2 = 0; a2 < kT; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < kT; ++b2) {
                    const int ky = iy % S2 + a2 * S2, kx = ix % S2 + b2 * S2;
                    const int oy = (iy - ky) / S2, ox = (ix - kx) / S2;
                    const bool ok = iy >= ky && oy < O2 && ix >= kx && ox < O2;
                    tok[a2 * kT + b2] = ok;
                    tv[a2 * kT + b2] = *reinterpret_cast<const float4 *>(dcol + (ok ? (oy - x20) * O2 + ox : 0) * LD2 + (ky * K2 + kx) * C1 + c4);
                }
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < kT * kT; ++q)
                if (tok[q]) { s.x += tv[q].x; s.y += tv[q].y; s.z += tv[q].z; s.w += tv[q].w; }
            s.x *= act_deriv(y1v[it].x, a.act); s.y *= act_deriv(y1v[it].y, a.act);
            s.z *= act_deriv(y1v[it].z, a.act); s.w *= act_deriv(y1v[it].w, a.act);
            *reinterpret_cast<float4 *>(dz1 + (size_t)item * 4) = s;
        }
    }
    if (stamp) { __syncthreads(); stamp[5] = wall_clock64(); }
}

unsigned long long *g_stamps = nullptr;
int g_tail16 = 1;           // rlx_conv32_tail_tiles
int g_prefetch = 1;         // rlx_conv32_prefetch

}  // namespace

extern "C" {

int rlx_conv32_input_grad_supported(int H, int W, int C, int k2, int s2, int c2, int k3, int s3, int c3) {
    return H == H1 && W == W1 && C == C1 && k2 == K2 && s2 == S2 && c2 == C2 && k3 == K3 && s3 == 1 && c3 == C3;
}

static int conv32_launch(const float *dz3, long long dz3_tower_stride, const float *w3, long long w3_tower_stride,
                         const float *y2, long long y2_tower_stride, float *dz2, long long dz2_tower_stride,
                         const float *w2, long long w2_tower_stride, const float *y1, long long y1_tower_stride,
                         float *dz1, long long dz1_tower_stride, int batch, int towers, int activation,
                         const rlx_per_update_desc *per, void *stream) {
    RLX_REQUIRE(dz3 && w3 && y2 && dz2 && w2 && y1 && dz1, "rlx_conv32_input_grad: null pointer");
    RLX_REQUIRE(batch >= 1 && towers >= 1 && (long long)batch * towers <= (1 << 20), "rlx_conv32_input_grad: bad batch / towers");
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_conv32_input_grad: unknown activation");
    RLX_REQUIRE((((uintptr_t)dz3 | (uintptr_t)w3 | (uintptr_t)y2 | (uintptr_t)dz2 | (uintptr_t)w2 | (uintptr_t)y1 |
                  (uintptr_t)dz1) & 15) == 0 &&
                    ((dz3_tower_stride | w3_tower_stride | y2_tower_stride | dz2_tower_stride | w2_tower_stride |
                      y1_tower_stride | dz1_tower_stride) & 3) == 0,
                "rlx_conv32_input_grad: operands must be 16-byte aligned");
    ConvBwdArgs a{dz3, dz3_tower_stride, w3, w3_tower_stride, y2, y2_tower_stride, dz2, dz2_tower_stride,
                  w2, w2_tower_stride, y1, y1_tower_stride, dz1, dz1_tower_stride, batch, towers, activation, g_stamps};
    const unsigned grid = 2u * batch * towers;
    hipStream_t s = rlx::as_stream(stream);
    PerRider r{};
    if (per) {
        RLX_REQUIRE(per->sum_tree && per->min_tree && per->max_tree && per->max_priority && per->status && per->idx &&
                    per->td_errors, "rlx_conv32_input_grad_per_update: null pointer in the priority update");
        RLX_REQUIRE(per->capacity > 0 && (per->capacity & (per->capacity - 1)) == 0,
                    "rlx_conv32_input_grad_per_update: capacity %d is not a power of two", per->capacity);
        RLX_REQUIRE(per->n >= 1 && per->n <= 64, "rlx_conv32_input_grad_per_update: 1 .. 64 leaves ride on the launch (got %d)", per->n);
        int levels = 0;
        while ((1 << levels) < per->capacity) ++levels;
        r = PerRider{per->sum_tree, per->min_tree, per->max_tree, per->max_priority, per->idx, per->td_errors, per->status,
                     per->alpha, per->epsilon, per->capacity, levels, per->n};
        if (g_tail16) RLX_LAUNCH((conv32_input_grad_kernel<true, 1, true>), grid + 1, kThreads, 0, s, a, r);
        else RLX_LAUNCH((conv32_input_grad_kernel<false, 1, true>), grid + 1, kThreads, 0, s, a, r);
    }
    else if (g_tail16 && g_prefetch == 2) RLX_LAUNCH((conv32_input_grad_kernel<true, 2>), grid, kThreads, 0, s, a, r);
    else if (g_tail16) RLX_LAUNCH((conv32_input_grad_kernel<true, 1>), grid, kThreads, 0, s, a, r);
    else if (g_prefetch == 2) RLX_LAUNCH((conv32_input_grad_kernel<false, 2>), grid, kThreads, 0, s, a, r);
    else RLX_LAUNCH((conv32_input_grad_kernel<false, 1>), grid, kThreads, 0, s, a, r);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv32_input_grad(const float *dz3, long long dz3_tower_stride, const float *w3, long long w3_tower_stride,
                          const float *y2, long long y2_tower_stride, float *dz2, long long dz2_tower_stride,
                          const float *w2, long long w2_tower_stride, const float *y1, long long y1_tower_stride,
                          float *dz1, long long dz1_tower_stride, int batch, int towers, int activation, void *stream) {
    return conv32_launch(dz3, dz3_tower_stride, w3, w3_tower_stride, y2, y2_tower_stride, dz2, dz2_tower_stride, w2,
                         w2_tower_stride, y1, y1_tower_stride, dz1, dz1_tower_stride, batch, towers, activation, nullptr, stream);
}

int rlx_conv32_input_grad_per_update(const float *dz3, long long dz3_tower_stride, const float *w3, long long w3_tower_stride,
                                     const float *y2, long long y2_tower_stride, float *dz2, long long dz2_tower_stride,
                                     const float *w2, long long w2_tower_stride, const float *y1, long long y1_tower_stride,
                                     float *dz1, long long dz1_tower_stride, int batch, int towers, int activation,
                                     const rlx_per_update_desc *per, void *stream) {
    RLX_REQUIRE(per != nullptr, "rlx_conv32_input_grad_per_update: null priority update");
    return conv32_launch(dz3, dz3_tower_stride, w3, w3_tower_stride, y2, y2_tower_stride, dz2, dz2_tower_stride, w2,
                         w2_tower_stride, y1, y1_tower_stride, dz1, dz1_tower_stride, batch, towers, activation, per, stream);
}

int rlx_conv32_tail_tiles(int sixteen_rows) {
    g_tail16 = sixteen_rows ? 1 : 0;
    return RLX_OK;
}

int rlx_conv32_prefetch(int jobs_ahead) {
    RLX_REQUIRE(jobs_ahead == 1 || jobs_ahead == 2, "rlx_conv32_prefetch: 1 or 2 jobs ahead");
    g_prefetch = jobs_ahead;
    return RLX_OK;
}

int rlx_conv32_debug_stamps(void *buffer) {
    g_stamps = static_cast<unsigned long long *>(buffer);
    return RLX_OK;
}

}  // extern "C"
