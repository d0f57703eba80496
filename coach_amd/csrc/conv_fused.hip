// conv2 -> conv3 of the Atari torso as ONE launch: the second convolution's output stays in LDS and feeds the third.
//
// Replaces, for the image embedder of rl_coach (architectures/embedder_parameters.py "Medium": conv 32 x 8 x 8 / 4,
// 64 x 4 x 4 / 2, 64 x 3 x 3 / 1; tensorflow_components/layers.py:108-121 tf.layers.conv2d VALID NHWC), the forward
// products of its second and third layer — two launches of the tiled implicit-im2col GEMM (gemm.hip), 15.7 us each in the
// Clipped-PPO minibatch update against 2.2 / 1.5 us of matrix work at the chip's fp32 MFMA rate.  What those launches
// wait for is operand delivery from beyond the XCD's L2: every workgroup pulls the whole
// weight matrix AND its slice of an activation tensor the previous launch has just written on other XCDs.  Here one
// workgroup owns HALF AN IMAGE of one tower end to end:
//
//   conv1 activations of its rows (contiguous in NHWC: 12 or 14 rows x 20 x 32 fp32, 31-36 KB)  -> LDS, once
//   conv2 on those rows: 45 / 54 output positions x 64 channels, K = 512           (A from LDS, W2 streamed through a ring)
//   its output -> LDS (and the rows this half OWNS -> global memory: the backward pass reads them)
//   conv3 on that: 21 / 28 positions x 64 channels, K = 576                         (A from LDS, W3 streamed through the ring)
//
// Halves: conv3 rows [0, 3) need conv2 rows [0, 5) need conv1 rows [0, 12); conv3 rows [3, 7) need conv2 rows [3, 9) need
// conv1 rows [6, 20): conv2 rows 3 and 4 are computed by both halves (11 % redundant conv2 work), each half stores the
// rows it owns ([0, 4) / [4, 9)).  2 x images x towers workgroups of 8 waves: 256 for the PPO minibatch (64 images, two
// towers) — one per CU — instead of 324 + 196 on two launches.  The conv2 activations never travel through memory on
// their way to conv3, one launch boundary (and one cold start of every operand) is gone.
//
// ARITHMETIC: bit-identical to the two tiled launches it replaces when those run on 32 x 64 tiles with the K slab split
// over two wave groups (gemm_dma_kernel<32, 64, 2, ...>, what rlx_gemm picks for these products at 64 images x 2 towers):
// per output element the same v_mfma_f32_32x32x2_f32 chain — slabs of 32 k in ascending order, inside a slab the k-quads
// {0, 1} by wave group 0 and {2, 3} by wave group 1, MFMA step t of a quad multiplying k = 8 q + t (half-wave 0) and
// 8 q + 4 + t (half-wave 1) — the two partial tiles added as group 0 + group 1, then bias, then the activation.  The
// host side (coach_amd/nn/graph.py) takes this kernel only where the tiled path would have run that configuration, so a
// network's trajectory does not move by a bit (tests/test_conv_fused.py compares the two paths with torch.equal).
//
// WITH conv1 IN FRONT (rlx_conv123_forward, template parameter SPLIT1 > 0): the same workgroup first computes the conv1
// rows it needs from the uint8 frame rows of its half image (52 / 60 rows x 336 bytes in LDS, bytes -> float through
// gemm.hip's 256-entry table, W1 through the same ring), writes them to the LDS image conv2 reads and the rows it owns to
// memory — in the order of sums of the register-staged tiled kernel that otherwise computes them (one chain over K = 256,
// or three 96-long chunks combined as the split-K reduce launch combines them).  8 / 9 position tiles on 8 waves; the ninth
// as four 16 x 16 blocks on four SIMDs (v_mfma_f32_16x16x4_f32: the same ascending-k chain; one dependent chain per wave issues
// every 44 cycles, two independent ones every 32 — the 32 x 32 x 2 flop rate — rlx_probe_mfma).
// 35.8 us against 21.3 + 19.3 us in the PPO update.
//
// LDS (105 KB, one workgroup per CU; 139 - 156 KB with conv1 in front): conv1 rows [position][32 + 4 pad] (the pad spreads the 16-byte operand reads of
// lanes that are 2 input columns apart over the banks), conv2 output [position][64 + 4], a ring of two 8 KB weight slabs
// filled by global_load_lds_dwordx4 (one 16-byte request per lane per slab, 512 lanes = 32 k x 64 n), 8 wave patches of
// 32 x 33 for the partial-tile exchange and the 16-byte epilogue.
// Bound: MFMA issue of a CU (conv2: 8 wave-jobs of 128 MFMAs on 4 SIMDs = 6.8 us; conv3: 4 wave-jobs of 144 MFMAs =
// 3.8 us) with the weight stream (275 KB per workgroup) underneath.
#include "rlx_common.hpp"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int H1 = 20, W1 = 20, C1 = 32;             // conv1 output = conv2 input
constexpr int K2 = 4, S2 = 2, C2 = 64, O2 = 9;       // conv2: 4 x 4 stride 2 -> 9 x 9 x 64
constexpr int K3 = 3, C3 = 64, O3 = 7;               // conv3: 3 x 3 stride 1 -> 7 x 7 x 64
constexpr int P1 = C1 + 4, P2 = C2 + 4;              // LDS position strides (floats)
constexpr int kIn1Rows = 14, kOut2Rows = 6;
constexpr int kIn1Floats = kIn1Rows * W1 * P1;       // 10 080
constexpr int kOut2Floats = kOut2Rows * O2 * P2;     // 3 672
constexpr int kSlabFloats = 32 * 64;                 // one weight slab: 32 k x 64 output channels
constexpr int kPatchFloats = 32 * 33;
// weight slabs in the ring (kDepth - 1 requests in flight per lane): a template parameter, rlx_conv23_depth() selects
// conv1 inside the launch (SPLIT1 > 0): the uint8 frame rows of this half (up to 60 rows x 84 x 4 bytes) + the byte -> float table
constexpr int H0 = 84, W0 = 84, C0 = 4, K1 = 8, S1 = 4;  // conv1: 8 x 8 stride 4 on 84 x 84 x 4 -> 20 x 20 x 32
constexpr int kRow0Bytes = W0 * C0;                      // 336
constexpr int kIn0Floats = (S1 * kIn1Rows + K1 - S1) * kRow0Bytes / 4;      // 60 rows: 5 040
constexpr int kSlabs1 = K1 * K1 * C0 * C1 / kSlabFloats; // conv1's weights as 4 ring slabs of 64 k x 32 channels
constexpr int smem_floats(int depth, int patches = 8, bool with1 = false) {
    return kIn1Floats + kOut2Floats + depth * kSlabFloats + patches * kPatchFloats + (with1 ? kIn0Floats + 256 : 0);
}
constexpr int kSlabs2 = K2 * K2 * C1 / 32;           // 16
constexpr int kSlabs3 = K3 * K3 * C2 / 32;           // 18
static_assert(S2 * (O2 - 1) + K2 == H1 && O2 - K3 + 1 == O3, "geometry");
static_assert(S1 * (H1 - 1) + K1 == H0 && kRow0Bytes % 16 == 0 && (H0 * kRow0Bytes) % 16 == 0 && kSlabs1 == 4, "conv1 geometry");
static_assert(smem_floats(8, 8) * 4 <= 160 * 1024 && smem_floats(4, 12, true) * 4 <= 160 * 1024, "LDS");

struct ConvPairArgs {
    const float *x1; long long x1_ts;                // [T][B * 400][32]  conv1 activations
    const float *w2; long long w2_ts;                // [T][512][64]
    const float *b2; long long b2_ts;                // [T][64]
    const float *w3; long long w3_ts;                // [T][576][64]
    const float *b3; long long b3_ts;
    float *y2; long long y2_ts;                      // [T][B * 81][64]
    float *y3; long long y3_ts;                      // [T][B * 49][64]
    int B, T, act;
    unsigned long long *stamps;                      // diagnostics (rlx_conv23_debug_stamps): [workgroup][8] 10 ns ticks, or null
    // conv1 in the same launch (rlx_conv123_forward; x1 is then its OUTPUT)
    const unsigned char *x0; long long x0_ts;        // [T or 1][B][84][84][4] uint8 frames (x0_ts = 0: the towers share them)
    const float *w1; long long w1_ts;                // [T][256][32]
    const float *b1; long long b1_ts;
    float a_div;                                     // frame byte / a_div = network input (observation rescaling, 255)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == RLX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RLX_ACT_TANH) return tanhf(v);
    return v;
}

// one 16-byte global -> LDS request per lane; lds_dst: wave-uniform LDS byte address of lane 0's 16 bytes (gemm.hip dma16)
__device__ __forceinline__ void dma16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// kDepth: weight slabs in the ring; S: slabs per sync point (one wait + barrier + S requests per S slabs; kDepth >= 2 S)
// KWG: wave groups per K slab of the TILED launches this one reproduces bit for bit — 2: 32 x 64 tiles, k-quads {0, 1} and
// {2, 3} each in one accumulator, partials added 0 + 1; 4: 32 x 32 tiles, one accumulator per k-quad, added ((0 + 1) + 2) + 3
// (what rlx_gemm picks for one tower of 64 images, or 32 images x 2 networks: acting, the DQN update).  The 8 waves then carry
// 2 quad-accumulators each in conv2 and one each in conv3 (all eight busy).
// SPLIT1 > 0: conv1 runs in front, in the same workgroup, from the uint8 frame rows of this half (rlx_conv123_forward) —
// its 256-long sums in the order of the register-staged tiled kernel that otherwise computes them (gemm_fast_body: slabs of
// 32 k ascending, MFMA step kk multiplying k = kk and kk + 1): SPLIT1 = 1 one chain per element (the folded two-tower
// product of the PPO update, 64 x 64 tiles, no K split); SPLIT1 = 3 three chains over k [0, 96), [96, 192), [192, 256)
// combined as splitk_reduce4_kernel<4> does (what rlx_gemm does on 128 x 32 tiles for 200 of them: acting, the DQN update).
template <int kDepth, int S, int KWG = 2, int SPLIT1 = 0>
__global__ void __launch_bounds__(kThreads, 2) conv23_forward_kernel(const ConvPairArgs a) {
    static_assert(kDepth >= 2 * S && kSlabs2 % S == 0 && kSlabs3 % S == 0, "ring depth / step");
    static_assert(KWG == 2 || KWG == 4, "two or four wave groups per K slab");
    static_assert(SPLIT1 == 0 || ((SPLIT1 == 1 || SPLIT1 == 3) && kDepth == 4 && S == 2), "conv1 in front: one or three K chunks, ring 4 / 2");
    constexpr int kPatches = KWG == 2 ? 8 : 12;
    constexpr int kPre = SPLIT1 ? kSlabs1 : 0;                 // ring slabs in front of conv2's
    __shared__ __attribute__((aligned(1024))) float smem[smem_floats(kDepth, kPatches, SPLIT1 > 0)];     // 105-156 KB static (gfx950: up to 160 KB)
    float *const ring = smem;                                  // (first: the DMA destinations stay 1 KB aligned)
    float *const patches = ring + kDepth * kSlabFloats;
    float *const in1 = patches + kPatches * kPatchFloats;
    float *const out2 = in1 + kIn1Floats;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int half = blockIdx.x & 1;
    const int img = (blockIdx.x >> 1) % a.B, t = (blockIdx.x >> 1) / a.B;
    // rows of this half: conv3 [r3, r3 + n3), conv2 [r2, r2 + n2) (owned: [own_lo, own_hi)), conv1 [r1, r1 + n1)
    const int r3 = half ? 3 : 0, n3 = half ? 4 : 3;
    const int r2 = half ? 3 : 0, n2 = half ? 6 : 5;
    const int r1 = half ? 6 : 0, n1 = half ? 14 : 12;
    const int own_lo = half ? 4 : 0, own_hi = half ? 9 : 4;
    const int np2 = n2 * O2, np3 = n3 * O3;                   // output positions of this half (45 / 54, 21 / 28)

    unsigned long long *const stamp = a.stamps && tid == 0 ? a.stamps + 8 * (size_t)blockIdx.x : nullptr;
    if (stamp) stamp[0] = wall_clock64();
    const float *const w2 = a.w2 + (size_t)t * a.w2_ts, *const w3 = a.w3 + (size_t)t * a.w3_ts;
    // weight slab g (0 .. 33: conv2's 16, then conv3's 18) -> ring buffer g & 1: lane tid moves 4 consecutive output
    // channels of k-row tid / 16
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(
        static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring)) + (unsigned)w * 1024u);
    const int wrow = tid >> 4, wcol = (tid & 15) * 4;
    // Every lane issues exactly ONE request per slab, in slab order, kDepth - 1 slabs ahead; slab g has landed for a lane
    // once all but its newest kDepth - 2 vector-memory operations are complete (they retire in order; other loads /
    // stores of the lane only make the wait longer).  Requests of non-existent slabs (g >= 34) are still issued — they
    // re-read the last slab into a buffer nobody reads again — so that the count is the same in every step.
    // (with conv1 in front: 4 more slabs ahead of them, 64 k x 32 channels each — the same 8 KB, lane tid's 16 bytes at
    // float 4 tid of the slab in all three weight matrices)
    constexpr int kSlabsAll = kPre + kSlabs2 + kSlabs3;
    const float *const w1 = SPLIT1 ? a.w1 + (size_t)t * a.w1_ts : nullptr;
    auto issue = [&](const int g) {
        const int gg = g < kSlabsAll ? g : kSlabsAll - 1;
        const float *src = gg < kPre + kSlabs2 ? w2 + (size_t)((gg - kPre) * 32 + wrow) * C2 + wcol
                                               : w3 + (size_t)((gg - kPre - kSlabs2) * 32 + wrow) * C3 + wcol;
        if constexpr (SPLIT1 > 0) {
            if (gg < kPre) src = w1 + (size_t)gg * kSlabFloats + tid * 4;
        }
        dma16(src, ring_lds + (unsigned)(g % kDepth) * (kSlabFloats * 4u));
    };
    // landed(g) for this lane -> its LDS reads of slab g - 1 are complete -> every wave: slab g may be read, the buffer
    // of slab g - 1 refilled
    // sync point of slab k (one per slab, in slab order): wait for it, let every wave finish its LDS reads of slab k - 1,
    // refill that buffer with slab k + kDepth - 1.  Between two sync points exactly kDepth - 1 requests are outstanding.
    // (S slabs per sync point: before it kDepth - S requests are outstanding, S of them must have landed, S are issued)
    int next_issue = kDepth - S;
    auto sync_point = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDepth - 2 * S) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < S; ++j) issue(next_issue++);
    };
    // a workgroup barrier that publishes LDS writes WITHOUT draining the weight requests in flight (__syncthreads() waits
    // for vmcnt(0) too)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    };
#pragma unroll
    for (int d = 0; d < kDepth - S; ++d) issue(d);

    // both epilogues' biases, requested now (a cold load in front of a tanh chain otherwise): conv2's epilogue is shared
    // by the wave pair (w & 3, (w & 3) + 4) that owns a tile — columns of tile w & 1; conv3's by all 8 waves — tile tid >> 8
    const float4 bias2 = *reinterpret_cast<const float4 *>(a.b2 + (size_t)t * a.b2_ts + (w & 1) * 32 + (lane & 7) * 4);
    const float4 bias3 = *reinterpret_cast<const float4 *>(a.b3 + (size_t)t * a.b3_ts + (tid >> 8) * 32 + (tid & 7) * 4);
    float4 y2v[2];                                             // this lane's share of the conv2 activations it owns
    long long y2off[2] = {-1, -1};                             // (stored at the very end: a store in front of the next
                                                               // slab wait would make that wait wait for its acknowledgement)

    // ---- conv1 activations of rows [r1, r1 + n1) -> in1[position][P1]
    float4 y1v[2][4];                                          // (SPLIT1: the conv1 activations this lane stores at the end; [1][0]: tile 8's)
    unsigned y1mask = 0;
    if constexpr (SPLIT1 == 0) {
        const float4 *src = reinterpret_cast<const float4 *>(a.x1 + (size_t)t * a.x1_ts +
                                                             ((size_t)img * (H1 * W1) + (size_t)r1 * W1) * C1);
        const int n4 = n1 * W1 * (C1 / 4);
        for (int f = tid; f < n4; f += kThreads) {
            const float4 v = src[f];
            *reinterpret_cast<float4 *>(in1 + (f >> 3) * P1 + (f & 7) * 4) = v;
        }
    } else {
        // ---- conv1 from the frame: rows [4 r1, 4 r1 + 4 n1 + 4) of the image as raw bytes in LDS; an output position's
        //      patch row ky is 32 CONTIGUOUS bytes there (8 pixels x 4 stacked frames = k 32 ky .. 32 ky + 31).
        //      9 (8) tiles of 32 positions x 32 channels on the 8 waves: wave w computes tile w.
        unsigned char *const in0 = reinterpret_cast<unsigned char *>(out2 + kOut2Floats);
        float *const lut = reinterpret_cast<float *>(in0) + kIn0Floats;
        const int np1 = n1 * W1;                               // 240 / 280 positions
        const float4 bias1 = *reinterpret_cast<const float4 *>(a.b1 + (size_t)t * a.b1_ts + (lane & 7) * 4);
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.x0 + (size_t)t * a.x0_ts + (size_t)img * (H0 * kRow0Bytes) +
                                                               (size_t)(S1 * r1) * kRow0Bytes);
            const int n16 = (S1 * n1 + K1 - S1) * (kRow0Bytes / 16);
            for (int f = tid; f < n16; f += kThreads) reinterpret_cast<uint4 *>(in0)[f] = src[f];
            if (tid < 256) lut[tid] = (float)tid / a.a_div;    // gemm.hip's table: the same division, once per byte value
        }
        // tile 8 (the lower half's positions 256 .. 279) as four 16 x 16 blocks on waves 0 .. 3 — one per SIMD — with
        // v_mfma_f32_16x16x4_f32 (lane group g = lane / 16 supplies k = 4 j + g of step j; the same ascending-k chain per
        // element): the busiest SIMD issues 2 x 8192 + 2048 cycles of MFMA instead of 3 x 8192
        const bool nine = np1 > 256;
        const int l15 = lane & 15, g4 = lane >> 4;
        int pb[2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int p = min(tl ? 256 + 16 * ((w >> 1) & 1) + l15 : w * 32 + l31, np1 - 1);
            const int oy = p / W1, ox = p - oy * W1;
            pb[tl] = (S1 * oy) * kRow0Bytes + (S1 * ox) * C0;
        }
        f32x16 ca[SPLIT1];
        f32x4 cs[SPLIT1];
#pragma unroll
        for (int c = 0; c < SPLIT1; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ca[c][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[c][r] = 0.f;
        }
        // unit u = kernel row ky = k [32 u, 32 u + 32): 16 MFMA steps per tile, step s multiplying k = 32 u + 2 s + hi
        auto unit = [&](auto U) {
            constexpr int u = decltype(U)::value;
            constexpr int chunk = SPLIT1 == 1 ? 0 : u / 3;     // K chunks of 96 = 3 kernel rows
            const float *bs = ring + ((u >> 1) % kDepth) * kSlabFloats + (u & 1) * 32 * C1;
            {
                const float *bp = bs + hi * C1 + l31;
                float b[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) b[s] = bp[2 * s * C1];
                const uint4 lo = *reinterpret_cast<const uint4 *>(in0 + pb[0] + u * kRow0Bytes);
                const uint4 up = *reinterpret_cast<const uint4 *>(in0 + pb[0] + u * kRow0Bytes + 16);
                const unsigned wd[8] = {lo.x, lo.y, lo.z, lo.w, up.x, up.y, up.z, up.w};
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const unsigned sh = wd[s >> 1] >> (8 * hi);                       // bytes 2 s' + hi of the word
                    const float x = lut[(s & 1) ? (sh >> 16) & 0xffu : sh & 0xffu];
                    ca[chunk] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b[s], ca[chunk], 0, 0, 0);
                }
            }
            if (nine && w < 4) {                               // (wave-uniform)
                const float *bp = bs + g4 * C1 + 16 * (w & 1) + l15;
                const uint4 lo = *reinterpret_cast<const uint4 *>(in0 + pb[1] + u * kRow0Bytes);
                const uint4 up = *reinterpret_cast<const uint4 *>(in0 + pb[1] + u * kRow0Bytes + 16);
                const unsigned wd[8] = {lo.x, lo.y, lo.z, lo.w, up.x, up.y, up.z, up.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = lut[(wd[j] >> (8 * g4)) & 0xffu];                 // byte g of word j: k = 4 j + g
                    cs[chunk] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bp[4 * j * C1], cs[chunk], 0, 0, 0);
                }
            }
        };
        sync_point();                                          // (publishes in0 and the table)
        if (stamp) stamp[6] = wall_clock64();
        unit(std::integral_constant<int, 0>()); unit(std::integral_constant<int, 1>());
        unit(std::integral_constant<int, 2>()); unit(std::integral_constant<int, 3>());
        sync_point();
        unit(std::integral_constant<int, 4>()); unit(std::integral_constant<int, 5>());
        unit(std::integral_constant<int, 6>()); unit(std::integral_constant<int, 7>());
        if (stamp) stamp[7] = wall_clock64();
        // epilogue, wave by wave through its own patch: (chunk sums as the reduce launch adds them) + bias, activation ->
        // in1 for conv2, and the rows this half owns ([0, 10) / [10, 20)) kept for the store at the end
        float *const patch = patches + w * kPatchFloats;
        const int own1_lo = half ? 10 : 0, own1_hi = half ? 20 : 10;
        auto rows = [&](const int tile_index, float4 (&keep)[4], const int mask_shift) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane;
                const int rl = idx >> 3, c4 = (idx & 7) * 4;
                const int pos = tile_index * 32 + rl;
                if (pos >= np1) continue;
                const float *sp = patch + rl * 33 + c4;
                float4 o = make_float4(sp[0], sp[1], sp[2], sp[3]);
                o.x += bias1.x; o.y += bias1.y; o.z += bias1.z; o.w += bias1.w;
                o.x = apply_act(o.x, a.act); o.y = apply_act(o.y, a.act);
                o.z = apply_act(o.z, a.act); o.w = apply_act(o.w, a.act);
                *reinterpret_cast<float4 *>(in1 + pos * P1 + c4) = o;
                const int row = r1 + pos / W1;
                if (row >= own1_lo && row < own1_hi) {
                    keep[it] = o;
                    y1mask |= 1u << (mask_shift + it);
                }
            }
        };
        // splitk_reduce4_kernel<4> on three partials: thread group q sums 0 + partial q (q = 3: nothing), then
        // ((s0 + s1) + s2) + s3
        {
            f32x16 v = ca[0];
            if constexpr (SPLIT1 == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = (((0.f + ca[0][r]) + (0.f + ca[1][r])) + (0.f + ca[2][r])) + 0.f;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = v[r];
            __builtin_amdgcn_wave_barrier();
            rows(w, y1v[0], 0);
        }
        if (nine) {                                            // (workgroup-uniform) tile 8: assembled in patch 0
            lds_barrier();
            if (w < 4) {
                f32x4 v = cs[0];
                if constexpr (SPLIT1 == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (((0.f + cs[0][r]) + (0.f + cs[1][r])) + (0.f + cs[2][r])) + 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)                      // D: row 4 g + r, column lane % 16 of the block
                    patches[(16 * ((w >> 1) & 1) + 4 * g4 + r) * 33 + 16 * (w & 1) + l15] = v[r];
            }
            lds_barrier();
            if (w < 4) {                                       // its 32 x 8 float4: one per lane of waves 0 .. 3
                const int rl = tid >> 3, c4 = (tid & 7) * 4;
                const int pos = 256 + rl;
                if (pos < np1) {
                    const float *sp = patches + rl * 33 + c4;
                    float4 o = make_float4(sp[0], sp[1], sp[2], sp[3]);
                    o.x += bias1.x; o.y += bias1.y; o.z += bias1.z; o.w += bias1.w;
                    o.x = apply_act(o.x, a.act); o.y = apply_act(o.y, a.act);
                    o.z = apply_act(o.z, a.act); o.w = apply_act(o.w, a.act);
                    *reinterpret_cast<float4 *>(in1 + pos * P1 + c4) = o;
                    const int row = r1 + pos / W1;
                    if (row >= own1_lo && row < own1_hi) {
                        y1v[1][0] = o;
                        y1mask |= 1u << 4;
                    }
                }
            }
        }
    }

    // ---- conv2: 2 row tiles x 2 column tiles x 2 wave groups = 8 wave-jobs
    {
        const int wk = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
        const int p = min(wm * 32 + l31, np2 - 1);            // rows past the last position repeat it (never stored)
        const int oy = p / O2, ox = p - oy * O2;
        const float *const arow = in1 + ((2 * oy) * W1 + 2 * ox) * P1 + 4 * hi;
        const int b_col = wn * 32 + l31;
        constexpr int NACC = KWG == 2 ? 1 : 2;                  // KWG 4: the wave's two k-quads sum separately
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
        // The slab step is software-pipelined through registers (as gemm.hip's ring loop): while the MFMA chain of slab g
        // runs, the sync point of slab g + 1 and the LDS reads of its operands happen in the MIDDLE of the chain — with the
        // barrier in front of the reads and the MFMAs behind them, the two waves of a SIMD did everything in lockstep
        // (23 us per launch against 11.7 us of MFMA issue).  The order of the sum is unchanged.
        auto load_ops = [&](const int g, float (&av)[2][4], float (&bv)[2][4]) {
            const int ky = g >> 2, kx = g & 3;
            const float *as = arow + (ky * W1 + kx) * P1;
            const float *bs = ring + ((g + kPre) % kDepth) * kSlabFloats;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = KWG == 2 ? wk * 2 + qq : wk + 2 * qq;      // (KWG 4: quads wk and wk + 2)
                const float4 v = *reinterpret_cast<const float4 *>(as + 8 * q);
                av[qq][0] = v.x; av[qq][1] = v.y; av[qq][2] = v.z; av[qq][3] = v.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[qq][i] = bs[(8 * q + 4 * hi + i) * C2 + b_col];
            }
        };
        // one STEP = S slabs: its 8 S MFMAs in slab order, the sync point + operand reads of the next step in their middle
        float ra[2][S][2][4], rb[2][S][2][4];
        auto load_step = [&](const int p, float (&av)[S][2][4], float (&bv)[S][2][4]) {
#pragma unroll
            for (int j = 0; j < S; ++j) load_ops(p * S + j, av[j], bv[j]);
        };
        auto mfma_half = [&](const float (&av)[S][2][4], const float (&bv)[S][2][4], const int half_) {
#pragma unroll
            for (int m = half_ * 4 * S; m < (half_ + 1) * 4 * S; ++m) {
                const float x = av[m >> 3][(m >> 2) & 1][m & 3], y = bv[m >> 3][(m >> 2) & 1][m & 3];
                if (NACC == 1 || ((m >> 2) & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc1, 0, 0, 0);
            }
        };
        auto do_step = [&](auto cur, const int p, const bool has_next) {
            constexpr int c = decltype(cur)::value;
            mfma_half(ra[c], rb[c], 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                sync_point();
                load_step(p + 1, ra[1 - c], rb[1 - c]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(ra[c], rb[c], 1);
        };
        constexpr int kSteps2 = kSlabs2 / S;
        sync_point();                                              // step 0 (also publishes in1)
        if (stamp) stamp[1] = wall_clock64();
        load_step(0, ra[0], rb[0]);
        for (int p = 0; p < kSteps2; p += 2) {
            do_step(std::integral_constant<int, 0>(), p, p + 1 < kSteps2);
            if (p + 1 < kSteps2) do_step(std::integral_constant<int, 1>(), p + 1, p + 2 < kSteps2);
        }
        if (stamp) stamp[2] = wall_clock64();
        // partial tile of wave group 1 -> its patch; wave group 0 adds it (group 0 + group 1, gemm.hip fast_epilogue)
        float *const patch = patches + w * kPatchFloats;
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[r];
            if (NACC == 2) {                                  // (quad 3's partial: patches 8 .. 11)
                float *const patch3 = patches + (w + 4) * kPatchFloats;
#pragma unroll
                for (int r = 0; r < 16; ++r) patch3[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc1[r];
            }
        }
        lds_barrier();
        if (wk == 0) {
            const float *src = patches + (w + 4) * kPatchFloats;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += src[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31];
            if (NACC == 2) {                                  // ((q0 + q1) + q2) + q3: q2 is this wave's second accumulator
                const float *src3 = patches + (w + 8) * kPatchFloats;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += src3[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31];
            }
            // C / D layout undone through the wave's own patch: a lane then owns 4 consecutive channels of a position
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[r];
        }
        lds_barrier();
        {
            // bias + activation by BOTH waves of the pair that computed the tile (rows it = 0, 1 by wave group 0, it = 2, 3
            // by wave group 1): half the tanh chain per wave
            const float *const tile = patches + (w & 3) * kPatchFloats;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int idx = (2 * wk + h2) * 64 + lane;
                const int rl = idx >> 3, c4 = (idx & 7) * 4;
                const int pos = wm * 32 + rl;
                if (pos >= np2) continue;
                const float *sp = tile + rl * 33 + c4;
                float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                v.x += bias2.x; v.y += bias2.y; v.z += bias2.z; v.w += bias2.w;
                v.x = apply_act(v.x, a.act); v.y = apply_act(v.y, a.act);
                v.z = apply_act(v.z, a.act); v.w = apply_act(v.w, a.act);
                *reinterpret_cast<float4 *>(out2 + pos * P2 + wn * 32 + c4) = v;
                const int row = r2 + pos / O2;                    // conv2 output row of this position
                if (row >= own_lo && row < own_hi) {
                    y2v[h2] = v;
                    y2off[h2] = (long long)(r2 * O2 + pos) * C2 + wn * 32 + c4;
                }
            }
        }
    }

    // ---- conv3: 1 row tile x 2 column tiles x 2 wave groups = 4 wave-jobs (waves 0 .. 3); waves 4 .. 7 keep the weight
    //      stream and the barriers going
    {
        const bool active = KWG == 4 || w < 4;                // (KWG 4: eight wave-jobs of one k-quad each)
        const int wk = KWG == 2 ? (w >> 1) & 1 : w >> 1, wn = w & 1;
        constexpr int NQ3 = KWG == 2 ? 2 : 1;                   // k-quads per wave and slab
        const int p = min(l31, np3 - 1);
        const int oy = p / O3, ox = p - oy * O3;
        const float *const arow = out2 + (oy * O2 + ox) * P2 + 4 * hi;
        const int b_col = wn * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        auto load_ops = [&](const int g, float (&av)[2][4], float (&bv)[2][4]) {
            if (!active) return;
            const int s3 = g - kSlabs2, tap = s3 >> 1;
            const int ky = tap / K3, kx = tap - ky * K3;
            const float *as = arow + (ky * O2 + kx) * P2 + (s3 & 1) * 32;
            const float *bs = ring + ((g + kPre) % kDepth) * kSlabFloats;
#pragma unroll
            for (int qq = 0; qq < NQ3; ++qq) {
                const int q = KWG == 2 ? wk * 2 + qq : wk;
                const float4 v = *reinterpret_cast<const float4 *>(as + 8 * q);
                av[qq][0] = v.x; av[qq][1] = v.y; av[qq][2] = v.z; av[qq][3] = v.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[qq][i] = bs[(8 * q + 4 * hi + i) * C3 + b_col];
            }
        };
        float ra[2][S][2][4], rb[2][S][2][4];
        auto load_step = [&](const int p, float (&av)[S][2][4], float (&bv)[S][2][4]) {
#pragma unroll
            for (int j = 0; j < S; ++j) load_ops(kSlabs2 + p * S + j, av[j], bv[j]);
        };
        // (a step's 4 NQ3 S MFMAs in slab order, quad by quad; first half / second half around the sync point)
        auto mfma_half = [&](const float (&av)[S][2][4], const float (&bv)[S][2][4], const int half_) {
            if (!active) return;
#pragma unroll
            for (int m = half_ * 2 * NQ3 * S; m < (half_ + 1) * 2 * NQ3 * S; ++m) {
                const int j = m / (4 * NQ3), qq = (m / 4) % NQ3, i = m & 3;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][qq][i], bv[j][qq][i], acc, 0, 0, 0);
            }
        };
        auto do_step = [&](auto cur, const int p, const bool has_next) {
            constexpr int c = decltype(cur)::value;
            mfma_half(ra[c], rb[c], 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                sync_point();
                load_step(p + 1, ra[1 - c], rb[1 - c]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(ra[c], rb[c], 1);
        };
        constexpr int kSteps3 = kSlabs3 / S;
        sync_point();                                              // conv3's step 0 (its barrier also publishes out2)
        if (stamp) stamp[3] = wall_clock64();
        load_step(0, ra[0], rb[0]);
        for (int p = 0; p < kSteps3; p += 2) {
            do_step(std::integral_constant<int, 0>(), p, p + 1 < kSteps3);
            if (p + 1 < kSteps3) do_step(std::integral_constant<int, 1>(), p + 1, p + 2 < kSteps3);
        }
        if (stamp) stamp[4] = wall_clock64();
        float *const patch = patches + w * kPatchFloats;
        lds_barrier();                                             // conv2's patches are no longer read
        if (active && wk >= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[r];
        }
        lds_barrier();
        if (active && wk == 0) {
#pragma unroll
            for (int q = 1; q < KWG; ++q) {                   // + group 1 (+ group 2 + group 3), in that order
                const float *src = patches + (w + 2 * q) * kPatchFloats;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += src[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 33 + l31] = acc[r];
        }
        lds_barrier();
        {
            // the two 32 x 32 tiles (patches 0 and 1) are 512 float4: one per lane of the workgroup
            const int tile = tid >> 8, idx = tid & 255;
            const int rl = idx >> 3, c4 = (idx & 7) * 4;
            if (rl < np3) {
                const float *sp = patches + tile * kPatchFloats + rl * 33 + c4;
                float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                v.x += bias3.x; v.y += bias3.y; v.z += bias3.z; v.w += bias3.w;
                v.x = apply_act(v.x, a.act); v.y = apply_act(v.y, a.act);
                v.z = apply_act(v.z, a.act); v.w = apply_act(v.w, a.act);
                float *const y3 = a.y3 + (size_t)t * a.y3_ts + ((size_t)img * (O3 * O3) + (size_t)r3 * O3) * C3;
                *reinterpret_cast<float4 *>(y3 + (size_t)rl * C3 + tile * 32 + c4) = v;
            }
        }
    }
    if constexpr (SPLIT1 > 0) {
        // (x1 is conv1's output here: [T][B * 400][32])
        float *const y1 = const_cast<float *>(a.x1) + (size_t)t * a.x1_ts + ((size_t)img * (H1 * W1) + (size_t)r1 * W1) * C1;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (y1mask & (1u << it)) {
                const int idx = it * 64 + lane;
                *reinterpret_cast<float4 *>(y1 + (size_t)(w * 32 + (idx >> 3)) * C1 + (idx & 7) * 4) = y1v[0][it];
            }
        if (y1mask & (1u << 4))                                // (tile 8: position 256 + tid / 8)
            *reinterpret_cast<float4 *>(y1 + (size_t)(256 + (tid >> 3)) * C1 + (tid & 7) * 4) = y1v[1][0];
    }
    {
        float *const y2 = a.y2 + (size_t)t * a.y2_ts + (size_t)img * (O2 * O2) * C2;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            if (y2off[h2] >= 0) *reinterpret_cast<float4 *>(y2 + y2off[h2]) = y2v[h2];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the requests of non-existent slabs still target the ring)
    if (stamp) { __syncthreads(); stamp[5] = wall_clock64(); }
}

int g_depth = 4, g_step = 2;       // measured on the C2 update: profiles/r05_ab_conv23.txt
unsigned long long *g_stamps = nullptr;

}  // namespace

extern "C" {

int rlx_conv23_forward_supported(int H, int W, int C, int k2, int s2, int c2, int k3, int s3, int c3) {
    return H == H1 && W == W1 && C == C1 && k2 == K2 && s2 == S2 && c2 == C2 && k3 == K3 && s3 == 1 && c3 == C3;
}

int rlx_conv23_forward(const float *x1, long long x1_tower_stride, const float *w2, long long w2_tower_stride,
                       const float *b2, long long b2_tower_stride, const float *w3, long long w3_tower_stride,
                       const float *b3, long long b3_tower_stride, float *y2, long long y2_tower_stride, float *y3,
                       long long y3_tower_stride, int batch, int towers, int activation, int wave_groups, void *stream) {
    RLX_REQUIRE(x1 && w2 && b2 && w3 && b3 && y2 && y3, "rlx_conv23_forward: null pointer");
    RLX_REQUIRE(batch >= 1 && towers >= 1 && (long long)batch * towers <= (1 << 20), "rlx_conv23_forward: bad batch / towers");
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_conv23_forward: unknown activation");
    RLX_REQUIRE(wave_groups == 2 || wave_groups == 4, "rlx_conv23_forward: wave_groups is 2 or 4");
    RLX_REQUIRE((((uintptr_t)x1 | (uintptr_t)w2 | (uintptr_t)b2 | (uintptr_t)w3 | (uintptr_t)b3 | (uintptr_t)y2 |
                  (uintptr_t)y3) & 15) == 0 &&
                    ((x1_tower_stride | w2_tower_stride | b2_tower_stride | w3_tower_stride | b3_tower_stride |
                      y2_tower_stride | y3_tower_stride) & 3) == 0,
                "rlx_conv23_forward: operands must be 16-byte aligned");
    ConvPairArgs a{x1, x1_tower_stride, w2, w2_tower_stride, b2, b2_tower_stride, w3, w3_tower_stride, b3, b3_tower_stride,
                   y2, y2_tower_stride, y3, y3_tower_stride, batch, towers, activation, g_stamps,
                   nullptr, 0, nullptr, 0, nullptr, 0, 1.f};
    const unsigned grid = 2u * batch * towers;
    hipStream_t s = rlx::as_stream(stream);
#define RLX_C23(D, SS) RLX_LAUNCH((conv23_forward_kernel<D, SS>), grid, kThreads, 0, s, a)
    if (wave_groups == 4) RLX_LAUNCH((conv23_forward_kernel<4, 2, 4>), grid, kThreads, 0, s, a);
    else if (g_step == 2) {
        if (g_depth <= 4) RLX_C23(4, 2);
        else if (g_depth == 6) RLX_C23(6, 2);
        else RLX_C23(8, 2);
    } else if (g_depth == 2) RLX_C23(2, 1);
    else if (g_depth == 3) RLX_C23(3, 1);
    else if (g_depth == 6) RLX_C23(6, 1);
    else if (g_depth == 8) RLX_C23(8, 1);
    else RLX_C23(4, 1);
#undef RLX_C23
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv123_forward_supported(int H, int W, int C, int k1, int s1, int c1) {
    return H == H0 && W == W0 && C == C0 && k1 == K1 && s1 == S1 && c1 == C1;
}

int rlx_conv123_forward(const unsigned char *x0, long long x0_tower_stride, float a_div, const float *w1,
                        long long w1_tower_stride, const float *b1, long long b1_tower_stride, float *y1,
                        long long y1_tower_stride, const float *w2, long long w2_tower_stride, const float *b2,
                        long long b2_tower_stride, const float *w3, long long w3_tower_stride, const float *b3,
                        long long b3_tower_stride, float *y2, long long y2_tower_stride, float *y3,
                        long long y3_tower_stride, int batch, int towers, int activation, int wave_groups,
                        int conv1_chunks, void *stream) {
    RLX_REQUIRE(x0 && w1 && b1 && y1 && w2 && b2 && w3 && b3 && y2 && y3, "rlx_conv123_forward: null pointer");
    RLX_REQUIRE(batch >= 1 && towers >= 1 && (long long)batch * towers <= (1 << 20), "rlx_conv123_forward: bad batch / towers");
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_conv123_forward: unknown activation");
    RLX_REQUIRE(wave_groups == 2 || wave_groups == 4, "rlx_conv123_forward: wave_groups is 2 or 4");
    RLX_REQUIRE(conv1_chunks == 1 || conv1_chunks == 3, "rlx_conv123_forward: conv1_chunks is 1 or 3");
    RLX_REQUIRE(a_div != 0.f, "rlx_conv123_forward: a_div must be non-zero");
    RLX_REQUIRE((((uintptr_t)x0 | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)y1 | (uintptr_t)w2 | (uintptr_t)b2 |
                  (uintptr_t)w3 | (uintptr_t)b3 | (uintptr_t)y2 | (uintptr_t)y3) & 15) == 0 &&
                    (x0_tower_stride & 15) == 0 &&
                    ((w1_tower_stride | b1_tower_stride | y1_tower_stride | w2_tower_stride | b2_tower_stride |
                      w3_tower_stride | b3_tower_stride | y2_tower_stride | y3_tower_stride) & 3) == 0,
                "rlx_conv123_forward: operands must be 16-byte aligned");
    ConvPairArgs a{y1, y1_tower_stride, w2, w2_tower_stride, b2, b2_tower_stride, w3, w3_tower_stride, b3, b3_tower_stride,
                   y2, y2_tower_stride, y3, y3_tower_stride, batch, towers, activation, g_stamps,
                   x0, x0_tower_stride, w1, w1_tower_stride, b1, b1_tower_stride, a_div};
    const unsigned grid = 2u * batch * towers;
    hipStream_t s = rlx::as_stream(stream);
    if (wave_groups == 2 && conv1_chunks == 1) RLX_LAUNCH((conv23_forward_kernel<4, 2, 2, 1>), grid, kThreads, 0, s, a);
    else if (wave_groups == 2) RLX_LAUNCH((conv23_forward_kernel<4, 2, 2, 3>), grid, kThreads, 0, s, a);
    else if (conv1_chunks == 1) RLX_LAUNCH((conv23_forward_kernel<4, 2, 4, 1>), grid, kThreads, 0, s, a);
    else RLX_LAUNCH((conv23_forward_kernel<4, 2, 4, 3>), grid, kThreads, 0, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_conv23_debug_stamps(void *buffer) {
    g_stamps = static_cast<unsigned long long *>(buffer);
    return RLX_OK;
}

int rlx_conv23_depth(int depth, int slabs_per_step) {
    RLX_REQUIRE(depth == 2 || depth == 3 || depth == 4 || depth == 6 || depth == 8, "rlx_conv23_depth: 2, 3, 4, 6 or 8");
    RLX_REQUIRE(slabs_per_step == 1 || (slabs_per_step == 2 && depth >= 4), "rlx_conv23_depth: 1 slab per step, or 2 with depth >= 4");
    g_depth = depth;
    g_step = slabs_per_step;
    return RLX_OK;
}

}  // extern "C"
