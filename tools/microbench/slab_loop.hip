// One wave per SIMD running the slab step of the tiled GEMM kernels WITHOUT its global loads: 16 x v_mfma_f32_32x32x2_f32 per
// 32-deep slab with the operands read from LDS (row pitches 33 / 36 as in gemm.hip), optionally re-staging the slab from
// registers (16 + 4 LDS writes per lane) every step.  Shader cycles per slab step, all 256 CUs busy.
//   variant 0: operands read pairwise right before their MFMAs (what the compiler emits for the plain loop)
//   variant 1: all 32 operand values read first, then the 16 MFMAs
//   +2       : with the staging writes of the next slab after / between the MFMAs
//   +4       : with the slab's operands really loaded from global memory (8 x 16 B per lane and slab, rows of 128 B at a 2 KB
//              pitch like an im2col gather; 64 KB per workgroup, so 16 MB in all: L2-resident), requested one slab (+8: two
//              slabs) ahead of their staging
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, LDA = 33, LDB = 36;

template <int VARIANT>
__global__ void __launch_bounds__(256) slab(long long *out, float *sink, int nslab, const float *src) {
    __shared__ float smem[4 * BK * (LDA + LDB)];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    float *as = smem + wid * BK * (LDA + LDB), *bs = as + BK * LDA;
    for (int i = lane; i < BK * (LDA + LDB); i += 64) as[i] = 1e-3f * i;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra[4], rb[4];
    for (int p = 0; p < 4; ++p) { ra[p] = make_float4(lane, p, 1.f, 2.f); rb[p] = make_float4(p, lane, 3.f, 4.f); }
    int a_r[4], a_k[4];
    for (int p = 0; p < 4; ++p) { const int e = lane + 64 * p; a_r[p] = e >> 3; a_k[p] = (e & 7) * 4; }
    const float *mine = src + (size_t)blockIdx.x * 16384 + wid * 4096;           // 16 KB per wave
    auto request = [&](int slab, float4 (&a)[4], float4 (&b)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            a[p] = *reinterpret_cast<const float4 *>(mine + ((a_r[p] * 512 + (slab & 3) * 32 + a_k[p]) & 4095));
            b[p] = *reinterpret_cast<const float4 *>(mine + ((a_r[p] * 32 + (slab & 3) * 1024 + a_k[p] + 2048) & 4095));
        }
    };
    float4 ra2[4], rb2[4];
    if (VARIANT & 4) request(0, ra, rb);
    if (VARIANT & 8) request(1, ra2, rb2);
    __syncthreads();
    const long long c0 = clock64();
    for (int s = 0; s < nslab; ++s) {
        const float *ap = (VARIANT & 16) ? as + l31 * LDB + hi : as + hi * LDA + l31, *bp = bs + hi * LDB + l31;
        constexpr int AS = (VARIANT & 16) ? 1 : LDA;          // distance of consecutive k of one row in LDS
        if (VARIANT & 16) {
            float av[16], bv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { av[q] = ap[2 * q * AS]; bv[q] = bp[2 * q * LDB]; }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        } else if (VARIANT & 1) {
            float av[16], bv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { av[q] = ap[2 * q * LDA]; bv[q] = bp[2 * q * LDB]; }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * q * LDA], bp[2 * q * LDB], acc, 0, 0, 0);
        }
        if (VARIANT & 2) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (VARIANT & 16) {
                    *reinterpret_cast<float4 *>(as + a_r[p] * LDB + a_k[p]) = ra[p];
                } else {
                    float *d = as + a_k[p] * LDA + a_r[p];
                    d[0] = ra[p].x; d[LDA] = ra[p].y; d[2 * LDA] = ra[p].z; d[3 * LDA] = ra[p].w;
                }
                *reinterpret_cast<float4 *>(bs + a_r[p] * LDB + a_k[p]) = rb[p];
                if (!(VARIANT & 4)) { ra[p].x += 1e-6f; rb[p].y += 1e-6f; }
            }
            if (VARIANT & 8) {
#pragma unroll
                for (int p = 0; p < 4; ++p) { ra[p] = ra2[p]; rb[p] = rb2[p]; }
                request(s + 2, ra2, rb2);
            } else if (VARIANT & 4) {
                request(s + 1, ra, rb);
            }
        }
    }
    const long long c1 = clock64();
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 123.456f) sink[0] = t;
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

static float *g_src;
// The software-pipelined step: requests for slab s + 2 first, all operand reads of slab s, then the MFMA chain with the
// staging of slab s + 1 (registers -> LDS) spread over its gaps.  Three register sets, unrolled by three.
__global__ void __launch_bounds__(256) slab_pipelined(long long *out, float *sink, int nslab, const float *src) {
    __shared__ float smem[4 * BK * (LDA + LDB)];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    float *as = smem + wid * BK * (LDA + LDB), *bs = as + BK * LDA;
    for (int i = lane; i < BK * (LDA + LDB); i += 64) as[i] = 1e-3f * i;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int a_r[4], a_k[4];
    for (int p = 0; p < 4; ++p) { const int e = lane + 64 * p; a_r[p] = e >> 3; a_k[p] = (e & 7) * 4; }
    const float *mine = src + (size_t)blockIdx.x * 16384 + wid * 4096;
    float4 ra[3][4], rb[3][4];
    auto request = [&](int slab, float4 (&a)[4], float4 (&b)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            a[p] = *reinterpret_cast<const float4 *>(mine + ((a_r[p] * 512 + (slab & 3) * 32 + a_k[p]) & 4095));
            b[p] = *reinterpret_cast<const float4 *>(mine + ((a_r[p] * 32 + (slab & 3) * 1024 + a_k[p] + 2048) & 4095));
        }
    };
    auto step = [&](const float4 (&a)[4], const float4 (&b)[4]) {
        const float *ap = as + hi * LDA + l31, *bp = bs + hi * LDB + l31;
        float av[16], bv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { av[q] = ap[2 * q * LDA]; bv[q] = bp[2 * q * LDB]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            if (q >= 4 && q < 8) {
                const int p = q - 4;
                float *d = as + a_k[p] * LDA + a_r[p];
                d[0] = a[p].x; d[LDA] = a[p].y; d[2 * LDA] = a[p].z; d[3 * LDA] = a[p].w;
            }
            if (q >= 8 && q < 12) {
                const int p = q - 8;
                *reinterpret_cast<float4 *>(bs + a_r[p] * LDB + a_k[p]) = b[p];
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q >= 4 && q < 12) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    request(0, ra[0], rb[0]); request(1, ra[1], rb[1]); request(2, ra[2], rb[2]);
    __syncthreads();
    const long long c0 = clock64();
    for (int s = 0; s < nslab; s += 3) {
        request(s + 3, ra[0], rb[0]); __builtin_amdgcn_sched_barrier(0); step(ra[1], rb[1]);
        request(s + 4, ra[1], rb[1]); __builtin_amdgcn_sched_barrier(0); step(ra[2], rb[2]);
        request(s + 5, ra[2], rb[2]); __builtin_amdgcn_sched_barrier(0); step(ra[0], rb[0]);
    }
    const long long c1 = clock64();
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 123.456f) sink[0] = t;
    if (threadIdx.x == 0) out[blockIdx.x] = (c1 - c0);
}

template <int V> void run(long long *d, float *sink, const char *what) {
    const int n = 64, g = 256;
    hipLaunchKernelGGL(slab<V>, dim3(g), dim3(256), 0, 0, d, sink, n, g_src);
    hipLaunchKernelGGL(slab<V>, dim3(g), dim3(256), 0, 0, d, sink, n, g_src);
    hipDeviceSynchronize();
    std::vector<long long> h(g);
    hipMemcpy(h.data(), d, g * sizeof(long long), hipMemcpyDeviceToHost);
    double c = 0;
    for (long long x : h) c += x;
    printf("%-70s %7.0f shader cycles per slab step (16 MFMAs = 1027)\n", what, c / g / n);
}

int main() {
    long long *d; float *sink;
    hipMalloc(&d, 4096 * sizeof(long long)); hipMalloc(&sink, 4);
    hipMalloc(&g_src, 256 * 16384 * sizeof(float) + 65536);
    hipMemset(g_src, 0, 256 * 16384 * sizeof(float) + 65536);
    run<0>(d, sink, "operands read next to their MFMAs");
    run<1>(d, sink, "all operand reads first, then the MFMAs");
    run<2>(d, sink, "reads next to MFMAs + staging writes behind the MFMAs");
    run<3>(d, sink, "all reads first + staging writes behind the MFMAs");
    run<6>(d, sink, "+ operands loaded from global memory, requested one slab ahead");
    run<14>(d, sink, "+ operands loaded from global memory, requested two slabs ahead");
    run<18>(d, sink, "A kept k-contiguous in LDS (4 x 16-byte writes), no global loads");
    run<22>(d, sink, "A kept k-contiguous in LDS + global loads one slab ahead");
    {
        const int n = 63, g = 256;
        hipLaunchKernelGGL(slab_pipelined, dim3(g), dim3(256), 0, 0, d, sink, n, g_src);
        hipLaunchKernelGGL(slab_pipelined, dim3(g), dim3(256), 0, 0, d, sink, n, g_src);
        hipDeviceSynchronize();
        std::vector<long long> h(g);
        hipMemcpy(h.data(), d, g * sizeof(long long), hipMemcpyDeviceToHost);
        double c = 0;
        for (long long x : h) c += x;
        printf("%-70s %7.0f shader cycles per slab step\n", "software-pipelined: loads 3 ahead, staging inside the MFMA chain", c / g / n);
    }
    return 0;
}
