// Where does a row-local layer spend its time?  dense_fwd / dense_bwdT of rowchain.hpp on the TD3 middle layer (400 x 300)
// in isolation: G workgroups each run the layer `reps` times on the same weights (L2-warm after the first), timed with
// events; variants by -DMODE: 0 = as in the library, 1 = no weight loads (registers), 2 = no arithmetic (loads only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I coach_amd/csrc tools/microbench/rowchain_probe.hip -o /tmp/rowchain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rowchain.hpp"

namespace rlx { void set_error(const char *, ...) {} Profiler g_prof; }
using namespace rlx_chain;

constexpr size_t kLds = sizeof(float) * (4 * R * kPitch + kPartFloats + 2 * kTileFloats);

__global__ void __launch_bounds__(T) fwd_kernel(const float *W, const float *bias, const float *x, float *out, int K, int N, int LDW, int reps,
                                                 long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem, *ys = xs + R * kPitch, *parts = ys + 3 * R * kPitch;
    load_rows(xs, kPitch, x, K, K, blockIdx.x * R, 1 << 20);
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; ++i) dense_fwd(xs, kPitch, K, W, LDW, bias, N, RLX_ACT_RELU, ys, kPitch, parts, nullptr, 0, 0, 4);
    const long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = ys[threadIdx.x * kPitch + 5];
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

__global__ void __launch_bounds__(T) bwd_kernel(const float *W, const float *x, float *out, int K, int N, int LDW, int reps, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dys = smem, *hs = dys + R * kPitch, *dxs = hs + R * kPitch, *wt = dxs + 2 * R * kPitch + kPartFloats;
    load_rows(dys, kPitch, x, N, N, blockIdx.x * R, 1 << 20);
    for (int e = threadIdx.x; e < R * kPitch; e += T) hs[e] = 1.f;
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; ++i) dense_bwdT(dys, kPitch, N, W, LDW, K, hs, kPitch, dxs, kPitch, wt, nullptr, 0, 0, 4);
    const long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = dxs[threadIdx.x * kPitch + 5];
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 400, LDW = argc > 2 ? atoi(argv[2]) : 300, N = argc > 3 ? atoi(argv[3]) : LDW;
    std::vector<float> hw((size_t)K * LDW), hx(4096 * 512);
    for (auto &v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto &v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    float *W, *b, *x, *out;
    long long *cyc;
    hipMalloc(&W, hw.size() * 4); hipMalloc(&b, 4096); hipMalloc(&x, hx.size() * 4); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, 4096);
    hipFuncSetAttribute((const void *)fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    hipFuncSetAttribute((const void *)bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {1, 8, 100}) {
        for (int reps : {1, 8}) {
            for (int which = 0; which < 2; ++which) {
                float best = 1e9f;
                std::vector<long long> hc(G);
                for (int t = 0; t < 5; ++t) {
                    hipEventRecord(e0);
                    if (which == 0) fwd_kernel<<<G, T, kLds>>>(W, b, x, out, K, N, LDW, reps, cyc);
                    else bwd_kernel<<<G, T, kLds>>>(W, x, out, K, N, LDW, reps, cyc);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                hipMemcpy(hc.data(), cyc, 8 * G, hipMemcpyDeviceToHost);
                long long mx = 0;
                for (auto c : hc) mx = c > mx ? c : mx;
                printf("%s %dx%d  (slice of %d) G=%3d reps=%d: kernel %.1f us, slowest workgroup %lld cycles = %.0f cycles per layer pass (%.2f us at 2.4 GHz)\n",
                       which ? "bwdT" : "fwd ", K, N, LDW, G, reps, best * 1e3f, mx, (double)mx / reps, (double)mx / reps / 2400.0);
            }
        }
    }
    hipError_t e = hipGetLastError();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
