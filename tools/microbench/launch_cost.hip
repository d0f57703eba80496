// Microbenchmarks behind DESIGN.md's launch-structure decisions (round 2):
//   (a) dependent trivial kernels launched eagerly from a C loop      -> us per kernel (GPU or host bound)
//   (b) the same chain captured into a hipGraph and replayed           -> us per node
//   (c) an in-kernel barrier among G co-resident workgroups (monotonic counter, agent-scope
//       release / acquire, bounded spin)                               -> us per barrier round
//   (d) the same chain with a 64 KB-per-WG dirty store before each boundary / barrier
// Build: hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void trivial(float *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}

__global__ void __launch_bounds__(256) barrier_rounds(unsigned *counter, float *buf, int rounds, int dirty_floats,
                                                      int *timeout_flag) {
    const int G = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (dirty_floats) {
            float *mine = buf + (size_t)blockIdx.x * dirty_floats;
            for (int i = threadIdx.x; i < dirty_floats; i += blockDim.x) mine[i] += 1.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * G;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *timeout_flag = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *p; CK(hipMalloc(&p, 1 << 26)); CK(hipMemset(p, 0, 1 << 26));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int blocks : {1, 256, 1024}) {
        const int n = blocks * 256, N = 2000;
        for (int i = 0; i < 50; ++i) trivial<<<blocks, 256, 0, s>>>(p, n);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) trivial<<<blocks, 256, 0, s>>>(p, n);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager  chain blocks=%4d : %.2f us per kernel\n", blocks, 1e3 * ms / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) trivial<<<blocks, 256, 0, s>>>(p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph  chain blocks=%4d : %.2f us per node\n", blocks, 1e3 * ms / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    unsigned *counter; int *flag; CK(hipMalloc(&counter, 4)); CK(hipMalloc(&flag, 4));
    for (int dirty : {0, 4096, 16384}) {
        for (int G : {8, 16, 32, 64, 128, 256}) {
            const int R = 200;
            CK(hipMemsetAsync(counter, 0, 4, s)); CK(hipMemsetAsync(flag, 0, 4, s));
            barrier_rounds<<<G, 256, 0, s>>>(counter, p, 10, dirty, flag);
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            barrier_rounds<<<G, 256, 0, s>>>(counter, p, R, dirty, flag);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            int f = 0; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
            printf("barrier G=%3d dirty=%5d floats/WG : %.2f us per round%s\n", G, dirty, 1e3 * ms / R, f ? "  (TIMEOUT)" : "");
        }
    }
    return 0;
}
