// Shared host-side plumbing for librlx.so (gfx950 only — no dual backend).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/rlx.h"

namespace rlx {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Number of CUs on MI355X; grids for streaming kernels are capped at CUS*8 blocks
// (cdna_hip_programming.md G11) and grid-stride the rest.
constexpr int kCUs = 256;
constexpr int kMaxStreamBlocks = kCUs * 8;

inline int grid_for(long long work_items, int block, int cap = kMaxStreamBlocks) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}


// In-process kernel timer (rlx_profile_*): while it is armed every launch of this library goes out through
// hipExtLaunchKernel with a private (start, stop) event pair, which the runtime fills from the dispatch packet's own
// begin / end timestamps — the duration a kernel trace (rocprofv3 --kernel-trace) reports for that dispatch, taken
// in situ: same stream, same neighbours, operands as the previous kernel left them.  Eager launches only (an armed
// timer inside a stream capture is refused by rlx_profile_begin's caller contract: the events are not graph nodes).
struct Profiler {
    bool active = false;
    int n = 0, cap = 0;         // records taken / allocated event pairs
    int limit = 0, dropped = 0; // max_records of the armed trace / launches that found it full (rlx_profile_end fails)
    hipEvent_t *start = nullptr, *stop = nullptr;
    const char **name = nullptr;
};
extern Profiler g_prof;

template <typename F, typename... Args>
inline void launch(const char *name, F kernel, dim3 grid, dim3 block, unsigned shmem, hipStream_t s,
                   Args... args) {
    if (g_prof.active && g_prof.n < g_prof.limit) {
        const int i = g_prof.n++;
        g_prof.name[i] = name;
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, s, g_prof.start[i], g_prof.stop[i], 0, args...);
    } else {
        if (g_prof.active) ++g_prof.dropped;
        kernel<<<grid, block, shmem, s>>>(args...);
    }
}
}  // namespace rlx

#define RLX_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            rlx::set_error(__VA_ARGS__);            \
            return RLX_ERR_INVALID_ARG;             \
        }                                           \
    } while (0)

#define RLX_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) {                                                   \
            rlx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                    \
            return RLX_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

// After a kernel launch: surfaces launch-configuration errors without syncing.
#define RLX_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            rlx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),  \
                           __FILE__, __LINE__);                                         \
            return RLX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

// Every kernel of the library is launched through this (see rlx::launch): `kernel` in parentheses, so that template
// argument lists with commas stay one macro argument.
#define RLX_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    rlx::launch(#kernel, kernel, grid, block, shmem, stream, ##__VA_ARGS__)
