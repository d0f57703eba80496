// Shared host-side plumbing for librlx.so (gfx950 only — no dual backend).
#pragma once
#include <hip/hip_runtime.h>
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
