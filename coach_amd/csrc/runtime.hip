// Runtime entry points of librlx.so: error reporting, device enumeration, HIP events.
#include "rlx_common.hpp"
#include <cstring>

namespace rlx {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
Profiler g_prof;
}  // namespace rlx

namespace {
// one lane follows `steps` dependent loads through a chain of indices: the duration / steps is the latency of whatever
// level of the memory hierarchy the chain's footprint reaches (bench.py's `box` block)
__global__ void probe_chase_kernel(const int *__restrict__ chain, int start, int steps, int *__restrict__ out) {
    int i = start;
    for (int s = 0; s < steps; ++s) i = chain[i];
    *out = i;
}
// every wave runs `iters` dependent v_mfma_f32_32x32x2_f32 (one accumulator: the chain of a GEMM slab) between two
// readings of the shader-clock counter and of the constant 100 MHz counter; lane 0 of every workgroup's wave 0 records
// both differences: cycles per MFMA, and the clock the shader actually ran at under that load
__global__ void __launch_bounds__(256) probe_mfma_kernel(int iters, long long *__restrict__ out, float *__restrict__ sink) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float a = 1.f + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) keep += acc[r];
    const long long c1 = clock64(), w1 = wall_clock64();
    if (keep == 12345.678f) *sink = keep;                    // never true: the chain stays live
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = w1 - w0;
    }
}
// The same for either fp32 MFMA shape with CHAINS independent accumulators per wave, issued round-robin with nothing
// between the MFMAs of a chain but the other chains' MFMAs: CHAINS = 1 measures the dependent-accumulator latency, CHAINS >= 2
// the issue rate of the pipe (the 68 cycles DESIGN quoted for 16x16x4 were one dependent chain with
// other instructions in it, not the issue rate).
template <bool SMALL, int CHAINS>
__global__ void __launch_bounds__(256) probe_mfma_shape_kernel(int iters, long long *__restrict__ out, float *__restrict__ sink) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 big[CHAINS];
    f32x4 small[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) big[c][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) small[c][r] = 0.f;
    }
    const float a = 1.f + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (SMALL) small[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, small[c], 0, 0, 0);
            else big[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[c], 0, 0, 0);
        }
    }
    float keep = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += big[c][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) keep += small[c][r];
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (keep == 12345.678f) *sink = keep;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = w1 - w0;
    }
}
// How long do K back-to-back 16-byte-per-lane loads of one wave take — as global -> LDS requests (global_load_lds_dwordx4)
// and as ordinary loads into registers?  Every wave of the workgroup issues K requests for lines of an L2-resident
// buffer and waits for all of them; lane 0 of wave 0 records the shader cycles.
template <int K, bool LDS_DMA>
__global__ void __launch_bounds__(256) probe_requests_kernel(const float *__restrict__ src, long long *__restrict__ out,
                                                             float *__restrict__ sink, int round) {
    __shared__ __attribute__((aligned(1024))) float ring[K * 1024 + 4];        // K x 4 KB (256 lanes x 16 bytes)
    const int tid = threadIdx.x, wid = tid >> 6;
    const float *p = src + ((size_t)blockIdx.x * 256 + tid) * 4 + (size_t)(round & 0xffff) * 64;
    if (wid >= (round >> 16) && (round >> 16) > 0) {          // round >> 16 = number of waves that issue (0: all four)
        __syncthreads();
        __syncthreads();
        return;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring)) + (unsigned)wid * 1024u);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v[K];
    __syncthreads();
    const long long c0 = clock64();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float *q = p + (size_t)k * 256 * 1024;           // another 1 MB page of the buffer per request
        if (LDS_DMA) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(q), "s"(lds0 + 4096u * k) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(q) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64();
    float keepv = 0.f;
    if (!LDS_DMA) {
#pragma unroll
        for (int k = 0; k < K; ++k) keepv += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    __syncthreads();
    if (LDS_DMA) keepv = ring[tid];
    if (keepv == 12345.678f) *sink = keepv;
    if (tid == 0) out[blockIdx.x] = c1 - c0;
}
}  // namespace

extern "C" {

int rlx_abi_version(void) { return 10; }   // 10: rlx_gemm_desc.kw_min_tiles, rlx_dqn_head_loss_backward; 9: rlx_conv32_input_grad_per_update / rlx_splitk_reduce_jobs_per_update (rlx_per_update_desc), rlx_imgreplay_gather_columns, conv_dw_u8 for one tower of 32 filters (2 B splits); 8: rlx_ppo_fc_rows / rlx_ppo_heads_tail / rlx_splitk_reduce_jobs_ppo_tail, fused TD3 / SAC updates, rlx_conv_dw_*; 7: rlx_conv123_forward, rlx_gemm_describe; 6: rlx_conv23_forward, rlx_gemm_multi_defer, rlx_gemm_big_tiles; 2: adam_tf1_norm / sac head accumulate; 3: gemm desc batch_inner, n_fold; 4: per_sample payload rows, libm pow; 5: rlx_adam_tf1_step ticket = RLX_ADAM_TICKET_WORDS words
const char *rlx_last_error(void) { return rlx::g_err; }
const char *rlx_build_arch(void) { return "gfx950"; }

// Replaces the reference's libcuda probing (rl_coach/coach.py:61-84).
int rlx_device_count(int *count_host) {
    RLX_REQUIRE(count_host != nullptr, "rlx_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count_host = n;
    return RLX_OK;
}

int rlx_stream_sync(void *stream) {
    RLX_HIP(hipStreamSynchronize(rlx::as_stream(stream)));
    return RLX_OK;
}

int rlx_event_create(void **event_host) {
    RLX_REQUIRE(event_host != nullptr, "rlx_event_create: null output");
    hipEvent_t ev;
    RLX_HIP(hipEventCreate(&ev));
    *event_host = ev;
    return RLX_OK;
}

int rlx_event_destroy(void *event) {
    RLX_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return RLX_OK;
}

int rlx_event_record(void *event, void *stream) {
    RLX_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), rlx::as_stream(stream)));
    return RLX_OK;
}

int rlx_event_elapsed_ms(void *start, void *stop, float *ms_host) {
    RLX_REQUIRE(ms_host != nullptr, "rlx_event_elapsed_ms: null output");
    RLX_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    RLX_HIP(hipEventElapsedTime(ms_host, reinterpret_cast<hipEvent_t>(start),
                                reinterpret_cast<hipEvent_t>(stop)));
    return RLX_OK;
}

// Dependent-load latency probe (measurement utility, no reference counterpart): follows chain[i] -> i for `steps`
// loads from `start` in ONE lane and writes the final index (so that the loads cannot be dropped).
int rlx_probe_chase(const int *chain, int start, int steps, int *out, void *stream) {
    RLX_REQUIRE(chain && out && start >= 0 && steps > 0, "rlx_probe_chase: bad arguments");
    RLX_LAUNCH((probe_chase_kernel), 1, 1, 0, rlx::as_stream(stream), chain, start, steps, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// Matrix-pipe clock probe (measurement utility, no reference counterpart): `workgroups` x 4 waves each run `iters`
// dependent fp32 32x32x2 MFMAs; out[2 w] = shader-clock cycles, out[2 w + 1] = ticks of the constant 100 MHz counter of
// workgroup w.  cycles / iters = cycles per MFMA of a dependent chain; cycles / ticks * 100 = the shader clock in MHz
// under that load (bench.py's `box` block).
int rlx_probe_mfma(int workgroups, int iters, long long *out, float *sink, void *stream) {
    RLX_REQUIRE(workgroups > 0 && iters > 0 && out && sink, "rlx_probe_mfma: bad arguments");
    RLX_LAUNCH((probe_mfma_kernel), workgroups, 256, 0, rlx::as_stream(stream), iters, out, sink);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// rlx_probe_mfma for either fp32 MFMA shape and 1 / 2 / 4 independent accumulator chains per wave (`iters` MFMAs PER CHAIN):
// cycles / (iters * chains) = cycles per MFMA at that interleave.
int rlx_probe_mfma_shape(int workgroups, int iters, int small_shape, int chains, long long *out, float *sink, void *stream) {
    RLX_REQUIRE(workgroups > 0 && iters > 0 && out && sink && (chains == 1 || chains == 2 || chains == 4),
                "rlx_probe_mfma_shape: bad arguments (chains must be 1, 2 or 4)");
    hipStream_t s = rlx::as_stream(stream);
#define RLX_PROBE_SHAPE(S, C)                                                                             \
    if ((small_shape != 0) == S && chains == C) {                                                         \
        RLX_LAUNCH((probe_mfma_shape_kernel<S, C>), workgroups, 256, 0, s, iters, out, sink);             \
        RLX_LAUNCH_CHECK();                                                                               \
        return RLX_OK;                                                                                    \
    }
    RLX_PROBE_SHAPE(false, 1) RLX_PROBE_SHAPE(false, 2) RLX_PROBE_SHAPE(false, 4)
    RLX_PROBE_SHAPE(true, 1) RLX_PROBE_SHAPE(true, 2) RLX_PROBE_SHAPE(true, 4)
#undef RLX_PROBE_SHAPE
    return RLX_ERR_INVALID_ARG;
}

// Request-pipelining probe (measurement utility): every wave of `workgroups` workgroups issues `requests` (1, 2, 4, 8)
// back-to-back 16-byte-per-lane loads — global -> LDS (lds_dma != 0) or into registers — from `src` (>= 9 MB + the
// workgroups' share) and waits for all of them; out[w] = shader cycles of workgroup w.
int rlx_probe_requests(int workgroups, int requests, int lds_dma, int round, const float *src, long long *out, float *sink,
                       void *stream) {
    RLX_REQUIRE(workgroups > 0 && src && out && sink, "rlx_probe_requests: bad arguments");
    hipStream_t s = rlx::as_stream(stream);
#define RLX_PROBE_REQ(K)                                                                                   \
    if (requests == K) {                                                                                   \
        if (lds_dma) RLX_LAUNCH((probe_requests_kernel<K, true>), workgroups, 256, 0, s, src, out, sink, round);  \
        else RLX_LAUNCH((probe_requests_kernel<K, false>), workgroups, 256, 0, s, src, out, sink, round);         \
        RLX_LAUNCH_CHECK();                                                                                \
        return RLX_OK;                                                                                     \
    }
    RLX_PROBE_REQ(1) RLX_PROBE_REQ(2) RLX_PROBE_REQ(4) RLX_PROBE_REQ(8)
#undef RLX_PROBE_REQ
    RLX_REQUIRE(false, "rlx_probe_requests: requests must be 1, 2, 4 or 8");
    return RLX_OK;
}

// ---- in-process kernel timer (rlx_common.hpp: rlx::launch) ------------------------------------------------------
int rlx_profile_begin(int max_records) {
    RLX_REQUIRE(max_records > 0 && max_records <= (1 << 20), "rlx_profile_begin: max_records out of range");
    rlx::Profiler &p = rlx::g_prof;
    RLX_REQUIRE(!p.active, "rlx_profile_begin: already armed");
    if (max_records > p.cap) {
        hipEvent_t *st = new hipEvent_t[max_records], *sp = new hipEvent_t[max_records];
        const char **nm = new const char *[max_records];
        int made = p.cap;                   // events [p.cap, made) of both arrays are new and ours to destroy on failure
        hipError_t err = hipSuccess;
        for (; made < max_records && err == hipSuccess; ++made) {
            err = hipEventCreate(&st[made]);
            if (err == hipSuccess) {
                err = hipEventCreate(&sp[made]);
                if (err != hipSuccess) (void)hipEventDestroy(st[made]);
            }
            if (err != hipSuccess) break;
        }
        if (err != hipSuccess) {
            for (int i = p.cap; i < made; ++i) {
                (void)hipEventDestroy(st[i]);
                (void)hipEventDestroy(sp[i]);
            }
            delete[] st;
            delete[] sp;
            delete[] nm;
            rlx::set_error("rlx_profile_begin: hipEventCreate failed: %s", hipGetErrorString(err));
            return RLX_ERR_HIP;
        }
        for (int i = 0; i < p.cap; ++i) {
            st[i] = p.start[i];
            sp[i] = p.stop[i];
        }
        delete[] p.start;
        delete[] p.stop;
        delete[] p.name;
        p.start = st;
        p.stop = sp;
        p.name = nm;
        p.cap = max_records;
    }
    p.n = 0;
    p.limit = max_records;                  // the bound the caller asked for, also when the arrays are larger
    p.dropped = 0;
    p.active = true;
    return RLX_OK;
}

int rlx_profile_end(int *n_records_host) {
    rlx::Profiler &p = rlx::g_prof;
    p.active = false;
    if (n_records_host) *n_records_host = p.n;
    // a trace that ran out of records is an error, not a shorter trace: sums over it would undercount
    RLX_REQUIRE(p.dropped == 0, "rlx_profile_end: %d launches beyond max_records = %d went out untimed", p.dropped, p.limit);
    return RLX_OK;
}

int rlx_profile_read(int index, const char **name_host, float *ms_host) {
    rlx::Profiler &p = rlx::g_prof;
    RLX_REQUIRE(!p.active && index >= 0 && index < p.n && name_host && ms_host,
                "rlx_profile_read: record %d of %d (timer %s)", index, p.n, p.active ? "still armed" : "stopped");
    RLX_HIP(hipEventSynchronize(p.stop[index]));
    RLX_HIP(hipEventElapsedTime(ms_host, p.start[index], p.stop[index]));
    *name_host = p.name[index];
    return RLX_OK;
}

}  // extern "C"
