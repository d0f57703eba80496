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
}  // namespace

extern "C" {

int rlx_abi_version(void) { return 5; }   // 2: adam_tf1_norm / sac head accumulate; 3: gemm desc batch_inner, n_fold; 4: per_sample payload rows, libm pow; 5: rlx_adam_tf1_step ticket = RLX_ADAM_TICKET_WORDS words
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

// ---- in-process kernel timer (rlx_common.hpp: rlx::launch) ------------------------------------------------------
int rlx_profile_begin(int max_records) {
    RLX_REQUIRE(max_records > 0 && max_records <= (1 << 20), "rlx_profile_begin: max_records out of range");
    rlx::Profiler &p = rlx::g_prof;
    RLX_REQUIRE(!p.active, "rlx_profile_begin: already armed");
    if (max_records > p.cap) {
        hipEvent_t *st = new hipEvent_t[max_records], *sp = new hipEvent_t[max_records];
        const char **nm = new const char *[max_records];
        for (int i = 0; i < p.cap; ++i) {
            st[i] = p.start[i];
            sp[i] = p.stop[i];
        }
        for (int i = p.cap; i < max_records; ++i) {
            RLX_HIP(hipEventCreate(&st[i]));
            RLX_HIP(hipEventCreate(&sp[i]));
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
    p.active = true;
    return RLX_OK;
}

int rlx_profile_end(int *n_records_host) {
    rlx::Profiler &p = rlx::g_prof;
    p.active = false;
    if (n_records_host) *n_records_host = p.n;
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
