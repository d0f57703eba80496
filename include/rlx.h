/*
 * rlx.h — C ABI of librlx.so, the MI355X (gfx950) hot-path library behind the
 * rl_coach-compatible Python adapters in coach_amd/.
 *
 * The reference (IntelLabs/coach, rl_coach 1.0.1) has NO C interface: its plug
 * points are Python classes.  Every entry point below therefore
 * cites the reference *Python* routine whose arithmetic it replaces (file:line under
 * /root/reference/rl_coach/), and INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add to call it.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Every function returns RLX_OK (0) or a negative rlx_status; the human readable
 *     reason is available from rlx_last_error() (thread local).
 *   - No function allocates device memory or synchronises the device unless its
 *     name says so (rlx_*_sync / rlx_event_elapsed_ms): all are hipGraph-capturable.
 *   - Layouts are row-major, innermost dimension last (NHWC for images).
 */
#ifndef RLX_H
#define RLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rlx_status {
    RLX_OK = 0,
    RLX_ERR_INVALID_ARG = -1, /* bad size / null pointer / unsupported combination */
    RLX_ERR_HIP = -2,         /* a HIP runtime call or a kernel launch failed       */
    RLX_ERR_UNSUPPORTED = -3
} rlx_status;

/* ------------------------------------------------------------------ runtime -- */
int rlx_abi_version(void);               /* bumps when a signature changes          */
const char *rlx_last_error(void);        /* message of the last failure (this thread) */
const char *rlx_build_arch(void);        /* "gfx950"                                  */
int rlx_device_count(int *count_host);   /* replaces coach.py:61-84 (cuDeviceGetCount) */
int rlx_stream_sync(void *stream);
int rlx_event_create(void **event_host);
int rlx_event_destroy(void *event);
int rlx_event_record(void *event, void *stream);
int rlx_event_elapsed_ms(void *start, void *stop, float *ms_host); /* syncs on stop */

/* --------------------------------------------- prioritized replay (K5 / K6) -- */
/* Trees are fp64 array-heaps of 2*capacity-1 nodes exactly as the reference's
 * SegmentTree (memories/non_episodic/prioritized_experience_replay.py:43-156);
 * capacity must be a power of two (:176-179).  max_priority is one device double
 * mirroring PrioritizedExperienceReplay.maximal_priority (:186,:201).  `status` is a
 * device int the kernels OR error bits into (1 = leaf index out of range -> the
 * reference's ValueError at :123-126; 2 = negative error -> ValueError at :195). */
int rlx_per_init(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                 double *max_priority, void *stream);               /* SegmentTree.__init__ :59-67 */
int rlx_per_store(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                  int start_leaf, int n, double alpha, double *max_priority, int *status,
                  void *stream);                                     /* .store :264-283 (n consecutive adds) */
int rlx_per_update(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                   const int *idx, const double *errors, int n, double alpha, double epsilon,
                   double *max_priority, int *status, void *stream); /* .update_priorities :203-217 */
int rlx_per_update_leaves(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                          const int *idx, const double *leaf_pa, const double *leaf_p, int n,
                          double *max_priority, int *status, void *stream); /* same, host-computed p**alpha */
int rlx_per_sample(const double *sum_tree, const double *min_tree, int capacity,
                   const double *uniforms, int batch, double num_transitions, double beta,
                   int *out_idx, double *out_weight, double *out_priority,
                   void *stream);                                    /* .sample :229-255; uniforms[i] = random.random() */

#ifdef __cplusplus
}
#endif
#endif /* RLX_H */
