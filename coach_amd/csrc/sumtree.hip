// K5/K6 — prioritized-replay segment trees on gfx950.
//
// Replaces rl_coach/memories/non_episodic/prioritized_experience_replay.py:
//   SegmentTree.update/_propagate (:63-74,:116-129), SegmentTree._retrieve (:76-92),
//   PrioritizedExperienceReplay._update_priority (:188-201), .sample (:229-255),
//   .store (:264-283).
//
// HBM layout (identical to the reference's numpy array-heap so that indices are
// bit-exact): each tree is 2*cap-1 fp64 values, root at 0, children of p at 2p+1,
// 2p+2, leaves at [cap-1, 2cap-1).  cap is a power of two (reference :176-179).
// Three trees (sum of p^alpha, min of p^alpha, max of p) are separate arrays.
//
// This file is compiled with -ffp-contract=off: the stratified draw
// a + (b-a)*u and the descent's val - tree[left] must round exactly like CPython's
// double arithmetic, an FMA would change the selected leaf in borderline cases.
//
// Bound: latency (log2(cap) dependent 8-byte loads per draw); algorithmic bytes are
// 8*log2(cap) per draw and 3*log2(cap)*(16 read + 8 write) per updated priority
// (SURVEY.md §8(d)).
#include "rlx_common.hpp"
#include "libm_pow.hpp"
#include "splitk_reduce_body.hpp"
#include "per_update_body.hpp"
#include <cmath>

namespace {

constexpr int kBlock = 256;

__global__ void per_init_kernel(double *__restrict__ sum, double *__restrict__ mn,
                                double *__restrict__ mx, long long n_nodes) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n_nodes; i += stride) {
        sum[i] = 0.0;                                   // Operation.SUM initial_value (:57)
        mn[i] = __builtin_huge_val();                   // Operation.MIN initial_value (:56)
        mx[i] = -__builtin_huge_val();                  // Operation.MAX initial_value (:55)
    }
}

// One workgroup walks all updated leaves to the root, one tree level per barrier.
// The reference applies updates one at a time; replaying them level-synchronously is
// equivalent because a parent is always recomputed from both children (:70), so the
// final tree only depends on the final leaf values.  Duplicate indices: the LAST
// occurrence wins, as in the sequential loop (:214-215).
//
// mode 0: leaves come from TD errors:  p = err + eps; sum/min <- p^alpha; max <- p
// mode 1: n consecutive leaves (ring, wrapping at cap) <- *max_priority  (store, :274-279)
// mode 2: caller supplies the leaf values directly (leaf_pa -> sum/min, leaf_p -> max);
//         used where p^alpha must be bit-identical to the host libm.
// mode 3: as mode 1 but the (single) stored value pair comes from the host: eps = p^alpha,
//         alpha = p  (bit-exact store when the adapter mirrors maximal_priority on the host).
__global__ void __launch_bounds__(1024)
per_update_kernel(double *__restrict__ sum, double *__restrict__ mn, double *__restrict__ mx,
                  int cap, int levels, const int *__restrict__ idx,
                  const double *__restrict__ err, const double *__restrict__ leaf_pa,
                  const double *__restrict__ leaf_p, int n, int start_leaf, double alpha,
                  double eps, double *__restrict__ max_priority, int mode,
                  int *__restrict__ status) {
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const double stored_priority = (mode == 1) ? *max_priority : alpha;
    const bool ring = (mode == 1 || mode == 3);

    // ---- leaves
    for (int i = tid; i < n; i += nthr) {
        int leaf;
        double pa, p;
        bool live = true;
        if (ring) {
            leaf = (start_leaf + i) & (cap - 1);
            p = stored_priority;
            int odd = 0;
            pa = (mode == 1) ? rlx::libm_pow(p, alpha, &odd) : eps;    // maximal_priority ** alpha (:274)
            if (odd) atomicOr(status, 4);
        } else {
            leaf = idx[i];
            if (leaf < 0 || leaf >= cap) {          // reference raises ValueError (:123-126)
                atomicOr(status, 1);
                live = false;
            }
            if (mode == 0) {
                double e = err[i];
                if (e < 0.0) {                      // "priorities must be non-negative" (:195)
                    atomicOr(status, 2);
                    live = false;
                }
                p = e + eps;
                int odd = 0;
                pa = rlx::libm_pow(p, alpha, &odd);                   // priority ** self.alpha (:197)
                if (odd) atomicOr(status, 4);
            } else {
                pa = leaf_pa[i];
                p = leaf_p[i];
            }
            // last occurrence of a duplicated index wins
            for (int j = i + 1; live && j < n; ++j)
                if (idx[j] == leaf) live = false;
        }
        if (live) {
            int node = leaf + cap - 1;
            sum[node] = pa;
            mn[node] = pa;
            mx[node] = p;
        }
    }
    __syncthreads();

    // ---- propagate, one level per barrier (leaf level first)
    for (int lvl = 0; lvl < levels; ++lvl) {
        for (int i = tid; i < n; i += nthr) {
            int leaf = ring ? ((start_leaf + i) & (cap - 1)) : idx[i];
            if (leaf < 0 || leaf >= cap) continue;
            int node = leaf + cap - 1;
            // ancestor of `node` that sits `lvl+1` levels above the leaf level
            int parent = ((node + 1) >> (lvl + 1)) - 1;
            int l = 2 * parent + 1, r = l + 1;
            double sl = sum[l], sr = sum[r];
            double ml = mn[l], mr = mn[r];
            double xl = mx[l], xr = mx[r];
            sum[parent] = sl + sr;                          // operator.add (:57)
            mn[parent] = (mr < ml) ? mr : ml;               // python min(a,b): b if b<a else a
            mx[parent] = (xr > xl) ? xr : xl;               // python max(a,b): b if b>a else a
        }
        __syncthreads();
    }
    if (tid == 0 && !ring) *max_priority = mx[0];          // (:201)
}

// The same update for n <= 256 leaves without a memory round trip per level.  The level-synchronous kernel above is
// a chain of `levels` dependent load -> store -> barrier steps, and a __syncthreads() behind global stores waits for
// their write acknowledgements: 17 us for 64 leaves of a 2^20-leaf tree, 23 % of the C3 GPU time
// (profiles/r03_call8_c3_kernel_stats.csv).  Here the updated leaves are first ordered by leaf index (a counting rank
// in LDS) and thread t takes the t-th of them: it carries the node of that leaf's path that sits on the current level,
// with its three values, in registers.  In that order the threads below one node form a run [lo, hi] of neighbours,
// all holding the same values, and the sibling of a left child can only be carried by the run that starts at hi + 1
// (of a right child: the run that ends at lo - 1).  So one level costs one look at ONE other thread's registers:
//   * n <= 64 (one wavefront, WAVE): eight ds_bpermute in flight together — one LDS-crossbar round trip, no barrier;
//   * n <= 256: a 48-byte record per thread in LDS, double-buffered, one barrier per level that waits for LDS traffic
//     alone.
// A sibling that no updated leaf lies below is untouched by this launch and was prefetched — the siblings of twenty
// levels in ONE round trip, issued before the pow.  The first thread of a run writes the node to the tree; those
// stores are never waited for.  Parents are recomputed from (left, right) exactly like _propagate (:63-74):
// bit-identical trees.  (A first version that found siblings through an LDS hash table was no faster than the kernel
// above — four to five dependent LDS round trips per level; profiles/r03_ab_per_update.txt.)
// (the device body: per_update_body.hpp, shared with conv_bwd_fused.hip)
using rlx_per::kPathChunk;
using rlx_per::kPathMaxLeaves;
using rlx_per::per_update_paths_body;
template <bool WAVE, bool CONTIG>
__global__ void __launch_bounds__(WAVE ? 64 : kPathMaxLeaves)
per_update_paths_kernel(double *__restrict__ sum, double *__restrict__ mn, double *__restrict__ mx,
                        int cap, int levels, const int *__restrict__ idx,
                        const double *__restrict__ err, const double *__restrict__ leaf_pa,
                        const double *__restrict__ leaf_p, int n, int start_leaf, double alpha,
                        double eps, double *__restrict__ max_priority, int mode,
                        int *__restrict__ status) {
    per_update_paths_body<WAVE, CONTIG>(sum, mn, mx, cap, levels, idx, err, leaf_pa, leaf_p, n, start_leaf, alpha, eps,
                                        max_priority, mode, status);
}

// update_priorities (:203-217) of <= 64 sampled leaves as the FIRST-dispatched workgroup of the backward pass's deferred
// split-K reduction launch (rlx_splitk_reduce_jobs_per_update): the TD errors exist since the head's loss kernel, the
// tree is not read again before the next sample() — the update needs no launch of its own in the chain of an off-policy
// update (10.3 us of the C3 update: one workgroup, a chain of dependent round trips).  One wave of the workgroup works;
// the other fifteen leave at once (s_barrier counts the surviving waves).
struct PerUpdateDev {
    double *sum, *mn, *mx, *max_priority;
    const int *idx;
    const double *err;
    int *status;
    double alpha, eps;
    int cap, levels, n;
};
__global__ void __launch_bounds__(1024) splitk_reduce_jobs_per_kernel(const rlx_reduce::ReduceJobs jobs, const PerUpdateDev u) {
    __shared__ float4 part[16][64];
    if (blockIdx.z > 0) {
        rlx_reduce::splitk_reduce_job_body(jobs.job[blockIdx.z - 1], part);
        return;
    }
    if (blockIdx.x != 0 || blockIdx.y != 0 || threadIdx.x >= 64) return;
    per_update_paths_body<true, false>(u.sum, u.mn, u.mx, u.cap, u.levels, u.idx, u.err, nullptr, nullptr, u.n, 0, u.alpha,
                                       u.eps, u.max_priority, 0, u.status);
}

// One draw per thread.  u[i] is CPython's random.random() drawn on the host, so that
// val == random.uniform(seg*i, seg*(i+1)) bit for bit (:240-244).
//
// _retrieve (:76-92) is a chain of log2(cap) dependent loads — 20 round trips to L2 / HBM for the 2^20-leaf tree, 8.5 us of
// the C3 update's critical path as a plain loop.  Here (same comparisons on the same values: bit-identical leaves)
//   * the top kTopSteps + 1 levels (<= 4095 nodes, 32 KB) are loaded into LDS by the whole block in ONE round trip,
//     together with u[i] and the roots — every draw's first kTopSteps steps run from LDS;
//   * below that a draw takes THREE levels per round trip: from node n it requests sum[] of the left child, of both
//     possible left grandchildren and of all eight possible great-grandchildren (11 loads in flight), then takes the
//     three steps from registers.  The number of LDS steps is chosen so that the rest is a multiple of three: the last
//     step of the last group holds both leaves, so the leaf value needs no load of its own.
// 4 round trips instead of 21 for the 2^20-leaf tree.
constexpr int kSampleThreads = 256, kTopMaxSteps = 11;
__global__ void __launch_bounds__(kSampleThreads)
per_sample_kernel(const double *__restrict__ sum, const double *__restrict__ mn,
                  int cap, int levels, int top_steps, const double *__restrict__ u, int batch,
                  double n_transitions, double beta, int *__restrict__ out_idx,
                  double *__restrict__ out_weight,
                  double *__restrict__ out_priority, long long stored_total,
                  long long payload_rows, int *__restrict__ out_rows) {
    extern __shared__ __attribute__((aligned(16))) double top[];         // nodes [0, 2^(top_steps + 1) - 1)
    __shared__ double log_tab[rlx::kLibmPowLogDoubles];                  // glibc's pow tables: filled in the same round trip
    __shared__ unsigned long long exp_tab[rlx::kLibmPowExpWords];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kSampleThreads + tid;
    const int n_nodes = 2 * cap - 1;
    const int n_top = (2 << top_steps) - 1;                              // <= n_nodes (top_steps <= levels)
    const double ui = i < batch ? u[i] : 0.0;
    const double mn0 = mn[0];
    {
        const double *lt = &rlx::libm_detail::kLogTab[0][0];
        static_assert(rlx::kLibmPowLogDoubles <= 2 * kSampleThreads && rlx::kLibmPowExpWords <= kSampleThreads, "pow tables");
        const double l0 = lt[tid], l1 = lt[min(tid + kSampleThreads, rlx::kLibmPowLogDoubles - 1)];
        const unsigned long long e0 = rlx::libm_detail::kExpTab[tid];
        log_tab[tid] = l0;
        if (tid + kSampleThreads < rlx::kLibmPowLogDoubles) log_tab[tid + kSampleThreads] = l1;
        exp_tab[tid] = e0;
    }
    for (int k0 = tid; k0 < n_top; k0 += 8 * kSampleThreads) {           // eight loads in flight per thread
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sum[min(k0 + j * kSampleThreads, n_top - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j * kSampleThreads < n_top) top[k0 + j * kSampleThreads] = v[j];
    }
    __syncthreads();
    if (i >= batch) return;
    const double total = top[0];
    const double segment_size = total / (double)batch;                   // (:232)
    const double min_probability = mn0 / total;                          // (:235)
    int odd = 0;                                  // (an empty tree gives inf/nan here, as in the reference)
    const double max_weight = rlx::libm_pow_tabs(min_probability * n_transitions, -beta, &odd, log_tab, exp_tab);  // (:236)
    const double a = segment_size * (double)i;                           // (:240)
    const double b = segment_size * (double)(i + 1);                     // (:241)
    double val = a + (b - a) * ui;                                       // random.uniform
    int node = 0;
    double leaf_value = total;                                           // (cap == 1: the root is the leaf)
    bool have = cap == 1;
#define RLX_PER_STEP(tl, tr, have_r)                                     \
    if (val <= (tl)) { node = left; leaf_value = (tl); have = true; }    \
    else { val = val - (tl); node = left + 1; leaf_value = (tr); have = (have_r); }
    for (int s = 0; s < top_steps; ++s) {                                // _retrieve (:76-92), levels held in LDS
        const int left = 2 * node + 1;
        const double tl = top[left], tr = top[left + 1];
        RLX_PER_STEP(tl, tr, true)
    }
    for (int s = top_steps; s < levels; s += 3) {                        // three levels per round trip
        const int l1 = 2 * node + 1;                                     // children l1, l1 + 1
        const int l2 = 2 * l1 + 1;                                       // grandchildren l2 .. l2 + 3
        const int l3 = 2 * l2 + 1;                                       // great-grandchildren l3 .. l3 + 7
        const bool two = s + 1 < levels, three = s + 2 < levels;
        const double c1 = sum[l1];
        double g[2] = {0.0, 0.0}, h[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (two) { g[0] = sum[l2]; g[1] = sum[l2 + 2]; }
        if (three) {
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = sum[l3 + k];
        }
        {
            const int left = l1;
            RLX_PER_STEP(c1, 0.0, false)
        }
        if (two) {
            const int left = 2 * node + 1;
            const double tl = node == l1 ? g[0] : g[1];
            RLX_PER_STEP(tl, 0.0, false)
        }
        if (three) {
            const int left = 2 * node + 1, k = left - l3;                // 0, 2, 4 or 6
            double tl = h[0], tr = h[1];
            if (k == 2) { tl = h[2]; tr = h[3]; }
            if (k == 4) { tl = h[4]; tr = h[5]; }
            if (k == 6) { tl = h[6]; tr = h[7]; }
            RLX_PER_STEP(tl, tr, true)
        }
    }
#undef RLX_PER_STEP
    if (!have) leaf_value = sum[node];
    (void)n_nodes;
    const double priority = leaf_value / total;                          // (:247)
    const double weight = rlx::libm_pow_tabs(n_transitions * priority, -beta, &odd, log_tab, exp_tab);   // (:248)
    out_idx[i] = node - cap + 1;
    out_weight[i] = weight / max_weight;                                 // (:249)
    if (out_priority) out_priority[i] = leaf_value;
    if (out_rows) {
        // The payload ring may have more rows than the tree has leaves (rows of a step that is
        // written but not visible yet).  Transition number g (0-based, in store order) owns leaf
        // g % cap and payload row g % payload_rows; the live owner of a leaf is the newest such g.
        const long long leaf = node - cap + 1;
        const long long back = ((stored_total - 1 - leaf) % cap + cap) % cap;
        const long long g = stored_total - 1 - back;
        out_rows[i] = g < 0 ? -1 : (int)(g % payload_rows);      // -1: a leaf that was never stored
    }
}

__global__ void libm_pow_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                double *__restrict__ out, int n, int *__restrict__ status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int odd = 0;
    out[i] = rlx::libm_pow(x[i], y[i], &odd);
    if (odd) atomicOr(status, 4);
}

inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" {

int rlx_per_init(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                 double *max_priority, void *stream) {
    RLX_REQUIRE(sum_tree && min_tree && max_tree && max_priority, "rlx_per_init: null tree");
    RLX_REQUIRE(is_pow2(capacity),
                "A segment tree size must be a positive power of 2. The given size is %d", capacity);
    long long n_nodes = 2LL * capacity - 1;
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((per_init_kernel), rlx::grid_for(n_nodes, kBlock), kBlock, 0, s, sum_tree, min_tree,
                                                                      max_tree, n_nodes);
    RLX_LAUNCH_CHECK();
    const double one = 1.0;   // self.maximal_priority = 1.0 (:186)
    RLX_HIP(hipMemcpyAsync(max_priority, &one, sizeof(double), hipMemcpyHostToDevice, s));
    return RLX_OK;
}

static int g_path_max_leaves = kPathMaxLeaves;

static int launch_update(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                         const int *idx, const double *err, const double *leaf_pa,
                         const double *leaf_p, int n, int start_leaf, double alpha, double eps,
                         double *max_priority, int mode, int *status, void *stream,
                         const char *who) {
    RLX_REQUIRE(sum_tree && min_tree && max_tree && max_priority && status, "%s: null pointer", who);
    RLX_REQUIRE(is_pow2(capacity), "%s: capacity %d is not a power of two", who, capacity);
    RLX_REQUIRE(n >= 0, "%s: negative count", who);
    if (n == 0) return RLX_OK;
    int threads = ((n + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    if (n <= g_path_max_leaves) {       // one leaf per thread: the path walk that meets only in LDS
        const bool contig = (mode == 1 || mode == 3) && n <= capacity && start_leaf >= 0 &&
                            (long long)start_leaf + n <= capacity && capacity <= (1 << 30);
        auto kernel = n <= 64 ? (contig ? per_update_paths_kernel<true, true> : per_update_paths_kernel<true, false>)
                              : (contig ? per_update_paths_kernel<false, true> : per_update_paths_kernel<false, false>);
        RLX_LAUNCH((kernel), 1, n <= 64 ? 64 : kPathMaxLeaves, 0, rlx::as_stream(stream), sum_tree, min_tree, max_tree, capacity, ilog2(capacity), idx, err, leaf_pa, leaf_p, n, start_leaf, alpha, eps,
            max_priority, mode, status);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    RLX_LAUNCH((per_update_kernel), 1, threads, 0, rlx::as_stream(stream), sum_tree, min_tree, max_tree, capacity, ilog2(capacity), idx, err, leaf_pa, leaf_p, n,
        start_leaf, alpha, eps, max_priority, mode, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_per_tuning(int path_max_leaves) {
    RLX_REQUIRE(path_max_leaves >= 0 && path_max_leaves <= kPathMaxLeaves,
                "rlx_per_tuning: path_max_leaves %d outside [0, %d]", path_max_leaves, kPathMaxLeaves);
    g_path_max_leaves = path_max_leaves;
    return RLX_OK;
}

int rlx_per_update(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                   const int *idx, const double *errors, int n, double alpha, double epsilon,
                   double *max_priority, int *status, void *stream) {
    RLX_REQUIRE(idx && errors, "rlx_per_update: null idx/errors");
    return launch_update(sum_tree, min_tree, max_tree, capacity, idx, errors, nullptr, nullptr, n,
                         0, alpha, epsilon, max_priority, 0, status, stream, "rlx_per_update");
}

int rlx_per_update_leaves(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                          const int *idx, const double *leaf_pa, const double *leaf_p, int n,
                          double *max_priority, int *status, void *stream) {
    RLX_REQUIRE(idx && leaf_pa && leaf_p, "rlx_per_update_leaves: null idx/values");
    return launch_update(sum_tree, min_tree, max_tree, capacity, idx, nullptr, leaf_pa, leaf_p, n,
                         0, 0.0, 0.0, max_priority, 2, status, stream, "rlx_per_update_leaves");
}

int rlx_per_store(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                  int start_leaf, int n, double alpha, double *max_priority, int *status,
                  void *stream) {
    RLX_REQUIRE(start_leaf >= 0 && start_leaf < capacity, "rlx_per_store: start leaf %d out of range",
                start_leaf);
    RLX_REQUIRE(n <= capacity, "rlx_per_store: %d stores exceed the capacity %d", n, capacity);
    return launch_update(sum_tree, min_tree, max_tree, capacity, nullptr, nullptr, nullptr, nullptr,
                         n, start_leaf, alpha, 0.0, max_priority, 1, status, stream,
                         "rlx_per_store");
}

int rlx_per_store_value(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                        int start_leaf, int n, double leaf_pa, double leaf_p,
                        double *max_priority, int *status, void *stream) {
    RLX_REQUIRE(start_leaf >= 0 && start_leaf < capacity,
                "rlx_per_store_value: start leaf %d out of range", start_leaf);
    RLX_REQUIRE(n <= capacity, "rlx_per_store_value: %d stores exceed the capacity %d", n, capacity);
    return launch_update(sum_tree, min_tree, max_tree, capacity, nullptr, nullptr, nullptr, nullptr,
                         n, start_leaf, /*alpha=p*/ leaf_p, /*eps=p^alpha*/ leaf_pa, max_priority, 3,
                         status, stream, "rlx_per_store_value");
}

int rlx_libm_pow(const double *x, const double *y, double *out, int n, int *status,
                 void *stream) {
    RLX_REQUIRE(x && y && out && status && n >= 0, "rlx_libm_pow: null pointer");
    if (n == 0) return RLX_OK;
    RLX_LAUNCH((libm_pow_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), x, y, out, n,
                                                                                     status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_splitk_reduce_jobs_per_update(const rlx_splitk_job *jobs_host, int n_jobs, double *sum_tree, double *min_tree,
                                      double *max_tree, int capacity, const int *idx, const double *td_errors, int n,
                                      double alpha, double epsilon, double *max_priority, int *status, void *stream) {
    RLX_REQUIRE(jobs_host && n_jobs >= 0 && n_jobs <= RLX_MAX_SPLITK_JOBS,
                "rlx_splitk_reduce_jobs_per_update: 0..%d jobs (got %d)", RLX_MAX_SPLITK_JOBS, n_jobs);
    RLX_REQUIRE(sum_tree && min_tree && max_tree && max_priority && status && idx && td_errors,
                "rlx_splitk_reduce_jobs_per_update: null pointer");
    RLX_REQUIRE(is_pow2(capacity), "rlx_splitk_reduce_jobs_per_update: capacity %d is not a power of two", capacity);
    RLX_REQUIRE(n >= 1 && n <= 64, "rlx_splitk_reduce_jobs_per_update: 1 .. 64 leaves ride on the launch (got %d)", n);
    rlx_reduce::ReduceJobs jobs;
    int m = 0, gx = 1, gy = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const rlx_splitk_job &j = jobs_host[i];
        if (j.splits <= 1) continue;
        RLX_REQUIRE(j.partials && j.C && j.M > 0 && j.N > 0 && j.N % 4 == 0 && j.batch > 0 &&
                    (long long)j.M * j.N < (1LL << 31) && (((uintptr_t)j.partials) & 15) == 0,
                    "rlx_splitk_reduce_jobs_per_update: job %d is not a float4-reducible product", i);
        jobs.job[m++] = j;
        const int bx = (int)(((long long)j.M * j.N / 4 + 63) / 64);
        gx = bx > gx ? bx : gx;
        gy = j.batch > gy ? j.batch : gy;
    }
    if (m == 0)
        return rlx_per_update(sum_tree, min_tree, max_tree, capacity, idx, td_errors, n, alpha, epsilon, max_priority, status,
                              stream);
    PerUpdateDev u;
    u.sum = sum_tree; u.mn = min_tree; u.mx = max_tree; u.max_priority = max_priority; u.idx = idx; u.err = td_errors;
    u.status = status; u.alpha = alpha; u.eps = epsilon; u.cap = capacity; u.levels = ilog2(capacity); u.n = n;
    RLX_LAUNCH((splitk_reduce_jobs_per_kernel), dim3(gx, gy, m + 1), 1024, 0, rlx::as_stream(stream), jobs, u);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int g_sample_top_steps = 8;            // tools/per_sample_bench.py: 8 levels from LDS + 4 groups of three beat 11 + 3 (9.8 against 10.4 us)
int rlx_per_sample_top_steps(int steps) {           /* A/B: the most tree levels rlx_per_sample descends from LDS (default 8) */
    RLX_REQUIRE(steps >= 0 && steps <= kTopMaxSteps, "rlx_per_sample_top_steps: 0 .. %d", kTopMaxSteps);
    g_sample_top_steps = steps;
    return RLX_OK;
}

int rlx_per_sample(const double *sum_tree, const double *min_tree, int capacity,
                   const double *uniforms, int batch, double num_transitions, double beta,
                   int *out_idx, double *out_weight, double *out_priority, long long stored_total,
                   long long payload_rows, int *out_rows, void *stream) {
    RLX_REQUIRE(sum_tree && min_tree && uniforms && out_idx && out_weight,
                "rlx_per_sample: null pointer");
    RLX_REQUIRE(is_pow2(capacity), "rlx_per_sample: capacity %d is not a power of two", capacity);
    RLX_REQUIRE(batch > 0, "rlx_per_sample: batch must be positive");
    RLX_REQUIRE(!out_rows || (stored_total >= 0 && payload_rows >= capacity),
                "rlx_per_sample: the payload ring (%lld rows) is smaller than the tree (%d leaves)",
                payload_rows, capacity);
    int levels = 0;
    while ((1 << levels) < capacity) ++levels;
    // steps taken from LDS: the most (<= kTopMaxSteps) that leave a multiple of three for the three-level groups
    int top_steps = levels < g_sample_top_steps ? levels : g_sample_top_steps;
    while (top_steps > 0 && (levels - top_steps) % 3 != 0) --top_steps;
    const size_t lds = sizeof(double) * (size_t)((2 << top_steps) - 1);
    RLX_LAUNCH((per_sample_kernel), (batch + kSampleThreads - 1) / kSampleThreads, kSampleThreads, lds, rlx::as_stream(stream),
               sum_tree, min_tree, capacity, levels, top_steps, uniforms, batch, num_transitions, beta, out_idx, out_weight,
               out_priority, stored_total, payload_rows, out_rows);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
