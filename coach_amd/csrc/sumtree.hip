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
constexpr int kPathChunk = 20, kPathMaxLeaves = 256;
#ifdef RLX_PER_PROFILE                  // tools/per_update_profile.hip: phase timestamps of thread 0 (100 MHz counter)
__device__ long long g_per_prof[8];
#define PER_PROF(k) do { if (threadIdx.x == 0) g_per_prof[k] = wall_clock64(); } while (0)
#else
#define PER_PROF(k) do {} while (0)
#endif
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double shfl_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl((int)(b & 0xffffffffll), lane), hi = __shfl((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct __attribute__((aligned(16))) PathRec {     // what a run shows its neighbours (n > 64)
    int node, lo, hi, pad;
    double s, m, x, pad2;
};

// CONTIG: a store of n consecutive leaves that does not wrap around the ring — already in leaf order, one run per leaf
template <bool WAVE, bool CONTIG>
__global__ void __launch_bounds__(WAVE ? 64 : kPathMaxLeaves)
per_update_paths_kernel(double *__restrict__ sum, double *__restrict__ mn, double *__restrict__ mx,
                        int cap, int levels, const int *__restrict__ idx,
                        const double *__restrict__ err, const double *__restrict__ leaf_pa,
                        const double *__restrict__ leaf_p, int n, int start_leaf, double alpha,
                        double eps, double *__restrict__ max_priority, int mode,
                        int *__restrict__ status) {
    constexpr int kThreads = WAVE ? 64 : kPathMaxLeaves;
    constexpr int kNoLeaf = 0x7fffffff;
    PER_PROF(0);
    __shared__ __attribute__((aligned(16))) int key[kThreads];
    __shared__ int sorted_leaf[kThreads], sorted_lo[kThreads], sorted_hi[kThreads];
    __shared__ double sorted_a[kThreads], sorted_b[kThreads];
    __shared__ PathRec rec[WAVE ? 1 : 2][WAVE ? 1 : kPathMaxLeaves];
    const int tid = threadIdx.x;
    const double stored_priority = (mode == 1) ? *max_priority : alpha;
    const bool ring = (mode == 1 || mode == 3);
    // ---- inputs in launch order; an out-of-range index sorts behind everything and carries no node
    int leaf = kNoLeaf;
    double in_a = 0.0, in_b = 0.0;
    if (tid < n) {
        if (ring) {
            leaf = (start_leaf + tid) & (cap - 1);
        } else {
            leaf = idx[tid];
            if (mode == 0) in_a = err[tid];
            else { in_a = leaf_pa[tid]; in_b = leaf_p[tid]; }
            if (leaf < 0 || leaf >= cap) {          // reference raises ValueError (:123-126)
                atomicOr(status, 1);
                leaf = kNoLeaf;
            }
        }
    }
    constexpr bool contiguous = CONTIG;
    key[tid] = leaf;
    lds_barrier();
    // ---- order by leaf: position = number of occurrences that sort before this one (ties by launch order).  The
    // occurrences of one leaf end up as one run whose LAST position is the last occurrence — the one that wins
    // (:214-215).
    if (!contiguous) {
        int lo = 0, same = 0, before = 0;
        const int n4 = (n + 3) & ~3;                      // keys behind n are kNoLeaf
        for (int j = 0; j < n4; j += 4) {
            const int4 k = *reinterpret_cast<const int4 *>(&key[j]);          // broadcast read
            const int kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo += (kk[q] < leaf);
                same += (kk[q] == leaf);
                before += (kk[q] == leaf && j + q < tid);
            }
        }
        if (tid < n) {
            const int pos = lo + before;
            sorted_leaf[pos] = leaf;
            sorted_lo[pos] = lo;
            sorted_hi[pos] = lo + same - 1;
            sorted_a[pos] = in_a;
            sorted_b[pos] = in_b;
        }
    }
    lds_barrier();
    // ---- from here on thread t IS position t
    int node = -1;                      // heap index of the node this thread carries; -1: none
    int lo = 0, hi = 0;
    bool valid = false;                 // this occurrence may write its leaf
    if (tid < n) {
        if (contiguous) {
            lo = hi = tid;
        } else {
            leaf = sorted_leaf[tid];
            lo = sorted_lo[tid];
            hi = sorted_hi[tid];
            in_a = sorted_a[tid];
            in_b = sorted_b[tid];
        }
        valid = leaf != kNoLeaf;
        if (valid) node = leaf + cap - 1;
    }
    // the loads of every sibling the path will need on the next kPathChunk levels depend on the index alone: they
    // are issued now and the pow and the leaf level run while they are in flight
    double ps[kPathChunk], pm[kPathChunk], px[kPathChunk];
    auto prefetch = [&](int lvl0) {
        int a = node;
#pragma unroll
        for (int j = 0; j < kPathChunk; ++j) {
            ps[j] = pm[j] = px[j] = 0.0;
            if (a > 0 && lvl0 + j < levels) {
                const int sib = (a & 1) ? a + 1 : a - 1;
                // contiguous store: a sibling with a stored leaf below it is carried by a neighbouring run — its old
                // value is never used, and near the leaves that is almost every sibling (the loads that remain are the
                // two edges of the range and, higher up, one address shared by all threads)
                bool covered = false;
                if (CONTIG) {                                             // (cap <= 2^30: everything fits an int)
                    const int h = lvl0 + j;                               // height of `a` and `sib` above the leaves
                    const int p = sib + 1 - (cap >> h);                   // position within its level
                    covered = (p << h) < start_leaf + n && ((p + 1) << h) > start_leaf;
                }
                if (!covered) { ps[j] = sum[sib]; pm[j] = mn[sib]; px[j] = mx[sib]; }
                a = (a - 1) >> 1;
            }
        }
    };
    PER_PROF(1);
    prefetch(0);
    PER_PROF(2);
    double s = 0.0, m = 0.0, x = 0.0;
    if (tid < n) {
        if (ring) {
            x = stored_priority;
            int odd = 0;
            s = (mode == 1) ? rlx::libm_pow(x, alpha, &odd) : eps;     // maximal_priority ** alpha (:274)
            if (odd) atomicOr(status, 4);
        } else if (mode == 0) {
            if (in_a < 0.0) {                       // "priorities must be non-negative" (:195)
                atomicOr(status, 2);
                valid = false;
            }
            x = in_a + eps;
            int odd = 0;
            s = rlx::libm_pow(x, alpha, &odd);                        // priority ** self.alpha (:197)
            if (odd) atomicOr(status, 4);
        } else {
            s = in_a;
            x = in_b;
        }
        m = s;
    }
    asm volatile("" : "+v"(s), "+v"(x));
    PER_PROF(3);
    // ---- leaf level: the winner shows its values, the other occurrences adopt them.  A rejected winner (negative
    // error) leaves the leaf as it is; its run walks up with the leaf's present values, which recomputes its
    // ancestors to what they already are — as the kernel above does.
    {
        const bool winner = (tid == hi);
        int w_ok;
        double ws, wm, wx;
        if (WAVE) {
            w_ok = __shfl((int)valid, hi);
            ws = shfl_f64(s, hi); wm = shfl_f64(m, hi); wx = shfl_f64(x, hi);
        } else {
            if (tid < n) {
                PathRec r;
                r.node = valid ? 1 : 0; r.lo = lo; r.hi = hi; r.pad = 0;
                r.s = s; r.m = m; r.x = x; r.pad2 = 0.0;
                rec[1][tid] = r;
            }
            lds_barrier();
            const PathRec w = rec[1][tid < n ? hi : 0];
            w_ok = w.node; ws = w.s; wm = w.m; wx = w.x;
        }
        if (node >= 0) {
            if (winner && valid) {
                sum[node] = s;
                mn[node] = m;
                mx[node] = x;
            } else if (w_ok) {
                s = ws; m = wm; x = wx;
            } else {
                s = sum[node]; m = mn[node]; x = mx[node];
            }
        }
    }
    PER_PROF(4);
    for (int lvl0 = 0; lvl0 < levels; lvl0 += kPathChunk) {
        if (lvl0 > 0) prefetch(lvl0);                     // trees deeper than kPathChunk levels: one more round trip
        // All prefetched values arrive HERE.  vmcnt counts loads and stores in issue order: a wait placed inside the
        // levels below would also wait for the stores of the level before it — the round trip per level this kernel
        // exists to avoid.
#pragma unroll
        for (int j = 0; j < kPathChunk; ++j) asm volatile("" : "+v"(ps[j]), "+v"(pm[j]), "+v"(px[j]));
        PER_PROF(5);
#pragma unroll
        for (int j = 0; j < kPathChunk; ++j) {
            if (lvl0 + j < levels) {                      // uniform
                const bool left = (node & 1) != 0;        // odd heap index = left child (2p + 1)
                const int sib = left ? node + 1 : node - 1;
                const int cand = left ? hi + 1 : lo - 1;  // where the sibling's run must be, if it exists
                const bool in_range = node > 0 && cand >= 0 && cand < n;
                const int src = in_range ? cand : tid;
                int r_node, r_lo, r_hi;
                double r_s, r_m, r_x;
                if (WAVE) {
                    r_node = __shfl(node, src);
                    const int ext = __shfl(lo | (hi << 8), src);
                    r_lo = ext & 0xff; r_hi = ext >> 8;
                    r_s = shfl_f64(s, src); r_m = shfl_f64(m, src); r_x = shfl_f64(x, src);
                } else {
                    const int buf = j & 1;                // (the leaf level used buffer 1; kPathChunk is even)
                    if (tid < n) {
                        PathRec r;
                        r.node = node; r.lo = lo; r.hi = hi; r.pad = 0;
                        r.s = s; r.m = m; r.x = x; r.pad2 = 0.0;
                        rec[buf][tid] = r;
                    }
                    lds_barrier();                        // one barrier per level: the buffers alternate
                    const PathRec r = rec[buf][src];
                    r_node = r.node; r_lo = r.lo; r_hi = r.hi; r_s = r.s; r_m = r.m; r_x = r.x;
                }
                if (node > 0) {
                    double bs = ps[j], bm = pm[j], bx = px[j];
                    if (in_range && r_node == sib) {
                        bs = r_s; bm = r_m; bx = r_x;
                        if (left) hi = r_hi; else lo = r_lo;
                    }
                    // (left, right) order does not matter to the result: IEEE addition commutes, and python's
                    // min(a, b) = b if b < a else a / max(a, b) = b if b > a else a pick between two values that are
                    // bit-identical whenever neither is smaller (priorities are positive; a NaN sets status bit 4).
                    node = (node - 1) >> 1;
                    s = s + bs;                           // operator.add (:57)
                    m = (bm < m) ? bm : m;
                    x = (bx > x) ? bx : x;
                    if (tid == lo) {                      // one thread per distinct node writes it to the tree
                        sum[node] = s;
                        mn[node] = m;
                        mx[node] = x;
                    }
                }
            }
        }
    }
    PER_PROF(6);
    // every path ends at the root with the root's values in registers: maximal_priority = max_tree root (:201)
    if (!ring && node == 0 && tid == lo) *max_priority = x;
}

// One draw per thread.  u[i] is CPython's random.random() drawn on the host, so that
// val == random.uniform(seg*i, seg*(i+1)) bit for bit (:240-244).
__global__ void per_sample_kernel(const double *__restrict__ sum, const double *__restrict__ mn,
                                  int cap, const double *__restrict__ u, int batch,
                                  double n_transitions, double beta, int *__restrict__ out_idx,
                                  double *__restrict__ out_weight,
                                  double *__restrict__ out_priority, long long stored_total,
                                  long long payload_rows, int *__restrict__ out_rows) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const int n_nodes = 2 * cap - 1;
    const double total = sum[0];
    const double segment_size = total / (double)batch;                   // (:232)
    const double min_probability = mn[0] / total;                        // (:235)
    int odd = 0;                                  // (an empty tree gives inf/nan here, as in the reference)
    const double max_weight = rlx::libm_pow(min_probability * n_transitions, -beta, &odd);  // (:236)
    const double a = segment_size * (double)i;                           // (:240)
    const double b = segment_size * (double)(i + 1);                     // (:241)
    double val = a + (b - a) * u[i];                                     // random.uniform
    int node = 0;
    while (true) {                                                       // _retrieve (:76-92)
        int left = 2 * node + 1;
        if (left >= n_nodes) break;
        double tl = sum[left];
        if (val <= tl) {
            node = left;
        } else {
            val = val - tl;
            node = left + 1;
        }
    }
    const double leaf_value = sum[node];
    const double priority = leaf_value / total;                          // (:247)
    const double weight = rlx::libm_pow(n_transitions * priority, -beta, &odd);   // (:248)
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
    int threads = 64;
    RLX_LAUNCH((per_sample_kernel), (batch + threads - 1) / threads, threads, 0, rlx::as_stream(stream), sum_tree, min_tree, capacity, uniforms, batch, num_transitions, beta, out_idx, out_weight,
        out_priority, stored_total, payload_rows, out_rows);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
