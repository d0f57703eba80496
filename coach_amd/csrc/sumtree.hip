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

// The same update for n <= 1024 leaves with ONE memory round trip per FIVE tree levels instead of one per level (the
// level-synchronous kernel above is a chain of `levels` dependent load -> store -> barrier steps: 17 us for 64 leaves of
// a 2^20-leaf tree, 23 % of the C3 GPU time in profiles/r03_call8_c3_kernel_stats.csv).
// Thread i carries the node of leaf i that sits on the current level, with its three values, in registers.  The other
// child of the node's parent either lies on another updated leaf's path — then some thread holds its NEW value, found
// through a per-level hash table in LDS (tag = node id, linear probing; threads that reached the same node computed
// the same values, the first inserts) — or it is untouched by this launch and its value was prefetched, five levels
// at a time, before the walk reached it.  Parents are recomputed from (left, right) exactly like _propagate (:63-74):
// bit-identical trees.  Every thread stores the nodes it computes (duplicates store identical values).
constexpr int kPathSlots = 2048, kPathChunk = 5;
__global__ void __launch_bounds__(1024)
per_update_paths_kernel(double *__restrict__ sum, double *__restrict__ mn, double *__restrict__ mx,
                        int cap, int levels, const int *__restrict__ idx,
                        const double *__restrict__ err, const double *__restrict__ leaf_pa,
                        const double *__restrict__ leaf_p, int n, int start_leaf, double alpha,
                        double eps, double *__restrict__ max_priority, int mode,
                        int *__restrict__ status) {
    __shared__ int tag[kPathSlots];
    __shared__ double vs[kPathSlots], vm[kPathSlots], vx[kPathSlots];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const double stored_priority = (mode == 1) ? *max_priority : alpha;
    const bool ring = (mode == 1 || mode == 3);
    // ---- the leaf of this thread (n <= blockDim.x)
    int node = -1;                      // -1: no valid leaf
    bool live = false;                  // this thread's occurrence of the leaf is the one that counts
    double s = 0.0, m = 0.0, x = 0.0;
    if (tid < n) {
        int leaf;
        live = true;
        if (ring) {
            leaf = (start_leaf + tid) & (cap - 1);
            x = stored_priority;
            int odd = 0;
            s = (mode == 1) ? rlx::libm_pow(x, alpha, &odd) : eps;     // maximal_priority ** alpha (:274)
            if (odd) atomicOr(status, 4);
        } else {
            leaf = idx[tid];
            if (leaf < 0 || leaf >= cap) {          // reference raises ValueError (:123-126)
                atomicOr(status, 1);
                live = false;
            }
            if (mode == 0) {
                const double e = err[tid];
                if (e < 0.0) {                      // "priorities must be non-negative" (:195)
                    atomicOr(status, 2);
                    live = false;
                }
                x = e + eps;
                int odd = 0;
                s = rlx::libm_pow(x, alpha, &odd);                    // priority ** self.alpha (:197)
                if (odd) atomicOr(status, 4);
            } else {
                s = leaf_pa[tid];
                x = leaf_p[tid];
            }
        }
        m = s;
        if (leaf >= 0 && leaf < cap) {
            node = leaf + cap - 1;
            // last occurrence of a duplicated index wins (:214-215); an earlier one adopts the winner's values below
            if (!ring)
                for (int j = tid + 1; live && j < n; ++j)
                    if (idx[j] == leaf) live = false;
        }
        if (live) {
            sum[node] = s;
            mn[node] = m;
            mx[node] = x;
        }
    }
    auto clear = [&]() {
        for (int i = tid; i < kPathSlots; i += nthr) tag[i] = -1;
    };
    auto insert = [&](int key, double a, double b, double c) {
        int slot = key & (kPathSlots - 1);
        while (true) {
            const int prev = atomicCAS(&tag[slot], -1, key);
            if (prev == -1) {
                vs[slot] = a; vm[slot] = b; vx[slot] = c;
                return;
            }
            if (prev == key) return;
            slot = (slot + 1) & (kPathSlots - 1);
        }
    };
    auto find = [&](int key) {
        int slot = key & (kPathSlots - 1);
        while (tag[slot] != -1) {
            if (tag[slot] == key) return slot;
            slot = (slot + 1) & (kPathSlots - 1);
        }
        return -1;
    };
    // level 0: publish the winning occurrences, the others read their leaf's final values back
    clear();
    __syncthreads();
    if (live) insert(node, s, m, x);
    __syncthreads();
    if (node >= 0 && !live) {
        const int slot = find(node);
        if (slot >= 0) { s = vs[slot]; m = vm[slot]; x = vx[slot]; }
        else node = -1;                                   // every occurrence of this leaf was rejected
    }
    for (int lvl0 = 0; lvl0 < levels; lvl0 += kPathChunk) {
        // siblings of this thread's ancestors on the next five levels: one round trip
        double ps[kPathChunk], pm[kPathChunk], px[kPathChunk];
        {
            int a = node;
#pragma unroll
            for (int j = 0; j < kPathChunk; ++j) {
                ps[j] = pm[j] = px[j] = 0.0;
                if (a > 0 && lvl0 + j < levels) {
                    const int sib = (a & 1) ? a + 1 : a - 1;
                    ps[j] = sum[sib]; pm[j] = mn[sib]; px[j] = mx[sib];
                    a = (a - 1) >> 1;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kPathChunk; ++j) {
            if (lvl0 + j >= levels) break;
            __syncthreads();                              // the previous level's lookups are done
            clear();
            __syncthreads();
            if (node > 0) insert(node, s, m, x);
            __syncthreads();
            if (node > 0) {
                const int sib = (node & 1) ? node + 1 : node - 1;
                double bs = ps[j], bm = pm[j], bx = px[j];
                const int slot = find(sib);
                if (slot >= 0) { bs = vs[slot]; bm = vm[slot]; bx = vx[slot]; }
                const bool left = (node & 1) != 0;        // odd heap index = left child (2p + 1)
                const double sl = left ? s : bs, sr = left ? bs : s;
                const double ml = left ? m : bm, mr = left ? bm : m;
                const double xl = left ? x : bx, xr = left ? bx : x;
                const int parent = (node - 1) >> 1;
                s = sl + sr;                              // operator.add (:57)
                m = (mr < ml) ? mr : ml;                  // python min(a,b): b if b<a else a
                x = (xr > xl) ? xr : xl;                  // python max(a,b): b if b>a else a
                sum[parent] = s;
                mn[parent] = m;
                mx[parent] = x;
                node = parent;
            }
        }
    }
    // every valid path ends at the root with the root's values in registers: maximal_priority = max_tree root (:201)
    if (!ring && node == 0) *max_priority = x;
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
    per_init_kernel<<<rlx::grid_for(n_nodes, kBlock), kBlock, 0, s>>>(sum_tree, min_tree,
                                                                      max_tree, n_nodes);
    RLX_LAUNCH_CHECK();
    const double one = 1.0;   // self.maximal_priority = 1.0 (:186)
    RLX_HIP(hipMemcpyAsync(max_priority, &one, sizeof(double), hipMemcpyHostToDevice, s));
    return RLX_OK;
}

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
    if (n <= 1024) {          // one leaf per thread: the path walk with batched round trips
        per_update_paths_kernel<<<1, threads, 0, rlx::as_stream(stream)>>>(
            sum_tree, min_tree, max_tree, capacity, ilog2(capacity), idx, err, leaf_pa, leaf_p, n,
            start_leaf, alpha, eps, max_priority, mode, status);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    per_update_kernel<<<1, threads, 0, rlx::as_stream(stream)>>>(
        sum_tree, min_tree, max_tree, capacity, ilog2(capacity), idx, err, leaf_pa, leaf_p, n,
        start_leaf, alpha, eps, max_priority, mode, status);
    RLX_LAUNCH_CHECK();
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
    libm_pow_kernel<<<(n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream)>>>(x, y, out, n,
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
    per_sample_kernel<<<(batch + threads - 1) / threads, threads, 0, rlx::as_stream(stream)>>>(
        sum_tree, min_tree, capacity, uniforms, batch, num_transitions, beta, out_idx, out_weight,
        out_priority, stored_total, payload_rows, out_rows);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
