// Device body of the prioritized replay's priority update for n <= 256 leaves (rlx_per_update; the algorithm is described in
// sumtree.hip, in front of per_update_paths_kernel), in a header so that the update can ride as one more workgroup on a launch
// of another translation unit: rlx_splitk_reduce_jobs_per_update (sumtree.hip) and rlx_conv32_input_grad's rider
// (conv_bwd_fused.hip).  Every p ** alpha goes through rlx::libm_pow, whose fusions are explicit: a translation unit that is
// not compiled with -ffp-contract=off includes this header (and libm_pow.hpp in front of it) under
// `#pragma clang fp contract(off)`.
#pragma once
#include "rlx_common.hpp"
#include "libm_pow.hpp"

namespace rlx_per {

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
__device__ __forceinline__ void
per_update_paths_body(double *__restrict__ sum, double *__restrict__ mn, double *__restrict__ mx,
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

}  // namespace rlx_per
