// The deferred split-K reduction's device body (rlx_splitk_reduce_jobs: gemm.hip), shared with sumtree.hip, where the
// prioritized replay's priority update rides on the same launch (rlx_splitk_reduce_jobs_per_update).  fp32 additions only:
// the result does not depend on the translation unit's -ffp-contract setting.
#pragma once
#include "rlx_common.hpp"

namespace rlx_reduce {

// blockIdx.y = batch entry of the job, blockIdx.x = 64 float4 output groups; the scheme of splitk_reduce4_kernel<16>
// (16 split groups per output group, partials of group q summed in increasing split order, groups combined in the
// fixed order ((s0 + s1) + s2) + ...).  Plain store: a weight gradient has no epilogue.
struct ReduceJobs {
    rlx_splitk_job job[RLX_MAX_SPLITK_JOBS];
};
__device__ __forceinline__ void splitk_reduce_job_body(const rlx_splitk_job &g, float4 (*part)[64]) {
    constexpr int SG = 16;
    const int batch = blockIdx.y;
    if (batch >= g.batch) return;
    const int mn4 = (g.M * g.N) >> 2;
    if ((int)blockIdx.x * 64 >= mn4 && !(blockIdx.x == 0 && g.colsum_out)) return;
    const int ox = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int gid = blockIdx.x * 64 + ox;
    const size_t mn = (size_t)g.M * g.N;
    const float4 *ws = reinterpret_cast<const float4 *>(g.partials + (size_t)batch * g.splits * mn);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gid < mn4) {
        int k = q;
        for (; k + 3 * SG < g.splits; k += 4 * SG) {    // 4 independent loads in flight per thread
            const float4 a = ws[(size_t)k * mn4 + gid];
            const float4 b = ws[(size_t)(k + SG) * mn4 + gid];
            const float4 c = ws[(size_t)(k + 2 * SG) * mn4 + gid];
            const float4 d = ws[(size_t)(k + 3 * SG) * mn4 + gid];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        }
        for (; k < g.splits; k += SG) {
            const float4 a = ws[(size_t)k * mn4 + gid];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    part[q][ox] = s;
    __syncthreads();
    if (q == 0 && gid < mn4) {
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int t = 1; t < SG; ++t) {
            const float4 p = part[t][ox];
            v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
        }
        const int i = gid << 2;
        const int row = i / g.N, col = i - row * g.N;
        const int tw = g.n_fold ? col / g.n_fold : batch, cl = g.n_fold ? col % g.n_fold : col;
        float *c = g.C + (size_t)tw * g.c_batch_stride + (size_t)row * g.ldc + cl;
        if ((g.ldc & 3) == 0 && (((uintptr_t)c) & 15) == 0) {
            *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3];
        }
    }
    if (g.colsum_out && blockIdx.x == 0) {
        float *cpart = reinterpret_cast<float *>(&part[0][0]);      // [SG][64] floats
        for (int n0 = 0; n0 < g.N; n0 += 64) {
            __syncthreads();
            const int n = n0 + ox;
            float t = 0.f;
            if (n < g.N)
                for (int k = q; k < g.splits; k += SG)
                    t += g.colsum_partials[((size_t)batch * g.splits + k) * g.N + n];
            cpart[q * 64 + ox] = t;
            __syncthreads();
            if (q == 0 && n < g.N) {
#pragma unroll
                for (int u = 1; u < SG; ++u) t += cpart[u * 64 + ox];
                if (g.n_fold)
                    g.colsum_out[(size_t)(n / g.n_fold) * g.colsum_batch_stride + n % g.n_fold] = t;
                else
                    g.colsum_out[(size_t)batch * g.colsum_batch_stride + n] = t;
            }
        }
    }
}

}  // namespace rlx_reduce
