// K11 — optimiser, target-network mixing and gradient-norm kernels over FLAT parameter buffers.
//
// Every network keeps its weights, gradients and Adam moments in one contiguous fp32 buffer each,
// so an update is one launch (and the data-parallel gradient exchange one RCCL all-reduce).
//
// Replaces, in the reference (paths under rl_coach/architectures/tensorflow_components/):
//   * tf.train.AdamOptimizer       general_network.py:390-394 (beta1/beta2/epsilon from
//     base_parameters.py:297-299).  TensorFlow 1.x is a pip dependency absent from
//     /root/reference (setup.py:69): the update is restated from TF 1.14's published ApplyAdam
//     kernel (tensorflow/core/kernels/training_ops.cc):
//         alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
//         m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2)
//         var -= (m * alpha) / (sqrt(v) + epsilon)          ("epsilon hat" outside the sqrt)
//     with beta^t carried as fp32 running products (beta1_power / beta2_power variables).
//   * TensorFlowArchitecture.set_weights    architecture.py:598-607
//         w_target = rate * w_online + (1 - rate) * w_target   (done on the HOST in the reference)
//   * tf.global_norm of the gradients       architecture.py:194
//   * apply_gradients' optional 1/num_workers scaling   architecture.py:485-488
//
// HBM-bound: Adam reads g, m, v, w and writes m, v, w = 28 B/parameter; mixing 12 B/parameter.
// float4 accesses, grid-stride, <= 2048 workgroups.  Compiled with -ffp-contract=off so that the
// mixing rounds like numpy's two products + add.
#include "rlx_common.hpp"

namespace {

constexpr int kBlock = 256;

// The fence-free publication below (partial sums stored at agent scope, completed by s_waitcnt / by the return of an
// exchange before the ticket is drawn) leans on how gfx942 / gfx950 perform agent-scope stores and atomics at the memory
// side; no other target may compile it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "optim.hip: the last-arriver ticket protocol is written for gfx950 (gfx942-compatible) only"
#endif

// "Am I the last workgroup of this launch?" without 1024 read-modify-writes on ONE word: an agent-scope atomic on one
// address is performed at the memory side, one after the other (measured ~10 ns each: 1024 of them were +10 us on the C2
// Adam launch, profiles/r04_ab_adam_norm_in_kernel.txt).  Two levels: workgroup b draws from group b % 32's word (32
// words on separate 128-byte lines, in parallel), the last of a group draws from the top word — 32 + 32 serialised
// operations instead of 1024.  Whatever a workgroup made visible at agent scope before its draw is visible to the
// workgroup this returns true for (the draws are chained through their return values).  Re-arms itself.
// ticket: RLX_ADAM_TICKET_WORDS zero-initialised 32-bit words.
constexpr unsigned kTicketGroups = 32, kTicketStride = 32;
static_assert(kTicketStride * (1 + kTicketGroups) == RLX_ADAM_TICKET_WORDS, "rlx.h: RLX_ADAM_TICKET_WORDS");
__device__ __forceinline__ bool draw_last_ticket(unsigned int *ticket) {
    const unsigned G = gridDim.x < kTicketGroups ? gridDim.x : kTicketGroups;
    const unsigned grp = blockIdx.x % G, gsize = (gridDim.x - grp + G - 1) / G;
    unsigned int *gw = ticket + kTicketStride * (1 + grp);
    if (__hip_atomic_fetch_add(gw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gsize - 1) return false;
    __hip_atomic_store(gw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != G - 1) return false;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// state[0] = beta1_power, state[1] = beta2_power (fp32, device).
template <bool NORM>
__global__ void adam_tf1_kernel(float *__restrict__ w, const float *__restrict__ g,
                                float *__restrict__ m, float *__restrict__ v, long long n,
                                float lr, float beta1, float beta2, float eps,
                                const float *__restrict__ state, float grad_scale,
                                float *__restrict__ sumsq_part) {
    __shared__ float red[kBlock];
    float ss = 0.f;
    const float b1p = state[0], b2p = state[1];
    const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 gw = reinterpret_cast<const float4 *>(g)[i];
        float4 mw = reinterpret_cast<float4 *>(m)[i];
        float4 vw = reinterpret_cast<float4 *>(v)[i];
        float4 ww = reinterpret_cast<float4 *>(w)[i];
        float *gp = &gw.x, *mp = &mw.x, *vp = &vw.x, *wp = &ww.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NORM) ss += gp[k] * gp[k];
            const float gr = gp[k] * grad_scale;
            mp[k] += (gr - mp[k]) * omb1;
            vp[k] += (gr * gr - vp[k]) * omb2;
            wp[k] -= (mp[k] * alpha) / (sqrtf(vp[k]) + eps);
        }
        reinterpret_cast<float4 *>(m)[i] = mw;
        reinterpret_cast<float4 *>(v)[i] = vw;
        reinterpret_cast<float4 *>(w)[i] = ww;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NORM) ss += g[i] * g[i];
        const float gr = g[i] * grad_scale;
        m[i] += (gr - m[i]) * omb1;
        v[i] += (gr * gr - v[i]) * omb2;
        w[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
    }
    if (NORM) {
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int d = kBlock >> 1; d > 0; d >>= 1) {
            if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) sumsq_part[blockIdx.x] = red[0];
    }
}


// Adam (+ per-block partial sums of squares for tf.global_norm) (+ the soft target update in the same elementwise pass).
// Without the norm the beta-power advance needs no second launch: the block that finishes LAST (a relaxed agent-scope
// ticket — nothing is published, so no fence) writes the advanced powers; every block read them at its start and has
// finished.  With the norm the partial sums have to be PUBLISHED to that block.  A release fence behind 28 B/parameter
// of stores costs far more than the finish launch it saves (measured: C2 +29 us per update, C3 +68 us,
// gpurun_out/r02_call24); round 4 publishes them without one: each partial is ONE agent-scope store (write-through),
// waited for (vmcnt) before the ticket is drawn, and the last block reads the partials with agent-scope loads.
// TICKET = false keeps the two-launch form (adam_finish_norm_kernel), which rlx_adam_norm_in_kernel(0) selects.
template <bool NORM, bool MIX, bool TICKET>
__global__ void __launch_bounds__(kBlock)
adam_step_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                 long long n, float lr, float beta1, float beta2, float eps, float *state, float grad_scale,
                 float *sumsq_part, float *__restrict__ target, float rate, float one_minus_rate,
                 unsigned int *ticket, float *norm_out, const float *acc_src, float *acc_dst, int n_acc) {
    __shared__ float red[kBlock];
    __shared__ int last_s;
    float ss = 0.f;
    const float b1p = state[0], b2p = state[1];
    const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 gw = reinterpret_cast<const float4 *>(g)[i];
        float4 mw = reinterpret_cast<float4 *>(m)[i];
        float4 vw = reinterpret_cast<float4 *>(v)[i];
        float4 ww = reinterpret_cast<float4 *>(w)[i];
        float4 tw = MIX ? reinterpret_cast<float4 *>(target)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float *gp = &gw.x, *mp = &mw.x, *vp = &vw.x, *wp = &ww.x, *tp = &tw.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NORM) ss += gp[k] * gp[k];
            const float gr = gp[k] * grad_scale;
            mp[k] += (gr - mp[k]) * omb1;
            vp[k] += (gr * gr - vp[k]) * omb2;
            wp[k] -= (mp[k] * alpha) / (sqrtf(vp[k]) + eps);
            if (MIX) tp[k] = rate * wp[k] + one_minus_rate * tp[k];
        }
        reinterpret_cast<float4 *>(m)[i] = mw;
        reinterpret_cast<float4 *>(v)[i] = vw;
        reinterpret_cast<float4 *>(w)[i] = ww;
        if (MIX) reinterpret_cast<float4 *>(target)[i] = tw;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NORM) ss += g[i] * g[i];
        const float gr = g[i] * grad_scale;
        m[i] += (gr - m[i]) * omb1;
        v[i] += (gr * gr - v[i]) * omb2;
        w[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
        if (MIX) target[i] = rate * w[i] + one_minus_rate * target[i];
    }
    if (NORM) {
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int d = kBlock >> 1; d > 0; d >>= 1) {
            if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            // with the ticket: an agent-scope store (written through to the memory side, past this XCD's L2) ...
            if (TICKET) __hip_atomic_store(&sumsq_part[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sumsq_part[blockIdx.x] = red[0];
        }
    }
    if (!TICKET) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // ... that has completed before the ticket is drawn: the partial sums are published without a release fence
        // (a fence here would first write back every dirty line of the 28 B/parameter this launch stores)
        if (NORM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool last = draw_last_ticket(ticket);
        if (last && !NORM) {
            state[0] = b1p * beta1;       // AdamOptimizer._finish: beta1_power *= beta1
            state[1] = b2p * beta2;
        }
        last_s = last;
    }
    if (!NORM) return;
    __syncthreads();
    if (!last_s) return;
    // the one workgroup that reads the other workgroups' partial sums pairs their publication with an agent-scope acquire
    // (one invalidate on one workgroup; the expensive release side stays elided, see above)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // adam_finish_norm_kernel's arithmetic, in its order, by the workgroup that drew the last ticket (agent-scope loads:
    // the partials come from the memory side, not from a stale line of this XCD's L2)
    float s = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock)
        s += __hip_atomic_load(&sumsq_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norm_out[0] = sqrtf(red[0]);
        state[0] = b1p * beta1;
        state[1] = b2p * beta2;
        for (int i = 0; i < n_acc; ++i) acc_dst[i] += acc_src[i];
    }
}

// Adam + tf.global_norm + the finish in ONE launch, for buffers a grid can hold in registers (<= kNormIt float4 per array
// and thread).  Why not adam_step_kernel<true, *, true>: there a block's sum of squares is complete only behind its
// stores, and waiting for the published partial (vmcnt is one in-order counter) means waiting for those stores to drain
// before the block gives up its slot — that form only breaks even with the finish launch it saves (212.3 vs 212.5 us per
// C2 update, profiles/r04_ab_adam_norm_in_kernel.txt).  Here every load is issued first, the sum of squares is complete
// BEFORE the first store, and it is published by an agent-scope atomic exchange issued in front of the stores; at the
// end of the block the exchange has long returned (the ticket draw is made to depend on its return value), and the
// block that draws the last ticket does adam_finish_norm_kernel's arithmetic in its order: 210.2 us per update.
constexpr int kNormIt = 4;
template <bool MIX>
__global__ void __launch_bounds__(kBlock)
adam_step_norm_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                      long long n, float lr, float beta1, float beta2, float eps, float *state, float grad_scale,
                      float *sumsq_part, float *__restrict__ target, float rate, float one_minus_rate,
                      unsigned int *ticket, float *norm_out, const float *acc_src, float *acc_dst, int n_acc) {
    __shared__ float red[kBlock];
    __shared__ int last_s;
    const float b1p = state[0], b2p = state[1];
    const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 gw[kNormIt], mw[kNormIt], vw[kNormIt], ww[kNormIt], tw[kNormIt];
#pragma unroll
    for (int it = 0; it < kNormIt; ++it) {
        const long long i = i0 + it * stride;
        if (i < n4) {
            gw[it] = reinterpret_cast<const float4 *>(g)[i];
            mw[it] = reinterpret_cast<float4 *>(m)[i];
            vw[it] = reinterpret_cast<float4 *>(v)[i];
            ww[it] = reinterpret_cast<float4 *>(w)[i];
            if (MIX) tw[it] = reinterpret_cast<float4 *>(target)[i];
        }
    }
    const long long it0 = (n4 << 2) + i0;            // the (at most 3) elements behind the last float4: one thread each
    const bool has_tail = it0 < n;
    const float gt = has_tail ? g[it0] : 0.f;
    float ss = 0.f;                                   // the additions of adam_step_kernel<true>, in its order
#pragma unroll
    for (int it = 0; it < kNormIt; ++it)
        if (i0 + it * stride < n4) {
            const float *gp = &gw[it].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) ss += gp[k] * gp[k];
        }
    if (has_tail) ss += gt * gt;
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    float old = 0.f;
    if (threadIdx.x == 0)
        old = __hip_atomic_exchange(&sumsq_part[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int it = 0; it < kNormIt; ++it) {
        const long long i = i0 + it * stride;
        if (i < n4) {
            float *gp = &gw[it].x, *mp = &mw[it].x, *vp = &vw[it].x, *wp = &ww[it].x, *tp = &tw[it].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gr = gp[k] * grad_scale;
                mp[k] += (gr - mp[k]) * omb1;
                vp[k] += (gr * gr - vp[k]) * omb2;
                wp[k] -= (mp[k] * alpha) / (sqrtf(vp[k]) + eps);
                if (MIX) tp[k] = rate * wp[k] + one_minus_rate * tp[k];
            }
            reinterpret_cast<float4 *>(m)[i] = mw[it];
            reinterpret_cast<float4 *>(v)[i] = vw[it];
            reinterpret_cast<float4 *>(w)[i] = ww[it];
            if (MIX) reinterpret_cast<float4 *>(target)[i] = tw[it];
        }
    }
    if (has_tail) {
        const float gr = gt * grad_scale;
        m[it0] += (gr - m[it0]) * omb1;
        v[it0] += (gr * gr - v[it0]) * omb2;
        w[it0] -= (m[it0] * alpha) / (sqrtf(v[it0]) + eps);
        if (MIX) target[it0] = rate * w[it0] + one_minus_rate * target[it0];
    }
    if (threadIdx.x == 0) {
        asm volatile("" ::"v"(old) : "memory");        // the draw is not issued before the exchange has returned
        last_s = draw_last_ticket(ticket);
    }
    __syncthreads();
    if (!last_s) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // pairs the publication of the partial sums (one workgroup only)
    float s = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock)
        s += __hip_atomic_load(&sumsq_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norm_out[0] = sqrtf(red[0]);
        state[0] = b1p * beta1;
        state[1] = b2p * beta2;
        for (int i = 0; i < n_acc; ++i) acc_dst[i] += acc_src[i];
    }
}

__global__ void adam_advance_kernel(float *state, float beta1, float beta2) {
    state[0] *= beta1;       // AdamOptimizer._finish: beta1_power *= beta1
    state[1] *= beta2;
}

// advance the beta powers AND finish the gradient norm (one workgroup, fixed summation order)
__global__ void adam_finish_norm_kernel(float *state, float beta1, float beta2,
                                        const float *__restrict__ part, int nparts,
                                        float *__restrict__ norm_out, const float *acc_src,
                                        float *acc_dst, int n_acc) {
    __shared__ float red[kBlock];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += kBlock) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norm_out[0] = sqrtf(red[0]);
        state[0] *= beta1;
        state[1] *= beta2;
        // running sums of the per-update signals (loss terms + the norm just written) for the
        // per-epoch means the agent logs: saves a separate elementwise launch per update
        for (int i = 0; i < n_acc; ++i) acc_dst[i] += acc_src[i];
    }
}

__global__ void mix_kernel(float *__restrict__ target, const float *__restrict__ online, long long n,
                           float rate, float one_minus_rate) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 o = reinterpret_cast<const float4 *>(online)[i];
        float4 t = reinterpret_cast<float4 *>(target)[i];
        t.x = rate * o.x + one_minus_rate * t.x;
        t.y = rate * o.y + one_minus_rate * t.y;
        t.z = rate * o.z + one_minus_rate * t.z;
        t.w = rate * o.w + one_minus_rate * t.w;
        reinterpret_cast<float4 *>(target)[i] = t;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        target[i] = rate * online[i] + one_minus_rate * target[i];
}

// sum of squares, two deterministic stages: per-workgroup partials then one workgroup.
__global__ void sumsq_partial_kernel(const float *__restrict__ x, long long n, float *__restrict__ part) {
    __shared__ float red[kBlock];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        s += x[i] * x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void sumsq_final_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out) {
    __shared__ float red[kBlock];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += kBlock) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = kBlock >> 1; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sqrtf(red[0]);
}

__global__ void clip_by_global_norm_kernel(float *__restrict__ g, long long n, const float *__restrict__ norm,
                                           float clip) {
    const float scale = clip * fminf(1.0f / *norm, 1.0f / clip);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= scale;
}

int g_norm_in_kernel = 2;       // rlx_adam_norm_in_kernel
}  // namespace

extern "C" {

int rlx_adam_init(float *m, float *v, long long n, float *state, float beta1, float beta2,
                  void *stream) {
    RLX_REQUIRE(m && v && state && n > 0, "rlx_adam_init: bad arguments");
    hipStream_t s = rlx::as_stream(stream);
    RLX_HIP(hipMemsetAsync(m, 0, sizeof(float) * n, s));
    RLX_HIP(hipMemsetAsync(v, 0, sizeof(float) * n, s));
    const float init[2] = {beta1, beta2};       // beta powers start at beta^1
    RLX_HIP(hipMemcpyAsync(state, init, sizeof(init), hipMemcpyHostToDevice, s));
    return RLX_OK;
}

int rlx_adam_tf1(float *weights, const float *grads, float *m, float *v, long long n,
                 float learning_rate, float beta1, float beta2, float epsilon, float *state,
                 float grad_scale, void *stream) {
    RLX_REQUIRE(weights && grads && m && v && state, "rlx_adam_tf1: null pointer");
    RLX_REQUIRE(n > 0, "rlx_adam_tf1: empty parameter buffer");
    RLX_REQUIRE((((uintptr_t)weights | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "rlx_adam_tf1: buffers must be 16-byte aligned");
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((adam_tf1_kernel<false>), rlx::grid_for(n / 4 + 1, kBlock), kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1, beta2, epsilon, state, grad_scale, nullptr);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((adam_advance_kernel), 1, 1, 0, s, state, beta1, beta2);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_adam_tf1_norm(float *weights, const float *grads, float *m, float *v, long long n,
                      float learning_rate, float beta1, float beta2, float epsilon, float *state,
                      float grad_scale, float *norm_out, float *workspace, long long workspace_floats,
                      const float *acc_src, float *acc_dst, int n_acc, void *stream) {
    RLX_REQUIRE(weights && grads && m && v && state && norm_out && workspace,
                "rlx_adam_tf1_norm: null pointer");
    RLX_REQUIRE(n > 0, "rlx_adam_tf1_norm: empty parameter buffer");
    RLX_REQUIRE(n_acc == 0 || (acc_src && acc_dst && n_acc > 0 && n_acc <= 64),
                "rlx_adam_tf1_norm: bad signal accumulation arguments");
    RLX_REQUIRE((((uintptr_t)weights | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "rlx_adam_tf1_norm: buffers must be 16-byte aligned");
    int blocks = rlx::grid_for(n / 4 + 1, kBlock, 1024);
    if (blocks > workspace_floats) blocks = (int)workspace_floats;
    RLX_REQUIRE(blocks >= 1, "rlx_adam_tf1_norm: workspace too small");
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((adam_tf1_kernel<true>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1, beta2,
                                                    epsilon, state, grad_scale, workspace);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((adam_finish_norm_kernel), 1, kBlock, 0, s, state, beta1, beta2, workspace, blocks, norm_out, acc_src,
                                                 acc_dst, n_acc);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_adam_norm_in_kernel(int on) {
    g_norm_in_kernel = on;
    return RLX_OK;
}

int rlx_adam_tf1_step(float *weights, const float *grads, float *m, float *v, long long n, float learning_rate,
                      float beta1, float beta2, float epsilon, float *state, float grad_scale, float *norm_out,
                      float *workspace, long long workspace_floats, const float *acc_src, float *acc_dst, int n_acc,
                      float *target, double mix_rate, unsigned int *ticket, void *stream) {
    RLX_REQUIRE(weights && grads && m && v && state && ticket, "rlx_adam_tf1_step: null pointer");
    RLX_REQUIRE(n > 0, "rlx_adam_tf1_step: empty parameter buffer");
    RLX_REQUIRE(!norm_out || workspace, "rlx_adam_tf1_step: the gradient norm needs a workspace");
    RLX_REQUIRE(n_acc == 0 || (acc_src && acc_dst && n_acc > 0 && n_acc <= 64),
                "rlx_adam_tf1_step: bad signal accumulation arguments");
    RLX_REQUIRE((((uintptr_t)weights | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v | (uintptr_t)target) & 15) == 0,
                "rlx_adam_tf1_step: buffers must be 16-byte aligned");
    int blocks = rlx::grid_for(n / 4 + 1, kBlock, 1024);
    if (norm_out && blocks > workspace_floats) blocks = (int)workspace_floats;
    RLX_REQUIRE(blocks >= 1, "rlx_adam_tf1_step: workspace too small");
    hipStream_t s = rlx::as_stream(stream);
    const float rate = (float)mix_rate, omr = (float)(1.0 - mix_rate);
#define RLX_ADAM_STEP(NORM, MIX, TICKET)                                                                      \
    RLX_LAUNCH((adam_step_kernel<NORM, MIX, TICKET>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1,  \
                                                                  beta2, epsilon, state, grad_scale, workspace, \
                                                                  target, rate, omr, ticket, nullptr, nullptr, nullptr, 0)
    // one launch with the norm as well (the finish by the last workgroup, see adam_step_kernel)
    if (norm_out && g_norm_in_kernel == 2 && (n >> 2) <= (long long)blocks * kBlock * kNormIt) {
        if (target)
            RLX_LAUNCH((adam_step_norm_kernel<true>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1, beta2,
                       epsilon, state, grad_scale, workspace, target, rate, omr, ticket, norm_out, acc_src, acc_dst, n_acc);
        else
            RLX_LAUNCH((adam_step_norm_kernel<false>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1, beta2,
                       epsilon, state, grad_scale, workspace, target, rate, omr, ticket, norm_out, acc_src, acc_dst, n_acc);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    if (norm_out && g_norm_in_kernel == 1) {
        if (target)
            RLX_LAUNCH((adam_step_kernel<true, true, true>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1,
                       beta2, epsilon, state, grad_scale, workspace, target, rate, omr, ticket, norm_out, acc_src, acc_dst, n_acc);
        else
            RLX_LAUNCH((adam_step_kernel<true, false, true>), blocks, kBlock, 0, s, weights, grads, m, v, n, learning_rate, beta1,
                       beta2, epsilon, state, grad_scale, workspace, target, rate, omr, ticket, norm_out, acc_src, acc_dst, n_acc);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    if (norm_out && target) RLX_ADAM_STEP(true, true, false);
    else if (norm_out) RLX_ADAM_STEP(true, false, false);
    else if (target) RLX_ADAM_STEP(false, true, true);
    else RLX_ADAM_STEP(false, false, true);
#undef RLX_ADAM_STEP
    RLX_LAUNCH_CHECK();
    if (norm_out) {
        RLX_LAUNCH((adam_finish_norm_kernel), 1, kBlock, 0, s, state, beta1, beta2, workspace, blocks, norm_out, acc_src,
                                                     acc_dst, n_acc);
        RLX_LAUNCH_CHECK();
    } else if (n_acc > 0) {
        RLX_REQUIRE(false, "rlx_adam_tf1_step: signal sums ride with the norm finish (give norm_out)");
    }
    return RLX_OK;
}

int rlx_mix_weights(float *target, const float *online, long long n, double rate, void *stream) {
    RLX_REQUIRE(target && online && n > 0, "rlx_mix_weights: bad arguments");
    RLX_REQUIRE((((uintptr_t)target | (uintptr_t)online) & 15) == 0,
                "rlx_mix_weights: buffers must be 16-byte aligned");
    // new_rate * new_weight + (1 - new_rate) * old_weights: python floats meet fp32 arrays, so
    // both coefficients are rounded to fp32 first (architecture.py:604-605).
    RLX_LAUNCH((mix_kernel), rlx::grid_for(n / 4 + 1, kBlock), kBlock, 0, rlx::as_stream(stream), target, online, n, (float)rate, (float)(1.0 - rate));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_global_norm(const float *x, long long n, float *norm_out, float *workspace,
                    long long workspace_floats, void *stream) {
    RLX_REQUIRE(x && norm_out && workspace && n > 0, "rlx_global_norm: bad arguments");
    int parts = rlx::grid_for(n, kBlock, 1024);
    if (parts > workspace_floats) parts = (int)workspace_floats;
    RLX_REQUIRE(parts >= 1, "rlx_global_norm: workspace too small");
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((sumsq_partial_kernel), parts, kBlock, 0, s, x, n, workspace);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((sumsq_final_kernel), 1, kBlock, 0, s, workspace, parts, norm_out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_clip_by_global_norm(float *grads, long long n, const float *global_norm, float clip_norm,
                            void *stream) {
    RLX_REQUIRE(grads && global_norm && n > 0, "rlx_clip_by_global_norm: bad arguments");
    RLX_REQUIRE(clip_norm > 0.f, "rlx_clip_by_global_norm: clip_norm must be positive (got %g)", (double)clip_norm);
    RLX_LAUNCH((clip_by_global_norm_kernel), rlx::grid_for(n, kBlock, 1024), kBlock, 0, rlx::as_stream(stream), grads, n, global_norm, clip_norm);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
