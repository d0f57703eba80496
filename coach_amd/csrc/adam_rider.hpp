// The TF1 Adam step of a parameter RANGE carried by workgroups of ANOTHER launch (rlx_adam_rider_*, include/rlx.h).
//
// Why: in the Clipped-PPO update the Adam pass over the dense layers' parameters (95 % of the bytes, 28 B per parameter,
// HBM-bound, ~17 us) and the convolution layers' backward products (latency-bound fp32 MFMA launches of 200-400
// workgroups that leave most of the chip's memory bandwidth idle, ~24 us each) need complementary resources, the dense
// layers' gradients are final before the convolution backward starts, and nothing reads those weights again before the
// next forward pass.  Overlapping the two through branches of the captured graph was measured three times and lost every
// time (a cross-stream edge pair costs more than it hides).  Here the Adam work is extra WORKGROUPS of the
// convolution layer's dW + dX pair launch: one dispatch, no edge, the hardware scheduler interleaves the two kinds of
// workgroups on the CUs.
//
// The element arithmetic is optim.hip's (adam_step_norm_kernel), expression by expression, under `fp contract(off)` — the
// flag optim.hip is compiled with — so a weight takes the same bits whichever kernel updates it.  The range's share of
// tf.global_norm goes to one partial sum of squares per rider workgroup; the Adam launch over the rest of the buffer adds
// them to its own (rlx_adam_tf1_step_ranges).
#pragma once
#include "rlx_common.hpp"

namespace rlx {

struct AdamRider {
    float *w;
    const float *g;
    float *m, *v;
    long long n4;               // float4 groups of the range (its length is a multiple of 4, its start 16-byte aligned)
    const float *state;         // {beta1_power, beta2_power}: read, not advanced (the closing Adam launch advances them)
    float *sumsq_part;          // [blocks] or null
    float lr, beta1, beta2, eps, grad_scale;
    int blocks;                 // rider workgroups of 256 threads
};

constexpr int kRiderIt = 4;     // float4 groups per array a thread has in flight

__device__ __forceinline__ void adam_rider_block(const AdamRider &r, const int rb, float *red /* >= 256 floats of LDS */) {
#pragma clang fp contract(off)
    const float b1p = r.state[0], b2p = r.state[1];
    const float alpha = r.lr * sqrtf(1.f - b2p) / (1.f - b1p);
    const float omb1 = 1.f - r.beta1, omb2 = 1.f - r.beta2;
    const long long stride = (long long)r.blocks * 256;
    float ss = 0.f;
    for (long long base = (long long)rb * 256 + threadIdx.x; base < r.n4; base += stride * kRiderIt) {
        float4 gw[kRiderIt], mw[kRiderIt], vw[kRiderIt], ww[kRiderIt];
#pragma unroll
        for (int it = 0; it < kRiderIt; ++it) {
            const long long i = base + it * stride;
            if (i < r.n4) {
                gw[it] = reinterpret_cast<const float4 *>(r.g)[i];
                mw[it] = reinterpret_cast<const float4 *>(r.m)[i];
                vw[it] = reinterpret_cast<const float4 *>(r.v)[i];
                ww[it] = reinterpret_cast<const float4 *>(r.w)[i];
            }
        }
#pragma unroll
        for (int it = 0; it < kRiderIt; ++it) {
            const long long i = base + it * stride;
            if (i < r.n4) {
                float *gp = &gw[it].x, *mp = &mw[it].x, *vp = &vw[it].x, *wp = &ww[it].x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ss += gp[k] * gp[k];
                    const float gr = gp[k] * r.grad_scale;
                    mp[k] += (gr - mp[k]) * omb1;
                    vp[k] += (gr * gr - vp[k]) * omb2;
                    wp[k] -= (mp[k] * alpha) / (sqrtf(vp[k]) + r.eps);
                }
                reinterpret_cast<float4 *>(r.m)[i] = mw[it];
                reinterpret_cast<float4 *>(r.v)[i] = vw[it];
                reinterpret_cast<float4 *>(r.w)[i] = ww[it];
            }
        }
    }
    if (r.sumsq_part) {
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) r.sumsq_part[rb] = red[0];
    }
}

// the pending rider of this host thread (armed by rlx_adam_rider_arm, taken by the next pair launch(es), optim.hip)
struct PendingRider {
    bool armed = false;
    AdamRider r;
    int launches_left = 0;      // pair launches the remaining range is to be divided over
    int parts_done = 0;         // partial sums already assigned (offset into sumsq_part)
};
PendingRider &pending_rider();
// the share of the pending rider the next carrying launch takes (false: nothing pending); advances the pending state
bool take_rider_share(AdamRider *out);

}  // namespace rlx
