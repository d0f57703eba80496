// Batch normalisation of dense-layer outputs on gfx950 — the `use_batchnorm=True` variant of the DDPG networks
// (rl_coach/agents/ddpg_agent.py:37-60: Dense -> BatchnormActivationDropout after the observation embedder, the FC
// middleware and, in the actor, the head's fc_mean; architectures/tensorflow_components/layers.py:26-55 ->
// tf.layers.batch_normalization(x, training=is_training): axis -1, momentum 0.99, epsilon 1e-3, gamma / beta).
//
// Arithmetic as TF 1.x's non-fused path for a [B, C] input (tf.nn.moments + tf.nn.batch_normalization):
//   mean = mean_b x,  var = mean_b (x - mean)^2                       (population variance, fp32)
//   inv  = rsqrt(var + eps) * gamma,   u = x * inv + (beta - mean * inv),   y = act(u)
//   inference: mean / var are the moving averages; they move at apply time (the UPDATE_OPS that gate
//   optimizer.apply_gradients, architecture.py:273-277) by  m -= (m - batch) * (1 - momentum).
// Backward of the batch-statistics form:  du = dy * act'(y),  xh = (x - mean) * rsqrt(var + eps),
//   dbeta = sum_b du,  dgamma = sum_b du * xh,  dx = gamma * rsqrt(var + eps) * (du - dbeta / B - xh * dgamma / B).
//
// B <= a few hundred rows, C <= a few hundred columns: one thread per column, rows in order (a fixed, reproducible
// summation order; column-adjacent threads read adjacent floats).  A few KB per launch — latency, not bandwidth.
#include "rlx_common.hpp"

namespace {

constexpr int kBlock = 64;

__device__ __forceinline__ float bn_act(float v, int act) {
    if (act == RLX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RLX_ACT_TANH) return tanhf(v);
    return v;
}
__device__ __forceinline__ float bn_act_deriv(float y, int act) {
    if (act == RLX_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == RLX_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

__global__ void __launch_bounds__(kBlock) bn_forward_kernel(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ moving_mean, const float *__restrict__ moving_var, int B, int C, float eps, int training,
    int act, float *__restrict__ y, float *__restrict__ save_mean, float *__restrict__ save_var) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(size_t)b * C + c];
        mean = s / (float)B;
        float q = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = x[(size_t)b * C + c] - mean;
            q += d * d;
        }
        var = q / (float)B;
        save_mean[c] = mean;
        save_var[c] = var;
    } else {
        mean = moving_mean[c];
        var = moving_var[c];
    }
    const float inv = rsqrtf(var + eps) * gamma[c];
    const float shift = beta[c] - mean * inv;
    for (int b = 0; b < B; ++b) y[(size_t)b * C + c] = bn_act(x[(size_t)b * C + c] * inv + shift, act);
}

__global__ void __launch_bounds__(kBlock) bn_backward_kernel(
    const float *__restrict__ dy, const float *__restrict__ y, const float *__restrict__ x,
    const float *__restrict__ gamma, const float *__restrict__ save_mean, const float *__restrict__ save_var, int B,
    int C, float eps, int act, float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    const float mean = save_mean[c], r = rsqrtf(save_var[c] + eps);
    float sb = 0.f, sg = 0.f;
    for (int b = 0; b < B; ++b) {
        const size_t i = (size_t)b * C + c;
        const float du = act ? dy[i] * bn_act_deriv(y[i], act) : dy[i];
        sb += du;
        sg += du * ((x[i] - mean) * r);
    }
    if (dgamma) {
        dgamma[c] = sg;
        dbeta[c] = sb;
    }
    const float k = gamma[c] * r, mb = sb / (float)B, mg = sg / (float)B;
    for (int b = 0; b < B; ++b) {
        const size_t i = (size_t)b * C + c;
        const float du = act ? dy[i] * bn_act_deriv(y[i], act) : dy[i];
        dx[i] = k * (du - mb - ((x[i] - mean) * r) * mg);
    }
}

__global__ void __launch_bounds__(kBlock) bn_update_moving_kernel(float *__restrict__ moving_mean,
                                                                  float *__restrict__ moving_var,
                                                                  const float *__restrict__ batch_mean,
                                                                  const float *__restrict__ batch_var, int C,
                                                                  float one_minus_momentum) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    moving_mean[c] = moving_mean[c] - (moving_mean[c] - batch_mean[c]) * one_minus_momentum;
    moving_var[c] = moving_var[c] - (moving_var[c] - batch_var[c]) * one_minus_momentum;
}

}  // namespace

extern "C" {

int rlx_bn_forward(const float *x, const float *gamma, const float *beta, const float *moving_mean,
                   const float *moving_var, int batch, int channels, double epsilon, int training, int activation,
                   float *y, float *save_mean, float *save_var, void *stream) {
    RLX_REQUIRE(x && gamma && beta && y && batch > 0 && channels > 0, "rlx_bn_forward: bad arguments");
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_bn_forward: unknown activation %d", activation);
    RLX_REQUIRE(training ? (save_mean && save_var) : (moving_mean && moving_var),
                "rlx_bn_forward: training needs save_mean / save_var, inference the moving statistics");
    RLX_REQUIRE(epsilon > 0, "rlx_bn_forward: epsilon must be positive");
    RLX_LAUNCH((bn_forward_kernel), (channels + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), x, gamma, beta, moving_mean, moving_var, batch, channels, (float)epsilon, training, activation, y, save_mean,
        save_var);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_bn_backward(const float *dy, const float *y, const float *x, const float *gamma, const float *save_mean,
                    const float *save_var, int batch, int channels, double epsilon, int activation, float *dx,
                    float *dgamma, float *dbeta, void *stream) {
    RLX_REQUIRE(dy && x && gamma && save_mean && save_var && dx && batch > 0 && channels > 0,
                "rlx_bn_backward: bad arguments");
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_bn_backward: unknown activation %d", activation);
    RLX_REQUIRE(activation == 0 || y, "rlx_bn_backward: the activation derivative needs the layer output");
    RLX_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "rlx_bn_backward: dgamma and dbeta go together");
    RLX_LAUNCH((bn_backward_kernel), (channels + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), dy, y, x, gamma, save_mean, save_var, batch, channels, (float)epsilon, activation, dx, dgamma, dbeta);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_bn_update_moving(float *moving_mean, float *moving_var, const float *batch_mean, const float *batch_var,
                         int channels, double momentum, void *stream) {
    RLX_REQUIRE(moving_mean && moving_var && batch_mean && batch_var && channels > 0,
                "rlx_bn_update_moving: bad arguments");
    RLX_REQUIRE(momentum >= 0 && momentum <= 1, "rlx_bn_update_moving: momentum outside [0, 1]");
    RLX_LAUNCH((bn_update_moving_kernel), (channels + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), moving_mean, moving_var, batch_mean, batch_var, channels, (float)(1.0 - momentum));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
