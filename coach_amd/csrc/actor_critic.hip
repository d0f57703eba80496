// Elementwise pieces of the continuous-control agents (DDPG / TD3 / SAC) on gfx950.
//
// Replaces, in the reference (paths under rl_coach/):
//   * tf.concat of the critic's observation / action embeddings        architectures/tensorflow_components/general_network.py:270-277
//     and the slicing of gradients_wrt_inputs['action']                 architectures/tensorflow_components/architecture.py:187-220
//       -> rlx_copy_2d (strided 2-D copy with a scale: concat, slice, negate, output_scale multiply
//          of DDPGActorHead, heads/ddpg_actor_head.py:48-56)
//   * qi_obs_emb + qi_act_emb of SACQHead                               heads/sac_q_head.py:63-67      -> rlx_axpby
//   * tf.minimum(q1, q2), reduce_mean and its gradient                  heads/sac_q_head.py:84-88, td3_v_head.py:54-58 -> rlx_min_pair
//   * SACPolicyHead: clip(log_std), MultivariateNormalDiag sample / log_prob, tanh squash and the
//     squash correction (appendix C), plus tf.gradients of outputs [5] (mean log-prob) and [3]
//     (squashed actions) down to the head's dense output                heads/sac_head.py:60-97, agents/soft_actor_critic_agent.py:186-229
//       -> rlx_sac_policy_head / rlx_sac_policy_head_backward
//   * ObservationStackingFilter-free vector envs: next state of a finished episode is the post-reset
//     observation (environment.py:276-327 reset-on-done)                -> rlx_select_rows
//
// TensorFlow (and tf.contrib.distributions) is not vendored: formulas restated from the head
// sources above -> "parity unpinned" for TF's op-level rounding; the standard-normal draws come from
// the host generator (TF's own Philox stream cannot be reproduced).
// All kernels are elementwise over <= a few hundred KB: latency-bound; they exist so that actions,
// Q values and gradients never leave the device between the network passes of one update.
#include "rlx_common.hpp"

namespace {

constexpr int kBlock = 256;

__global__ void copy_2d_kernel(const float *__restrict__ src, long long src_ld,
                               float *__restrict__ dst, long long dst_ld, int rows, int cols,
                               float scale) {
    const long long total = (long long)rows * cols;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / cols, c = t - r * cols;
        dst[r * dst_ld + c] = scale * src[r * src_ld + c];
    }
}

__global__ void axpby_kernel(float *__restrict__ out, float a, const float *__restrict__ x, float b,
                             const float *__restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// out_min = min(q1, q2); g1/g2 = grad_scale * d mean(min)/d q_i  (tf.minimum: x <= y -> x gets it)
__global__ void min_pair_kernel(const float *__restrict__ q1, const float *__restrict__ q2,
                                float *__restrict__ out_min, float *__restrict__ g1,
                                float *__restrict__ g2, float grad_scale, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = q1[i], b = q2[i];
    const bool first = a <= b;
    if (out_min) out_min[i] = first ? a : b;
    if (g1) g1[i] = first ? grad_scale : 0.f;
    if (g2) g2[i] = first ? 0.f : grad_scale;
}

// min_pair + sac_value_targets in one launch: out_min = min(q1, q2), value_targets = out_min - logprob (fp32, as the
// numpy arrays of soft_actor_critic_agent.py:244), g_i = grad_scale * d sum(min) / d q_i.
__global__ void sac_min_targets_kernel(const float *__restrict__ q1, const float *__restrict__ q2,
                                       const float *__restrict__ logprob, float grad_scale, int n,
                                       float *__restrict__ out_min, float *__restrict__ value_targets,
                                       float *__restrict__ g1, float *__restrict__ g2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = q1[i], b = q2[i];
    const bool first = a <= b;
    const float m = first ? a : b;
    out_min[i] = m;
    value_targets[i] = m - logprob[i];
    g1[i] = first ? grad_scale : 0.f;
    g2[i] = first ? 0.f : grad_scale;
}

__global__ void select_rows_kernel(const unsigned char *__restrict__ mask,
                                   const unsigned char *__restrict__ if_set,
                                   const unsigned char *__restrict__ if_clear,
                                   unsigned char *__restrict__ out, int n, long long row_bytes) {
    const long long total = (long long)n * row_bytes;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / row_bytes;
        out[t] = mask[r] ? if_set[t] : if_clear[t];
    }
}

// out[b][a] = exp(log_std[a]): tf.tile(tf.exp(policy_logstd), [batch, 1])  (ppo_head.py:139)
__global__ void exp_rows_kernel(const float *__restrict__ log_std, float *__restrict__ out, int batch, int A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < batch * A) out[t] = expf(log_std[t % A]);
}

constexpr float kLogSigCapMin = -20.f, kLogSigCapMax = 2.f;       // sac_head.py:26-27
constexpr float kEpsF32 = 1.1920928955078125e-07f;                 // np.finfo(np.float32).eps (utils.py:38)
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// One thread per sample (A <= 64 action dims looped): logp needs the sum over dims.
__global__ void sac_policy_head_kernel(const float *__restrict__ mu_logsig, long long ld,
                                       const double *__restrict__ normals, int batch, int A,
                                       float *__restrict__ out_mean, float *__restrict__ out_logstd,
                                       float *__restrict__ out_raw, float *__restrict__ out_act,
                                       float *__restrict__ out_logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float *row = mu_logsig + (size_t)b * ld;
    float lp = 0.f, corr = 0.f;
    for (int a = 0; a < A; ++a) {
        const float mu = row[a];
        const float ls = fminf(fmaxf(row[A + a], kLogSigCapMin), kLogSigCapMax);   // :65-66
        const float sd = expf(ls);
        const float e = (float)normals[(size_t)b * A + a];
        const float raw = mu + sd * e;                                             // sample() :79
        const float t = tanhf(raw);                                                // :82
        const float z = (raw - mu) / sd;
        lp += -0.5f * z * z - ls - kHalfLog2Pi;                                    // log_prob :90
        corr += logf(1.f - t * t + kEpsF32);                                       // :58
        const size_t o = (size_t)b * A + a;
        if (out_mean) out_mean[o] = mu;
        if (out_logstd) out_logstd[o] = ls;
        if (out_raw) out_raw[o] = raw;
        if (out_act) out_act[o] = t;
    }
    if (out_logp) out_logp[b] = lp - corr;
}

// d_out[b][0:A]  = d/d mu,  d_out[b][A:2A] = d/d (unclipped) log_std  of
//     logp_weight * mean_b(logp_b)  +  sum_{b,a} act_weight[b][a] * tanh(raw[b][a])
// through the reparameterised sample raw = mu + exp(log_std) * normal.
__global__ void sac_policy_head_backward_kernel(const float *__restrict__ mu_logsig, long long ld,
                                                const double *__restrict__ normals, int batch, int A,
                                                float logp_weight,
                                                const float *__restrict__ act_weight,
                                                float act_weight_scale, float *__restrict__ d_out,
                                                long long ld_out, int accumulate) {
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
    if (t_ >= batch * A) return;
    const int b = t_ / A, a = t_ - b * A;
    const float *row = mu_logsig + (size_t)b * ld;
    const float mu = row[a];
    const float ls_raw = row[A + a];
    const bool inside = ls_raw >= kLogSigCapMin && ls_raw <= kLogSigCapMax;   // clip_by_value grad
    const float ls = fminf(fmaxf(ls_raw, kLogSigCapMin), kLogSigCapMax);
    const float sd = expf(ls);
    const float e = (float)normals[(size_t)b * A + a];
    const float raw = mu + sd * e;
    const float t = tanhf(raw);
    const float one_m = 1.f - t * t;
    const float w_lp = logp_weight / (float)batch;
    // d logp / d raw: the Gaussian terms cancel through the reparameterisation, the squash
    // correction -log(1 - t^2 + eps) leaves 2 t (1 - t^2) / (1 - t^2 + eps)
    const float dcorr = 2.f * t * one_m / (one_m + kEpsF32);
    float g_raw = w_lp * dcorr;
    if (act_weight) g_raw += act_weight_scale * act_weight[(size_t)b * A + a] * one_m;
    float d_mu = g_raw;
    float d_ls = g_raw * sd * e + w_lp * (-1.f);
    if (!inside) d_ls = 0.f;
    float *o = d_out + (size_t)b * ld_out;
    o[a] = accumulate ? o[a] + d_mu : d_mu;
    o[A + a] = accumulate ? o[A + a] + d_ls : d_ls;
}

}  // namespace

extern "C" {

int rlx_copy_2d(const float *src, long long src_ld, float *dst, long long dst_ld, int rows, int cols,
                float scale, void *stream) {
    RLX_REQUIRE(src && dst, "rlx_copy_2d: null pointer");
    RLX_REQUIRE(rows > 0 && cols > 0 && src_ld >= cols && dst_ld >= cols, "rlx_copy_2d: bad shape");
    RLX_LAUNCH((copy_2d_kernel), rlx::grid_for((long long)rows * cols, kBlock), kBlock, 0, rlx::as_stream(stream), src, src_ld, dst, dst_ld, rows, cols, scale);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_exp_rows(const float *log_std, float *out, int batch, int action_dim, void *stream) {
    RLX_REQUIRE(log_std && out && batch > 0 && action_dim > 0, "rlx_exp_rows: bad arguments");
    const int n = batch * action_dim;
    RLX_LAUNCH((exp_rows_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), log_std, out, batch, action_dim);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_axpby(float *out, float a, const float *x, float b, const float *y, long long n, void *stream) {
    RLX_REQUIRE(out && x && n > 0, "rlx_axpby: bad arguments");
    RLX_LAUNCH((axpby_kernel), rlx::grid_for(n, kBlock), kBlock, 0, rlx::as_stream(stream), out, a, x, b, y, n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_min_pair(const float *q1, const float *q2, float *out_min, float *grad1, float *grad2,
                 float grad_scale, int n, void *stream) {
    RLX_REQUIRE(q1 && q2 && n > 0, "rlx_min_pair: bad arguments");
    RLX_LAUNCH((min_pair_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), q1, q2, out_min, grad1, grad2, grad_scale, n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_sac_min_targets(const float *q1, const float *q2, const float *sampled_logprob, float grad_scale, int n,
                        float *out_min, float *value_targets, float *grad1, float *grad2, void *stream) {
    RLX_REQUIRE(q1 && q2 && sampled_logprob && out_min && value_targets && grad1 && grad2 && n > 0,
                "rlx_sac_min_targets: bad arguments");
    RLX_LAUNCH((sac_min_targets_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), q1, q2, sampled_logprob, grad_scale, n, out_min, value_targets, grad1, grad2);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_select_rows(const unsigned char *mask, const void *if_set, const void *if_clear, void *out,
                    int n, long long row_bytes, void *stream) {
    RLX_REQUIRE(mask && if_set && if_clear && out && n > 0 && row_bytes > 0,
                "rlx_select_rows: bad arguments");
    RLX_LAUNCH((select_rows_kernel), rlx::grid_for((long long)n * row_bytes, kBlock), kBlock, 0,
                         rlx::as_stream(stream), mask, static_cast<const unsigned char *>(if_set), static_cast<const unsigned char *>(if_clear),
        static_cast<unsigned char *>(out), n, row_bytes);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_sac_policy_head(const float *mu_logsig, long long ld, const double *standard_normals, int batch,
                        int action_dim, float *out_mean, float *out_log_std, float *out_raw_actions,
                        float *out_actions, float *out_logprob, void *stream) {
    RLX_REQUIRE(mu_logsig && standard_normals, "rlx_sac_policy_head: null pointer");
    RLX_REQUIRE(batch > 0 && action_dim > 0 && ld >= 2 * action_dim, "rlx_sac_policy_head: bad shape");
    RLX_LAUNCH((sac_policy_head_kernel), (batch + 63) / 64, 64, 0, rlx::as_stream(stream), mu_logsig, ld, standard_normals, batch, action_dim, out_mean, out_log_std, out_raw_actions,
        out_actions, out_logprob);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_sac_policy_head_backward(const float *mu_logsig, long long ld, const double *standard_normals,
                                 int batch, int action_dim, float logprob_mean_weight,
                                 const float *action_weights, float action_weight_scale,
                                 float *d_mu_logsig, long long ld_grad, int accumulate, void *stream) {
    RLX_REQUIRE(mu_logsig && standard_normals && d_mu_logsig, "rlx_sac_policy_head_backward: null pointer");
    RLX_REQUIRE(batch > 0 && action_dim > 0 && ld >= 2 * action_dim && ld_grad >= 2 * action_dim,
                "rlx_sac_policy_head_backward: bad shape");
    const int n = batch * action_dim;
    RLX_LAUNCH((sac_policy_head_backward_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), mu_logsig, ld, standard_normals, batch, action_dim, logprob_mean_weight, action_weights,
        action_weight_scale, d_mu_logsig, ld_grad, accumulate);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
