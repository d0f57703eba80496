// K9 — head losses and their gradients w.r.t. the head outputs, fused into one launch per head.
//
// Replaces the TensorFlow graph pieces in rl_coach/architectures/tensorflow_components/heads/:
//   * Head.set_loss            head.py:143-186   loss_h = mean_b( loss_weight * w_b * sum_dims l(target, out) )
//     with l = tf.losses.mean_squared_error (q_head.py, v_head.py:43-52) or tf.losses.huber_loss
//     (delta = 1; q_head.py when replace_mse_with_huber_loss)
//   * PPOHead._build_module    ppo_head.py:52-98 + _build_discrete_net :100-116
//     (Categorical log-prob, likelihood ratio, clipped surrogate, entropy bonus, KL fetch)
// plus tf.gradients of those losses down to the head's dense output (architecture.py:187-194).
//
// TensorFlow is not vendored in the reference tree and cannot be installed here: these formulas
// are restated from the head sources above and pinned only by the reference's MXNet-twin
// known-answer tests (tests/architectures/mxnet_components/heads/test_ppo_head.py:141-183,363-376)
// -> "parity unpinned" for TF's exact op-level rounding.
//
// One workgroup handles the whole minibatch (B <= 1024 rows x A actions): per-sample math in
// registers, the batch means by an LDS tree reduction, gradients written once.  Latency-bound
// (a few KB); the point is that logits, targets and gradients never leave HBM/L2.
#include "losses_body.hpp"

namespace {
using namespace rlx_losses;

// kind 0: mean squared error (t - o)^2;  kind 1: huber, delta = 1.
// out/target [B, D]; weights [B] or null; grad = d loss / d out.  scalars[0] = loss.
__device__ __forceinline__ void
regression_loss_body(const float *__restrict__ out, const float *__restrict__ target,
                       const float *__restrict__ weights, int batch, int dim, long long ld_out,
                       long long ld_target, int kind, float loss_weight, float grad_scale,
                       float *__restrict__ grad, long long ld_grad, float *__restrict__ scalars) {
    __shared__ float red[kMaxBlock];
    float local = 0.f;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        const float w = loss_weight * (weights ? weights[b] : 1.f);
        float row = 0.f;
        for (int j = 0; j < dim; ++j) {
            const float e = out[(size_t)b * ld_out + j] - target[(size_t)b * ld_target + j];
            float l, g;
            regression_terms(e, kind, l, g);
            row += l;
            if (grad) grad[(size_t)b * ld_grad + j] = regression_grad(grad_scale, w, g, batch);
        }
        local += w * row;
    }
    const float s = block_sum(local, red);
    if (threadIdx.x == 0 && scalars) scalars[0] = s / (float)batch;
}

// Row-wise softmax (tf.nn.softmax, ppo_head.py:108): probs = exp(z - max) / sum.
__global__ void softmax_kernel(const float *__restrict__ logits, int batch, int n, long long ld,
                               float *__restrict__ probs, long long ld_out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float *z = logits + (size_t)b * ld;
    float mx = z[0];
    for (int j = 1; j < n; ++j) mx = fmaxf(mx, z[j]);
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += expf(z[j] - mx);
    for (int j = 0; j < n; ++j) probs[(size_t)b * ld_out + j] = expf(z[j] - mx) / s;
}

// Discrete Clipped-PPO head.  scalars: [0] surrogate loss, [1] mean entropy, [2] mean KL(old||new),
// [3] total head loss = surrogate - beta * entropy.
__device__ __forceinline__ void
ppo_discrete_loss_body(const float *__restrict__ logits, long long ld, const int *__restrict__ actions,
                         const float *__restrict__ advantages, const float *__restrict__ old_probs,
                         long long ld_old, int batch, int n, float clip_eps, float beta,
                         float grad_scale, float *__restrict__ dlogits, long long ld_grad,
                         float *__restrict__ scalars, float *__restrict__ ratio_out,
                         float *__restrict__ clipped_out, int *__restrict__ status) {
    __shared__ float red[kMaxBlock];
    float l_sur = 0.f, l_ent = 0.f, l_kl = 0.f;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        PpoRowTerms t;
        if (!ppo_discrete_row(logits + (size_t)b * ld, old_probs + (size_t)b * ld_old, actions[b], n, advantages[b],
                              clip_eps, beta, grad_scale, batch, dlogits ? dlogits + (size_t)b * ld_grad : nullptr,
                              ratio_out ? ratio_out + b : nullptr, clipped_out ? clipped_out + b : nullptr, t)) {
            atomicOr(status, 1);
            continue;
        }
        l_sur += t.sur;
        l_ent += t.ent;
        l_kl += t.kl;
    }
    const float sur = block_sum(l_sur, red);
    const float ent = block_sum(l_ent, red);
    const float kl = block_sum(l_kl, red);
    if (threadIdx.x == 0 && scalars) ppo_discrete_scalars(sur, ent, kl, beta, batch, scalars);
}

__global__ void __launch_bounds__(kMaxBlock)
regression_loss_kernel(const float *__restrict__ out, const float *__restrict__ target,
                       const float *__restrict__ weights, int batch, int dim, long long ld_out,
                       long long ld_target, int kind, float loss_weight, float grad_scale,
                       float *__restrict__ grad, long long ld_grad, float *__restrict__ scalars) {
    regression_loss_body(out, target, weights, batch, dim, ld_out, ld_target, kind, loss_weight, grad_scale,
                         grad, ld_grad, scalars);
}

// TD targets + the critic regression losses of an actor-critic update in ONE launch (one workgroup, batch <= 1024):
//   q_next = min(q_next1, q_next2)  (TD3 output #2, td3_agent.py:168; q_next2 null: q_next1 as is)
//   y      = r + (1 - done) * discount * q_next   in fp64, optional clip, cast to fp32   (ac_targets_kernel)
//   stream t: loss_t = loss_weight * mean((q_t - y)^2), dq_t = d loss_t / d q_t             (regression_loss_body)
//   loss[n_streams] = loss_0 + loss_1 + ...  (the logged total, added in stream order)
// — the arithmetic of min_pair + rlx_ac_td_targets + n_streams x rlx_regression_loss, which it replaces.
struct AcCriticLossArgs {
    const float *q_next1, *q_next2, *rewards;
    const unsigned char *dones;
    double discount, clip_lo, clip_hi;
    int nonzero_terminal_discount, has_clip;
    const float *q;
    int n_streams, batch;
    float loss_weight;
    float *q_min_out, *td_targets, *dq, *loss;
};
__global__ void __launch_bounds__(kMaxBlock) ac_critic_losses_kernel(const AcCriticLossArgs a) {
    __shared__ float y_s[kMaxBlock];
    for (int i = threadIdx.x; i < a.batch; i += blockDim.x) {
        float qn = a.q_next1[i];
        if (a.q_next2) {
            const float b = a.q_next2[i];
            qn = qn <= b ? qn : b;
        }
        if (a.q_min_out) a.q_min_out[i] = qn;
        const double q = (double)qn;
        double t;
        if (a.nonzero_terminal_discount)
            t = (double)a.rewards[i] + a.discount * q;
        else
            t = (double)a.rewards[i] + (1.0 - (a.dones[i] ? 1.0 : 0.0)) * a.discount * q;
        if (a.has_clip) t = fmin(fmax(t, a.clip_lo), a.clip_hi);
        const float y = (float)t;
        y_s[i] = y;
        a.td_targets[i] = y;
    }
    __syncthreads();
    float total = 0.f;
    for (int t = 0; t < a.n_streams; ++t) {
        regression_loss_body(a.q + (size_t)t * a.batch, y_s, nullptr, a.batch, 1, 1, 1, 0, a.loss_weight, 1.f,
                             a.dq + (size_t)t * a.batch, 1, a.loss + t);
        __syncthreads();
        if (threadIdx.x == 0) {
            total += a.loss[t];
            if (t == a.n_streams - 1) a.loss[a.n_streams] = total;
        }
    }
}

__global__ void __launch_bounds__(kMaxBlock)
ppo_discrete_loss_kernel(const float *__restrict__ logits, long long ld, const int *__restrict__ actions,
                         const float *__restrict__ advantages, const float *__restrict__ old_probs,
                         long long ld_old, int batch, int n, float clip_eps, float beta,
                         float grad_scale, float *__restrict__ dlogits, long long ld_grad,
                         float *__restrict__ scalars, float *__restrict__ ratio_out,
                         float *__restrict__ clipped_out, int *__restrict__ status,
                         const float *__restrict__ clip_scale) {
    if (clip_scale) clip_eps *= *clip_scale;      // clip_param_rescaler as a device scalar (fp32 product, like TF)
    ppo_discrete_loss_body(logits, ld, actions, advantages, old_probs, ld_old, batch, n, clip_eps, beta,
                           grad_scale, dlogits, ld_grad, scalars, ratio_out, clipped_out, status);
}

// Both heads of Clipped PPO in one launch: workgroup 0 = PPOHead, workgroup 1 = VHead (MSE, weight 1).
struct PpoValueLossArgs {
    const float *logits; long long ld; const int *actions; const float *advantages;
    const float *old_probs; long long ld_old; int batch, n; float clip_eps, beta, grad_scale;
    float *dlogits; long long ld_grad; float *scalars; float *ratio_out; float *clipped_out; int *status;
    const float *v; const float *v_target; float *dv; float *v_scalar;
    const float *clip_scale;
};
__global__ void __launch_bounds__(kMaxBlock) ppo_value_losses_kernel(const PpoValueLossArgs a) {
    if (blockIdx.x == 0)
        ppo_discrete_loss_body(a.logits, a.ld, a.actions, a.advantages, a.old_probs, a.ld_old, a.batch, a.n,
                               a.clip_scale ? a.clip_eps * *a.clip_scale : a.clip_eps, a.beta, a.grad_scale, a.dlogits, a.ld_grad, a.scalars,
                               a.ratio_out, a.clipped_out, a.status);
    else
        regression_loss_body(a.v, a.v_target, nullptr, a.batch, 1, 1, 1, 0, 1.f, a.grad_scale, a.dv, 1,
                             a.v_scalar);
}

// Continuous Clipped-PPO head (ppo_head.py:118-144 + :58-98): policy = MultivariateNormalDiag(mean,
// exp(log_std) + eps) with ONE state-independent log_std vector; old policy = MVN(old_mean,
// old_std + eps) where old_std is the network's policy_std output.  eps = np.finfo(float32).eps.
// scalars as the discrete kernel.  d_log_std[a] (sum over the batch) is reduced in the workgroup.
__global__ void __launch_bounds__(kMaxBlock)
ppo_continuous_loss_kernel(const float *__restrict__ mean, long long ld, const float *__restrict__ log_std,
                           const float *__restrict__ actions, const float *__restrict__ advantages,
                           const float *__restrict__ old_mean, const float *__restrict__ old_std,
                           long long ld_old, int batch, int A, float clip_eps, float beta,
                           float grad_scale, float *__restrict__ dmean, long long ld_grad,
                           float *__restrict__ dlog_std, float *__restrict__ scalars,
                           float *__restrict__ ratio_out, float *__restrict__ clipped_out,
                           const float *__restrict__ clip_scale) {
    __shared__ float red[kMaxBlock];
    if (clip_scale) clip_eps *= *clip_scale;
    constexpr float kEps = 1.1920928955078125e-07f, kHalfLog2Pi = 0.91893853320467274178f;
    float l_sur = 0.f, l_kl = 0.f;
    float ent = 0.f;                                   // identical for every sample
    for (int a = 0; a < A; ++a) ent += 0.5f + kHalfLog2Pi + logf(expf(log_std[a]) + kEps);
    float my_glp = 0.f;                                 // batch <= blockDim: one sample per thread
    const int b = threadIdx.x;
    if (b < batch) {
        const float *mu = mean + (size_t)b * ld, *x = actions + (size_t)b * A;
        const float *mo = old_mean + (size_t)b * ld_old, *so = old_std + (size_t)b * ld_old;
        float logp = 0.f, logp_old = 0.f, kl = 0.f;
        for (int a = 0; a < A; ++a) {
            const float sd = expf(log_std[a]) + kEps, sdo = so[a] + kEps;
            const float z = (x[a] - mu[a]) / sd, zo = (x[a] - mo[a]) / sdo;
            logp += -0.5f * z * z - logf(sd) - kHalfLog2Pi;
            logp_old += -0.5f * zo * zo - logf(sdo) - kHalfLog2Pi;
            const float dm = mo[a] - mu[a];
            kl += logf(sd / sdo) + (sdo * sdo + dm * dm) / (2.f * sd * sd) - 0.5f;
        }
        const float ratio = expf(logp - logp_old);
        const float lo = 1.f - clip_eps, hi = 1.f + clip_eps;
        const float clipped = fminf(fmaxf(ratio, lo), hi);
        const float adv = advantages[b];
        const float s1 = ratio * adv, s2 = clipped * adv;
        l_sur = fminf(s1, s2);
        l_kl = kl;
        if (ratio_out) ratio_out[b] = ratio;
        if (clipped_out) clipped_out[b] = clipped;
        if (dmean) {
            float g_logp = (s1 <= s2 || (ratio >= lo && ratio <= hi)) ? -adv * ratio : 0.f;
            g_logp *= grad_scale / (float)batch;
            my_glp = g_logp;
            for (int a = 0; a < A; ++a) {
                const float sd = expf(log_std[a]) + kEps;
                dmean[(size_t)b * ld_grad + a] = g_logp * (x[a] - mu[a]) / (sd * sd);
            }
        }
    }
    const float sur = block_sum(l_sur, red);
    const float kls = block_sum(l_kl, red);
    if (dlog_std) {
        // d loss / d log_std[a] = sum_b g_logp_b * (d^2 / sd^3 - 1 / sd) * exp(log_std) - beta * dH/d log_std:
        // one fixed-order workgroup reduction per action dimension (reproducible, no atomics)
        for (int a = 0; a < A; ++a) {
            const float e = expf(log_std[a]);
            const float sd = e + kEps;
            float part = 0.f;
            if (b < batch) {
                const float d = actions[(size_t)b * A + a] - mean[(size_t)b * ld + a];
                part = my_glp * (d * d / (sd * sd * sd) - 1.f / sd) * e;
            }
            const float tot = block_sum(part, red);
            if (threadIdx.x == 0) dlog_std[a] = tot - grad_scale * beta * e / sd;
        }
    }
    if (threadIdx.x == 0 && scalars) {
        const float inv = 1.f / (float)batch;
        scalars[0] = -sur * inv;
        scalars[1] = ent;
        scalars[2] = kls * inv;
        scalars[3] = -sur * inv - beta * ent;
    }
}

inline int block_for(int batch) {
    int t = 64;
    while (t < batch && t < kMaxBlock) t <<= 1;
    return t;
}

}  // namespace

extern "C" {

int rlx_regression_loss(const float *out, long long ld_out, const float *target,
                        long long ld_target, const float *importance_weights, int batch, int dim,
                        int kind, float loss_weight, float grad_scale, float *grad,
                        long long ld_grad, float *loss_scalar, void *stream) {
    RLX_REQUIRE(out && target, "rlx_regression_loss: null pointer");
    RLX_REQUIRE(batch > 0 && dim > 0 && ld_out >= dim && ld_target >= dim,
                "rlx_regression_loss: bad shape (batch=%d dim=%d)", batch, dim);
    RLX_REQUIRE(kind == 0 || kind == 1, "rlx_regression_loss: kind must be 0 (mse) or 1 (huber)");
    RLX_REQUIRE(!grad || ld_grad >= dim, "rlx_regression_loss: bad gradient pitch");
    RLX_LAUNCH((regression_loss_kernel), 1, block_for(batch), 0, rlx::as_stream(stream), out, target, importance_weights, batch, dim, ld_out, ld_target, kind, loss_weight, grad_scale,
        grad, ld_grad, loss_scalar);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ac_critic_losses(const float *q_next1, const float *q_next2, const float *rewards,
                         const unsigned char *game_overs, double discount,
                         int use_non_zero_discount_for_terminal_states, int has_clip, double clip_low,
                         double clip_high, const float *q, int n_streams, int batch, float loss_weight,
                         float *q_min_out, float *td_targets, float *dq, float *loss, void *stream) {
    RLX_REQUIRE(q_next1 && rewards && game_overs && q && td_targets && dq && loss,
                "rlx_ac_critic_losses: null pointer");
    RLX_REQUIRE(batch > 0 && batch <= kMaxBlock && n_streams >= 1 && n_streams <= 4,
                "rlx_ac_critic_losses: bad sizes (batch=%d <= %d, n_streams=%d <= 4)", batch, kMaxBlock, n_streams);
    AcCriticLossArgs a;
    a.q_next1 = q_next1; a.q_next2 = q_next2; a.rewards = rewards; a.dones = game_overs;
    a.discount = discount; a.clip_lo = clip_low; a.clip_hi = clip_high;
    a.nonzero_terminal_discount = use_non_zero_discount_for_terminal_states; a.has_clip = has_clip;
    a.q = q; a.n_streams = n_streams; a.batch = batch; a.loss_weight = loss_weight;
    a.q_min_out = q_min_out; a.td_targets = td_targets; a.dq = dq; a.loss = loss;
    RLX_LAUNCH((ac_critic_losses_kernel), 1, block_for(batch), 0, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_softmax(const float *logits, long long ld, int batch, int n, float *probs,
                long long ld_out, void *stream) {
    RLX_REQUIRE(logits && probs, "rlx_softmax: null pointer");
    RLX_REQUIRE(batch > 0 && n > 0 && ld >= n && ld_out >= n, "rlx_softmax: bad shape");
    RLX_LAUNCH((softmax_kernel), (batch + 63) / 64, 64, 0, rlx::as_stream(stream), logits, batch, n, ld, probs,
                                                                       ld_out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ppo_discrete_loss(const float *logits, long long ld, const int *actions,
                          const float *advantages, const float *old_probs, long long ld_old,
                          int batch, int n_actions, float clip_epsilon, float beta_entropy,
                          float grad_scale, float *dlogits, long long ld_grad, float *scalars,
                          float *likelihood_ratio, float *clipped_likelihood_ratio, int *status,
                          const float *clip_scale, void *stream) {
    RLX_REQUIRE(logits && actions && advantages && old_probs && status,
                "rlx_ppo_discrete_loss: null pointer");
    RLX_REQUIRE(batch > 0 && n_actions > 0 && ld >= n_actions && ld_old >= n_actions,
                "rlx_ppo_discrete_loss: bad shape");
    RLX_REQUIRE(!dlogits || ld_grad >= n_actions, "rlx_ppo_discrete_loss: bad gradient pitch");
    RLX_LAUNCH((ppo_discrete_loss_kernel), 1, block_for(batch), 0, rlx::as_stream(stream), logits, ld, actions, advantages, old_probs, ld_old, batch, n_actions, clip_epsilon,
        beta_entropy, grad_scale, dlogits, ld_grad, scalars, likelihood_ratio,
        clipped_likelihood_ratio, status, clip_scale);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ppo_continuous_loss(const float *mean, long long ld, const float *log_std, const float *actions,
                            const float *advantages, const float *old_mean, const float *old_std,
                            long long ld_old, int batch, int action_dim, float clip_epsilon,
                            float beta_entropy, float grad_scale, float *dmean, long long ld_grad,
                            float *dlog_std, float *scalars, float *likelihood_ratio,
                            float *clipped_likelihood_ratio, const float *clip_scale, void *stream) {
    RLX_REQUIRE(mean && log_std && actions && advantages && old_mean && old_std,
                "rlx_ppo_continuous_loss: null pointer");
    RLX_REQUIRE(batch > 0 && batch <= kMaxBlock && action_dim > 0 && ld >= action_dim && ld_old >= action_dim,
                "rlx_ppo_continuous_loss: bad shape (minibatch must be <= %d rows)", kMaxBlock);
    RLX_REQUIRE((dmean == nullptr) == (dlog_std == nullptr) && (!dmean || ld_grad >= action_dim),
                "rlx_ppo_continuous_loss: give both gradient outputs or neither");
    RLX_LAUNCH((ppo_continuous_loss_kernel), 1, block_for(batch), 0, rlx::as_stream(stream), mean, ld, log_std, actions, advantages, old_mean, old_std, ld_old, batch, action_dim, clip_epsilon,
        beta_entropy, grad_scale, dmean, ld_grad, dlog_std, scalars, likelihood_ratio,
        clipped_likelihood_ratio, clip_scale);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ppo_discrete_value_losses(const float *logits, long long ld, const int *actions,
                                  const float *advantages, const float *old_probs, long long ld_old,
                                  int batch, int n_actions, float clip_epsilon, float beta_entropy,
                                  float grad_scale, float *dlogits, long long ld_grad, float *scalars,
                                  float *likelihood_ratio, float *clipped_likelihood_ratio, int *status,
                                  const float *values, const float *value_targets, float *dvalues,
                                  float *value_loss_scalar, const float *clip_scale, void *stream) {
    RLX_REQUIRE(logits && actions && advantages && old_probs && status && values && value_targets,
                "rlx_ppo_discrete_value_losses: null pointer");
    RLX_REQUIRE(batch > 0 && n_actions > 0 && ld >= n_actions && ld_old >= n_actions,
                "rlx_ppo_discrete_value_losses: bad shape");
    RLX_REQUIRE(!dlogits || ld_grad >= n_actions, "rlx_ppo_discrete_value_losses: bad gradient pitch");
    PpoValueLossArgs a{logits, ld, actions, advantages, old_probs, ld_old, batch, n_actions, clip_epsilon,
                       beta_entropy, grad_scale, dlogits, ld_grad, scalars, likelihood_ratio,
                       clipped_likelihood_ratio, status, values, value_targets, dvalues, value_loss_scalar,
                       clip_scale};
    RLX_LAUNCH((ppo_value_losses_kernel), 2, block_for(batch), 0, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
