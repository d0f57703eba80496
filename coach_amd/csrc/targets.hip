// K7 / K10 — bootstrapped targets of the value-based and actor-critic agents on gfx950.
//
// Replaces the per-sample Python / numpy target arithmetic in (paths under rl_coach/agents/):
//   * DQNAgent.learn_from_batch             dqn_agent.py:92-103   (python loop over the batch)
//     DDQNAgent.select_actions              ddqn_agent.py:43      (argmax of the ONLINE net)
//   * DDPGAgent.learn_from_batch            ddpg_agent.py:156-164 (TD targets, optional clip)
//   * TD3Agent.learn_from_batch             td3_agent.py:162-180  (target-policy smoothing + min Q)
//   * SoftActorCriticAgent.learn_from_batch soft_actor_critic_agent.py:244,265-266
//
// The reference does this arithmetic in fp64 (python floats / numpy promote the fp32 network
// outputs) and casts to fp32 when the result is fed back to the network; the kernels do the same.
// Compiled with -ffp-contract=off so the fp64 expressions round like numpy (no FMA).
// Elementwise, HBM-bound, a few hundred bytes per launch at the BASELINE batch sizes: the point
// of these kernels is that Q values never leave the device between the two network passes.
#include "rlx_common.hpp"
#include "dense_small_body.hpp"

namespace {

constexpr int kBlock = 256;

// One thread per sample.  q_sel selects the greedy action (target net for DQN, online net for
// DDQN); np.argmax returns the FIRST maximum.
__global__ void dqn_targets_kernel(const float *__restrict__ q_next, const float *__restrict__ q_sel,
                                   float *__restrict__ td_targets, const int *__restrict__ actions,
                                   const float *__restrict__ rewards,
                                   const unsigned char *__restrict__ dones, double discount,
                                   int batch, int n_actions, double *__restrict__ td_errors,
                                   int *__restrict__ status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const float *qs = q_sel + (size_t)i * n_actions;
    int best = 0;
    float bv = qs[0];
    for (int a = 1; a < n_actions; ++a) {
        float v = qs[a];
        if (v > bv) {
            bv = v;
            best = a;
        }
    }
    const int act = actions[i];
    if (act < 0 || act >= n_actions) {
        atomicOr(status, 1);
        return;
    }
    // new_target = r + (1.0 - game_over) * discount * q_st_plus_1[i][a*]   (dqn_agent.py:100-101)
    const double qn = (double)q_next[(size_t)i * n_actions + best];
    const double new_target =
        (double)rewards[i] + (1.0 - (dones[i] ? 1.0 : 0.0)) * discount * qn;
    const size_t o = (size_t)i * n_actions + act;
    if (td_errors) td_errors[i] = fabs(new_target - (double)td_targets[o]);   // (:102)
    td_targets[o] = (float)new_target;                                          // (:103)
}

// DQN learn_from_batch from the Q values to the head gradient in ONE launch (dqn_agent.py:92-113 +
// heads/q_head.py loss, head.py:172-181): greedy next action, fp64 TD target, |TD error| for the
// prioritized replay, and — because TD_targets equals Q_online except at the taken action — the
// MSE / Huber loss and its gradient, which are non-zero only at [b, a_b]:
//   loss = mean_b( w_b * l(y_b, Q(s_b, a_b)) ),   dQ[b][a_b] = grad_scale * w_b * l'(.) / B.
// One workgroup (B <= 1024): per-sample work in registers, the batch mean by a fixed-order LDS tree.
__global__ void __launch_bounds__(1024)
dqn_head_loss_kernel(const float *__restrict__ q, long long ld_q, const float *__restrict__ q_next,
                     const float *__restrict__ q_sel, long long ld_next, const int *__restrict__ actions,
                     const float *__restrict__ rewards, const unsigned char *__restrict__ dones,
                     const double *__restrict__ weights, double discount, int batch, int n_actions,
                     int huber, float grad_scale, float *__restrict__ dq, long long ld_dq,
                     double *__restrict__ td_errors, float *__restrict__ td_targets, long long ld_t,
                     float *__restrict__ loss, int *__restrict__ status) {
    __shared__ float red[1024];
    const int i = threadIdx.x;
    float term = 0.f;
    if (i < batch) {
        const float *qs = q_sel + (size_t)i * ld_next;
        int best = 0;
        float bv = qs[0];
        for (int a = 1; a < n_actions; ++a)
            if (qs[a] > bv) { bv = qs[a]; best = a; }            // np.argmax: first maximum
        const int act = actions[i];
        if (act < 0 || act >= n_actions) {
            atomicOr(status, 1);
        } else {
            const double qn = (double)q_next[(size_t)i * ld_next + best];
            const double y = (double)rewards[i] + (1.0 - (dones[i] ? 1.0 : 0.0)) * discount * qn;   // :100-101
            const float qa = q[(size_t)i * ld_q + act];
            if (td_errors) td_errors[i] = fabs(y - (double)qa);                                     // :102
            const float y32 = (float)y;                                                             // :103
            const float e = qa - y32;
            const float w = weights ? (float)weights[i] : 1.f;   // fp64 PER weights fed to an fp32 placeholder
            float l, g;
            if (!huber) { l = e * e; g = 2.f * e; }
            else { const float ae = fabsf(e); l = ae <= 1.f ? 0.5f * e * e : ae - 0.5f; g = fminf(fmaxf(e, -1.f), 1.f); }
            term = w * l;
            for (int a = 0; a < n_actions; ++a) {
                dq[(size_t)i * ld_dq + a] = a == act ? grad_scale * w * g / (float)batch : 0.f;
                if (td_targets) td_targets[(size_t)i * ld_t + a] = a == act ? y32 : q[(size_t)i * ld_q + a];
            }
        }
    }
    red[i] = term;
    __syncthreads();
    for (int d = blockDim.x >> 1; d > 0; d >>= 1) {
        if (i < d) red[i] += red[i + d];
        __syncthreads();
    }
    if (i == 0 && loss) loss[0] = red[0] / (float)batch;
}

// The same TD targets / |TD errors| / loss / dQ AND the Q head's backward pass (dW, db, dx with the lower layer's
// activation derivative: tf.gradients through the head's Dense layer, architecture.py:312-385) in ONE launch
// (rlx_dqn_head_loss_backward) — as rlx_ppo_heads_loss_backward does for the Clipped-PPO heads: the loss kernel is 1-2 us of
// work behind a launch boundary, and its output dQ [B, A] is what every workgroup of the head's backward stages into LDS
// first.  Each workgroup computes the rows itself (one row per thread, the arithmetic above: one rounding), under the
// latency of its weight and input loads; the workgroup of feature block 0 writes dQ, the TD errors and the loss scalar
// (the same tree over `red_threads` terms as the stand-alone kernel's blockDim: bit-identical scalar).
struct DqnHeadBwdArgs {
    rlx_small::SmallDenseBwd head;               // .dy receives dQ (written by feature block 0)
    const float *q; long long ld_q;
    const float *q_next, *q_sel; long long ld_next;
    const int *actions;
    const float *rewards;
    const unsigned char *dones;
    const double *weights;
    double discount;
    double *td_errors;
    float *loss;
    int *status;
    int batch, n_actions, huber, red_threads;
    float grad_scale;
};

template <int NN>
__global__ void __launch_bounds__(256) dqn_head_loss_bwd_kernel(const DqnHeadBwdArgs a) {
    extern __shared__ float smem[];
    __shared__ float red[256];
    const bool writer = blockIdx.x == 0;
    const int i = threadIdx.x, B = a.batch, A = a.n_actions;
    float *dz = smem;                                         // [B][A]
    auto stage = [&]() {
        float term = 0.f;
        if (i < B) {
            const float *qs = a.q_sel + (size_t)i * a.ld_next;
            int best = 0;
            float bv = qs[0];
            for (int k = 1; k < A; ++k)
                if (qs[k] > bv) { bv = qs[k]; best = k; }              // np.argmax: first maximum
            const int act = a.actions[i];
            if (act < 0 || act >= A) {
                if (writer) atomicOr(a.status, 1);
                for (int k = 0; k < A; ++k) dz[(size_t)i * A + k] = 0.f;
            } else {
                const double qn = (double)a.q_next[(size_t)i * a.ld_next + best];
                const double y = (double)a.rewards[i] + (1.0 - (a.dones[i] ? 1.0 : 0.0)) * a.discount * qn;   // :100-101
                const float qa = a.q[(size_t)i * a.ld_q + act];
                if (writer && a.td_errors) a.td_errors[i] = fabs(y - (double)qa);                              // :102
                const float y32 = (float)y;                                                                     // :103
                const float e = qa - y32;
                const float w = a.weights ? (float)a.weights[i] : 1.f;
                float l, g;
                if (!a.huber) { l = e * e; g = 2.f * e; }
                else { const float ae = fabsf(e); l = ae <= 1.f ? 0.5f * e * e : ae - 0.5f; g = fminf(fmaxf(e, -1.f), 1.f); }
                term = w * l;
                for (int k = 0; k < A; ++k) dz[(size_t)i * A + k] = k == act ? a.grad_scale * w * g / (float)B : 0.f;
            }
        }
        if (writer) {                                         // (block-uniform) the stand-alone kernel's tree over its blockDim
            red[i] = i < a.red_threads ? term : 0.f;
            __syncthreads();
            for (int d = a.red_threads >> 1; d > 0; d >>= 1) {
                if (i < d) red[i] += red[i + d];
                __syncthreads();
            }
            if (i == 0 && a.loss) a.loss[0] = red[0] / (float)B;
        }
        __syncthreads();
        if (writer && a.head.dy) {                            // dQ, for whoever reads q.grad
            float *dy = const_cast<float *>(a.head.dy);
            for (int k = threadIdx.x; k < B * A; k += 256) dy[k] = dz[k];
        }
    };
    rlx_small::dense_small_bwd_body<NN, 16, 16, true>(a.head, blockIdx.x, 0, smem, stage);
}

// TD = r + (1 - done) * discount * q      (or r + discount * q), optionally clipped.
__global__ void ac_targets_kernel(const float *__restrict__ rewards,
                                  const unsigned char *__restrict__ dones,
                                  const float *__restrict__ q_next, int q_stride, double discount,
                                  int nonzero_terminal_discount, int has_clip, double clip_lo,
                                  double clip_hi, int batch, float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const double q = (double)q_next[(size_t)i * q_stride];
    double t;
    if (nonzero_terminal_discount)
        t = (double)rewards[i] + discount * q;
    else
        t = (double)rewards[i] + (1.0 - (dones[i] ? 1.0 : 0.0)) * discount * q;
    if (has_clip) t = fmin(fmax(t, clip_lo), clip_hi);        // np.clip
    out[i] = (float)t;
}

// next_actions = clip_action_to_space(mu_target(s') + clip(N(0, sigma), -c, c))   td3_agent.py:162-165
// `noise` holds the host-drawn np.random.normal(0, sigma, shape) values (fp64) BEFORE clipping so
// that the host RNG stream is consumed exactly like the reference's.
__global__ void td3_smooth_kernel(const float *__restrict__ next_actions,
                                  const double *__restrict__ noise, double noise_clip,
                                  const float *__restrict__ low, const float *__restrict__ high,
                                  int batch, int act_dim, float *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * act_dim) return;
    const int a = t % act_dim;
    double nz = fmin(fmax(noise[t], -noise_clip), noise_clip);
    double v = (double)next_actions[t] + nz;
    v = fmin(fmax(v, (double)low[a]), (double)high[a]);       // spaces.py:379 np.clip
    out[t] = (float)v;
}

// The two critic inputs of a DDPG / TD3 update, merged = concat(action, observation) (general_network.py:251,270-277),
// in one launch: row block 0 = [batch actions | s], row block 1 = [smoothed target actions | s'] — what
// td3_smooth_kernel + three strided copies produced.  noise == null: the target actions are taken as they are (DDPG).
__global__ void ac_merge_inputs_kernel(const float *__restrict__ actions, const float *__restrict__ obs,
                                       const float *__restrict__ next_actions, const double *__restrict__ noise,
                                       double noise_clip, const float *__restrict__ low,
                                       const float *__restrict__ high, const float *__restrict__ next_obs,
                                       int batch, int A, int D, float *__restrict__ merged2,
                                       float *__restrict__ merged_obs_only) {
    const int W = A + D;
    const long long total = (merged_obs_only ? 3LL : 2LL) * batch * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int half = (int)(t / ((long long)batch * W));
        const int r = (int)((t / W) % batch), c = (int)(t % W);
        if (half == 2) {                    // third block: only the observation columns ([. | s], actions filled later)
            if (c >= A) merged_obs_only[(size_t)r * W + c] = obs[(size_t)r * D + (c - A)];
            continue;
        }
        float v;
        if (c >= A) {
            v = (half ? next_obs : obs)[(size_t)r * D + (c - A)];
        } else if (!half) {
            v = actions[(size_t)r * A + c];
        } else if (!noise) {
            v = next_actions[(size_t)r * A + c];
        } else {
            const double nz = fmin(fmax(noise[(size_t)r * A + c], -noise_clip), noise_clip);
            double x = (double)next_actions[(size_t)r * A + c] + nz;
            x = fmin(fmax(x, (double)low[c]), (double)high[c]);
            v = (float)x;
        }
        merged2[t] = v;
    }
}

// value_targets = min(Q1,Q2)(s, a~pi) - log pi(a|s)     soft_actor_critic_agent.py:244
__global__ void sac_value_targets_kernel(const float *__restrict__ q_min,
                                         const float *__restrict__ logprob, int batch,
                                         float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    out[i] = q_min[i] - logprob[i];          // both fp32 numpy arrays in the reference
}

// DuelingQHead (dueling_q_head.py:33-48): one thread per sample, A is a handful of actions.
__global__ void dueling_combine_kernel(const float *__restrict__ v, const float *__restrict__ adv, int B, int A,
                                       float *__restrict__ q) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *a = adv + (size_t)b * A;
    float s = 0.f;
    for (int i = 0; i < A; ++i) s += a[i];
    const float mean = s / (float)A;                 // tf.reduce_mean(axis=1, keepdims=True)
    const float sv = v[b];
    for (int i = 0; i < A; ++i) q[(size_t)b * A + i] = sv + (a[i] - mean);
}

__global__ void dueling_combine_bwd_kernel(const float *__restrict__ dq, int B, int A, float *__restrict__ dv,
                                           float *__restrict__ dadv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *g = dq + (size_t)b * A;
    float s = 0.f;
    for (int i = 0; i < A; ++i) s += g[i];
    dv[b] = s;                                       // V is broadcast over the actions
    const float mean = s / (float)A;
    for (int i = 0; i < A; ++i) dadv[(size_t)b * A + i] = g[i] - mean;
}

}  // namespace

extern "C" {

int rlx_dqn_targets(const float *q_next_target, const float *q_next_selector, float *td_targets,
                    const int *actions, const float *rewards, const unsigned char *game_overs,
                    double discount, int batch, int n_actions, double *td_errors, int *status,
                    void *stream) {
    RLX_REQUIRE(q_next_target && td_targets && actions && rewards && game_overs && status,
                "rlx_dqn_targets: null pointer");
    RLX_REQUIRE(batch > 0 && n_actions > 0, "rlx_dqn_targets: bad sizes (batch=%d actions=%d)",
                batch, n_actions);
    if (!q_next_selector) q_next_selector = q_next_target;
    RLX_LAUNCH((dqn_targets_kernel), (batch + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), q_next_target, q_next_selector, td_targets, actions, rewards, game_overs, discount, batch,
        n_actions, td_errors, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dqn_head_loss(const float *q_online, long long ld_q, const float *q_next_target,
                      const float *q_next_selector, long long ld_next, const int *actions,
                      const float *rewards, const unsigned char *game_overs,
                      const double *importance_weights, double discount, int batch, int n_actions,
                      int huber, float grad_scale, float *dq, long long ld_dq, double *td_errors,
                      float *td_targets, long long ld_targets, float *loss_scalar, int *status,
                      void *stream) {
    RLX_REQUIRE(q_online && q_next_target && actions && rewards && game_overs && dq && status,
                "rlx_dqn_head_loss: null pointer");
    RLX_REQUIRE(batch > 0 && batch <= 1024 && n_actions > 0 && ld_q >= n_actions && ld_next >= n_actions &&
                    ld_dq >= n_actions,
                "rlx_dqn_head_loss: bad sizes (batch=%d <= 1024, actions=%d)", batch, n_actions);
    if (!q_next_selector) q_next_selector = q_next_target;
    int threads = 64;
    while (threads < batch) threads <<= 1;
    RLX_LAUNCH((dqn_head_loss_kernel), 1, threads, 0, rlx::as_stream(stream), q_online, ld_q, q_next_target, q_next_selector, ld_next, actions, rewards, game_overs,
        importance_weights, discount, batch, n_actions, huber, grad_scale, dq, ld_dq, td_errors,
        td_targets, ld_targets, loss_scalar, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dqn_head_loss_backward(const rlx_small_dense_problem *q_head, const float *q_online, long long ld_q,
                               const float *q_next_target, const float *q_next_selector, long long ld_next,
                               const int *actions, const float *rewards, const unsigned char *game_overs,
                               const double *importance_weights, double discount, int batch, int n_actions, int huber,
                               float grad_scale, double *td_errors, float *loss_scalar, int *status, void *stream) {
    RLX_REQUIRE(q_head && q_online && q_next_target && actions && rewards && game_overs && status,
                "rlx_dqn_head_loss_backward: null pointer");
    RLX_REQUIRE(q_head->x && q_head->w && q_head->dy && (q_head->dw || q_head->dx), "rlx_dqn_head_loss_backward: null pointer in the head");
    RLX_REQUIRE(q_head->towers == 1 && q_head->M == batch && q_head->N == n_actions && q_head->K > 0 && q_head->activation == 0 &&
                    q_head->lower_activation >= 0 && q_head->lower_activation <= 2,
                "rlx_dqn_head_loss_backward: the head must be one linear tower [batch] x [K] -> [n_actions]");
    RLX_REQUIRE(batch >= 1 && batch <= 256 && n_actions >= 1 && n_actions <= rlx_small::kMaxN && ld_q >= n_actions &&
                    ld_next >= n_actions,
                "rlx_dqn_head_loss_backward: 1 <= batch <= 256 (one row per thread), 1 <= actions <= %d", rlx_small::kMaxN);
    if (!q_next_selector) q_next_selector = q_next_target;
    const int NN = n_actions <= 1 ? 1 : n_actions <= 4 ? 4 : n_actions <= 8 ? 8 : 16;
    const size_t smem = ((size_t)batch * n_actions + 16 * 16 * NN) * sizeof(float);
    RLX_REQUIRE(smem <= 64 * 1024, "rlx_dqn_head_loss_backward: batch x actions exceeds the LDS budget");
    int red_threads = 64;
    while (red_threads < batch) red_threads <<= 1;       // blockDim of the stand-alone loss kernel
    DqnHeadBwdArgs a{rlx_small::SmallDenseBwd{q_head->x, q_head->x_tower_stride, q_head->w, q_head->w_tower_stride, q_head->dy,
                                               q_head->dy_tower_stride, nullptr, 0, q_head->dw, q_head->dw_tower_stride,
                                               q_head->db, q_head->db_tower_stride, q_head->dx, q_head->dx_tower_stride,
                                               q_head->M, q_head->K, q_head->N, 0, q_head->lower_activation},
                     q_online, ld_q, q_next_target, q_next_selector, ld_next, actions, rewards, game_overs,
                     importance_weights, discount, td_errors, loss_scalar, status, batch, n_actions, huber, red_threads,
                     grad_scale};
    const dim3 grid((q_head->K + 15) / 16);
    hipStream_t s = rlx::as_stream(stream);
    if (NN == 1) RLX_LAUNCH((dqn_head_loss_bwd_kernel<1>), grid, 256, smem, s, a);
    else if (NN == 4) RLX_LAUNCH((dqn_head_loss_bwd_kernel<4>), grid, 256, smem, s, a);
    else if (NN == 8) RLX_LAUNCH((dqn_head_loss_bwd_kernel<8>), grid, 256, smem, s, a);
    else RLX_LAUNCH((dqn_head_loss_bwd_kernel<16>), grid, 256, smem, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dueling_combine(const float *state_value, const float *action_advantage, int batch,
                        int n_actions, float *q, void *stream) {
    RLX_REQUIRE(state_value && action_advantage && q, "rlx_dueling_combine: null pointer");
    RLX_REQUIRE(batch > 0 && n_actions > 0, "rlx_dueling_combine: bad sizes (batch=%d actions=%d)", batch, n_actions);
    RLX_LAUNCH((dueling_combine_kernel), (batch + 63) / 64, 64, 0, rlx::as_stream(stream), state_value, action_advantage,
                                                                                  batch, n_actions, q);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dueling_combine_backward(const float *dq, int batch, int n_actions, float *dstate_value,
                                 float *daction_advantage, void *stream) {
    RLX_REQUIRE(dq && dstate_value && daction_advantage, "rlx_dueling_combine_backward: null pointer");
    RLX_REQUIRE(batch > 0 && n_actions > 0, "rlx_dueling_combine_backward: bad sizes (batch=%d actions=%d)", batch,
                n_actions);
    RLX_LAUNCH((dueling_combine_bwd_kernel), (batch + 63) / 64, 64, 0, rlx::as_stream(stream), dq, batch, n_actions,
                                                                                      dstate_value, daction_advantage);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ac_td_targets(const float *rewards, const unsigned char *game_overs, const float *q_next,
                      int q_stride, double discount, int use_non_zero_discount_for_terminal_states,
                      int has_clip, double clip_low, double clip_high, int batch,
                      float *td_targets, void *stream) {
    RLX_REQUIRE(rewards && game_overs && q_next && td_targets, "rlx_ac_td_targets: null pointer");
    RLX_REQUIRE(batch > 0 && q_stride > 0, "rlx_ac_td_targets: bad sizes");
    RLX_LAUNCH((ac_targets_kernel), (batch + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), rewards, game_overs, q_next, q_stride, discount, use_non_zero_discount_for_terminal_states,
        has_clip, clip_low, clip_high, batch, td_targets);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_td3_smooth_actions(const float *next_actions, const double *noise, double noise_clipping,
                           const float *action_low, const float *action_high, int batch,
                           int action_dim, float *out, void *stream) {
    RLX_REQUIRE(next_actions && noise && action_low && action_high && out,
                "rlx_td3_smooth_actions: null pointer");
    RLX_REQUIRE(batch > 0 && action_dim > 0, "rlx_td3_smooth_actions: bad sizes");
    int n = batch * action_dim;
    RLX_LAUNCH((td3_smooth_kernel), (n + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), next_actions, noise, noise_clipping, action_low, action_high, batch, action_dim, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_ac_merge_inputs(const float *actions, const float *obs, const float *next_actions, const double *noise,
                        double noise_clipping, const float *action_low, const float *action_high,
                        const float *next_obs, int batch, int action_dim, int obs_dim, float *merged2,
                        float *merged_obs_only, void *stream) {
    RLX_REQUIRE(actions && obs && next_actions && next_obs && merged2, "rlx_ac_merge_inputs: null pointer");
    RLX_REQUIRE(!noise || (action_low && action_high), "rlx_ac_merge_inputs: smoothing needs the action bounds");
    RLX_REQUIRE(batch > 0 && action_dim > 0 && obs_dim > 0, "rlx_ac_merge_inputs: bad sizes");
    const long long total = 3LL * batch * (action_dim + obs_dim);
    RLX_LAUNCH((ac_merge_inputs_kernel), rlx::grid_for(total, kBlock), kBlock, 0, rlx::as_stream(stream), actions, obs, next_actions, noise, noise_clipping, action_low, action_high, next_obs, batch, action_dim,
        obs_dim, merged2, merged_obs_only);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_sac_value_targets(const float *q_min, const float *sampled_logprob, int batch,
                          float *value_targets, void *stream) {
    RLX_REQUIRE(q_min && sampled_logprob && value_targets, "rlx_sac_value_targets: null pointer");
    RLX_REQUIRE(batch > 0, "rlx_sac_value_targets: bad batch");
    RLX_LAUNCH((sac_value_targets_kernel), (batch + kBlock - 1) / kBlock, kBlock, 0, rlx::as_stream(stream), q_min, sampled_logprob, batch, value_targets);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
