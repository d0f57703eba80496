// Exploration policies vectorised over the envs of one GPU (thin, per env-step).
//
// Replaces, in the reference (paths under rl_coach/exploration_policies/):
//   * Categorical.get_action      categorical.py:45-56   np.random.choice(actions, p=probs)
//   * EGreedy.get_action          e_greedy.py:84-101     random action w.p. epsilon, else argmax
//                                                        with random tie-break among isclose maxima
//   * AdditiveNoise.get_action    additive_noise.py:75-111  np.random.normal(mean, std) (+ clip by
//                                                        the agent, agents/agent.py clip_action_to_space)
//
// The host draws the random numbers from the SAME generators in the same order as the reference
// (the global legacy np.random stream also feeds replay sampling, so its consumption order is part
// of the bit-exact index contract — SURVEY.md §7.3.1) and ships the draws; the kernels restate
// numpy's arithmetic on them:
//   np.random.choice(n, p=p): cdf = p.astype(f64).cumsum(); cdf /= cdf[-1];
//                             idx = cdf.searchsorted(u, side='right')       (numpy mtrand.pyx)
// Compiled with -ffp-contract=off.
#include "rlx_common.hpp"

namespace {

__global__ void categorical_sample_kernel(const float *__restrict__ probs, long long ld,
                                          const double *__restrict__ u, int n_env, int n_actions,
                                          int *__restrict__ actions) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    const float *p = probs + (size_t)e * ld;
    double total = 0.0;
    for (int j = 0; j < n_actions; ++j) total += (double)p[j];   // cumsum()[-1]
    const double uu = u[e];
    double c = 0.0;
    int idx = n_actions;                                  // searchsorted(..., side='right')
    for (int j = 0; j < n_actions; ++j) {
        c += (double)p[j];
        if (uu < c / total) {                             // first j with cdf[j] > u
            idx = j;
            break;
        }
    }
    actions[e] = idx < n_actions ? idx : n_actions - 1;
}

// softmax of the policy head's logits (tf.nn.softmax, ppo_head.py:108: losses.hip softmax_kernel's arithmetic, one row per
// thread) and the categorical draw above on the probabilities it just produced, in ONE launch: what an acting step of
// Clipped PPO does with the head's output.  probs_out may be null (acting needs only the action).
__global__ void softmax_categorical_sample_kernel(const float *__restrict__ logits, long long ld,
                                                  const double *__restrict__ u, int n_env, int n_actions,
                                                  float *__restrict__ probs_out, long long ld_out,
                                                  int *__restrict__ actions) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    const float *z = logits + (size_t)e * ld;
    float mx = z[0];
    for (int j = 1; j < n_actions; ++j) mx = fmaxf(mx, z[j]);
    float s = 0.f;
    for (int j = 0; j < n_actions; ++j) s += expf(z[j] - mx);
    double total = 0.0;
    for (int j = 0; j < n_actions; ++j) {
        const float pj = expf(z[j] - mx) / s;
        if (probs_out) probs_out[(size_t)e * ld_out + j] = pj;
        total += (double)pj;                              // cumsum()[-1]
    }
    const double uu = u[e];
    double c = 0.0;
    int idx = n_actions;                                  // searchsorted(..., side='right')
    for (int j = 0; j < n_actions; ++j) {
        c += (double)(expf(z[j] - mx) / s);
        if (uu < c / total) {                             // first j with cdf[j] > u
            idx = j;
            break;
        }
    }
    actions[e] = idx < n_actions ? idx : n_actions - 1;
}

// explore_u[e]   = the policy's current_random_value (np.random.rand())
// random_act[e]  = action_space.sample() drawn by the host for exploring envs (ignored otherwise)
// tie_rand[e][a] = np.random.random(action_values.shape) for greedy envs
__global__ void egreedy_kernel(const float *__restrict__ q, long long ld,
                               const double *__restrict__ explore_u,
                               const int *__restrict__ random_act,
                               const double *__restrict__ tie_rand, double epsilon, int n_env,
                               int n_actions, int *__restrict__ actions) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    if (explore_u[e] < epsilon) {                          // e_greedy.py:88
        actions[e] = random_act[e];
        return;
    }
    const float *qe = q + (size_t)e * ld;
    float mx = qe[0];
    for (int a = 1; a < n_actions; ++a) mx = fmaxf(mx, qe[a]);
    // np.argmax(np.random.random(shape) * np.isclose(action_values, action_values.max()))  (:93-94)
    // isclose on fp32 inputs (numpy 2 keeps fp32): |a - b| <= atol + rtol * |b|
    const float tol = 1e-8f + 1e-5f * fabsf(mx);
    int best = 0;
    double bv = -1.0;
    for (int a = 0; a < n_actions; ++a) {
        const bool close = fabsf(qe[a] - mx) <= tol;
        const double v = close ? tie_rand[(size_t)e * n_actions + a] : 0.0;
        if (v > bv) {
            bv = v;
            best = a;
        }
    }
    actions[e] = best;
}

// action = clip(mean + std * z, low, high), z = standard normal drawn on the host in the order
// np.random.normal(mean, std) consumes them.  std: per-dimension array (noise percentage of the
// action range, additive_noise.py:86-89) or per-sample network output.
__global__ void gaussian_action_kernel(const float *__restrict__ mean, const float *__restrict__ std_dim,
                                       const float *__restrict__ std_full, const double *__restrict__ z,
                                       const float *__restrict__ low, const float *__restrict__ high,
                                       int n_env, int act_dim, float *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_env * act_dim) return;
    const int a = t % act_dim;
    const double sd = std_full ? (double)std_full[t] : (double)std_dim[a];
    double v = (double)mean[t] + sd * z[t];
    if (low) v = fmin(fmax(v, (double)low[a]), (double)high[a]);
    out[t] = (float)v;
}

// np.argmax per row: the FIRST maximal entry (Categorical.get_action outside TRAIN, categorical.py:50-52)
__global__ void argmax_rows_kernel(const float *__restrict__ v, long long ld, int n_rows, int n_cols, int *out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows) return;
    const float *r = v + (size_t)e * ld;
    int best = 0;
    float bv = r[0];
    for (int a = 1; a < n_cols; ++a)
        if (r[a] > bv) {
            bv = r[a];
            best = a;
        }
    out[e] = best;
}

}  // namespace

extern "C" {

int rlx_argmax_rows(const float *values, long long ld, int n_rows, int n_cols, int *out, void *stream) {
    RLX_REQUIRE(values && out, "rlx_argmax_rows: null pointer");
    RLX_REQUIRE(n_rows > 0 && n_cols > 0 && ld >= n_cols, "rlx_argmax_rows: bad shape");
    RLX_LAUNCH((argmax_rows_kernel), (n_rows + 63) / 64, 64, 0, rlx::as_stream(stream), values, ld, n_rows, n_cols, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_categorical_sample(const float *probs, long long ld, const double *uniforms, int n_env,
                           int n_actions, int *actions, void *stream) {
    RLX_REQUIRE(probs && uniforms && actions, "rlx_categorical_sample: null pointer");
    RLX_REQUIRE(n_env > 0 && n_actions > 0 && ld >= n_actions, "rlx_categorical_sample: bad shape");
    RLX_LAUNCH((categorical_sample_kernel), (n_env + 63) / 64, 64, 0, rlx::as_stream(stream), probs, ld, uniforms, n_env, n_actions, actions);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_softmax_categorical_sample(const float *logits, long long ld, const double *uniforms, int n_env,
                                   int n_actions, float *probs_out, long long ld_out, int *actions, void *stream) {
    RLX_REQUIRE(logits && uniforms && actions, "rlx_softmax_categorical_sample: null pointer");
    RLX_REQUIRE(n_env > 0 && n_actions > 0 && ld >= n_actions && (!probs_out || ld_out >= n_actions),
                "rlx_softmax_categorical_sample: bad shape");
    RLX_LAUNCH((softmax_categorical_sample_kernel), (n_env + 63) / 64, 64, 0, rlx::as_stream(stream), logits, ld, uniforms,
               n_env, n_actions, probs_out, ld_out, actions);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_egreedy(const float *q_values, long long ld, const double *explore_uniforms,
                const int *random_actions, const double *tie_break_uniforms, double epsilon,
                int n_env, int n_actions, int *actions, void *stream) {
    RLX_REQUIRE(q_values && explore_uniforms && random_actions && tie_break_uniforms && actions,
                "rlx_egreedy: null pointer");
    RLX_REQUIRE(n_env > 0 && n_actions > 0 && ld >= n_actions, "rlx_egreedy: bad shape");
    RLX_LAUNCH((egreedy_kernel), (n_env + 63) / 64, 64, 0, rlx::as_stream(stream), q_values, ld, explore_uniforms, random_actions, tie_break_uniforms, epsilon, n_env,
        n_actions, actions);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_gaussian_action(const float *mean, const float *std_per_dim, const float *std_per_sample,
                        const double *standard_normals, const float *action_low,
                        const float *action_high, int n_env, int action_dim, float *actions,
                        void *stream) {
    RLX_REQUIRE(mean && standard_normals && actions && (std_per_dim || std_per_sample),
                "rlx_gaussian_action: null pointer");
    RLX_REQUIRE((action_low == nullptr) == (action_high == nullptr),
                "rlx_gaussian_action: give both bounds or neither");
    RLX_REQUIRE(n_env > 0 && action_dim > 0, "rlx_gaussian_action: bad shape");
    int n = n_env * action_dim;
    RLX_LAUNCH((gaussian_action_kernel), (n + 255) / 256, 256, 0, rlx::as_stream(stream), mean, std_per_dim, std_per_sample, standard_normals, action_low, action_high, n_env,
        action_dim, actions);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
