// CartPole-v0 / -v1 for N environments per GPU — the learnable environment behind the reference's CartPole presets
// (rl_coach/presets/CartPole_DQN.py:46, CartPole_ClippedPPO.py:59 `GymVectorEnvironment(level='CartPole-v0')`, stepped
// through rl_coach/environments/gym_environment.py:418-474).
//
// The physics are gym 0.12.5's (requirements.txt:10; gym/envs/classic_control/cartpole.py `step`, Euler integrator,
// fp64 like gym's Python floats) behind gym's TimeLimit wrapper (done after max_episode_steps steps: 200 for -v0,
// 500 for -v1):
//     force = +-10;  temp = (force + 0.05 * theta_dot^2 * sin(theta)) / 1.1
//     thetaacc = (9.8 * sin(theta) - cos(theta) * temp) / (0.5 * (4/3 - 0.1 * cos(theta)^2 / 1.1))
//     xacc = temp - 0.05 * thetaacc * cos(theta) / 1.1
//     x += 0.02 x_dot; x_dot += 0.02 xacc; theta += 0.02 theta_dot; theta_dot += 0.02 thetaacc
//     done = |x| > 2.4 or |theta| > 12 degrees;  reward = 1.0 on every step
// gym is a third-party dependency that is not vendored in the reference and not installable here: the CPU restatement
// is oracle/cartpole.py (Python floats + math.sin / math.cos, the operations gym executes), and this kernel follows it
// BIT FOR BIT: every operation in gym's order without contraction (-ffp-contract=off), sin / cos through
// rlx::libm_sin / libm_cos (glibc's algorithm on glibc's table, libm_sincos.hpp).
// Reset states: gym draws uniform(-0.05, 0.05, 4) from a per-env MT19937; here they come from the counter-based
// Philox stream the synthetic envs use (key = (seed, env id), counter = (episode, word pair, 0, kStreamReset)), 53-bit
// uniforms built like numpy's random_sample — a pure function of (seed, env, episode), so that N envs reset on
// different steps without a host round trip and shard across GPUs without communication.
// One thread per env: ~60 flops and 100 bytes per env-step — latency-bound plumbing, not a roofline kernel.
#include "rlx_common.hpp"
#include "libm_sincos.hpp"

namespace {

struct U4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
constexpr uint32_t kStreamReset = 2;

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {        // numpy random_sample: (a >> 5, b >> 6)
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
__device__ __forceinline__ void reset_state(uint32_t seed, uint32_t env, uint32_t ep, double (&s)[4]) {
    const U4 r0 = philox4x32_10(ep, 0u, 0u, kStreamReset, seed, env);
    const U4 r1 = philox4x32_10(ep, 1u, 0u, kStreamReset, seed, env);
    s[0] = -0.05 + 0.1 * u53(r0.x, r0.y);
    s[1] = -0.05 + 0.1 * u53(r0.z, r0.w);
    s[2] = -0.05 + 0.1 * u53(r1.x, r1.y);
    s[3] = -0.05 + 0.1 * u53(r1.z, r1.w);
}

__global__ void cartpole_reset_kernel(double *state, float *obs, int *episode, int *steps, int n_env, uint32_t seed,
                                      uint32_t env_id0, int next_episode) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    const int ep = next_episode ? episode[e] + 1 : 0;       // a forced reset mid-episode starts the NEXT episode's draw
    double s[4];
    reset_state(seed, env_id0 + e, (uint32_t)ep, s);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        state[4 * e + j] = s[j];
        obs[4 * e + j] = (float)s[j];
    }
    episode[e] = ep;
    steps[e] = 0;
}

__global__ void cartpole_step_kernel(const int *__restrict__ action, double *state, int *episode, int *steps,
                                     float *next_obs, float *reset_obs, double *next_state64, float *reward,
                                     unsigned char *done, int n_env, int max_episode_steps, uint32_t seed,
                                     uint32_t env_id0, int *status) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    double x = state[4 * e], x_dot = state[4 * e + 1], theta = state[4 * e + 2], theta_dot = state[4 * e + 3];
    const int a = action[e];
    if (a != 0 && a != 1) atomicOr(status, 2);              // gym: assert self.action_space.contains(action)
    const double force = a == 1 ? 10.0 : -10.0;
    int dom = 0;
    const double costheta = rlx::libm_cos(theta, &dom), sintheta = rlx::libm_sin(theta, &dom);
    if (dom) atomicOr(status, 1);
    // gym's constants as Python evaluates them: total_mass = 0.1 + 1.0, polemass_length = 0.1 * 0.5
    const double total_mass = 0.1 + 1.0, polemass_length = 0.1 * 0.5, length = 0.5, masspole = 0.1, gravity = 9.8;
    const double tau = 0.02;
    const double temp = (force + polemass_length * theta_dot * theta_dot * sintheta) / total_mass;
    const double thetaacc = (gravity * sintheta - costheta * temp) /
                            (length * (4.0 / 3.0 - masspole * costheta * costheta / total_mass));
    const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    const double theta_threshold = 12 * 2 * 3.141592653589793 / 360, x_threshold = 2.4;
    const int t = steps[e] + 1;
    const bool fell = x < -x_threshold || x > x_threshold || theta < -theta_threshold || theta > theta_threshold;
    const bool is_done = fell || t >= max_episode_steps;     // TimeLimit: _max_episode_steps <= _elapsed_steps
    const double ns[4] = {x, x_dot, theta, theta_dot};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        next_obs[4 * e + j] = (float)ns[j];
        if (next_state64) next_state64[4 * e + j] = ns[j];
    }
    reward[e] = 1.0f;
    done[e] = is_done ? 1 : 0;
    if (is_done) {
        const int ep = episode[e] + 1;
        double s[4];
        reset_state(seed, env_id0 + e, (uint32_t)ep, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            state[4 * e + j] = s[j];
            reset_obs[4 * e + j] = (float)s[j];
        }
        episode[e] = ep;
        steps[e] = 0;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) state[4 * e + j] = ns[j];
        steps[e] = t;
    }
}

__global__ void libm_sincos_kernel(const double *x, double *s, double *c, int n, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int dom = 0;
    s[i] = rlx::libm_sin(x[i], &dom);
    c[i] = rlx::libm_cos(x[i], &dom);
    if (dom) atomicOr(status, 1);
}

}  // namespace

extern "C" {

int rlx_cartpole_reset(double *state, float *obs, int *episode, int *steps, int n_env, unsigned int seed,
                       unsigned int env_id0, int next_episode, void *stream) {
    RLX_REQUIRE(state && obs && episode && steps, "rlx_cartpole_reset: null pointer");
    RLX_REQUIRE(n_env > 0, "rlx_cartpole_reset: n_env must be positive (got %d)", n_env);
    RLX_LAUNCH((cartpole_reset_kernel), (n_env + 63) / 64, 64, 0, rlx::as_stream(stream), state, obs, episode, steps, n_env,
                                                                                seed, env_id0, next_episode);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_cartpole_step(const int *action, double *state, int *episode, int *steps, float *next_obs, float *reset_obs,
                      double *next_state64, float *reward, unsigned char *game_over, int n_env,
                      int max_episode_steps, unsigned int seed, unsigned int env_id0, int *status, void *stream) {
    RLX_REQUIRE(action && state && episode && steps && next_obs && reset_obs && reward && game_over && status,
                "rlx_cartpole_step: null pointer");
    RLX_REQUIRE(n_env > 0 && max_episode_steps > 0, "rlx_cartpole_step: bad sizes (n_env %d, max_episode_steps %d)",
                n_env, max_episode_steps);
    RLX_LAUNCH((cartpole_step_kernel), (n_env + 63) / 64, 64, 0, rlx::as_stream(stream), action, state, episode, steps, next_obs, reset_obs, next_state64, reward, game_over, n_env,
        max_episode_steps, seed, env_id0, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_libm_sincos(const double *x, double *sin_out, double *cos_out, int n, int *status, void *stream) {
    RLX_REQUIRE(x && sin_out && cos_out && status && n > 0, "rlx_libm_sincos: bad arguments");
    RLX_LAUNCH((libm_sincos_kernel), (n + 255) / 256, 256, 0, rlx::as_stream(stream), x, sin_out, cos_out, n, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
