// Synthetic vectorised environments for the throughput / parity workloads.
//
// The reference steps ONE simulator per process through Environment.step
// (rl_coach/environments/environment.py:276-327); gym / ALE / MuJoCo are not installable here, so
// the BASELINE configs run on fixed-length synthetic episodes whose observations and rewards come
// from a counter-based generator: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy
// as 1, 2, 3", SC'11), key = (seed, env id), counter = (episode, step, block, stream).  Being a pure
// function of (seed, env, episode, step) the oracle (oracle/synth_env.py, numpy) produces the same
// bytes on the CPU, the env can be sharded across GPUs without communication, and actions are
// consumed but do not influence the dynamics.
//   image env : obs = obs_elems uniform bytes (84*84 = 7056 per frame), reward in {-1, 0, +1} with
//               P = {.05, .9, .05}
//   vector env: obs = obs_elems fp32 values ~ unit-variance Irwin-Hall(4) (sum of 4 uniforms, exact
//               IEEE adds -> bit-identical on CPU and GPU), reward likewise
// game_over is raised on the last step of every episode_len-step episode.
// HBM-bound writer: one 16-byte store per Philox call.  Compiled with -ffp-contract=off.
#include "rlx_common.hpp"

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

constexpr uint32_t kStreamObs = 0, kStreamReward = 1;

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

__device__ __forceinline__ float irwin_hall(const U4 r) {
    const float s = ((u01(r.x) + u01(r.y)) + u01(r.z)) + u01(r.w);
    return (s - 2.0f) * 1.7320508075688772f;          // var(sum of 4 U(0,1)) = 1/3
}

__device__ __forceinline__ float reward_of(int kind, uint32_t seed, uint32_t env, uint32_t ep, uint32_t t) {
    const U4 r = philox4x32_10(ep, t, 0u, kStreamReward, seed, env);
    if (kind == 0) {
        const float u = u01(r.x);
        return u < 0.05f ? -1.f : (u < 0.95f ? 0.f : 1.f);
    }
    return irwin_hall(r);
}

// writes observation O(ep, t) of env `env` into dst (obs_elems elements)
__device__ __forceinline__ void write_obs(int kind, void *dst, int obs_elems, uint32_t seed,
                                          uint32_t env, uint32_t ep, uint32_t t, int first, int step) {
    if (kind == 0) {
        const int blocks = obs_elems >> 4;            // 16 bytes per Philox call
        uint4 *d = static_cast<uint4 *>(dst);
        for (int j = first; j < blocks; j += step) {
            const U4 r = philox4x32_10(ep, t, (uint32_t)j, kStreamObs, seed, env);
            d[j] = make_uint4(r.x, r.y, r.z, r.w);
        }
    } else {
        float *d = static_cast<float *>(dst);
        for (int j = first; j < obs_elems; j += step) {
            const U4 r = philox4x32_10(ep, t, (uint32_t)j, kStreamObs, seed, env);
            d[j] = irwin_hall(r);
        }
    }
}

__global__ void env_reset_kernel(int kind, void *obs, int *ep, int *t, int n_env, int obs_elems,
                                 uint32_t seed, uint32_t env_id0) {
    const int e = blockIdx.y;
    const size_t esz = kind == 0 ? 1 : 4;
    write_obs(kind, static_cast<char *>(obs) + (size_t)e * obs_elems * esz, obs_elems, seed,
              env_id0 + e, 0u, 0u, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ep[e] = 0;
        t[e] = 0;
    }
}

// advance != 0 (only when one workgroup serves an env, gridDim.x == 1): the env's episode / step counters are
// advanced here, after every thread has read them, instead of by a second launch.
__global__ void env_step_kernel(int kind, void *next_obs, void *reset_obs, float *reward,
                                unsigned char *done, int *ep_in, int *t_in, int n_env,
                                int obs_elems, int episode_len, const int *__restrict__ len_per_env,
                                uint32_t seed, uint32_t env_id0, int advance) {
    const int e = blockIdx.y;
    const uint32_t ep = (uint32_t)ep_in[e], t = (uint32_t)t_in[e];
    const bool is_done = (int)t + 1 >= (len_per_env ? len_per_env[e] : episode_len);
    const size_t esz = kind == 0 ? 1 : 4;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    write_obs(kind, static_cast<char *>(next_obs) + (size_t)e * obs_elems * esz, obs_elems, seed,
              env_id0 + e, ep, t + 1, first, step);
    if (is_done)
        write_obs(kind, static_cast<char *>(reset_obs) + (size_t)e * obs_elems * esz, obs_elems,
                  seed, env_id0 + e, ep + 1, 0u, first, step);
    if (first == 0) {
        reward[e] = reward_of(kind, seed, env_id0 + e, ep, t);
        done[e] = is_done ? 1 : 0;
    }
    if (advance) {
        __syncthreads();
        if (threadIdx.x == 0) {
            ep_in[e] = is_done ? (int)ep + 1 : (int)ep;
            t_in[e] = is_done ? 0 : (int)t + 1;
        }
    }
}

__global__ void env_advance_kernel(int *ep, int *t, const unsigned char *done, int n_env) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    if (done[e]) {
        ep[e] += 1;
        t[e] = 0;
    } else {
        t[e] += 1;
    }
}

}  // namespace

extern "C" {

int rlx_synth_env_reset(int kind, void *obs, int *episode, int *step, int n_env, int obs_elems,
                        unsigned int seed, unsigned int env_id0, void *stream) {
    RLX_REQUIRE(obs && episode && step, "rlx_synth_env_reset: null pointer");
    RLX_REQUIRE(kind == 0 || kind == 1, "rlx_synth_env_reset: kind must be 0 (image) or 1 (vector)");
    RLX_REQUIRE(n_env > 0 && obs_elems > 0 && (kind == 1 || obs_elems % 16 == 0),
                "rlx_synth_env_reset: image observations must be a multiple of 16 bytes (got %d)",
                obs_elems);
    const int work = kind == 0 ? obs_elems / 16 : obs_elems;
    dim3 grid(rlx::grid_for(work, 64, 16), n_env);
    RLX_LAUNCH((env_reset_kernel), grid, 64, 0, rlx::as_stream(stream), kind, obs, episode, step, n_env,
                                                             obs_elems, seed, env_id0);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int synth_step(int kind, void *next_obs, void *reset_obs, float *reward, unsigned char *game_over,
                      int *episode, int *step, int n_env, int obs_elems, int episode_len,
                      const int *len_per_env, unsigned int seed, unsigned int env_id0, void *stream);

int rlx_synth_env_step(int kind, void *next_obs, void *reset_obs, float *reward,
                       unsigned char *game_over, int *episode, int *step, int n_env, int obs_elems,
                       int episode_len, unsigned int seed, unsigned int env_id0, void *stream) {
    return synth_step(kind, next_obs, reset_obs, reward, game_over, episode, step, n_env, obs_elems, episode_len,
                      nullptr, seed, env_id0, stream);
}

int rlx_synth_env_step_lengths(int kind, void *next_obs, void *reset_obs, float *reward,
                               unsigned char *game_over, int *episode, int *step, int n_env, int obs_elems,
                               const int *episode_len_per_env, unsigned int seed, unsigned int env_id0,
                               void *stream) {
    RLX_REQUIRE(episode_len_per_env, "rlx_synth_env_step_lengths: null length table");
    return synth_step(kind, next_obs, reset_obs, reward, game_over, episode, step, n_env, obs_elems, 1,
                      episode_len_per_env, seed, env_id0, stream);
}

static int synth_step(int kind, void *next_obs, void *reset_obs, float *reward, unsigned char *game_over,
                      int *episode, int *step, int n_env, int obs_elems, int episode_len,
                      const int *len_per_env, unsigned int seed, unsigned int env_id0, void *stream) {
    RLX_REQUIRE(next_obs && reset_obs && reward && game_over && episode && step,
                "rlx_synth_env_step: null pointer");
    RLX_REQUIRE(kind == 0 || kind == 1, "rlx_synth_env_step: kind must be 0 (image) or 1 (vector)");
    RLX_REQUIRE(n_env > 0 && obs_elems > 0 && episode_len > 0 && (kind == 1 || obs_elems % 16 == 0),
                "rlx_synth_env_step: bad sizes");
    hipStream_t s = rlx::as_stream(stream);
    const int work = kind == 0 ? obs_elems / 16 : obs_elems;
    dim3 grid(rlx::grid_for(work, 64, 16), n_env);
    const int fused = grid.x == 1;
    RLX_LAUNCH((env_step_kernel), grid, 64, 0, s, kind, next_obs, reset_obs, reward, game_over, episode, step,
                                        n_env, obs_elems, episode_len, len_per_env, seed, env_id0, fused);
    RLX_LAUNCH_CHECK();
    if (!fused) {
        RLX_LAUNCH((env_advance_kernel), (n_env + 63) / 64, 64, 0, s, episode, step, game_over, n_env);
        RLX_LAUNCH_CHECK();
    }
    return RLX_OK;
}

}  // extern "C"
