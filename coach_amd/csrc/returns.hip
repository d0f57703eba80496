// K8 / K12 — returns, generalized advantage estimation, advantage standardisation and per-env
// episode statistics on gfx950.
//
// Replaces, in the reference (paths under rl_coach/):
//   * ActorCriticAgent.discount + get_general_advantage_estimation_values
//       agents/actor_critic_agent.py:108-125   (scipy.signal.lfilter per episode, fp64)
//   * ClippedPPOAgent.fill_advantages episode loop + standardisation
//       agents/clipped_ppo_agent.py:181-201    (np.append per episode; (adv-mean)/std, no epsilon)
//   * Episode.update_discounted_rewards       core_types.py:771-801 (O(T^2) pad-and-add)
//   * Signal accumulation / Agent.handle_episode_ended / update_log episode totals
//       utils.py:162-212, agents/agent.py:509-601
//
// The recurrences  A_t = delta_t + (gamma*lambda) * A_{t+1}   and   R_t = r_t + gamma * R_{t+1}
// (both reset where game_over is set) are first-order linear, i.e. a scan over affine maps
// x -> b + a*x, which is associative: one workgroup per trajectory runs a reverse wave64 scan
// (shuffle-based Hillis-Steele inside a wave, LDS across the 4 waves, running carry across tiles).
// All arithmetic is fp64 like the reference; inputs/outputs are the fp32 arrays the network uses.
// HBM-bound: 17 B per step (r, V, done in; advantage, value target out) — SURVEY.md §8(d).
// Compiled with -ffp-contract=off (see Makefile).
#include "rlx_common.hpp"

namespace {

constexpr int kScanBlock = 256;
constexpr int kItems = 4;
constexpr int kTile = kScanBlock * kItems;

struct Affine {
    double a, b;   // x -> b + a * x
};

// later ∘ earlier  (apply `earlier` first)
__device__ __forceinline__ Affine compose(const Affine later, const Affine earlier) {
    Affine r;
    r.a = later.a * earlier.a;
    r.b = later.b + later.a * earlier.b;
    return r;
}

__device__ __forceinline__ Affine wave_inclusive_scan(Affine v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double pa = __shfl_up(v.a, d, 64);
        double pb = __shfl_up(v.b, d, 64);
        if (lane >= d) {
            Affine e{pa, pb};
            v = compose(v, e);
        }
    }
    return v;
}

// mode 0: GAE.  in: rewards, values (V(s_t)), dones;   out: adv (fp64), vtarget (fp32)
// mode 1: discounted returns. in: rewards, dones;      out: ret (fp64 in `adv`), optional fp32 copy
// Sequence q occupies [q*seq_len, (q+1)*seq_len).  Element j of the reversed order is t = L-1-j.
__global__ void __launch_bounds__(kScanBlock)
reverse_scan_kernel(const float *__restrict__ rewards, const float *__restrict__ values,
                    const unsigned char *__restrict__ dones,
                    const float *__restrict__ bootstrap, long long seq_len, double gamma,
                    double coef /* gamma*lambda (mode 0) or gamma (mode 1) */, int mode,
                    double *__restrict__ out64, float *__restrict__ out32) {
    __shared__ Affine wave_tot[kScanBlock / 64];
    __shared__ double carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long base = (long long)blockIdx.x * seq_len;
    rewards += base;
    dones += base;
    if (values) values += base;
    out64 += base;
    if (out32) out32 += base;
    const double boot = bootstrap ? (double)bootstrap[blockIdx.x] : 0.0;
    if (tid == 0) carry_s = 0.0;
    __syncthreads();

    for (long long tile0 = 0; tile0 < seq_len; tile0 += kTile) {
        Affine el[kItems];
        Affine local{1.0, 0.0};
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const long long j = tile0 + (long long)tid * kItems + k;
            Affine f{1.0, 0.0};                       // identity for the padding past the end
            if (j < seq_len) {
                const long long t = seq_len - 1 - j;
                const bool done = dones[t] != 0;
                const double r = (double)rewards[t];
                if (mode == 0) {
                    const double v = (double)values[t];
                    // values[-1] of the episode is the bootstrap 0 (clipped_ppo_agent.py:188-189)
                    const double vnext =
                        done ? 0.0 : (t + 1 < seq_len ? (double)values[t + 1] : boot);
                    f.b = r + gamma * vnext - v;      // deltas (actor_critic_agent.py:118)
                } else {
                    f.b = r;
                }
                f.a = done ? 0.0 : coef;
            }
            el[k] = f;
            local = compose(f, local);
        }
        // inclusive scan of the per-thread aggregates across the block
        Affine incl = wave_inclusive_scan(local, lane);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        Affine prefix{1.0, 0.0};                       // everything before this thread in the tile
        for (int w = 0; w < wid; ++w) prefix = compose(wave_tot[w], prefix);
        {
            double pa = __shfl_up(incl.a, 1, 64), pb = __shfl_up(incl.b, 1, 64);
            if (lane > 0) prefix = compose(Affine{pa, pb}, prefix);
        }
        const double carry = carry_s;                  // value at the element just before the tile
        double x = prefix.b + prefix.a * carry;
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const long long j = tile0 + (long long)tid * kItems + k;
            if (j < seq_len) {
                x = el[k].b + el[k].a * x;
                const long long t = seq_len - 1 - j;
                out64[t] = x;
                if (out32) out32[t] = (float)(mode == 0 ? x + (double)values[t] : x);
            }
        }
        __syncthreads();                               // everyone has read carry_s / wave_tot
        if (tid == kScanBlock - 1) carry_s = x;        // last element of the tile
        __syncthreads();
    }
}

// (x - mean) / std  with numpy's definitions: mean = sum/n, std = sqrt(mean(|x-mean|^2)).
// One workgroup; deterministic tree reductions in fp64.
__global__ void __launch_bounds__(1024)
standardize_kernel(const double *__restrict__ x, long long n, float *__restrict__ out32,
                   double *__restrict__ out64, double *__restrict__ stats) {
    __shared__ double red[1024];
    __shared__ double mean_s, std_s;
    const int tid = threadIdx.x, nt = blockDim.x;
    double s = 0.0;
    for (long long i = tid; i < n; i += nt) s += x[i];
    red[tid] = s;
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    if (tid == 0) mean_s = red[0] / (double)n;
    __syncthreads();
    const double mean = mean_s;
    s = 0.0;
    for (long long i = tid; i < n; i += nt) {
        double d = x[i] - mean;
        s += d * d;
    }
    red[tid] = s;
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        std_s = sqrt(red[0] / (double)n);
        if (stats) {
            stats[0] = mean;
            stats[1] = std_s;
        }
    }
    __syncthreads();
    const double sd = std_s;
    for (long long i = tid; i < n; i += nt) {
        double v = (x[i] - mean) / sd;
        if (out32) out32[i] = (float)v;
        if (out64) out64[i] = v;
    }
}

// K12: per-env running episode totals, segmented at game_over.
// acc[0..7] = {episodes, sum_return, sumsq_return, max_return, min_return, sum_len, max_len, min_len}
// (one workgroup; ends with every thread past its last barrier)
__device__ __forceinline__ void
episode_stats_body(const float *__restrict__ reward, const unsigned char *__restrict__ done,
                   double *__restrict__ ep_return, int *__restrict__ ep_len, int n_env,
                   double *__restrict__ acc, double *__restrict__ last_return, int *__restrict__ last_len) {
    __shared__ double s_cnt[1024], s_sum[1024], s_sq[1024], s_max[1024], s_min[1024];
    __shared__ double s_len[1024], s_lmax[1024], s_lmin[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    double cnt = 0, sum = 0, sq = 0, mx = -__builtin_huge_val(), mn = __builtin_huge_val();
    double ln = 0, lmx = -__builtin_huge_val(), lmn = __builtin_huge_val();
    for (int e = tid; e < n_env; e += nt) {
        double ret = ep_return[e] + (double)reward[e];
        int len = ep_len[e] + 1;
        if (done[e]) {
            cnt += 1.0;
            sum += ret;
            sq += ret * ret;
            mx = fmax(mx, ret);
            mn = fmin(mn, ret);
            ln += (double)len;
            lmx = fmax(lmx, (double)len);
            lmn = fmin(lmn, (double)len);
            if (last_return) last_return[e] = ret;
            if (last_len) last_len[e] = len;
            ret = 0.0;
            len = 0;
        }
        ep_return[e] = ret;
        ep_len[e] = len;
    }
    s_cnt[tid] = cnt; s_sum[tid] = sum; s_sq[tid] = sq; s_max[tid] = mx; s_min[tid] = mn;
    s_len[tid] = ln; s_lmax[tid] = lmx; s_lmin[tid] = lmn;
    __syncthreads();
    for (int d = nt >> 1; d > 0; d >>= 1) {
        if (tid < d) {
            s_cnt[tid] += s_cnt[tid + d];
            s_sum[tid] += s_sum[tid + d];
            s_sq[tid] += s_sq[tid + d];
            s_max[tid] = fmax(s_max[tid], s_max[tid + d]);
            s_min[tid] = fmin(s_min[tid], s_min[tid + d]);
            s_len[tid] += s_len[tid + d];
            s_lmax[tid] = fmax(s_lmax[tid], s_lmax[tid + d]);
            s_lmin[tid] = fmin(s_lmin[tid], s_lmin[tid + d]);
        }
        __syncthreads();
    }
    if (tid == 0 && s_cnt[0] > 0) {
        acc[0] += s_cnt[0];
        acc[1] += s_sum[0];
        acc[2] += s_sq[0];
        acc[3] = fmax(acc[3], s_max[0]);
        acc[4] = fmin(acc[4], s_min[0]);
        acc[5] += s_len[0];
        acc[6] = fmax(acc[6], s_lmax[0]);
        acc[7] = fmin(acc[7], s_lmin[0]);
    }
}

__global__ void __launch_bounds__(1024)
episode_stats_kernel(const float *__restrict__ reward, const unsigned char *__restrict__ done,
                     double *__restrict__ ep_return, int *__restrict__ ep_len, int n_env,
                     double *__restrict__ acc, double *__restrict__ last_return,
                     int *__restrict__ last_len) {
    episode_stats_body(reward, done, ep_return, ep_len, n_env, acc, last_return, last_len);
}

// What an agent does with an environment response, for a handful of vector-observation envs, in ONE launch of one
// workgroup: reward filter (rlx_reward_filter) -> episode totals (rlx_episode_stats_step) -> the transition store
// (rlx_copy_columns: action, filtered reward, game_over, current state, next state -> replay row dst_rows[e]) ->
// the envs' new current state (rlx_select_rows: reset observation where the episode ended).  Same arithmetic and
// the same order as the four launches it replaces; the phases are separated by workgroup barriers.
struct ObserveArgs {
    const float *reward_in; float *reward_out; double rescale; int use_hi; double hi; int use_lo; double lo;
    const unsigned char *done, *stored_done;
    double *ep_return; int *ep_len; double *acc; double *last_return; int *last_len;
    const unsigned char *actions; long long action_row_bytes;
    unsigned char *cur_state; const unsigned char *next_obs, *reset_obs; long long obs_row_bytes;
    unsigned char *mem_action, *mem_reward, *mem_done, *mem_obs, *mem_next_obs;
    const int *dst_rows; long long mem_rows; int *status; int n_env;
};

__device__ __forceinline__ void observe_copy_rows(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                                                  long long row_bytes, int n, const int *__restrict__ dst_rows,
                                                  long long dst_limit, int *__restrict__ status) {
    const bool words = ((row_bytes | (long long)(uintptr_t)src | (long long)(uintptr_t)dst) & 3) == 0;
    const long long per_row = words ? row_bytes >> 2 : row_bytes;
    for (long long t = threadIdx.x; t < per_row * n; t += blockDim.x) {
        const long long r = t / per_row, e = t - r * per_row;
        const long long d = dst_rows ? (long long)dst_rows[r] : r;
        if (d < 0 || d >= dst_limit) {
            if (e == 0) atomicOr(status, 1);
            continue;
        }
        if (words)
            reinterpret_cast<uint32_t *>(dst)[d * per_row + e] = reinterpret_cast<const uint32_t *>(src)[r * per_row + e];
        else
            dst[d * per_row + e] = src[r * per_row + e];
    }
}

// The same for a lockstep ROLLOUT buffer (Clipped PPO's acting step): reward filter -> episode totals -> the step's action /
// filtered reward / game_over columns at rows [row0, row0 + n_env) (the frames go through rlx_imgreplay_append, vector
// observations through their own copy).  Uses ObserveArgs with dst_rows = null and the column pointers already at row0.
__global__ void __launch_bounds__(1024) rollout_observe_kernel(const ObserveArgs a) {
    for (int e = threadIdx.x; e < a.n_env; e += blockDim.x) {
        double r = (double)a.reward_in[e] * a.rescale;
        if (a.use_hi) r = fmin(r, a.hi);
        if (a.use_lo) r = fmax(r, a.lo);
        a.reward_out[e] = (float)r;
    }
    __syncthreads();
    episode_stats_body(a.reward_out, a.done, a.ep_return, a.ep_len, a.n_env, a.acc, a.last_return, a.last_len);
    observe_copy_rows(a.actions, a.mem_action, a.action_row_bytes, a.n_env, nullptr, a.mem_rows, a.status);
    observe_copy_rows(reinterpret_cast<const unsigned char *>(a.reward_out), a.mem_reward, 4, a.n_env, nullptr, a.mem_rows,
                      a.status);
    observe_copy_rows(a.stored_done, a.mem_done, 1, a.n_env, nullptr, a.mem_rows, a.status);
}

__global__ void __launch_bounds__(1024) observe_step_kernel(const ObserveArgs a) {
    for (int e = threadIdx.x; e < a.n_env; e += blockDim.x) {
        double r = (double)a.reward_in[e] * a.rescale;
        if (a.use_hi) r = fmin(r, a.hi);
        if (a.use_lo) r = fmax(r, a.lo);
        a.reward_out[e] = (float)r;
    }
    __syncthreads();
    episode_stats_body(a.reward_out, a.done, a.ep_return, a.ep_len, a.n_env, a.acc, a.last_return, a.last_len);
    if (a.mem_obs) {
        observe_copy_rows(a.actions, a.mem_action, a.action_row_bytes, a.n_env, a.dst_rows, a.mem_rows, a.status);
        observe_copy_rows(reinterpret_cast<const unsigned char *>(a.reward_out), a.mem_reward, 4, a.n_env, a.dst_rows,
                          a.mem_rows, a.status);
        observe_copy_rows(a.stored_done, a.mem_done, 1, a.n_env, a.dst_rows, a.mem_rows, a.status);
        observe_copy_rows(a.cur_state, a.mem_obs, a.obs_row_bytes, a.n_env, a.dst_rows, a.mem_rows, a.status);
        observe_copy_rows(a.next_obs, a.mem_next_obs, a.obs_row_bytes, a.n_env, a.dst_rows, a.mem_rows, a.status);
    }
    __syncthreads();
    const bool words = ((a.obs_row_bytes | (long long)(uintptr_t)a.cur_state | (long long)(uintptr_t)a.next_obs |
                         (long long)(uintptr_t)a.reset_obs) & 3) == 0;
    const long long per_row = words ? a.obs_row_bytes >> 2 : a.obs_row_bytes;
    for (long long t = threadIdx.x; t < per_row * a.n_env; t += blockDim.x) {
        const long long r = t / per_row;
        if (words)
            reinterpret_cast<uint32_t *>(a.cur_state)[t] =
                a.done[r] ? reinterpret_cast<const uint32_t *>(a.reset_obs)[t] : reinterpret_cast<const uint32_t *>(a.next_obs)[t];
        else
            a.cur_state[t] = a.done[r] ? a.reset_obs[t] : a.next_obs[t];
    }
}

}  // namespace

namespace {
// Signal statistics (utils.py:162-212 `Signal`: a list of samples per episode, get_mean / get_stdev / get_max /
// get_min at log time, agent.py:548-552).  Device form: one fp64 record per signal,
//   {count, sum, sum of squares, max, min},
// that every update folds its new samples into; the host reads the few records once per logged episode.
// One workgroup per launch, up to 8 sources, each reduced in a fixed order (reproducible).
struct SignalSources {
    const void *ptr[8];
    int n[8];
    int is_f64[8];
    int row[8];
    int count;
};

__global__ void __launch_bounds__(256) signals_accumulate_kernel(SignalSources src, double *__restrict__ table) {
    __shared__ double red[3][256];
    __shared__ double ext[2][256];
    const int tid = threadIdx.x;
    for (int s = 0; s < src.count; ++s) {
        double sum = 0.0, sq = 0.0, mx = -__builtin_huge_val(), mn = __builtin_huge_val();
        for (int i = tid; i < src.n[s]; i += 256) {
            const double v = src.is_f64[s] ? static_cast<const double *>(src.ptr[s])[i]
                                           : (double)static_cast<const float *>(src.ptr[s])[i];
            sum += v;
            sq += v * v;
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
        red[0][tid] = sum; red[1][tid] = sq; ext[0][tid] = mx; ext[1][tid] = mn;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (tid < d) {
                red[0][tid] += red[0][tid + d];
                red[1][tid] += red[1][tid + d];
                ext[0][tid] = ext[0][tid + d] > ext[0][tid] ? ext[0][tid + d] : ext[0][tid];
                ext[1][tid] = ext[1][tid + d] < ext[1][tid] ? ext[1][tid + d] : ext[1][tid];
            }
            __syncthreads();
        }
        if (tid == 0 && src.n[s] > 0) {
            double *r = table + 5 * src.row[s];
            r[0] += (double)src.n[s];
            r[1] += red[0][0];
            r[2] += red[1][0];
            r[3] = ext[0][0] > r[3] ? ext[0][0] : r[3];
            r[4] = ext[1][0] < r[4] ? ext[1][0] : r[4];
        }
        __syncthreads();
    }
}

__global__ void signals_reset_kernel(double *table, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double *r = table + 5 * i;
    r[0] = 0.0; r[1] = 0.0; r[2] = 0.0; r[3] = -__builtin_huge_val(); r[4] = __builtin_huge_val();
}
}  // namespace

namespace {
// Episode.update_discounted_rewards (core_types.py:771-801) for ONE completed episode whose T transitions sit in a
// time-major ring (row of step k = ((s0 + k) mod ring_steps) * n_env + env).  The reference adds
// current_discount * rewards[i:] for i = 1 .. n-1 onto a float64 copy of the rewards, current_discount growing by
// repeated multiplication: out[t] = ((r[t] + g1 r[t+1]) + g2 r[t+2]) + ...  with g_j = g_{j-1} * discount — the same
// order and the same products here, so the column is bit-identical to the reference's n_step_discounted_rewards.
// (-ffp-contract=off: no fused multiply-add.)  O(T * n) like the reference; one thread per transition.
__global__ void episode_nstep_returns_kernel(const float *__restrict__ rewards, double *__restrict__ out,
                                             double *__restrict__ compact, long long s0, int T, int env,
                                             int n_env, long long ring_steps, double discount, int n_step) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int n = (n_step == -1 || n_step > T) ? T : n_step;
    auto row = [&](int k) { return ((s0 + k) % ring_steps) * n_env + env; };
    double acc = (double)rewards[row(t)];
    double g = discount;
    for (int j = 1; j < n; ++j) {
        if (t + j < T) acc += g * (double)rewards[row(t + j)];
        g *= discount;
    }
    if (out) out[row(t)] = acc;
    if (compact) compact[t] = acc;
}
}  // namespace

extern "C" {

int rlx_gae(const float *rewards, const float *values, const unsigned char *game_overs,
            const float *bootstrap_values, int n_seq, long long seq_len, double discount,
            double gae_lambda, double *advantages, float *value_targets, void *stream) {
    RLX_REQUIRE(rewards && values && game_overs && advantages, "rlx_gae: null pointer");
    RLX_REQUIRE(n_seq > 0 && seq_len > 0, "rlx_gae: empty input (n_seq=%d seq_len=%lld)", n_seq,
                seq_len);
    RLX_LAUNCH((reverse_scan_kernel), n_seq, kScanBlock, 0, rlx::as_stream(stream), rewards, values, game_overs, bootstrap_values, seq_len, discount, discount * gae_lambda, 0,
        advantages, value_targets);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_discounted_returns(const float *rewards, const unsigned char *game_overs, int n_seq,
                           long long seq_len, double discount, double *returns64,
                           float *returns32, void *stream) {
    RLX_REQUIRE(rewards && game_overs && returns64, "rlx_discounted_returns: null pointer");
    RLX_REQUIRE(n_seq > 0 && seq_len > 0, "rlx_discounted_returns: empty input");
    RLX_LAUNCH((reverse_scan_kernel), n_seq, kScanBlock, 0, rlx::as_stream(stream), rewards, nullptr, game_overs, nullptr, seq_len, discount, discount, 1, returns64,
        returns32);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_signals_reset(double *table, int n_signals, void *stream) {
    RLX_REQUIRE(table && n_signals > 0, "rlx_signals_reset: bad arguments");
    RLX_LAUNCH((signals_reset_kernel), (n_signals + 63) / 64, 64, 0, rlx::as_stream(stream), table, n_signals);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_signals_accumulate(const rlx_signal_source *sources_host, int n_sources, double *table, int n_signals,
                           void *stream) {
    RLX_REQUIRE(sources_host && table && n_sources > 0 && n_sources <= 8 && n_signals > 0,
                "rlx_signals_accumulate: 1..8 sources, got %d", n_sources);
    SignalSources src;
    src.count = n_sources;
    for (int i = 0; i < n_sources; ++i) {
        RLX_REQUIRE(sources_host[i].values && sources_host[i].n >= 0 && sources_host[i].signal >= 0 &&
                        sources_host[i].signal < n_signals,
                    "rlx_signals_accumulate: source %d is invalid", i);
        src.ptr[i] = sources_host[i].values;
        src.n[i] = sources_host[i].n;
        src.is_f64[i] = sources_host[i].is_f64;
        src.row[i] = sources_host[i].signal;
    }
    RLX_LAUNCH((signals_accumulate_kernel), 1, 256, 0, rlx::as_stream(stream), src, table);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_episode_nstep_returns(const float *rewards, double *out, double *compact_out, long long first_step,
                              int length, int env, int n_env, long long ring_steps, double discount, int n_step,
                              void *stream) {
    RLX_REQUIRE(rewards && (out || compact_out), "rlx_episode_nstep_returns: null pointer");
    RLX_REQUIRE(length > 0 && n_env > 0 && env >= 0 && env < n_env && ring_steps >= length && first_step >= 0,
                "rlx_episode_nstep_returns: bad episode geometry (length=%d ring_steps=%lld)", length, ring_steps);
    RLX_REQUIRE(n_step == -1 || n_step >= 1,
                "n-step should be an integer with value >= 1, or set to -1 for always setting to episode length.");
    RLX_LAUNCH((episode_nstep_returns_kernel), (length + 255) / 256, 256, 0, rlx::as_stream(stream), rewards, out, compact_out, first_step, length, env, n_env, ring_steps, discount, n_step);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_standardize(const double *x, long long n, float *out32, double *out64, double *mean_std,
                    void *stream) {
    RLX_REQUIRE(x && (out32 || out64), "rlx_standardize: null pointer");
    RLX_REQUIRE(n > 0, "rlx_standardize: empty input");
    RLX_LAUNCH((standardize_kernel), 1, 1024, 0, rlx::as_stream(stream), x, n, out32, out64, mean_std);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_episode_stats_init(double *ep_return, int *ep_len, int n_env, double *acc, void *stream) {
    RLX_REQUIRE(ep_return && ep_len && n_env > 0, "rlx_episode_stats_init: bad arguments");
    hipStream_t s = rlx::as_stream(stream);
    RLX_HIP(hipMemsetAsync(ep_return, 0, sizeof(double) * n_env, s));
    RLX_HIP(hipMemsetAsync(ep_len, 0, sizeof(int) * n_env, s));
    if (!acc) return RLX_OK;      // acc == NULL: only the running episodes restart (a forced reset)
    const double inf = __builtin_huge_val();
    const double init[8] = {0, 0, 0, -inf, inf, 0, -inf, inf};
    RLX_HIP(hipMemcpyAsync(acc, init, sizeof(init), hipMemcpyHostToDevice, s));
    return RLX_OK;
}

int rlx_episode_stats_step(const float *reward, const unsigned char *game_over, double *ep_return,
                           int *ep_len, int n_env, double *acc, double *last_return,
                           int *last_len, void *stream) {
    RLX_REQUIRE(reward && game_over && ep_return && ep_len && acc,
                "rlx_episode_stats_step: null pointer");
    RLX_REQUIRE(n_env > 0, "rlx_episode_stats_step: n_env must be positive");
    int threads = n_env >= 1024 ? 1024 : 64;
    while (threads < n_env && threads < 1024) threads <<= 1;
    RLX_LAUNCH((episode_stats_kernel), 1, threads, 0, rlx::as_stream(stream), reward, game_over, ep_return, ep_len, n_env, acc, last_return, last_len);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_rollout_observe_step(const float *reward, float *filtered_reward, double reward_rescale, int has_clip,
                             double clip_low, double clip_high, const unsigned char *game_over, double *ep_return,
                             int *ep_len, double *acc, double *last_return, int *last_len, const void *actions,
                             long long action_row_bytes, void *mem_action, float *mem_reward,
                             unsigned char *mem_game_over, long long row0, long long mem_rows, int n_env, int *status,
                             void *stream) {
    RLX_REQUIRE(reward && filtered_reward && game_over && ep_return && ep_len && acc,
                "rlx_rollout_observe_step: null reward / episode-statistics pointer");
    RLX_REQUIRE(actions && action_row_bytes > 0 && mem_action && mem_reward && mem_game_over && status,
                "rlx_rollout_observe_step: null column pointer");
    RLX_REQUIRE(n_env > 0 && n_env <= 1024, "rlx_rollout_observe_step: 1 <= n_env <= 1024 (got %d)", n_env);
    RLX_REQUIRE(row0 >= 0 && mem_rows > 0, "rlx_rollout_observe_step: bad rows");
    ObserveArgs a{};
    a.reward_in = reward; a.reward_out = filtered_reward; a.rescale = reward_rescale;
    a.use_hi = has_clip && clip_high != 0.0; a.hi = clip_high;        // (a bound equal to 0 is not applied: rlx_reward_filter)
    a.use_lo = has_clip && clip_low != 0.0; a.lo = clip_low;
    a.done = game_over; a.stored_done = game_over;
    a.ep_return = ep_return; a.ep_len = ep_len; a.acc = acc; a.last_return = last_return; a.last_len = last_len;
    a.actions = static_cast<const unsigned char *>(actions); a.action_row_bytes = action_row_bytes;
    a.mem_action = static_cast<unsigned char *>(mem_action) + row0 * action_row_bytes;
    a.mem_reward = reinterpret_cast<unsigned char *>(mem_reward + row0);
    a.mem_done = mem_game_over + row0;
    a.dst_rows = nullptr; a.mem_rows = mem_rows - row0;               // rows left from row0: beyond -> status bit 1
    a.status = status; a.n_env = n_env;
    int threads = n_env >= 1024 ? 1024 : 64;                          // (rlx_episode_stats_step's: the same reduction tree)
    while (threads < n_env && threads < 1024) threads <<= 1;
    RLX_LAUNCH((rollout_observe_kernel), 1, threads, 0, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_observe_step(const rlx_observe_desc *d, void *stream) {
    RLX_REQUIRE(d, "rlx_observe_step: null descriptor");
    RLX_REQUIRE(d->n_env > 0 && d->n_env <= 1024, "rlx_observe_step: 1 <= n_env <= 1024 (got %d)", d->n_env);
    RLX_REQUIRE(d->reward && d->filtered_reward && d->game_over && d->ep_return && d->ep_len && d->acc,
                "rlx_observe_step: null reward / episode-statistics pointer");
    RLX_REQUIRE(d->cur_state && d->next_obs && d->reset_obs && d->obs_row_bytes > 0,
                "rlx_observe_step: null observation pointer");
    RLX_REQUIRE((long long)d->n_env * d->obs_row_bytes <= (1 << 20),
                "rlx_observe_step: one workgroup moves the observations; %lld bytes is too much",
                (long long)d->n_env * d->obs_row_bytes);
    const bool store = d->mem_obs != nullptr;
    RLX_REQUIRE(!store || (d->actions && d->action_row_bytes > 0 && d->mem_action && d->mem_reward && d->mem_game_over &&
                           d->mem_next_obs && d->stored_game_over && d->mem_rows > 0 && d->status),
                "rlx_observe_step: incomplete replay columns");
    ObserveArgs a;
    a.reward_in = d->reward; a.reward_out = d->filtered_reward; a.rescale = d->reward_rescale;
    // `if self.clipping_high:` / `if self.clipping_low:` — a bound equal to 0 is not applied (rlx_reward_filter)
    a.use_hi = d->has_clip && d->clip_high != 0.0; a.hi = d->clip_high;
    a.use_lo = d->has_clip && d->clip_low != 0.0; a.lo = d->clip_low;
    a.done = d->game_over; a.stored_done = d->stored_game_over;
    a.ep_return = d->ep_return; a.ep_len = d->ep_len; a.acc = d->acc; a.last_return = d->last_return;
    a.last_len = d->last_len;
    a.actions = static_cast<const unsigned char *>(d->actions); a.action_row_bytes = d->action_row_bytes;
    a.cur_state = static_cast<unsigned char *>(d->cur_state);
    a.next_obs = static_cast<const unsigned char *>(d->next_obs);
    a.reset_obs = static_cast<const unsigned char *>(d->reset_obs);
    a.obs_row_bytes = d->obs_row_bytes;
    a.mem_action = static_cast<unsigned char *>(d->mem_action);
    a.mem_reward = reinterpret_cast<unsigned char *>(d->mem_reward);
    a.mem_done = d->mem_game_over;
    a.mem_obs = static_cast<unsigned char *>(d->mem_obs);
    a.mem_next_obs = static_cast<unsigned char *>(d->mem_next_obs);
    a.dst_rows = d->dst_rows; a.mem_rows = d->mem_rows; a.status = d->status; a.n_env = d->n_env;
    int threads = 64;
    const long long work = (long long)d->n_env * (d->obs_row_bytes / 4 + 1);
    while (threads < work && threads < 1024) threads <<= 1;
    RLX_LAUNCH((observe_step_kernel), 1, threads, 0, rlx::as_stream(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
