/*
 * rlx.h — C ABI of librlx.so, the MI355X (gfx950) hot-path library behind the
 * rl_coach-compatible Python adapters in coach_amd/.
 *
 * The reference (IntelLabs/coach, rl_coach 1.0.1) has NO C interface: its plug
 * points are Python classes.  Every entry point below therefore
 * cites the reference *Python* routine whose arithmetic it replaces (file:line under
 * /root/reference/rl_coach/), and INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add to call it.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Every function returns RLX_OK (0) or a negative rlx_status; the human readable
 *     reason is available from rlx_last_error() (thread local).
 *   - No function allocates device memory or synchronises the device unless its
 *     name says so (rlx_*_sync / rlx_event_elapsed_ms): all are hipGraph-capturable.
 *   - Layouts are row-major, innermost dimension last (NHWC for images).
 */
#ifndef RLX_H
#define RLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rlx_status {
    RLX_OK = 0,
    RLX_ERR_INVALID_ARG = -1, /* bad size / null pointer / unsupported combination */
    RLX_ERR_HIP = -2,         /* a HIP runtime call or a kernel launch failed       */
    RLX_ERR_UNSUPPORTED = -3
} rlx_status;

/* ------------------------------------------------------------------ runtime -- */
int rlx_abi_version(void);               /* bumps when a signature changes          */
const char *rlx_last_error(void);        /* message of the last failure (this thread) */
const char *rlx_build_arch(void);        /* "gfx950"                                  */
int rlx_device_count(int *count_host);   /* replaces coach.py:61-84 (cuDeviceGetCount) */
int rlx_stream_sync(void *stream);
int rlx_event_create(void **event_host);
int rlx_event_destroy(void *event);
int rlx_event_record(void *event, void *stream);
int rlx_event_elapsed_ms(void *start, void *stop, float *ms_host); /* syncs on stop */

/* Latency probe for bench.py's `box` block (no reference counterpart): one lane follows `steps` dependent 4-byte loads
 * i = chain[i] from `start` and stores the last index in *out; time it with events. */
int rlx_probe_chase(const int *chain, int start, int steps, int *out, void *stream);
/* Matrix-pipe clock probe (measurement utility): `workgroups` x 4 waves each run `iters` dependent
 * v_mfma_f32_32x32x2_f32; out[2 w] = shader-clock cycles and out[2 w + 1] = ticks of the constant 100 MHz counter that
 * workgroup w's chain took: cycles per MFMA, and the clock the shader really ran at under that load. */
int rlx_probe_mfma(int workgroups, int iters, long long *out, float *sink, void *stream);
/* The same for either fp32 MFMA shape (small_shape != 0: v_mfma_f32_16x16x4_f32) with `chains` (1, 2 or 4) independent
 * accumulators per wave issued round-robin, `iters` MFMAs per chain: chains = 1 is the dependent-accumulator latency,
 * chains >= 2 the issue rate of the matrix pipe. */
int rlx_probe_mfma_shape(int workgroups, int iters, int small_shape, int chains, long long *out, float *sink, void *stream);
/* Request-pipelining probe (measurement utility): every wave of `workgroups` workgroups issues `requests` (1, 2, 4 or 8)
 * back-to-back 16-byte-per-lane loads from src — global -> LDS requests (lds_dma != 0: what the GEMM ring issues) or
 * ordinary loads into registers — and waits for all of them; out[w] = shader-clock cycles of workgroup w.  src: at least
 * 8 MB + 4 KB * workgroups + 256 * round bytes. */
int rlx_probe_requests(int workgroups, int requests, int lds_dma, int round, const float *src, long long *out, float *sink,
                       void *stream);

/* In-process kernel timer: between rlx_profile_begin and rlx_profile_end every kernel this library launches (EAGER
 * launches on any stream; do not arm it inside a stream capture) carries its own start / stop event pair filled from
 * the dispatch's begin / end timestamps — the per-dispatch duration a kernel trace reports (the reference has no
 * counterpart: its TF timeline is the closest thing, architectures/tensorflow_components/architecture.py has none on
 * this path).  rlx_profile_read(i) returns the kernel's name (a static string: the template instance as written at
 * the launch site) and its duration in ms, in launch order; it waits for that kernel.  max_records bounds the trace:
 * launches beyond it go out untimed and rlx_profile_end then FAILS (the records taken stay readable). */
int rlx_profile_begin(int max_records);
int rlx_profile_end(int *n_records_host);
int rlx_profile_read(int index, const char **name_host, float *ms_host);

/* The second and third convolution of the Atari torso (embedder "Medium": 64 x 4 x 4 / 2 on 20 x 20 x 32, then 64 x 3 x 3 / 1)
 * as ONE launch: half an image of one tower per workgroup, the second layer's output stays in LDS and feeds the third
 * (tf.layers.conv2d twice, architectures/tensorflow_components/layers.py:108-121).  y2 / y3 receive both layers'
 * activations (the backward pass reads them).  wave_groups = 2 / 4: bit-identical to the two rlx_gemm launches it replaces
 * where those run on 32 x 64 tiles with two wave groups per K slab / on 32 x 32 tiles with four (coach_amd/csrc/conv_fused.hip).  rlx_conv23_forward_supported: 1 for
 * the geometry the kernel is compiled for. */
int rlx_conv23_forward_supported(int H, int W, int C, int k2, int s2, int c2, int k3, int s3, int c3);
/* weight slabs in rlx_conv23_forward's LDS ring (2, 3, 4, 6, 8) and slabs per synchronisation step (1, or 2 with depth
 * >= 4): process-wide, for same-process A/Bs */
int rlx_conv23_depth(int depth, int slabs_per_step);
/* diagnostics: while buffer != NULL every workgroup of rlx_conv23_forward records 10 ns ticks at [8 w + 0 .. 5]: entry,
 * conv2's first slab and the conv1 rows staged, conv2's slab loop done, conv3's first slab staged (conv2's epilogue done),
 * conv3's slab loop done, exit.  buffer: 16 * batch * towers 64-bit words. */
int rlx_conv23_debug_stamps(void *buffer);
int rlx_conv23_forward(const float *x1, long long x1_tower_stride, const float *w2, long long w2_tower_stride,
                       const float *b2, long long b2_tower_stride, const float *w3, long long w3_tower_stride,
                       const float *b3, long long b3_tower_stride, float *y2, long long y2_tower_stride, float *y3,
                       long long y3_tower_stride, int batch, int towers, int activation, int wave_groups, void *stream);
/* ... and the FIRST convolution (32 x 8 x 8 / 4 on 84 x 84 x 4 uint8 frames, rescaled by 1 / a_div:
 * architectures/embedder_parameters.py + observation rescaling filter) in front of them in the same launch: the frame rows
 * of a half image go to LDS as bytes, conv1's output feeds conv2 from LDS and is also written to y1 (the backward pass reads
 * it).  x0: [towers or 1][batch][84][84][4] (x0_tower_stride in BYTES, 0 = every tower reads the same frames).
 * conv1_chunks = 1 / 3: bit-identical to rlx_gemm's conv1 launch where that sums K = 256 in one chain (64 x 64 tiles) / in
 * three chains over 96-long K chunks combined by its split-K reduce launch (128 x 32 tiles, 200 of them) — with operands
 * staged by registers (rlx_gemm_describe reports all of that; coach_amd/nn/graph.py asks it before taking this launch). */
int rlx_conv123_forward_supported(int H, int W, int C, int k1, int s1, int c1);
int rlx_conv123_forward(const unsigned char *x0, long long x0_tower_stride, float a_div, const float *w1,
                        long long w1_tower_stride, const float *b1, long long b1_tower_stride, float *y1,
                        long long y1_tower_stride, const float *w2, long long w2_tower_stride, const float *b2,
                        long long b2_tower_stride, const float *w3, long long w3_tower_stride, const float *b3,
                        long long b3_tower_stride, float *y2, long long y2_tower_stride, float *y3,
                        long long y3_tower_stride, int batch, int towers, int activation, int wave_groups,
                        int conv1_chunks, void *stream);

/* The INPUT-GRADIENT chain of the same two layers in the backward pass as one launch (tf.gradients through the two
 * tf.layers.conv2d, layers.py:108-121; architecture.py:312-385 accumulate_gradients): per half image of a tower
 * dcol3 = dz3 W3^T -> dz2 = col2im(dcol3) * act'(y2) -> dcol2 = dz2 W2^T -> dz1 = col2im(dcol2) * act'(y1), both column
 * matrices in LDS (coach_amd/csrc/conv_bwd_fused.hip).  dz3: gradient at conv3's pre-activation [towers][batch * 49][64];
 * y2 / y1: the stored activations of conv2 / conv1; dz2 / dz1 (out): gradients at their pre-activations (what the layers'
 * weight-gradient products read).  Bit-identical to rlx_gemm (dcol) + rlx_col2im per layer where rlx_gemm_describe
 * reports the LDS-DMA ring without a K split for the dcol products. */
int rlx_conv32_input_grad_supported(int H, int W, int C, int k2, int s2, int c2, int k3, int s3, int c3);
int rlx_conv32_input_grad(const float *dz3, long long dz3_tower_stride, const float *w3, long long w3_tower_stride,
                          const float *y2, long long y2_tower_stride, float *dz2, long long dz2_tower_stride,
                          const float *w2, long long w2_tower_stride, const float *y1, long long y1_tower_stride,
                          float *dz1, long long dz1_tower_stride, int batch, int towers, int activation, void *stream);
/* The same launch with PrioritizedExperienceReplay.update_priorities (prioritized_experience_replay.py:203-217; called right
 * behind learn_from_batch, agents/dqn_agent.py:106-109) of the batch's n <= 64 sampled leaves as ONE MORE workgroup: the
 * arguments of rlx_per_update as a structure.  The DQN update (batch 32, one tower) fills 64 of 256 CUs with this launch; the
 * priority update — a 10 us chain of dependent round trips, so far a launch of its own behind every update — runs beside it.
 * Its inputs (the TD errors) exist since the head's loss and nothing reads the trees before the next sample(): trees
 * bit-identical to rlx_per_update's (same device code: coach_amd/csrc/per_update_body.hpp). */
typedef struct rlx_per_update_desc {
    double *sum_tree, *min_tree, *max_tree;
    int capacity;
    const int *idx;
    const double *td_errors;
    int n;
    double alpha, epsilon;
    double *max_priority;
    int *status;
} rlx_per_update_desc;
int rlx_conv32_input_grad_per_update(const float *dz3, long long dz3_tower_stride, const float *w3, long long w3_tower_stride,
                                     const float *y2, long long y2_tower_stride, float *dz2, long long dz2_tower_stride,
                                     const float *w2, long long w2_tower_stride, const float *y1, long long y1_tower_stride,
                                     float *dz1, long long dz1_tower_stride, int batch, int towers, int activation,
                                     const rlx_per_update_desc *per, void *stream);
/* diagnostics as rlx_conv23_debug_stamps: [8 w + 0 .. 5] = entry, first product's operands staged, its K loop done,
 * second product's operands staged (= first gather done), its K loop done, exit */
int rlx_conv32_debug_stamps(void *buffer);
/* Process-wide (default 1): 1 = the second row tile of both products (3 of 35, 13 of 45 positions) runs as a 16-row tile
 * on v_mfma_f32_16x16x4_f32 with two accumulator chains per wave — half the MFMA cycles of the padded 32-row tile; rows
 * >= 32 of a column matrix then sum k in another order (last bits: -3 us per Clipped-PPO update, profiles/r06_conv32_tail16.txt); 0 = 32-row tiles throughout, bit-identical to
 * rlx_gemm + rlx_col2im. */
int rlx_conv32_tail_tiles(int sixteen_rows);
/* how many jobs ahead a wave of that launch requests its weight operands: 1 (default) or 2 (one more register set; results
 * unchanged) */
int rlx_conv32_prefetch(int jobs_ahead);

/* --------------------------------------------- prioritized replay (K5 / K6) -- */
/* Trees are fp64 array-heaps of 2*capacity-1 nodes exactly as the reference's
 * SegmentTree (memories/non_episodic/prioritized_experience_replay.py:43-156);
 * capacity must be a power of two (:176-179).  max_priority is one device double
 * mirroring PrioritizedExperienceReplay.maximal_priority (:186,:201).  `status` is a
 * device int the kernels OR error bits into (1 = leaf index out of range -> the
 * reference's ValueError at :123-126; 2 = negative error -> ValueError at :195;
 * 4 = a priority outside the domain of the bit-exact pow, see rlx_libm_pow).
 * Every `p ** alpha` / `(N*P) ** -beta` below is evaluated with rlx::libm_pow
 * (csrc/libm_pow.hpp): glibc's pow algorithm on glibc's tables, so leaves and importance
 * weights are bit-identical to the reference's CPython `**` (= libm pow) — no host round trip. */
int rlx_per_init(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                 double *max_priority, void *stream);               /* SegmentTree.__init__ :59-67 */
int rlx_per_store(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                  int start_leaf, int n, double alpha, double *max_priority, int *status,
                  void *stream);                                     /* .store :264-283 (n consecutive adds) */
int rlx_per_store_value(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                        int start_leaf, int n, double leaf_pa, double leaf_p,
                        double *max_priority, int *status,
                        void *stream);                               /* same, host-computed (p**alpha, p) */
int rlx_per_update(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                   const int *idx, const double *errors, int n, double alpha, double epsilon,
                   double *max_priority, int *status, void *stream); /* .update_priorities :203-217 */
/* Updates of at most path_max_leaves leaves (default and maximum 256; 0 = never) run on the kernel whose threads meet
 * only in LDS instead of the level-synchronous one.  Both write bit-identical trees; the knob exists for same-process
 * A/B timing (tools/ab_per_update.py). */
int rlx_per_tuning(int path_max_leaves);
int rlx_per_update_leaves(double *sum_tree, double *min_tree, double *max_tree, int capacity,
                          const int *idx, const double *leaf_pa, const double *leaf_p, int n,
                          double *max_priority, int *status, void *stream); /* same, host-computed p**alpha */
int rlx_per_sample(const double *sum_tree, const double *min_tree, int capacity,
                   const double *uniforms, int batch, double num_transitions, double beta,
                   int *out_idx, double *out_weight, double *out_priority, long long stored_total,
                   long long payload_rows, int *out_rows,
                   void *stream);                                    /* .sample :229-255; uniforms[i] = random.random();
                                                                        out_rows (optional): payload-ring row of every sampled
                                                                        leaf when the ring holds payload_rows >= capacity rows
                                                                        and stored_total transitions were stored so far */
int rlx_per_sample_top_steps(int steps);   /* A/B knob: the most tree levels rlx_per_sample descends from LDS (0 .. 11, default 8) */

int rlx_libm_pow(const double *x, const double *y, double *out, int n, int *status,
                 void *stream);  /* out[i] = x[i] ** y[i] as the host libm rounds it (test hook of
                                    the priority arithmetic above; status bit 4 = outside its domain) */

/* ------------------------------------------------- replay storage (K1 / K4) -- */
/* A column of a struct-of-arrays transition store: `row_bytes` bytes per transition. */
#define RLX_MAX_COLUMNS 8
typedef struct rlx_column {
    const void *src;      /* device base of the source array      */
    void *dst;            /* device base of the destination array */
    long long row_bytes;  /* bytes per row (same on both sides)   */
} rlx_column;

/* Row r (0 <= r < n) of every column is copied from source row
 *   src_idx ? src_idx[r] : (src_start + r) mod src_rows      to destination row
 *   dst_idx ? dst_idx[r] : (dst_start + r) mod dst_rows.
 * gather  = ExperienceReplay.sample's `[self.transitions[i] for i in idx]` + Batch collation
 *           (memories/non_episodic/experience_replay.py:80-90, core_types.py:488-649);
 * append  = ExperienceReplay.store + _enforce_max_length as a ring (:117-150).
 * status bit 1 is set for an out-of-range row (the reference's IndexError). */
int rlx_copy_columns(const rlx_column *columns_host, int ncols, const int *src_idx,
                     const int *dst_idx, long long src_start, long long dst_start,
                     long long src_rows, long long dst_rows, int n, int *status, void *stream);

/* Frame-dedup image replay (ObservationStackingFilter + LazyStack,
 * filters/observation/observation_stacking_filter.py:27-41,89-101).
 * ring: u8 [n_env][ring_frames][frame_bytes]; env_fpos/env_epoff: int32[n_env] describe every
 * env's current stacked state; t_fpos (int32) / t_epoff (u8): per stored transition. */
int rlx_imgreplay_reset(unsigned char *ring, int *env_fpos, int *env_epoff,
                        const unsigned char *first_frame, int n_env, int ring_frames,
                        int frame_bytes, void *stream);              /* first observation: stack = [f0]*stack (:90-91) */
int rlx_imgreplay_append(unsigned char *ring, int *env_fpos, int *env_epoff, int *t_fpos,
                         unsigned char *t_epoff, const unsigned char *next_frame,
                         const unsigned char *reset_frame, const unsigned char *done, int n_env,
                         int ring_frames, int frame_bytes, int stack, long long cursor,
                         long long capacity, int record, void *stream); /* deque.append (:93-94) for n_env envs */
int rlx_imgreplay_gather(const unsigned char *ring, const int *t_fpos,
                         const unsigned char *t_epoff, const int *env_fpos, const int *env_epoff,
                         const int *idx, int batch, int n_env, int ring_frames, int frame_bytes,
                         int stack, long long capacity, unsigned char *out_state,
                         unsigned char *out_next, int *status,
                         void *stream);                              /* LazyStack.__array__ (:37-41) for a batch */
/* rlx_imgreplay_gather of sampled rows (stack == 4) + rlx_copy_columns(columns, src_idx = idx, n = batch) of the batch's
 * small columns (actions, rewards, game_overs: core_types.py:488-649) as ONE launch: the columns are gathered by one
 * more workgroup of the same grid (row_bytes * batch <= 65536 per column; the columns' tables have `capacity` rows). */
int rlx_imgreplay_gather_columns(const unsigned char *ring, const int *t_fpos, const unsigned char *t_epoff,
                                 const int *idx, int batch, int n_env, int ring_frames, int frame_bytes, int stack,
                                 long long capacity, unsigned char *out_state, unsigned char *out_next,
                                 const rlx_column *columns_host, int ncols, int *status, void *stream);

/* ------------------------------------- returns / GAE / episode stats (K8 / K12) -- */
/* n_seq independent trajectories of seq_len steps, contiguous; game_overs cut episodes inside a
 * trajectory.  fp64 arithmetic like the reference; `values` are V(s_t) as predicted (fp32). */
int rlx_gae(const float *rewards, const float *values, const unsigned char *game_overs,
            const float *bootstrap_values, int n_seq, long long seq_len, double discount,
            double gae_lambda, double *advantages, float *value_targets,
            void *stream);   /* agents/actor_critic_agent.py:108-125 + clipped_ppo_agent.py:181-196 */
int rlx_discounted_returns(const float *rewards, const unsigned char *game_overs, int n_seq,
                           long long seq_len, double discount, double *returns64,
                           float *returns32, void *stream);          /* core_types.py:771-801 (n_step = -1) */
/* Signal statistics (utils.py:162-212, agent.py:548-552): table = fp64 [n_signals][5] = {count, sum, sum of
 * squares, max, min}; accumulate folds up to 8 device arrays (fp32 or fp64) into their signals' records in ONE
 * launch; the host reads the table when it logs an episode and resets it. */
typedef struct rlx_signal_source {
    const void *values;   /* device array */
    int n;                /* samples */
    int is_f64;           /* element type: 0 = float, 1 = double */
    int signal;           /* table row */
} rlx_signal_source;
int rlx_signals_reset(double *table, int n_signals, void *stream);
int rlx_signals_accumulate(const rlx_signal_source *sources_host, int n_sources, double *table, int n_signals,
                           void *stream);
int rlx_episode_nstep_returns(const float *rewards, double *out, double *compact_out, long long first_step,
                              int length, int env, int n_env, long long ring_steps, double discount, int n_step,
                              void *stream);   /* core_types.py:771-801 for any n_step, ONE completed episode stored
                                                  time-major (row = ((first_step + k) mod ring_steps) * n_env + env);
                                                  fp64, the reference's summation order (bit-identical).  out: a
                                                  column indexed like `rewards` (or NULL); compact_out: [length]
                                                  (or NULL) — the 'Discounted Return' signal's samples */
int rlx_standardize(const double *x, long long n, float *out32, double *out64, double *mean_std,
                    void *stream);                                   /* clipped_ppo_agent.py:201 (no epsilon) */
int rlx_episode_stats_init(double *ep_return, int *ep_len, int n_env, double *acc, void *stream);
int rlx_episode_stats_step(const float *reward, const unsigned char *game_over, double *ep_return,
                           int *ep_len, int n_env, double *acc, double *last_return,
                           int *last_len, void *stream);             /* agents/agent.py:558-601,509-556 totals */

/* Everything an agent does with one environment response (Agent.observe, agents/agent.py:905-973: reward filters,
 * episode totals, the transition store, the next current state) for up to 1024 vector-observation envs as ONE
 * launch: rlx_reward_filter -> rlx_episode_stats_step -> rlx_copy_columns (action, filtered reward, stored_game_over,
 * current state, next state -> replay row dst_rows[e]) -> rlx_select_rows (cur_state <- reset_obs where game_over).
 * mem_obs == NULL skips the store (evaluation).  Identical arithmetic to the four entry points it replaces. */
typedef struct rlx_observe_desc {
    const float *reward;                 /* [n_env] env rewards                                  */
    float *filtered_reward;              /* [n_env] out                                          */
    double reward_rescale;
    int has_clip;
    double clip_low, clip_high;
    const unsigned char *game_over;      /* [n_env] episode ends (statistics, next current state) */
    const unsigned char *stored_game_over; /* [n_env] what the replay records (TD3 clears time-limit ends) */
    double *ep_return; int *ep_len; double *acc; double *last_return; int *last_len;   /* rlx_episode_stats_step */
    const void *actions; long long action_row_bytes;
    void *cur_state; const void *next_obs; const void *reset_obs; long long obs_row_bytes;
    void *mem_action; float *mem_reward; unsigned char *mem_game_over; void *mem_obs; void *mem_next_obs;
    const int *dst_rows;                 /* [n_env] device: replay row of every env's transition  */
    long long mem_rows;
    int *status;                         /* bit 1: destination row out of range                   */
    int n_env;
} rlx_observe_desc;
int rlx_observe_step(const rlx_observe_desc *desc_host, void *stream);
/* The same for a lockstep rollout buffer (ClippedPPOAgent's acting step, agents/agent.py:905-973 + the episodic memory's
 * store): rlx_reward_filter -> rlx_episode_stats_step -> the step's action / filtered reward / game_over columns at rows
 * [row0, row0 + n_env) as ONE launch (frames go through rlx_imgreplay_append).  status bit 1: a row beyond mem_rows. */
int rlx_rollout_observe_step(const float *reward, float *filtered_reward, double reward_rescale, int has_clip,
                             double clip_low, double clip_high, const unsigned char *game_over, double *ep_return,
                             int *ep_len, double *acc, double *last_return, int *last_len, const void *actions,
                             long long action_row_bytes, void *mem_action, float *mem_reward,
                             unsigned char *mem_game_over, long long row0, long long mem_rows, int n_env, int *status,
                             void *stream);

/* --------------------------------------------------- agent targets (K7 / K10) -- */
/* td_targets holds Q_online(s,.) on entry (fp32 [batch, n_actions]); column actions[i] of row i is
 * replaced by r + (1-done)*discount*Q_target(s', argmax_a q_next_selector(s',a)); td_errors (fp64,
 * optional) receives |new_target - Q_online(s,a)|.  q_next_selector = NULL -> DQN (target net
 * selects, agents/dqn_agent.py:78-79), = Q_online(s',.) -> DDQN (agents/ddqn_agent.py:43). */
int rlx_dqn_targets(const float *q_next_target, const float *q_next_selector, float *td_targets,
                    const int *actions, const float *rewards, const unsigned char *game_overs,
                    double discount, int batch, int n_actions, double *td_errors, int *status,
                    void *stream);                                   /* agents/dqn_agent.py:92-103 */
/* The same targets, the QHead loss (MSE / Huber, importance weighted) and its gradient w.r.t.
 * Q_online in ONE launch: TD_targets equals Q_online except at the taken action, so loss and gradient
 * live at [b, actions[b]] only (dqn_agent.py:92-113, heads/q_head.py, head.py:172-181).
 * importance_weights: the fp64 weights of rlx_per_sample (rounded to fp32 like the TF placeholder) or
 * NULL.  td_errors / td_targets / loss_scalar are optional outputs. */
int rlx_dqn_head_loss(const float *q_online, long long ld_q, const float *q_next_target,
                      const float *q_next_selector, long long ld_next, const int *actions,
                      const float *rewards, const unsigned char *game_overs,
                      const double *importance_weights, double discount, int batch, int n_actions,
                      int huber, float grad_scale, float *dq, long long ld_dq, double *td_errors,
                      float *td_targets, long long ld_targets, float *loss_scalar, int *status,
                      void *stream);
/* DuelingQHead merge (architectures/tensorflow_components/heads/dueling_q_head.py:33-48):
 * q[b,a] = V[b] + (A[b,a] - mean_a A[b,:]); the backward entry point maps dL/dq to dL/dV (row sums)
 * and dL/dA (dq - row mean). */
int rlx_dueling_combine(const float *state_value, const float *action_advantage, int batch,
                        int n_actions, float *q, void *stream);
int rlx_dueling_combine_backward(const float *dq, int batch, int n_actions, float *dstate_value,
                                 float *daction_advantage, void *stream);
int rlx_ac_td_targets(const float *rewards, const unsigned char *game_overs, const float *q_next,
                      int q_stride, double discount, int use_non_zero_discount_for_terminal_states,
                      int has_clip, double clip_low, double clip_high, int batch,
                      float *td_targets, void *stream);              /* agents/ddpg_agent.py:156-164, td3_agent.py:171-180, soft_actor_critic_agent.py:265-266 */
/* One launch for min(q_next1, q_next2) (q_next2 NULL: q_next1), rlx_ac_td_targets and the mean-squared-error
 * loss + gradient of n_streams critic outputs q [n_streams][batch] against those targets (rlx_regression_loss with
 * loss_weight, grad_scale 1): loss[t] per stream, loss[n_streams] = their sum in stream order (the logged total
 * critic loss, td3_agent.py:176-180 / sac_q_head.py:91-95).  q_min_out may be NULL.  batch <= 1024. */
int rlx_ac_critic_losses(const float *q_next1, const float *q_next2, const float *rewards,
                         const unsigned char *game_overs, double discount,
                         int use_non_zero_discount_for_terminal_states, int has_clip, double clip_low,
                         double clip_high, const float *q, int n_streams, int batch, float loss_weight,
                         float *q_min_out, float *td_targets, float *dq, float *loss, void *stream);
/* The critic's merged inputs (concat(action, observation embedding), general_network.py:251,270-277) of the online
 * pass on (s, a) and the target pass on (s', a'): merged2 [2][batch][action_dim + obs_dim]; a' = next_actions with
 * the TD3 target-policy smoothing of rlx_td3_smooth_actions when noise != NULL.  merged_obs_only (may be NULL)
 * [batch][action_dim + obs_dim]: only its observation columns are written (the input of the action-gradient pass,
 * whose action columns the caller fills with the online actor's output). */
int rlx_ac_merge_inputs(const float *actions, const float *obs, const float *next_actions, const double *noise,
                        double noise_clipping, const float *action_low, const float *action_high,
                        const float *next_obs, int batch, int action_dim, int obs_dim, float *merged2,
                        float *merged_obs_only, void *stream);
int rlx_td3_smooth_actions(const float *next_actions, const double *noise, double noise_clipping,
                           const float *action_low, const float *action_high, int batch,
                           int action_dim, float *out, void *stream); /* agents/td3_agent.py:162-165 */
int rlx_sac_value_targets(const float *q_min, const float *sampled_logprob, int batch,
                          float *value_targets, void *stream);       /* agents/soft_actor_critic_agent.py:244 */

/* --------------------------------------- observation / reward filters (K2 / K3) -- */
int rlx_rgb_to_y_u8(const unsigned char *rgb, unsigned char *out, long long n_pixels,
                    double input_low, double input_high,
                    void *stream);   /* filters/observation/observation_rgb_to_y_filter.py:41-47 + observation_to_uint8_filter.py:51-60 */
int rlx_resize_bilinear_u8(const unsigned char *in, unsigned char *out, int n, int H, int W, int C,
                           int OH, int OW,
                           void *stream);   /* filters/observation/observation_rescale_to_size_filter.py:62-79 (skimage resize, order 1) */
int rlx_max_over_frames_u8(const unsigned char *frames, unsigned char *out, int n_env, int n_frames,
                           long long frame_bytes,
                           void *stream);   /* environments/gym_environment.py:148-175: np.max over the newest frames of a frame-skip step */
int rlx_running_stats_push(const void *samples, int samples_are_f64, long long n, int dim,
                           double *sum, double *sum_squares, double *count, double *mean,
                           double *std, double epsilon,
                           void *stream);                            /* utilities/shared_running_stats.py:130-140 */
/* Data-parallel running statistics (the reference shares them between workers through Redis pub/sub,
 * utilities/shared_running_stats.py:46-67): `delta` = [sum(dim) | sum_squares(dim) | count(1)] summed
 * over all ranks (one all-reduce of 2*dim+1 doubles) is added to the running totals and mean / std
 * are recomputed with the formulas of :137-140. */
int rlx_running_stats_merge(const double *delta, int dim, double *sum, double *sum_squares,
                            double *count, double *mean, double *std, double epsilon, void *stream);
int rlx_running_stats_normalize(const void *x, int x_is_f64, long long n, int dim,
                                const double *mean, const double *std, double clip_low,
                                double clip_high, float *out32, double *out64,
                                void *stream);                       /* utilities/shared_running_stats.py:162-164 */
int rlx_reward_filter(const float *rewards, float *out, long long n, double rescale_factor,
                      int has_clip, double clipping_low, double clipping_high,
                      void *stream);   /* filters/reward/reward_rescale_filter.py:37-39, reward_clipping_filter.py:41-49 */

/* ------------------------------------------- dense / conv layers (fp32 MFMA) -- */
enum { RLX_ACT_NONE = 0, RLX_ACT_RELU = 1, RLX_ACT_TANH = 2 };

/* C[M,N] (+)= epilogue(A[M,K] * B[K,N]) for `batch` independent problems (strided).
 * A(m,k) = A[off_m + off_k] with off_* = index*stride or a lookup in an int32 table (implicit
 * im2col: rlx_conv_tables); one of the two indices must be contiguous in groups of 4.
 * epilogue: v = act(acc + bias[n]); if deriv_aux: v *= act'(deriv_aux[m][n]) (derivative written in
 * terms of the activation OUTPUT); C = accumulate ? C + v : v.
 * Replaces tf.layers.dense / tf.layers.conv2d and their gradients
 * (architectures/tensorflow_components/layers.py:108-121,168-185). */
struct rlx_small_dense_problem;   /* below: the narrow layers (heads) */
typedef struct rlx_gemm_desc {
    const void *A;            /* float (or uint8 when a_is_u8) */
    const int *a_row_tab;     /* offset per m (elements) or NULL -> m * a_row_stride */
    const int *a_k_tab;       /* offset per k or NULL -> k * a_k_stride            */
    const float *B;
    float *C;
    const float *bias;        /* [N] or NULL */
    const float *deriv_aux;   /* [M, aux_ld] or NULL */
    float *workspace;         /* split-K partials, or NULL to forbid splitting */
    float *colsum_out;        /* optional [N]: sum_k B[k][n] (the bias gradient rides on the dW GEMM) */
    long long a_row_stride, a_k_stride, a_batch_stride;
    long long b_k_stride, b_n_stride, b_batch_stride;
    long long ldc, c_batch_stride, bias_batch_stride;
    long long aux_ld, aux_batch_stride;
    long long workspace_floats;
    long long colsum_batch_stride;
    int M, N, K, batch;
    int a_is_u8;              /* A holds bytes; value = byte / a_div (embedders/embedder.py:107-108) */
    int a_vec_along_k;        /* with tables: 1 = 4 consecutive k are contiguous, 0 = 4 consecutive m */
    int a_tab_vec_ok;         /* with tables: groups of 4 are contiguous AND 16-byte (4-byte for u8) aligned */
    int activation, deriv_kind, accumulate;
    float a_div;
    /* two-level batch addressing of A, B and bias (0 = off): batch index b = bo * batch_inner + bi,
     * operand offset = bo * *_batch_stride2 + bi * *_batch_stride.  Lets the online and the target
     * copy of a multi-stream layer run as one launch (network_wrapper.py:188-213 parallel_prediction). */
    int batch_inner;
    long long a_batch_stride2, b_batch_stride2, bias_batch_stride2;
    /* n_fold > 0 (with batch == 1): N counts the columns of N / n_fold towers that SHARE the A operand;
     * column n uses tower n / n_fold: B, C, bias and colsum_out are addressed at
     * tower * *_batch_stride + (n % n_fold).  The separate value / policy networks of Clipped PPO
     * (clipped_ppo_agent.py:50) read the same observation: their first layer gathers it once. */
    int n_fold;
    /* Narrow layers that read ROWS of this product's output (the value / policy heads on the last dense layer of their
     * tower, heads/v_head.py:43-48, ppo_head.py:100-116): row_heads[i].x must be C + t * c_batch_stride for a batch entry
     * t (one head per entry at most), K = N of this product, M = M, towers = 1, N <= 16.  When the product is split over
     * K, the reduction pass that finishes a row (bias, activation) computes its head outputs right there — the row is in
     * LDS — instead of a separate launch; otherwise rlx_gemm runs rlx_dense_small_forward_multi(row_heads) behind the
     * product.  Either way y of every head is complete in stream order, with identical values.  NULL / 0: none. */
    const struct rlx_small_dense_problem *row_heads;
    int n_row_heads;
    /* 0: the library's rule (rlx_gemm_tuning's kw_min_tiles, default 192).  > 0: for THIS product, the fewest 32 x 64 / 32 x 32
     * tiles for which K is split over the waves of a workgroup (no partials, no reduce launch) rather than over workgroups.
     * A dense layer's input gradient at batch 32 (M = 32, N = 3136, K = 512: 98 tiles of 32 x 32) sits in the chain of the
     * update in front of the convolutions' backward pass: with 96 it is one launch of 13.7 us instead of 12.4 + a 4.7 us reduce
     * launch (coach_amd/nn/graph.py Dense.backward; profiles/r06_ab_c3_kw_min_tiles.txt). */
    int kw_min_tiles;
} rlx_gemm_desc;

int rlx_gemm(const rlx_gemm_desc *desc_host, void *stream);
int rlx_gemm_workspace_floats(int M, int N, int K, int batch, long long *floats_host);
/* Path selection of rlx_gemm (process-wide; defaults 192 / 192 / -1): a problem whose 64 x 64 tiling has fewer than
 * kw_below_tiles tiles runs on 32 x 64 / 32 x 32 tiles with the K slab split over the waves of a workgroup, if that
 * tiling has at least kw_min_tiles tiles.  xcd_mode: how the tiled kernels hand the tile order to the 8 XCDs (one L2
 * each): 0 = every 8th tile (the hardware's order), G = 2^k > 0: groups of G consecutive tiles round-robin, -1 = one
 * contiguous share per XCD — so that tiles sharing operand rows share an L2.
 * An explicit knob for same-process A/B measurements (tools/ab_c2.py); results are identical either way. */
int rlx_gemm_tuning(int kw_below_tiles, int kw_min_tiles, int xcd_mode);
/* the current values (NULL: not wanted) — what a caller that mirrors rlx_gemm's tile rule (coach_amd/nn/graph.py
 * _kw2_tiling: where rlx_conv23_forward is bit-identical to the tiled launches) must read instead of assuming the defaults */
int rlx_gemm_tuning_get(int *kw_below_tiles, int *kw_min_tiles, int *xcd_mode);
/* What rlx_gemm WOULD launch for this descriptor, without launching: out8 = {vector-load tiled kernel (else: thin / generic
 * paths, the other fields are then meaningless), tile rows, tile columns, wave groups per K slab, K chunks over workgroups
 * (> 1: partials + a reduce launch), chunk length, operands through the LDS-DMA ring (0: staged by registers — the order
 * of the fp32 sum inside a slab differs, see rlx_gemm_pipeline), thin kernel}.  For callers that replace a launch by a
 * fused one ONLY where the sums stay bit-identical (rlx_conv123_forward). */
int rlx_gemm_describe(const rlx_gemm_desc *desc, int *out8);
/* Main loop of the fast tiled kernels: 1 (default) = operand slabs go global -> LDS by DMA into a ring of two buffers,
 * one barrier per slab, three workgroups per CU (csrc/gemm.hip gemm_dma_body); 0 = the register-staged two-set pipeline
 * of rounds 1-3 (128 x 32 tiles always use it); 2 = the ring for uint8 operands too (4-byte requests; measured equal to 1).  Same products, same tiles; the order of the fp32 sum inside a 32-deep slab differs between the two
 * (both deterministic).  Process-wide, read at launch (or capture) time: exists for same-process A/B measurements. */
int rlx_gemm_pipeline(int lds_dma_ring);
/* 64 x 64 (or 64 x 32) per wave (2 x 2 / 2 x 1 accumulator tiles; 128 x 128 or 128 x 64 per workgroup) for products of at least 256 such tiles
 * without a K split: 1 (default: 128 x 128 for N >= 128) / 0 / 2 (also 128 x 64 for N <= 64, measured neutral on the
 * convolution layers of the whole-dataset passes).  Bit-identical results either way
 * (every element stays one chain over K); process-wide, for same-process A/Bs and tests. */
int rlx_gemm_big_tiles(int on);
/* The most K splits rlx_gemm cuts one product into (default 64; the workspace given with the descriptor bounds it as
 * well).  Process-wide, read at launch / capture time: a knob for same-process A/B measurements. */
int rlx_gemm_split_cap(int max_splits);

/* Input gradient of a VALID-padding NHWC convolution (tf.gradients of tf.layers.conv2d,
 * architectures/tensorflow_components/layers.py:108-121, architecture.py:187-220) as ONE product that gathers dY
 * directly — no column matrix in memory, no col2im pass:
 *   dx[t][b, iy, ix, c] = act'(x_out[t][b, iy, ix, c]) * sum_{ky, kx, co : iy = s*oy + ky, ix = s*ox + kx}
 *                         dy[t][b, oy, ox, co] * weights[t][ky, kx, c, co]
 * for `towers` independent networks (tower strides in elements).  The s*s phases (iy % s, ix % s) of the input are the
 * row blocks of one GEMM with M = s*s*rows, N = C, K = (KH/s)*(KW/s)*Co whose A operand is a windowed gather of dy
 * (fp32 MFMA, exact products).  Needs KH % s == 0, KW % s == 0, C % 4 == 0, Co % 4 == 0 and K <= 1024.
 * `tables`: int32 device buffer of rlx_conv_input_grad_tables_ints(...) entries filled once per geometry by
 * rlx_conv_input_grad_tables.  deriv_kind != 0 multiplies by the activation derivative of the producing layer's OUTPUT
 * x_out (same layout as dx). */
int rlx_conv_input_grad_tables_ints(int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                                    long long *ints_host);
int rlx_conv_input_grad_tables(int *tables, int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                               void *stream);
int rlx_conv_input_grad(const float *dy, const float *weights, float *dx, const float *x_out, int deriv_kind,
                        const int *tables, int batch, int H, int W, int C, int KH, int KW, int stride, int Co,
                        int towers, long long dy_tower_stride, long long w_tower_stride, long long dx_tower_stride,
                        void *stream);
/* A layer's weight gradient (dW = X^T dY) and input gradient (dX = dY W^T) — two independent products of the same
 * dY — as ONE launch when both take the tiled kernel or both the thin kernel (two launches otherwise).
 * Each descriptor is exactly what rlx_gemm would get; when both split K they need disjoint workspaces. */
int rlx_gemm_pair(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad, void *stream);

/* Deferred split-K reduction.  A weight gradient dW = X^T dY has a tiny output and a long reduction (the batch), so
 * its K is split over workgroups into partial sums [batch][split][M][N] in the workspace, and summing them is a second
 * launch.  Nobody reads a weight gradient before the optimiser does, so the backward pass of a network can leave
 * the partials of ALL its layers where they are and sum them in ONE launch at its end: rlx_gemm_defer /
 * rlx_gemm_pair_defer run the product(s) and, if K was split, describe the outstanding reduction in *job (job->splits
 * > 1) instead of launching it; rlx_splitk_reduce_jobs sums up to RLX_MAX_SPLITK_JOBS of them (fixed summation order:
 * reproducible).  The workspace of a deferred product must stay untouched until its job has run.  Only plain
 * products qualify (no bias / activation / derivative / accumulate epilogue — a weight gradient has none). */
#define RLX_MAX_SPLITK_JOBS 8
typedef struct rlx_splitk_job {
    const float *partials;        /* [batch][splits][M][N] */
    const float *colsum_partials; /* [batch][splits][N] or NULL */
    float *C;
    float *colsum_out;
    long long ldc, c_batch_stride, colsum_batch_stride;
    int M, N, batch, splits;      /* splits <= 1: nothing outstanding */
    int n_fold;
} rlx_splitk_job;
int rlx_gemm_defer(const rlx_gemm_desc *desc_host, rlx_splitk_job *job_host, void *stream);
int rlx_gemm_pair_defer(const rlx_gemm_desc *weight_grad, const rlx_gemm_desc *input_grad,
                        rlx_splitk_job *weight_grad_job_host, void *stream);
int rlx_splitk_reduce_jobs(const rlx_splitk_job *jobs_host, int n_jobs, void *stream);
/* Up to three weight-gradient products (the convolution layers' dW = cols^T dz of one backward pass,
 * architectures/tensorflow_components/architecture.py:187-220 tf.gradients) as ONE launch, each with the tiling and K split
 * rlx_gemm would give it alone (bit-identical sums) and its split-K reduction deferred into jobs[i] as rlx_gemm_defer
 * does (jobs[i].splits = 0: nothing to reduce).  Descriptors that do not qualify for the one-launch form (anything but
 * 64 x 64 tiles of an im2col-gathered A^T) are launched one by one. */
int rlx_gemm_multi_defer(const rlx_gemm_desc *descs, int n, rlx_splitk_job *jobs, void *stream);
/* Diagnostics (tools/gemm_timeline.py): while a device buffer of `capacity_u64` 64-bit words is registered, every
 * tiled-kernel launch of rlx_gemm records, per workgroup, four wall-clock ticks (10 ns: entry, first slab staged,
 * main loop done, exit) in its own region of the buffer; rlx_gemm_debug_calls lists the regions as rows of
 * {M, N, K, batch, splits, grid.x, grid.y, grid.z, offset}.  buffer == NULL switches the stamps off (the default). */
int rlx_gemm_debug_stamps(void *buffer, long long capacity_u64);
int rlx_gemm_debug_calls(long long *out_host, int max_calls, int *n_calls_host);
int rlx_colsum(const float *x, int M, int N, long long ld, float *out, int accumulate,
               float *workspace, long long workspace_floats, void *stream);  /* bias gradients */
int rlx_act_backward(float *dy, const float *y, long long n, int kind, void *stream); /* dy *= act'(y) */
int rlx_conv_tables(int *rowbase, int *koff, int batch, int H, int W, int C, int KH, int KW,
                    int stride, void *stream);      /* VALID-padding NHWC im2col offsets */
int rlx_col2im(const float *dcol, float *dx, const float *x_out, int deriv_kind, int batch, int H,
               int W, int C, int KH, int KW, int stride, void *stream); /* conv input gradient */

/* Batch normalisation after a dense layer — the `use_batchnorm=True` DDPG networks (agents/ddpg_agent.py:37-60;
 * architectures/tensorflow_components/layers.py:26-55 -> tf.layers.batch_normalization: momentum 0.99, epsilon 1e-3).
 * x, y, dy, dx: [batch][channels] fp32, rows contiguous; gamma, beta, statistics: [channels].
 *   rlx_bn_forward   training != 0: batch mean / population variance (written to save_mean / save_var), else the moving
 *                    statistics;  y = act(x * inv + (beta - mean * inv)),  inv = rsqrt(var + eps) * gamma
 *   rlx_bn_backward  dy = d loss / d y: multiplies by act'(y) (activation 0: dy is already d loss / d u), writes dx and,
 *                    unless NULL, dgamma / dbeta (gradients_wrt_inputs passes need no weight gradients)
 *   rlx_bn_update_moving  m -= (m - batch) * (1 - momentum): the UPDATE_OPS that run before apply_gradients
 *                    (architecture.py:273-277) on the batch fed with it (ddpg_agent.py:178-193) */
int rlx_bn_forward(const float *x, const float *gamma, const float *beta, const float *moving_mean,
                   const float *moving_var, int batch, int channels, double epsilon, int training, int activation,
                   float *y, float *save_mean, float *save_var, void *stream);
int rlx_bn_backward(const float *dy, const float *y, const float *x, const float *gamma, const float *save_mean,
                    const float *save_var, int batch, int channels, double epsilon, int activation, float *dx,
                    float *dgamma, float *dbeta, void *stream);
int rlx_bn_update_moving(float *moving_mean, float *moving_var, const float *batch_mean, const float *batch_var,
                         int channels, double momentum, void *stream);

/* Fused small-MLP DQN update: DQNAgent.learn_from_batch (agents/dqn_agent.py:81-113) for a vector-observation
 * Q network  obs -> Dense(h1, relu) -> Dense(h2, relu) -> Dense(n_actions)  in ONE launch: online / target (/ Double
 * DQN selector) forward passes, TD targets, |TD errors|, MSE or Huber loss with importance weights
 * (heads/q_head.py, head.py:143-186), backward, TF1 Adam (general_network.py:390-394) and tf.global_norm.
 * h2 / 32 workgroups each own a 32-column slice of the wide layer; two in-kernel exchanges (csrc/mlp_fused.hip).
 * All pointers are device pointers; weights / target_weights / adam_m / adam_v are the network's flat buffers and
 * off_* the element offsets of the six tensors ([obs,h1] [h1] [h1,h2] [h2] [h2,A] [A], row-major) inside them.
 * sync_words: 3 zero-initialised uint32 owned by this network (the kernel leaves them zero).  status bits: 1 = action
 * out of range, 8 = an in-kernel barrier timed out (results invalid). */
typedef struct rlx_mlp_dqn_desc {
    float *weights; const float *target_weights; float *adam_m; float *adam_v; float *adam_state;
    const float *states; const float *next_states; const int *actions; const float *rewards;
    const unsigned char *game_overs; const double *importance_weights;   /* fp64 [batch] or NULL */
    float *workspace; unsigned int *sync_words;
    float *loss_out; float *norm_out; double *td_errors; int *status;
    long long workspace_floats;
    long long off_w1, off_b1, off_w2, off_b2, off_w3, off_b3;
    double discount;
    int batch, obs_dim, h1, h2, n_actions;
    int huber, double_dqn;
    float learning_rate, beta1, beta2, epsilon, grad_scale;
} rlx_mlp_dqn_desc;
int rlx_mlp_dqn_supported(int batch, int obs_dim, int h1, int h2, int n_actions);   /* 1 / 0 (a value, not a status) */
int rlx_mlp_dqn_workspace_floats(int h1, int h2, int n_actions, long long *floats_host);
int rlx_mlp_dqn_update(const rlx_mlp_dqn_desc *desc_host, void *stream);
/* Acting with the same network: q_out [n_env][n_actions] = Q(states) (may be NULL) and, when actions != NULL, the
 * epsilon-greedy choice of rlx_egreedy on those values — one launch of one workgroup for n_env <= 8 envs
 * (agents/dqn_agent.py choose_action + exploration_policies/e_greedy.py:84-101).  Weights at the element offsets
 * off_* of the flat parameter buffer, [in][out] storage. */
int rlx_mlp_q_act_supported(int n_env, int obs_dim, int h1, int h2, int n_actions);   /* 1 / 0 (a value) */
int rlx_mlp_q_act(const float *weights, long long off_w1, long long off_b1, long long off_w2, long long off_b2,
                  long long off_w3, long long off_b3, const float *states, int n_env, int obs_dim, int h1, int h2,
                  int n_actions, const double *explore_uniforms, const int *random_actions,
                  const double *tie_break_uniforms, double epsilon, float *q_out, int *actions, void *stream);

/* Narrow dense layers (1 <= N <= 16 outputs: value / policy / Q heads) as coalesced fp32 FMA
 * kernels instead of MFMA tiles (heads/v_head.py:43-48, ppo_head.py:100-116, q_head.py,
 * ddpg_actor_head.py:48-56, td3_v_head.py:40-60).  Tower t of every operand sits at
 * base + t * tower_stride (x_tower_stride = 0: one input shared by all towers).
 * forward : y = act(x W + b)
 * backward: dz = dy * act'(y) (y may be NULL when activation == NONE); dw = x^T dz; db = 1^T dz;
 *           dx = dz W^T, multiplied by lower_activation'(x) when lower_activation != NONE (dx is then
 *           the lower layer's dz).  dw/db or dx may be NULL. */
int rlx_dense_small_forward(const float *x, long long x_tower_stride, const float *w,
                            long long w_tower_stride, const float *bias, long long bias_tower_stride,
                            float *y, long long y_tower_stride, int towers, int M, int K, int N,
                            int activation, void *stream);
int rlx_dense_small_backward(const float *x, long long x_tower_stride, const float *w,
                             long long w_tower_stride, const float *dy, long long dy_tower_stride,
                             const float *y, long long y_tower_stride, float *dw,
                             long long dw_tower_stride, float *db, long long db_tower_stride, float *dx,
                             long long dx_tower_stride, int towers, int M, int K, int N, int activation,
                             int lower_activation, void *stream);

/* Up to 4 narrow layers of the same depth in ONE launch (e.g. the value and the policy head of
 * Clipped PPO, clipped_ppo_agent.py:41-58: separate towers, widths 1 and A).  Field meaning as the
 * single-layer calls; `y` is the forward output, and in backward the layer output needed for the
 * activation derivative (may be NULL when activation == NONE). */
typedef struct rlx_small_dense_problem {
    const float *x; long long x_tower_stride;
    const float *w; long long w_tower_stride;
    const float *bias; long long bias_tower_stride;
    float *y; long long y_tower_stride;
    const float *dy; long long dy_tower_stride;
    float *dw; long long dw_tower_stride;
    float *db; long long db_tower_stride;
    float *dx; long long dx_tower_stride;
    int towers, M, K, N, activation, lower_activation;
} rlx_small_dense_problem;
/* rlx_dqn_head_loss + the Q head's backward pass (rlx_dense_small_backward of q_head: dW, db, dx with the lower layer's
 * activation derivative; tf.gradients through the head, architecture.py:312-385) as ONE launch.  q_head: x, w, dy (receives
 * dQ), dw / db / dx as for rlx_dense_small_backward_multi, one linear tower with M = batch <= 256, N = n_actions <= 16.
 * Values: those of the two launches bit for bit (the same row arithmetic, the same reduction trees). */
int rlx_dqn_head_loss_backward(const rlx_small_dense_problem *q_head, const float *q_online, long long ld_q,
                               const float *q_next_target, const float *q_next_selector, long long ld_next,
                               const int *actions, const float *rewards, const unsigned char *game_overs,
                               const double *importance_weights, double discount, int batch, int n_actions, int huber,
                               float grad_scale, double *td_errors, float *loss_scalar, int *status, void *stream);
int rlx_dense_small_forward_multi(const rlx_small_dense_problem *problems_host, int n_problems, void *stream);
int rlx_dense_small_backward_multi(const rlx_small_dense_problem *problems_host, int n_problems, void *stream);
/* Discrete Clipped-PPO: the losses of both heads (heads/ppo_head.py:52-116, v_head.py:43-52, head.py:143-186) AND the
 * heads' backward pass (their share of accumulate_gradients, architecture.py:312-385) in ONE launch — what
 * rlx_ppo_discrete_value_losses followed by rlx_dense_small_backward_multi({value_head, policy_head}) computes, bit for bit
 * (every workgroup of the backward recomputes the batch's loss-gradient rows instead of reading them behind a launch
 * boundary).  Heads: one linear tower each over the `batch` <= 256 rows, value N = 1, policy N = n_actions <= 16; their .dy
 * (optional) receives dV / dlogits, .dw / .db / .dx as in the backward call.  scalars[5] = {surrogate loss, mean entropy,
 * mean KL(old||new), policy head total, value loss}; clip_scale: device scalar or NULL (see rlx_ppo_discrete_loss). */
int rlx_ppo_heads_loss_backward(const rlx_small_dense_problem *value_head, const rlx_small_dense_problem *policy_head,
                                const float *values, const float *value_targets, const float *logits,
                                const int *actions, const float *advantages, const float *old_probs, long long ld_old,
                                int batch, float clip_epsilon, const float *clip_scale, float beta_entropy,
                                float grad_scale, float *scalars, float *likelihood_ratio,
                                float *clipped_likelihood_ratio, int *status, void *stream);
/* The same update with the heads' work split by what it depends on (round 6).  Everything of rlx_ppo_heads_loss_backward that
 * is LOCAL TO A ROW — the head's forward on the finished row, the row's loss terms (heads/ppo_head.py:52-116, v_head.py:43-52,
 * head.py:143-186), the gradient at the head's outputs and dz = act'(h) (dy W_head^T), the gradient at the last dense layer's
 * pre-activation output (architecture.py:312-385) — runs in the workgroup that finishes that row of the dense layer
 * (layers.py:168-185: its K-split reduction, one workgroup per row and tower): rlx_ppo_fc_rows = rlx_gemm(fc) with
 * row_heads = {value_head, policy_head} + the row-local part.  What needs ALL rows and is read by nobody before the
 * optimizer step — dW / db of both heads, scalars[5] — is rlx_ppo_heads_tail, a launch of its own, or
 * rlx_splitk_reduce_jobs_ppo_tail: extra workgroups of the backward pass's deferred-reduction launch (rlx_splitk_reduce_jobs).
 * Two launches of the update's chain become none.  Values: those of rlx_gemm(row_heads) + rlx_ppo_heads_loss_backward bit for
 * bit (the same per-row arithmetic, the same chains and reduction trees).
 * value_head / policy_head as for rlx_ppo_heads_loss_backward, all of x, w, y, dy, dw, dx required: x = the tower's rows of
 * fc->C, y receives V / the logits, dy dV / dlogits, dx dz of the tower.  row_terms: [batch][4] floats of scratch that carries
 * {value loss term, surrogate, entropy, KL} of each row from the rows launch to the tail.
 * rlx_ppo_fc_rows_supported(fc, n_actions) -> 1 where rlx_gemm would split fc (two towers, <= 256 rows, N % 4 == 0,
 * N <= 1024, no row_heads of its own) into more than 16 K chunks, i.e. where the row-finishing reduction runs. */
typedef struct rlx_ppo_rows_desc {
    rlx_small_dense_problem value_head, policy_head;
    const float *value_targets;
    const int *actions;
    const float *advantages;
    const float *old_probs; long long ld_old;
    const float *clip_scale;               /* device scalar or NULL (see rlx_ppo_discrete_loss) */
    float clip_epsilon, beta_entropy, grad_scale;
    int batch;
    float *row_terms;
    float *scalars;                        /* [5] as in rlx_ppo_heads_loss_backward */
    float *likelihood_ratio, *clipped_likelihood_ratio;   /* [batch] or NULL */
    int *status;
} rlx_ppo_rows_desc;
int rlx_ppo_fc_rows_supported(const rlx_gemm_desc *fc, int n_actions);
int rlx_ppo_fc_rows(const rlx_gemm_desc *fc, const rlx_ppo_rows_desc *rows, void *stream);
int rlx_ppo_heads_tail(const rlx_ppo_rows_desc *rows, void *stream);
int rlx_splitk_reduce_jobs_ppo_tail(const rlx_splitk_job *jobs_host, int n_jobs, const rlx_ppo_rows_desc *rows, void *stream);
/* rlx_splitk_reduce_jobs + rlx_per_update(idx, td_errors, n <= 64) as ONE launch: PrioritizedExperienceReplay.
 * update_priorities (memories/non_episodic/prioritized_experience_replay.py:203-217; called right behind learn_from_batch,
 * agents/dqn_agent.py:106-109) is one workgroup, dispatched first, of the backward pass's deferred-reduction launch — the TD
 * errors exist since the head's loss, and nothing reads the trees before the next sample().  Trees bit-identical to
 * rlx_per_update's.  No deferred job with splits > 1: rlx_per_update as a launch of its own. */
int rlx_splitk_reduce_jobs_per_update(const rlx_splitk_job *jobs_host, int n_jobs, double *sum_tree, double *min_tree,
                                      double *max_tree, int capacity, const int *idx, const double *td_errors, int n,
                                      double alpha, double epsilon, double *max_priority, int *status, void *stream);

/* -------------------------------------------------------- head losses (K9) -- */
/* loss = mean_b(loss_weight * w_b * sum_j l(target, out)); kind 0 = MSE, 1 = Huber(delta 1).
 * grad (optional) = grad_scale * d loss / d out.  heads/head.py:143-186, q_head.py, v_head.py:43-52 */
int rlx_regression_loss(const float *out, long long ld_out, const float *target,
                        long long ld_target, const float *importance_weights, int batch, int dim,
                        int kind, float loss_weight, float grad_scale, float *grad,
                        long long ld_grad, float *loss_scalar, void *stream);
int rlx_softmax(const float *logits, long long ld, int batch, int n, float *probs,
                long long ld_out, void *stream);                    /* heads/ppo_head.py:108 */
/* scalars[4] = {surrogate loss, mean entropy, mean KL(old||new), total head loss}.
 * The clip range is clip_epsilon (clip_likelihood_ratio_using_epsilon) times the clip_param_rescaler placeholder
 * (heads/ppo_head.py:52-116, clipped_ppo_agent.py:266-268): pass the rescaler either folded into clip_epsilon
 * (clip_scale NULL) or as a DEVICE scalar clip_scale[0] — a captured hipGraph then serves a decaying
 * clipping_decay_schedule (presets/Mujoco_ClippedPPO.py) without being re-captured for every new value. */
int rlx_ppo_discrete_loss(const float *logits, long long ld, const int *actions,
                          const float *advantages, const float *old_probs, long long ld_old,
                          int batch, int n_actions, float clip_epsilon, float beta_entropy,
                          float grad_scale, float *dlogits, long long ld_grad, float *scalars,
                          float *likelihood_ratio, float *clipped_likelihood_ratio, int *status,
                          const float *clip_scale, void *stream);

/* rlx_ppo_discrete_loss and the VHead MSE loss (rlx_regression_loss, dim 1, weight 1) of the same
 * minibatch in ONE launch: values / value_targets / dvalues are [batch]. */
int rlx_ppo_discrete_value_losses(const float *logits, long long ld, const int *actions,
                                  const float *advantages, const float *old_probs, long long ld_old,
                                  int batch, int n_actions, float clip_epsilon, float beta_entropy,
                                  float grad_scale, float *dlogits, long long ld_grad, float *scalars,
                                  float *likelihood_ratio, float *clipped_likelihood_ratio, int *status,
                                  const float *values, const float *value_targets, float *dvalues,
                                  float *value_loss_scalar, const float *clip_scale, void *stream);

/* Continuous policy (heads/ppo_head.py:118-144): MultivariateNormalDiag(mean, exp(log_std) + eps),
 * log_std one state-independent vector [action_dim]; old_std is the old network's policy_std output.
 * dlog_std[action_dim] receives the batch-summed gradient.  scalars as rlx_ppo_discrete_loss. */
int rlx_ppo_continuous_loss(const float *mean, long long ld, const float *log_std, const float *actions,
                            const float *advantages, const float *old_mean, const float *old_std,
                            long long ld_old, int batch, int action_dim, float clip_epsilon,
                            float beta_entropy, float grad_scale, float *dmean, long long ld_grad,
                            float *dlog_std, float *scalars, float *likelihood_ratio,
                            float *clipped_likelihood_ratio, const float *clip_scale, void *stream);

/* ------------------------------------------ optimiser / target mixing (K11) -- */
/* state = {beta1_power, beta2_power} (2 device floats).  tf.train.AdamOptimizer as built in
 * architectures/tensorflow_components/general_network.py:390-394. */
int rlx_adam_init(float *m, float *v, long long n, float *state, float beta1, float beta2,
                  void *stream);
int rlx_adam_tf1(float *weights, const float *grads, float *m, float *v, long long n,
                 float learning_rate, float beta1, float beta2, float epsilon, float *state,
                 float grad_scale, void *stream);
/* rlx_adam_tf1 that also returns tf.global_norm of the (unscaled) gradients it consumed: the
 * per-workgroup sums of squares ride on the Adam pass (no second read of the gradients) and the
 * one-thread finish kernel that advances the beta powers also takes the square root and, when
 * n_acc > 0, adds acc_src[0..n_acc) (loss terms / the norm just written) onto the running sums
 * acc_dst (the per-epoch signal means of agents/agent.py:743-767, utils.py:162-212). */
int rlx_adam_tf1_norm(float *weights, const float *grads, float *m, float *v, long long n,
                      float learning_rate, float beta1, float beta2, float epsilon, float *state,
                      float grad_scale, float *norm_out, float *workspace, long long workspace_floats,
                      const float *acc_src, float *acc_dst, int n_acc, void *stream);
/* rlx_adam_tf1 / rlx_adam_tf1_norm with (target != NULL) the soft target update
 * target = rate * w_new + (1 - rate) * target (rlx_mix_weights' arithmetic) in the same elementwise pass.  Without
 * norm_out it is ONE launch: the last workgroup to finish (device ticket, no fence) advances the beta powers.  With
 * norm_out it is one launch as well when a workgroup's share fits its registers (up to 4.2 M parameters; larger buffers
 * take the two-launch form): the sums of squares are complete before the first store and are published without a
 * release fence (agent-scope atomics), and the workgroup that draws the last ticket does the finish — same arithmetic,
 * same order, bit-identical to the two-launch form.
 * ticket: RLX_ADAM_TICKET_WORDS zero-initialised 32-bit device words owned by this optimiser (a two-level last-arriver
 * count: 32 group words + 1, each on its own 128-byte line); the kernel re-arms them. */
#define RLX_ADAM_TICKET_WORDS 1056
int rlx_adam_tf1_step(float *weights, const float *grads, float *m, float *v, long long n, float learning_rate,
                      float beta1, float beta2, float epsilon, float *state, float grad_scale, float *norm_out,
                      float *workspace, long long workspace_floats, const float *acc_src, float *acc_dst, int n_acc,
                      float *target, double mix_rate, unsigned int *ticket, void *stream);
/* How rlx_adam_tf1_step finishes the gradient norm: 2 (default) inside the Adam launch, the workgroup's share held in
 * registers (falls back to 0 beyond 4.2 M parameters); 1: inside the grid-stride Adam launch (any size; measured equal to
 * 0); 0: the separate finish launch.  Process-wide, read at launch / capture time: for same-process A/Bs and tests. */
int rlx_adam_norm_in_kernel(int on);
int rlx_mix_weights(float *target, const float *online, long long n, double rate,
                    void *stream);   /* architectures/tensorflow_components/architecture.py:598-607 */
int rlx_global_norm(const float *x, long long n, float *norm_out, float *workspace,
                    long long workspace_floats, void *stream);      /* tf.global_norm, architecture.py:194 */

/* tf.clip_by_global_norm (architectures/tensorflow_components/architecture.py:196-200, clip method
 * ClipByGlobalNorm): grads *= clip_norm * min(1 / global_norm, 1 / clip_norm), in place; global_norm is
 * the device scalar rlx_global_norm wrote. */
int rlx_clip_by_global_norm(float *grads, long long n, const float *global_norm, float clip_norm,
                            void *stream);

/* ------------------------------------ continuous-control agents (DDPG / TD3 / SAC) -- */
/* dst[r][c] = scale * src[r][c]: concat / slice / negate / DDPGActorHead output_scale
 * (general_network.py:270-277, heads/ddpg_actor_head.py:48-56). */
int rlx_copy_2d(const float *src, long long src_ld, float *dst, long long dst_ld, int rows, int cols,
                float scale, void *stream);
int rlx_exp_rows(const float *log_std, float *out, int batch, int action_dim,
                 void *stream);                /* policy_std = tile(exp(policy_logstd)), heads/ppo_head.py:139 */
int rlx_axpby(float *out, float a, const float *x, float b, const float *y, long long n,
              void *stream);                   /* out = a*x + b*y (y may be NULL); heads/sac_q_head.py:63-67 */
/* out_min = min(q1,q2); grad_i = grad_scale * d sum(min)/d q_i (tf.minimum: ties go to q1).
 * heads/sac_q_head.py:84-88, heads/td3_v_head.py:54-58 */
int rlx_min_pair(const float *q1, const float *q2, float *out_min, float *grad1, float *grad2,
                 float grad_scale, int n, void *stream);
/* rlx_min_pair + rlx_sac_value_targets in one launch (soft_actor_critic_agent.py:198-200,216-217,244):
 * out_min = min(q1, q2), value_targets = out_min - sampled_logprob, grad_i = grad_scale * d sum(min) / d q_i. */
int rlx_sac_min_targets(const float *q1, const float *q2, const float *sampled_logprob, float grad_scale, int n,
                        float *out_min, float *value_targets, float *grad1, float *grad2, void *stream);
int rlx_select_rows(const unsigned char *mask, const void *if_set, const void *if_clear, void *out,
                    int n, long long row_bytes, void *stream);   /* out[r] = mask[r] ? if_set[r] : if_clear[r] */
/* SACPolicyHead (heads/sac_head.py:60-97): mu_logsig [batch, 2*action_dim] is the head's dense
 * output; standard_normals [batch, action_dim] are host draws (fp64).  Any output may be NULL. */
int rlx_sac_policy_head(const float *mu_logsig, long long ld, const double *standard_normals, int batch,
                        int action_dim, float *out_mean, float *out_log_std, float *out_raw_actions,
                        float *out_actions, float *out_logprob, void *stream);
/* d/d mu_logsig of  logprob_mean_weight * mean_b(logprob_b) + action_weight_scale *
 * sum(action_weights * actions)  (weighted_gradients[5] / [3], soft_actor_critic_agent.py:210-229);
 * accumulate != 0 adds onto d_mu_logsig (the passes of one update share the deterministic torso
 * output, so their head gradients can be summed before ONE backward pass through the torso). */
int rlx_sac_policy_head_backward(const float *mu_logsig, long long ld, const double *standard_normals,
                                 int batch, int action_dim, float logprob_mean_weight,
                                 const float *action_weights, float action_weight_scale,
                                 float *d_mu_logsig, long long ld_grad, int accumulate, void *stream);

/* -------------------------------------------------------- exploration policies -- */
int rlx_categorical_sample(const float *probs, long long ld, const double *uniforms, int n_env,
                           int n_actions, int *actions, void *stream); /* exploration_policies/categorical.py:45-48 */
/* rlx_softmax (tf.nn.softmax of the policy head's logits, heads/ppo_head.py:108) + rlx_categorical_sample on the
 * probabilities it produces, as ONE launch — the acting step of discrete Clipped PPO.  probs_out may be NULL. */
int rlx_softmax_categorical_sample(const float *logits, long long ld, const double *uniforms, int n_env,
                                   int n_actions, float *probs_out, long long ld_out, int *actions, void *stream);
int rlx_argmax_rows(const float *values, long long ld, int n_rows, int n_cols, int *out,
                    void *stream);            /* np.argmax per row: exploration_policies/categorical.py:50-52 */
int rlx_egreedy(const float *q_values, long long ld, const double *explore_uniforms,
                const int *random_actions, const double *tie_break_uniforms, double epsilon,
                int n_env, int n_actions, int *actions, void *stream); /* exploration_policies/e_greedy.py:84-101 */
int rlx_gaussian_action(const float *mean, const float *std_per_dim, const float *std_per_sample,
                        const double *standard_normals, const float *action_low,
                        const float *action_high, int n_env, int action_dim, float *actions,
                        void *stream);                              /* exploration_policies/additive_noise.py:75-111 */

/* ------------------------------------------------ synthetic vector environment -- */
/* Device-resident stand-in for Environment.step (environments/environment.py:276-327) on the
 * BASELINE workloads: fixed-length episodes, Philox4x32-10 observations/rewards keyed by
 * (seed, env id) and counted by (episode, step).  kind 0 = uint8 image frames, 1 = fp32 vectors.
 * episode/step: int32[n_env] state.  reset_obs[e] is written only where game_over[e] is set. */
int rlx_synth_env_reset(int kind, void *obs, int *episode, int *step, int n_env, int obs_elems,
                        unsigned int seed, unsigned int env_id0, void *stream);
int rlx_synth_env_step(int kind, void *next_obs, void *reset_obs, float *reward,
                       unsigned char *game_over, int *episode, int *step, int n_env, int obs_elems,
                       int episode_len, unsigned int seed, unsigned int env_id0, void *stream);

/* the same with a time limit per env (device int32[n_env]): envs whose episodes end on different steps, the
 * case a real simulator front end produces (environment.py:276-327 episode bookkeeping is per env) */
int rlx_synth_env_step_lengths(int kind, void *next_obs, void *reset_obs, float *reward,
                               unsigned char *game_over, int *episode, int *step, int n_env, int obs_elems,
                               const int *episode_len_per_env, unsigned int seed, unsigned int env_id0,
                               void *stream);

/* ------------------------------------------------------------ CartPole-v0 / -v1 -- */
/* N CartPole environments per GPU: gym 0.12.5's physics (gym/envs/classic_control/cartpole.py `step`, fp64, Euler)
 * behind gym's TimeLimit (max_episode_steps), i.e. what `GymVectorEnvironment(level='CartPole-v0')` steps through
 * rl_coach/environments/gym_environment.py:418-474 (presets/CartPole_DQN.py:46, CartPole_ClippedPPO.py:59).
 * state: fp64[n_env][4] (x, x_dot, theta, theta_dot) — after a step it holds the state the NEXT step starts from
 * (the new episode's first state where game_over is set).  next_obs / reset_obs: fp32[n_env][4] views of the stepped
 * state / the new episode's first state (reset_obs is written only where game_over is set); next_state64 (optional):
 * the stepped state in fp64.  reward = 1.0 per step.  Reset states are Philox draws keyed by (seed, env_id0 + e)
 * and counted by the episode number.  status: bit 0 = a pole angle outside rlx::libm_sin's table domain, bit 1 = an
 * action outside {0, 1}.  next_episode != 0: forced reset mid-episode (every env starts its next episode now). */
int rlx_cartpole_reset(double *state, float *obs, int *episode, int *steps, int n_env, unsigned int seed,
                       unsigned int env_id0, int next_episode, void *stream);
int rlx_cartpole_step(const int *action, double *state, int *episode, int *steps, float *next_obs, float *reset_obs,
                      double *next_state64, float *reward, unsigned char *game_over, int n_env,
                      int max_episode_steps, unsigned int seed, unsigned int env_id0, int *status, void *stream);
/* sin / cos rounded like the host libm's (csrc/libm_sincos.hpp) — exposed for tests/test_cartpole.py */
int rlx_libm_sincos(const double *x, double *sin_out, double *cos_out, int n, int *status, void *stream);


/* ------------------------------------------------------------ fused continuous-control updates (ac_fused.hip) -- */
/* A 2-hidden-layer MLP inside a flat parameter buffer: float offsets of the three layers' kernels ([in][out], TF layout)
 * and biases for tower 0, and the offset between the towers of each layer's parameter group (twin critics). */
typedef struct rlx_mlp3 {
    long long off_w1, off_b1, off_w2, off_b2, off_w3, off_b3;
    long long tower_stride1, tower_stride2, tower_stride3;
    int d_in, h1, h2, d_out;
} rlx_mlp3;
/* One network's flat buffers and optimiser as the fused updates drive them (TF1 Adam of rlx_adam_tf1; mix_rate >= 0: the
 * soft target update w_t <- rate w + (1 - rate) w_t rides in the same pass; norm_out: tf.global_norm of the update's raw
 * gradients or null; ticket: >= 1 zeroed 32-bit word, re-armed by every launch). */
typedef struct rlx_fused_net {
    float *weights, *target_weights, *adam_m, *adam_v, *adam_state, *grads, *norm_out;
    unsigned int *ticket;
    float learning_rate, beta1, beta2, epsilon, grad_scale, mix_rate;
} rlx_fused_net;
/* TD3Agent.learn_from_batch (rl_coach/agents/td3_agent.py:148-209) on MLP actor / twin-critic networks (td3_agent.py:36-68:
 * actor obs -> h1 -> h2 -> tanh head x actor_scale, critic concat(action, obs) -> h1 -> h2 -> Dense(1) per stream; relu).
 * obs / next_obs [batch][obs_dim], actions [batch][act_dim], rewards [batch], game_overs [batch] (as stored),
 * noise [batch][act_dim] fp64: the host's np.random.normal(0, policy_noise) draw BEFORE clipping (:162) or null.
 * Outputs: td_targets [batch], q_min [batch] (output #2 of the target critic), loss [3] = stream losses and their sum,
 * neg_action_grad [batch][act_dim] = -actor_scale * d mean(Q_1) / d a (actor step). */
typedef struct rlx_td3_fused_desc {
    rlx_fused_net actor, critic;
    rlx_mlp3 actor_mlp, critic_mlp;
    const float *obs, *next_obs, *actions, *rewards;
    const unsigned char *game_overs;
    const double *noise;
    const float *action_low, *action_high;
    double noise_clip, discount, clip_low, clip_high;
    int use_non_zero_discount_for_terminal_states, has_clip;
    float actor_scale;
    int batch, obs_dim, act_dim;
    float *workspace; long long workspace_floats;
    float *td_targets, *q_min, *loss, *neg_action_grad;
} rlx_td3_fused_desc;
int rlx_td3_fused_supported(const rlx_td3_fused_desc *desc_host);           /* 1 / 0: shapes only (pointers not read) */
/* Measurement switch (no reference counterpart): workgroup 0 of every chain kernel records s_memtime at its phase
 * boundaries into the LAST 96 int64 words of the workspace. */
int rlx_fused_phase_stamps(int enable);
int rlx_td3_fused_workspace_floats(const rlx_td3_fused_desc *desc_host, long long *floats_host);
/* The critic half (:157-184) as three launches: target chain + online forward, loss + input-gradient chain, weight
 * gradients + Adam (+ norm, + soft target update when critic.mix_rate >= 0).  write_grads != 0: the weight gradients go
 * to critic.grads and no optimiser step is taken (data parallel: the caller all-reduces and steps). */
int rlx_td3_fused_critic_update(const rlx_td3_fused_desc *desc_host, int write_grads, void *stream);
/* The actor half (:186-207) as two launches; uses the critic's CURRENT (already updated) online weights. */
int rlx_td3_fused_actor_update(const rlx_td3_fused_desc *desc_host, int write_grads, void *stream);


/* SoftActorCriticAgent.learn_from_batch (rl_coach/agents/soft_actor_critic_agent.py:168-280) on the Mujoco_SAC topology:
 * policy obs -> h1 -> h2 -> [mu | log sigma] (heads/sac_head.py:60-97), V obs -> h1 -> h2 -> 1 with a target copy, twin Q
 * towers relu(obs_fc s) + relu(act_fc a) -> fc -> 1 (heads/sac_q_head.py:46-96; q_* = float offsets of tower 0's tensors in
 * q.weights and the towers' strides).  normals [3][batch][act_dim] fp64: the host's standard-normal draws of the three
 * policy passes (resample_noise_per_pass == 0: only the first is used).  Outputs: value_targets, log_target = min(Q1, Q2),
 * td_targets [batch], dq_da [batch][act_dim], q_loss [3] = the towers' losses and their sum, v_loss [1].
 * policy.ticket: >= 1 zeroed word.  Six launches. */
typedef struct rlx_sac_fused_desc {
    rlx_fused_net policy, q, v;
    rlx_mlp3 policy_mlp, v_mlp;
    long long q_off_obs_w, q_off_obs_b, q_off_act_w, q_off_act_b, q_off_fc_w, q_off_fc_b, q_off_out_w, q_off_out_b;
    long long q_stride_obs, q_stride_act, q_stride_fc, q_stride_out;
    const float *obs, *next_obs, *actions, *rewards;
    const unsigned char *game_overs;
    const double *normals;
    double discount;
    int resample_noise_per_pass, batch, obs_dim, act_dim, q_hidden, reserved;
    float *workspace; long long workspace_floats;
    float *value_targets, *log_target, *td_targets, *dq_da, *q_loss, *v_loss;
} rlx_sac_fused_desc;
int rlx_sac_fused_supported(const rlx_sac_fused_desc *desc_host);
int rlx_sac_fused_workspace_floats(const rlx_sac_fused_desc *desc_host, long long *floats_host);
int rlx_sac_fused_update(const rlx_sac_fused_desc *desc_host, int write_grads, void *stream);

/* ------------------------------------------------- first convolution's weight gradient from uint8 frames -- */
/* dW1 = cols(frames)^T dz1 and db1 = column sums of dz1 for the first convolution of an image torso whose towers read the
 * SAME uint8 frames (rl_coach/architectures/tensorflow_components/embedders/image_embedder.py:33-40, layers.py:108-121;
 * the tf.gradients pass of architecture.py:187-220) — what rlx_gemm computed as an implicit-im2col product on its
 * register-staged uint8 loop.  frames [B][H][W][C] uint8 (network input = byte / a_div), dz [towers][B * OH * OW][filters]
 * (tower stride dz_tower_stride floats).  One workgroup per (image, pair of kernel rows); the images' partial sums go to
 * `workspace` and *job_host describes their outstanding reduction (one split per image, summed in image order) for
 * rlx_splitk_reduce_jobs: dw [KH * KW * C][filters] per tower (tower stride dw_tower_stride), db [filters] per tower or
 * NULL.  Shapes: KW * C == 32, KH even, towers * filters == 64, 2 <= B <= 128 (rlx_conv_dw_u8_supported). */
int rlx_conv_dw_u8_supported(int B, int H, int W, int C, int KH, int KW, int S, int filters, int towers);   /* 1 / 0 */
int rlx_conv_dw_u8_workspace_floats(int B, int H, int W, int C, int KH, int KW, int S, int filters, int towers,
                                    long long *floats_host);
int rlx_conv_dw_u8(const unsigned char *frames, float a_div, const float *dz, long long dz_tower_stride, int B, int H, int W,
                   int C, int KH, int KW, int S, int filters, int towers, float *dw, long long dw_tower_stride, float *db,
                   long long db_tower_stride, float *workspace, long long workspace_floats, rlx_splitk_job *job_host,
                   void *stream);
/* Measurement switch (no reference counterpart): workgroup 0 records s_memtime at its phase boundaries into 5 int64 words. */
int rlx_conv_dw_u8_stamps(long long *device_words5);

/* The same for an inner convolution layer with fp32 input activations x [towers][B][H][W][C] (tower stride x_tower_stride)
 * and 64 filters — the Atari torso's conv2 (4 x 4 x 32 stride 2 on 20 x 20) and conv3 (3 x 3 x 64 stride 1 on 9 x 9): one
 * workgroup per (tower, pair of images, kernel row), one deferred split per image pair.  dw [KH * KW * C][64] per tower. */
int rlx_conv_dw_f32_supported(int B, int H, int W, int C, int KH, int KW, int S, int filters, int towers);   /* 1 / 0 */
int rlx_conv_dw_f32_workspace_floats(int B, int H, int W, int C, int KH, int KW, int S, int filters, int towers,
                                     long long *floats_host);
int rlx_conv_dw_f32(const float *x, long long x_tower_stride, const float *dz, long long dz_tower_stride, int B, int H, int W,
                    int C, int KH, int KW, int S, int filters, int towers, float *dw, long long dw_tower_stride, float *db,
                    long long db_tower_stride, float *workspace, long long workspace_floats, rlx_splitk_job *job_host,
                    void *stream);
int rlx_conv_dw_f32_stamps(long long *device_words4);

/* The weight gradients of up to three convolution layers of one backward pass — independent once the input-gradient chain
 * has passed — as ONE launch: items[i] is what rlx_conv_dw_u8 (x_is_u8, x = frames) or rlx_conv_dw_f32 would get, jobs[i]
 * receives its outstanding reduction.  The one-launch form takes at most one uint8 item of the Atari geometry and two fp32
 * items; any other combination is issued item by item (same results either way). */
typedef struct rlx_conv_dw_item {
    const void *x; long long x_tower_stride;
    int x_is_u8; float a_div;
    const float *dz; long long dz_tower_stride;
    int B, H, W, C, KH, KW, S, filters, towers;
    float *dw; long long dw_tower_stride;
    float *db; long long db_tower_stride;
    float *workspace; long long workspace_floats;
} rlx_conv_dw_item;
int rlx_conv_dw_multi(const rlx_conv_dw_item *items_host, rlx_splitk_job *jobs_host, int n_items, void *stream);
/* Process-wide (default 2): 2 = the one-launch form stages conv1's and conv2's operands in TWO passes over the output rows
 * (79 KB of LDS per workgroup instead of 157 KB: two workgroups per CU, one's fills under the other's products).  Weight
 * gradients identical; the inner layer's bias gradient sums its positions in another grouping (last bits).  -6 us per Clipped-PPO
 * update (profiles/r06_ab_conv_dw_passes.txt).  1 = the bodies of rlx_conv_dw_u8 / rlx_conv_dw_f32 as they are. */
int rlx_conv_dw_passes(int passes);
/* Process-wide, multi-pass forms only: image pairs an fp32 item's workgroup takes one after the other into the same
 * accumulators.  2: half the splits of the conv2 / conv3 weight gradients — 8.9 MB less partial sums written and read back per
 * Clipped-PPO minibatch update at the same launch time — at half the workgroups; 1: a split per pair; 0 (default): 2 where an
 * item has >= 64 (tower, pair) units, else 1 (profiles/r06_ab_conv_dw_pairs.txt). */
int rlx_conv_dw_pairs_per_workgroup(int pairs);

/* ------------------------------------------------- Clipped PPO: last dense layer + heads + losses, one launch -- */
/* The middleware's Dense(units) of both towers (tower 0 = value, tower 1 = policy; layers.py:168-185), VHead / discrete
 * PPOHead forward (heads/v_head.py:43-52, heads/ppo_head.py:52-116), both head losses (head.py:143-186) and the heads'
 * backward pass (architecture.py:312-385) — what rlx_gemm (row_heads) + rlx_ppo_heads_loss_backward did in three launches.
 * x [2][batch][in_features]; weights [in][units] / bias [units] per tower; value_w [units][1], policy_w [units][n_actions].
 * Outputs: h (the layer's output) and dz = d loss / d (its pre-activation) [2][batch][units]; values [batch], logits
 * [batch][n_actions] and their gradients; the heads' weight / bias gradients; scalars [5] = surrogate, entropy, KL, policy
 * total, value loss.  workspace / tickets: rlx_ppo_fc_heads_workspace (tickets zeroed once; every launch re-arms them). */
typedef struct rlx_ppo_fc_heads_desc {
    const float *x; long long x_tower_stride;
    const float *weights; long long weight_tower_stride;
    const float *bias; long long bias_tower_stride;
    const float *value_w, *value_b, *policy_w, *policy_b;
    const float *value_targets, *advantages, *old_probs; long long ld_old;
    const int *actions;
    const float *clip_scale;                 /* device scalar multiplying clip_epsilon, or null */
    float clip_epsilon, beta_entropy, grad_scale;
    int batch, in_features, units, n_actions, activation;
    float *h, *dz, *values, *logits, *dvalues, *dlogits;
    float *d_value_w, *d_value_b, *d_policy_w, *d_policy_b;
    float *scalars, *likelihood_ratio, *clipped_likelihood_ratio;
    int *status;
    float *workspace; long long workspace_floats;
    unsigned int *tickets;
} rlx_ppo_fc_heads_desc;
int rlx_ppo_fc_heads_supported(int batch, int in_features, int units, int n_actions);      /* 1 / 0 */
int rlx_ppo_fc_heads_workspace(int units, long long *floats_host, long long *ticket_words_host);
int rlx_ppo_fc_heads(const rlx_ppo_fc_heads_desc *desc_host, void *stream);
/* Measurement switch: workgroup 0 and the last arrivers record s_memtime into the workspace's last 16 int64. */
int rlx_ppo_fc_heads_stamps(int enable);

#ifdef __cplusplus
}
#endif
#endif /* RLX_H */
