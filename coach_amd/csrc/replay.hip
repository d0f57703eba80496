// K1 / K4 — device-resident replay storage: row gather/scatter and the frame-dedup image ring.
//
// Replaces, in the reference (paths under rl_coach/):
//   * ExperienceReplay.store/_enforce_max_length/sample  memories/non_episodic/experience_replay.py:71-150
//     (Python list append + `del list[0]` FIFO + list-comprehension gather) and
//     Batch.states/next_states/actions/rewards/game_overs collation  core_types.py:488-649
//       -> rlx_copy_columns (one launch gathers every column of the sampled batch; the FIFO is a
//          ring: logical index i of the reference's list == physical row (head + i) % capacity).
//   * ObservationStackingFilter deque + LazyStack.__array__ np.stack(axis=-1)
//     filters/observation/observation_stacking_filter.py:27-41,89-101
//       -> rlx_imgreplay_* : every frame is stored ONCE per env in a ring; a stacked state is
//          materialised on demand from `stack` ring rows (first frame of an episode replicated,
//          :90-91), so a sampled transition costs (stack+1) frame reads instead of 2*stack.
//
// All kernels are HBM-bound byte movers: 16-byte accesses where alignment permits, grid capped at
// 2048 blocks and grid-strided (cdna_hip_programming.md G11/G13).  No inter-block reuse, so no
// XCD remap is needed.
#include "rlx_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxCols = RLX_MAX_COLUMNS;

struct ColumnSet {
    const unsigned char *src[kMaxCols];
    unsigned char *dst[kMaxCols];
    long long row_bytes[kMaxCols];
    int ncols;
};

__device__ __forceinline__ long long wrap(long long v, long long n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// Row r of every column:  src row = src_idx ? src_idx[r] : (src_start + r) mod src_rows
//                         dst row = dst_idx ? dst_idx[r] : (dst_start + r) mod dst_rows
// blockIdx.y = column.  W = access width in bytes (16 / 4 / 1), chosen per launch so that every
// column's row_bytes and base pointers are W-aligned.
template <typename T>
__global__ void copy_columns_kernel(ColumnSet cs, const int *__restrict__ src_idx,
                                    const int *__restrict__ dst_idx, long long src_start,
                                    long long dst_start, long long src_rows, long long dst_rows,
                                    int n, int *__restrict__ status) {
    const int c = blockIdx.y;
    const long long row_elems = cs.row_bytes[c] / (long long)sizeof(T);
    const T *__restrict__ src = reinterpret_cast<const T *>(cs.src[c]);
    T *__restrict__ dst = reinterpret_cast<T *>(cs.dst[c]);
    const long long total = row_elems * n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const long long r = t / row_elems;
        const long long e = t - r * row_elems;
        long long s = src_idx ? (long long)src_idx[r] : wrap(src_start + r, src_rows);
        long long d = dst_idx ? (long long)dst_idx[r] : wrap(dst_start + r, dst_rows);
        if (s < 0 || s >= src_rows || d < 0 || d >= dst_rows) {   // IndexError in the reference
            if (e == 0) atomicOr(status, 1);
            continue;
        }
        dst[d * row_elems + e] = src[s * row_elems + e];
    }
}

// ------------------------------------------------------------------------------ image replay
// State per env e (device arrays of length n_env):
//   fpos[e]   ring row holding the newest frame of the env's CURRENT stacked state
//   epoff[e]  number of earlier frames of the current episode available (clipped to stack-1)
// Transition table (capacity rows, written in env order each vector step at rows
// (cursor + e) mod capacity):  t_fpos, t_epoff (state descriptor), action, reward, done.
// Ring: u8 [n_env][ring_frames][frame_bytes].

// One thread per 4 bytes of a frame; blockIdx.y = env.
__global__ void img_append_kernel(unsigned char *__restrict__ ring, int *__restrict__ fpos,
                                  int *__restrict__ epoff, int *__restrict__ t_fpos,
                                  unsigned char *__restrict__ t_epoff,
                                  const unsigned char *__restrict__ next_frame,
                                  const unsigned char *__restrict__ reset_frame,
                                  const unsigned char *__restrict__ done, int n_env,
                                  int ring_frames, int frame_bytes, int stack,
                                  long long cursor, long long capacity, int record) {
    const int e = blockIdx.y;
    const int pos = fpos[e];
    const int off = epoff[e];
    const bool is_done = done[e] != 0;
    const int p1 = (pos + 1) % ring_frames;
    const int p2 = (pos + 2) % ring_frames;
    const int words = frame_bytes >> 2;
    const uint32_t *src1 = reinterpret_cast<const uint32_t *>(next_frame + (size_t)e * frame_bytes);
    uint32_t *dst1 =
        reinterpret_cast<uint32_t *>(ring + ((size_t)e * ring_frames + p1) * frame_bytes);
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x)
        dst1[w] = src1[w];
    if (is_done && reset_frame) {
        const uint32_t *src2 =
            reinterpret_cast<const uint32_t *>(reset_frame + (size_t)e * frame_bytes);
        uint32_t *dst2 =
            reinterpret_cast<uint32_t *>(ring + ((size_t)e * ring_frames + p2) * frame_bytes);
        for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x)
            dst2[w] = src2[w];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (record) {
            long long slot = (cursor + e) % capacity;
            t_fpos[slot] = pos;
            t_epoff[slot] = (unsigned char)off;
        }
    }
}

// Second, tiny launch (after the copy kernel has consumed fpos/epoff): advance env state.
__global__ void img_advance_kernel(int *__restrict__ fpos, int *__restrict__ epoff,
                                   const unsigned char *__restrict__ done, int n_env,
                                   int ring_frames, int stack, int has_reset) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    int pos = fpos[e], off = epoff[e];
    if (done[e] && has_reset) {
        fpos[e] = (pos + 2) % ring_frames;     // the post-reset first frame
        epoff[e] = 0;
    } else {
        fpos[e] = (pos + 1) % ring_frames;
        epoff[e] = min(off + 1, stack - 1);
    }
}

// Writes stacked states (B, H*W, stack) u8 with the stack index innermost (np.stack(axis=-1)).
// Fast path stack == 4: a thread reads one dword (4 pixels) from each of the 4 source frames,
// transposes the 4x4 byte tile in registers and stores 16 bytes.
// which: bit0 = state, bit1 = next_state.   rows: idx[b] = transition-table row, or (idx == null)
// the CURRENT state of env b (acting path; only `state` is produced).
// COLS: the sampled rows of the batch's small columns (action, reward, game_over ...: Batch collation,
// core_types.py:488-649) ride along as the workgroup of one more y slice — rlx_copy_columns' gather without a launch of
// its own in front of every off-policy update (4.4 us in the C3 update's chain).
template <bool COLS>
__global__ void img_gather4_kernel(const unsigned char *__restrict__ ring,
                                   const int *__restrict__ t_fpos,
                                   const unsigned char *__restrict__ t_epoff,
                                   const int *__restrict__ env_fpos,
                                   const int *__restrict__ env_epoff,
                                   const int *__restrict__ idx, int n_env, int ring_frames,
                                   int frame_bytes, long long capacity,
                                   unsigned char *__restrict__ out_state,
                                   unsigned char *__restrict__ out_next, int which,
                                   int *__restrict__ status, const ColumnSet cs, int batch) {
    if (COLS && (int)blockIdx.y == batch) {
        if (blockIdx.x != 0) return;
        for (int c = 0; c < cs.ncols; ++c) {
            const int rb = (int)cs.row_bytes[c], total = rb * batch;
            const unsigned char *__restrict__ src = cs.src[c];
            unsigned char *__restrict__ dst = cs.dst[c];
            for (int t = threadIdx.x; t < total; t += blockDim.x) {
                const int r = t / rb, e = t - r * rb;
                const long long row = idx[r];
                if (row < 0 || row >= capacity) {             // IndexError in the reference
                    if (e == 0) atomicOr(status, 1);
                    continue;
                }
                dst[(size_t)r * rb + e] = src[(size_t)row * rb + e];
            }
        }
        return;
    }
    const int b = blockIdx.y;
    int e, pos, off;
    if (idx) {
        const int row = idx[b];
        if (row < 0 || row >= capacity) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1);
            return;
        }
        e = row % n_env;
        pos = t_fpos[row];
        off = t_epoff[row];
    } else {
        e = b;
        pos = env_fpos[b];
        off = env_epoff[b];
    }
    const size_t env_base = (size_t)e * ring_frames * frame_bytes;
    const int words = frame_bytes >> 2;
    for (int pass = 0; pass < 2; ++pass) {
        if (!(which & (1 << pass))) continue;
        const int newest = pass == 0 ? pos : pos + 1;
        const int o = pass == 0 ? off : min(off + 1, 3);
        const uint32_t *f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int back = min(3 - k, o);                         // first frame replicated (:90-91)
            int fr = ((newest - back) % ring_frames + ring_frames) % ring_frames;
            f[k] = reinterpret_cast<const uint32_t *>(ring + env_base + (size_t)fr * frame_bytes);
        }
        uint4 *out = reinterpret_cast<uint4 *>((pass == 0 ? out_state : out_next) +
                                               (size_t)b * frame_bytes * 4);
        for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words;
             w += gridDim.x * blockDim.x) {
            const uint32_t a0 = f[0][w], a1 = f[1][w], a2 = f[2][w], a3 = f[3][w];
            // 4x4 byte transpose with v_perm_b32: perm(hi, lo, sel) selects from bytes
            // {lo.b0..b3 = 0..3, hi.b0..b3 = 4..7}.
            const uint32_t lo01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);  // a0b0 a1b0 a0b1 a1b1
            const uint32_t hi01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);  // a0b2 a1b2 a0b3 a1b3
            const uint32_t lo23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
            const uint32_t hi23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
            uint4 v;
            v.x = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);   // pixel 0: a0 a1 a2 a3
            v.y = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);   // pixel 1
            v.z = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);   // pixel 2
            v.w = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);   // pixel 3
            out[w] = v;
        }
    }
}

// Generic stack size (byte-granular, slower): out[b][p][k].
__global__ void img_gather_generic_kernel(const unsigned char *__restrict__ ring,
                                          const int *__restrict__ t_fpos,
                                          const unsigned char *__restrict__ t_epoff,
                                          const int *__restrict__ env_fpos,
                                          const int *__restrict__ env_epoff,
                                          const int *__restrict__ idx, int n_env, int ring_frames,
                                          int frame_bytes, int stack, long long capacity,
                                          unsigned char *__restrict__ out_state,
                                          unsigned char *__restrict__ out_next, int which,
                                          int *__restrict__ status) {
    const int b = blockIdx.y;
    int e, pos, off;
    if (idx) {
        const int row = idx[b];
        if (row < 0 || row >= capacity) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1);
            return;
        }
        e = row % n_env;
        pos = t_fpos[row];
        off = t_epoff[row];
    } else {
        e = b;
        pos = env_fpos[b];
        off = env_epoff[b];
    }
    const size_t env_base = (size_t)e * ring_frames * frame_bytes;
    const long long total = (long long)frame_bytes * stack;
    for (int pass = 0; pass < 2; ++pass) {
        if (!(which & (1 << pass))) continue;
        const int newest = pass == 0 ? pos : pos + 1;
        const int o = pass == 0 ? off : min(off + 1, stack - 1);
        unsigned char *out = (pass == 0 ? out_state : out_next) + (size_t)b * total;
        for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
             t += (long long)gridDim.x * blockDim.x) {
            const int k = (int)(t % stack);
            const long long p = t / stack;
            int back = min(stack - 1 - k, o);
            int fr = ((newest - back) % ring_frames + ring_frames) % ring_frames;
            out[t] = ring[env_base + (size_t)fr * frame_bytes + p];
        }
    }
}

// First frame of a new episode in every env (the very first one, or a reset in the middle of an
// episode — GraphManager.reset_internal_state(force_environment_reset=True) before an evaluation).
// The frame goes BEHIND the newest frame of the env, never to a fixed row: stored transitions
// address absolute ring rows, so restarting at row 0 would overwrite the frames of live transitions.
// An episode that has not stepped yet (epoff == 0: nothing refers to its first frame) is restarted
// in place, so repeated resets do not consume ring rows.  On a fresh ring (fpos = epoff = 0) this is
// row 0.  Second, tiny launch: the position update (the copy blocks all read fpos/epoff).
__global__ void img_reset_kernel(unsigned char *__restrict__ ring, const int *__restrict__ fpos,
                                 const int *__restrict__ epoff,
                                 const unsigned char *__restrict__ first_frame, int ring_frames,
                                 int frame_bytes) {
    const int e = blockIdx.y;
    const int words = frame_bytes >> 2;
    const int p = epoff[e] == 0 ? fpos[e] : (fpos[e] + 1) % ring_frames;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(first_frame + (size_t)e * frame_bytes);
    uint32_t *dst =
        reinterpret_cast<uint32_t *>(ring + ((size_t)e * ring_frames + p) * frame_bytes);
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x)
        dst[w] = src[w];
}

__global__ void img_reset_advance_kernel(int *__restrict__ fpos, int *__restrict__ epoff, int n_env,
                                         int ring_frames) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    if (epoff[e] != 0) fpos[e] = (fpos[e] + 1) % ring_frames;
    epoff[e] = 0;
}

}  // namespace

extern "C" {

int rlx_copy_columns(const rlx_column *columns_host, int ncols, const int *src_idx,
                     const int *dst_idx, long long src_start, long long dst_start,
                     long long src_rows, long long dst_rows, int n, int *status, void *stream) {
    RLX_REQUIRE(columns_host != nullptr && ncols > 0 && ncols <= kMaxCols,
                "rlx_copy_columns: need 1..%d columns, got %d", kMaxCols, ncols);
    RLX_REQUIRE(src_rows > 0 && dst_rows > 0 && n >= 0 && status,
                "rlx_copy_columns: bad row counts (src_rows=%lld dst_rows=%lld n=%d)", src_rows,
                dst_rows, n);
    if (n == 0) return RLX_OK;
    ColumnSet cs;
    cs.ncols = ncols;
    int width = 16;
    long long max_row = 0;
    for (int c = 0; c < ncols; ++c) {
        const rlx_column &col = columns_host[c];
        RLX_REQUIRE(col.src && col.dst && col.row_bytes > 0, "rlx_copy_columns: column %d is empty",
                    c);
        cs.src[c] = static_cast<const unsigned char *>(col.src);
        cs.dst[c] = static_cast<unsigned char *>(col.dst);
        cs.row_bytes[c] = col.row_bytes;
        uintptr_t bits = (uintptr_t)col.src | (uintptr_t)col.dst | (uintptr_t)col.row_bytes;
        while (width > 1 && (bits & (uintptr_t)(width - 1))) width = width == 16 ? 4 : 1;
        if (col.row_bytes > max_row) max_row = col.row_bytes;
    }
    for (int c = ncols; c < kMaxCols; ++c) {
        cs.src[c] = nullptr;
        cs.dst[c] = nullptr;
        cs.row_bytes[c] = 0;
    }
    dim3 grid(rlx::grid_for(max_row / width * n, kBlock, rlx::kMaxStreamBlocks / ncols + 1), ncols);
    hipStream_t s = rlx::as_stream(stream);
    if (width == 16)
        RLX_LAUNCH((copy_columns_kernel<uint4>), grid, kBlock, 0, s, cs, src_idx, dst_idx, src_start,
                                                           dst_start, src_rows, dst_rows, n, status);
    else if (width == 4)
        RLX_LAUNCH((copy_columns_kernel<uint32_t>), grid, kBlock, 0, s, cs, src_idx, dst_idx, src_start,
                                                              dst_start, src_rows, dst_rows, n,
                                                              status);
    else
        RLX_LAUNCH((copy_columns_kernel<unsigned char>), grid, kBlock, 0, s, cs, src_idx, dst_idx, src_start, dst_start, src_rows, dst_rows, n, status);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_imgreplay_reset(unsigned char *ring, int *env_fpos, int *env_epoff,
                        const unsigned char *first_frame, int n_env, int ring_frames,
                        int frame_bytes, void *stream) {
    RLX_REQUIRE(ring && env_fpos && env_epoff && first_frame, "rlx_imgreplay_reset: null pointer");
    RLX_REQUIRE(n_env > 0 && ring_frames >= 2 && frame_bytes > 0 && frame_bytes % 4 == 0,
                "rlx_imgreplay_reset: frame_bytes must be a positive multiple of 4 (got %d)",
                frame_bytes);
    dim3 grid(rlx::grid_for(frame_bytes / 4, kBlock, 8), n_env);
    hipStream_t s = rlx::as_stream(stream);
    RLX_LAUNCH((img_reset_kernel), grid, kBlock, 0, s, ring, env_fpos, env_epoff, first_frame, ring_frames,
                                             frame_bytes);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((img_reset_advance_kernel), (n_env + 63) / 64, 64, 0, s, env_fpos, env_epoff, n_env,
                                                              ring_frames);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_imgreplay_append(unsigned char *ring, int *env_fpos, int *env_epoff, int *t_fpos,
                         unsigned char *t_epoff, const unsigned char *next_frame,
                         const unsigned char *reset_frame, const unsigned char *done, int n_env,
                         int ring_frames, int frame_bytes, int stack, long long cursor,
                         long long capacity, int record, void *stream) {
    RLX_REQUIRE(ring && env_fpos && env_epoff && next_frame && done,
                "rlx_imgreplay_append: null pointer");
    RLX_REQUIRE(!record || (t_fpos && t_epoff && capacity > 0 && cursor >= 0),
                "rlx_imgreplay_append: recording needs the transition table");
    RLX_REQUIRE(n_env > 0 && ring_frames >= stack + 2 && stack >= 1 && stack <= 255,
                "rlx_imgreplay_append: ring of %d frames is too small for stack %d", ring_frames,
                stack);
    RLX_REQUIRE(frame_bytes > 0 && frame_bytes % 4 == 0,
                "rlx_imgreplay_append: frame_bytes must be a positive multiple of 4 (got %d)",
                frame_bytes);
    hipStream_t s = rlx::as_stream(stream);
    dim3 grid(rlx::grid_for(frame_bytes / 4, kBlock, 8), n_env);
    RLX_LAUNCH((img_append_kernel), grid, kBlock, 0, s, ring, env_fpos, env_epoff, t_fpos, t_epoff,
                                              next_frame, reset_frame, done, n_env, ring_frames,
                                              frame_bytes, stack, cursor, capacity, record);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((img_advance_kernel), (n_env + 63) / 64, 64, 0, s, env_fpos, env_epoff, done, n_env,
                                                        ring_frames, stack, reset_frame != nullptr);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_imgreplay_gather(const unsigned char *ring, const int *t_fpos,
                         const unsigned char *t_epoff, const int *env_fpos, const int *env_epoff,
                         const int *idx, int batch, int n_env, int ring_frames, int frame_bytes,
                         int stack, long long capacity, unsigned char *out_state,
                         unsigned char *out_next, int *status, void *stream) {
    RLX_REQUIRE(ring && status, "rlx_imgreplay_gather: null pointer");
    RLX_REQUIRE(batch > 0 && n_env > 0 && stack >= 1, "rlx_imgreplay_gather: bad sizes");
    RLX_REQUIRE(out_state || out_next, "rlx_imgreplay_gather: nothing to produce");
    if (idx) {
        RLX_REQUIRE(t_fpos && t_epoff && capacity > 0,
                    "rlx_imgreplay_gather: sampling needs the transition table");
    } else {
        RLX_REQUIRE(env_fpos && env_epoff && batch == n_env && !out_next,
                    "rlx_imgreplay_gather: idx == NULL gathers the current state of every env "
                    "(batch must equal n_env, no next_state)");
    }
    RLX_REQUIRE(frame_bytes > 0 && frame_bytes % 4 == 0,
                "rlx_imgreplay_gather: frame_bytes must be a positive multiple of 4 (got %d)",
                frame_bytes);
    const int which = (out_state ? 1 : 0) | (out_next ? 2 : 0);
    hipStream_t s = rlx::as_stream(stream);
    if (stack == 4) {
        dim3 grid(rlx::grid_for(frame_bytes / 4, kBlock, 8), batch);
        ColumnSet none{};
        RLX_LAUNCH((img_gather4_kernel<false>), grid, kBlock, 0, s, ring, t_fpos, t_epoff, env_fpos, env_epoff, idx,
                                                          n_env, ring_frames, frame_bytes, capacity,
                                                          out_state, out_next, which, status, none, batch);
    } else {
        dim3 grid(rlx::grid_for((long long)frame_bytes * stack, kBlock, 16), batch);
        RLX_LAUNCH((img_gather_generic_kernel), grid, kBlock, 0, s, ring, t_fpos, t_epoff, env_fpos,
                                                          env_epoff, idx, n_env, ring_frames,
                                                          frame_bytes, stack, capacity, out_state,
                                                          out_next, which, status);
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_imgreplay_gather_columns(const unsigned char *ring, const int *t_fpos, const unsigned char *t_epoff,
                                 const int *idx, int batch, int n_env, int ring_frames, int frame_bytes, int stack,
                                 long long capacity, unsigned char *out_state, unsigned char *out_next,
                                 const rlx_column *columns_host, int ncols, int *status, void *stream) {
    RLX_REQUIRE(ring && status && idx && t_fpos && t_epoff && capacity > 0, "rlx_imgreplay_gather_columns: null pointer");
    RLX_REQUIRE(batch > 0 && n_env > 0 && stack == 4, "rlx_imgreplay_gather_columns: stacks of 4 frames");
    RLX_REQUIRE(out_state || out_next, "rlx_imgreplay_gather_columns: nothing to produce");
    RLX_REQUIRE(frame_bytes > 0 && frame_bytes % 4 == 0,
                "rlx_imgreplay_gather_columns: frame_bytes must be a positive multiple of 4 (got %d)", frame_bytes);
    RLX_REQUIRE(columns_host != nullptr && ncols > 0 && ncols <= kMaxCols,
                "rlx_imgreplay_gather_columns: need 1..%d columns, got %d", kMaxCols, ncols);
    ColumnSet cs{};
    cs.ncols = ncols;
    for (int c = 0; c < ncols; ++c) {
        const rlx_column &col = columns_host[c];
        RLX_REQUIRE(col.src && col.dst && col.row_bytes > 0 && col.row_bytes * batch <= (1 << 16),
                    "rlx_imgreplay_gather_columns: column %d is empty or too wide for the riding workgroup", c);
        cs.src[c] = static_cast<const unsigned char *>(col.src);
        cs.dst[c] = static_cast<unsigned char *>(col.dst);
        cs.row_bytes[c] = col.row_bytes;
    }
    const int which = (out_state ? 1 : 0) | (out_next ? 2 : 0);
    dim3 grid(rlx::grid_for(frame_bytes / 4, kBlock, 8), batch + 1);
    RLX_LAUNCH((img_gather4_kernel<true>), grid, kBlock, 0, rlx::as_stream(stream), ring, t_fpos, t_epoff, nullptr, nullptr,
               idx, n_env, ring_frames, frame_bytes, capacity, out_state, out_next, which, status, cs, batch);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
