// K2 / K3 — observation and reward filters on gfx950 (vectorised over envs / batch rows).
//
// Replaces, in the reference (paths under rl_coach/filters/ unless noted):
//   * ObservationRGBToYFilter.filter        observation/observation_rgb_to_y_filter.py:41-47
//   * ObservationToUInt8Filter.filter       observation/observation_to_uint8_filter.py:51-60
//   * ObservationNormalizationFilter.filter observation/observation_normalization_filter.py:71-78
//     + NumpySharedRunningStats.push_val / normalize   utilities/shared_running_stats.py:130-164
//   * RewardRescaleFilter.filter            reward/reward_rescale_filter.py:37-39
//   * RewardClippingFilter.filter           reward/reward_clipping_filter.py:41-49 (truthiness quirk:
//     a bound of 0 is ignored)
//
// fp64 arithmetic in the reference's operation order; compiled with -ffp-contract=off so the
// uint8 truncation and the running sums are bit-identical to numpy.  HBM-bound streaming kernels.
#include "rlx_common.hpp"

namespace {

constexpr int kBlock = 256;

// 4 pixels per thread: 12 input bytes -> 4 output bytes.
__global__ void rgb_to_y_u8_kernel(const unsigned char *__restrict__ rgb,
                                   unsigned char *__restrict__ out, long long n_pixels,
                                   double low, double high) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long groups = n_pixels >> 2;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(rgb) + g * 3;
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        unsigned char b[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = (w0 >> (8 * k)) & 0xff;
            b[4 + k] = (w1 >> (8 * k)) & 0xff;
            b[8 + k] = (w2 >> (8 * k)) & 0xff;
        }
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double y = 0.2989 * (double)b[3 * k] + 0.5870 * (double)b[3 * k + 1] +
                       0.1140 * (double)b[3 * k + 2];                       // rgb_to_y :44-45
            double v = (y - low) / (high - low);                             // to_uint8 :53
            v *= 255;                                                        // :56
            o |= ((uint32_t)(unsigned char)(int)v) << (8 * k);               // astype('uint8') :58
        }
        reinterpret_cast<uint32_t *>(out)[g] = o;
    }
    // tail (n_pixels not a multiple of 4)
    for (long long i = (groups << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
         i < n_pixels; i += stride) {
        double y = 0.2989 * (double)rgb[3 * i] + 0.5870 * (double)rgb[3 * i + 1] +
                   0.1140 * (double)rgb[3 * i + 2];
        double v = (y - low) / (high - low);
        v *= 255;
        out[i] = (unsigned char)(int)v;
    }
}

// One thread per feature column; rows are accumulated sequentially in fp64 — the same order
// numpy uses for `samples.sum(axis=0)` on a C-contiguous (n, D) array, so the sums are
// bit-identical to the reference.  Loads are coalesced across the D columns.
template <typename T>
__global__ void running_stats_push_kernel(const T *__restrict__ samples, long long n, int dim,
                                          double *__restrict__ sum, double *__restrict__ sumsq,
                                          const double *__restrict__ count_in,
                                          double *__restrict__ mean, double *__restrict__ stdv,
                                          double epsilon) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dim) return;
    double s = 0.0, q = 0.0;
    const T *p = samples + j;
    for (long long i = 0; i < n; ++i) {
        double v = (double)p[i * dim];
        s += v;
        q += v * v;                                    // np.square(samples).sum(axis=0)  (:135)
    }
    const double new_sum = sum[j] + s;                 // self._sum += ...             (:134)
    const double new_sq = sumsq[j] + q;
    const double cnt = *count_in + (double)n;          // self._count += n              (:136)
    const double m = new_sum / cnt;                    // (:137)
    double var = (new_sq - cnt * (m * m)) / fmax(cnt - 1.0, 1.0);   // (:138-140)
    sum[j] = new_sum;
    sumsq[j] = new_sq;
    mean[j] = m;
    stdv[j] = sqrt(fmax(var, epsilon));
}

__global__ void running_stats_count_kernel(double *count, double n) { *count += n; }

// count is updated by thread (0,0) AFTER every thread of the (single) workgroup has read it
__global__ void running_stats_merge_kernel(const double *__restrict__ delta, int dim,
                                           double *__restrict__ sum, double *__restrict__ sumsq,
                                           double *__restrict__ count, double *__restrict__ mean,
                                           double *__restrict__ stdv, double epsilon) {
    const double cnt = *count + delta[2 * dim];
    __syncthreads();
    for (int j = threadIdx.x; j < dim; j += blockDim.x) {
        const double new_sum = sum[j] + delta[j];
        const double new_sq = sumsq[j] + delta[dim + j];
        const double m = new_sum / cnt;
        const double var = (new_sq - cnt * (m * m)) / fmax(cnt - 1.0, 1.0);
        sum[j] = new_sum;
        sumsq[j] = new_sq;
        mean[j] = m;
        stdv[j] = sqrt(fmax(var, epsilon));
    }
    if (threadIdx.x == 0) *count = cnt;
}

template <typename T>
__global__ void running_stats_normalize_kernel(const T *__restrict__ x, long long total, int dim,
                                               const double *__restrict__ mean,
                                               const double *__restrict__ stdv, double clip_lo,
                                               double clip_hi, float *__restrict__ out32,
                                               double *__restrict__ out64) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int j = (int)(t % dim);
        double v = ((double)x[t] - mean[j]) / (stdv[j] + 1e-15);             // (:163)
        v = fmin(fmax(v, clip_lo), clip_hi);                                 // np.clip (:164)
        if (out32) out32[t] = (float)v;
        if (out64) out64[t] = v;
    }
}

// reward = float(reward) * rescale ; then the clipping filter's truthiness semantics.
__global__ void reward_filter_kernel(const float *__restrict__ in, float *__restrict__ out,
                                     long long n, double rescale, int use_hi, double hi,
                                     int use_lo, double lo) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double r = (double)in[i] * rescale;
        if (use_hi) r = fmin(r, hi);
        if (use_lo) r = fmax(r, lo);
        out[i] = (float)r;
    }
}

// K2 (a4) — ObservationRescaleToSizeFilter: skimage.transform.resize(obs, out_shape, order=1,
// mode='reflect', anti_aliasing=False, preserve_range=True).astype('uint8')
// (filters/observation/observation_rescale_to_size_filter.py:62-79).  scikit-image is an un-vendored dependency
// (requirements.txt:9, >= 0.13); since 0.19 it executes this call as scipy.ndimage.zoom(order=1, mode='mirror',
// grid_mode=True), whose arithmetic (ndimage/src/ni_interpolation.c NI_ZoomShift) is followed here operation by
// operation in fp64 — no contraction (this file is compiled with -ffp-contract=off):
//   cc = (o + 0.5) * (in / out) - 0.5;  support floor(cc), floor(cc) + 1 mirrored about 0 and in - 1;
//   weights (1 - y, y), y = cc - floor(cc);  t = (v00 wr0) wc0 + (v01 wr0) wc1 + (v10 wr1) wc0 + (v11 wr1) wc1, in that order.
// oracle/filters.py restates the same and is pinned to scipy bit for bit on every pixel; the kernel equals the oracle
// bit for bit (tests/test_filter_classes.py).  scikit-image itself cannot be installed here.
__device__ __forceinline__ int mirror_index(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i > n - 1) i = 2 * (n - 1) - i;
    return i;
}
__global__ void resize_bilinear_u8_kernel(const unsigned char *__restrict__ in,
                                          unsigned char *__restrict__ out, int n, int H, int W, int C,
                                          int OH, int OW) {
    const long long total = (long long)n * OH * OW * C;
    const double sr = (double)H / (double)OH, sc = (double)W / (double)OW;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int ox = (int)((t / C) % OW);
        const int oy = (int)((t / ((long long)C * OW)) % OH);
        const int b = (int)(t / ((long long)C * OW * OH));
        const double r = ((double)oy + 0.5) * sr - 0.5;
        const double q = ((double)ox + 0.5) * sc - 0.5;
        const double fr = floor(r), fq = floor(q);
        const double yr = r - fr, yq = q - fq;
        const double wr0 = 1.0 - yr, wr1 = yr, wc0 = 1.0 - yq, wc1 = yq;
        const int r0 = mirror_index((int)fr, H), r1 = mirror_index((int)fr + 1, H);
        const int c0 = mirror_index((int)fq, W), c1 = mirror_index((int)fq + 1, W);
        const unsigned char *img = in + (size_t)b * H * W * C;
        const double p00 = img[((size_t)r0 * W + c0) * C + c], p01 = img[((size_t)r0 * W + c1) * C + c];
        const double p10 = img[((size_t)r1 * W + c0) * C + c], p11 = img[((size_t)r1 * W + c1) * C + c];
        double v = (p00 * wr0) * wc0;
        v = v + (p01 * wr0) * wc1;
        v = v + (p10 * wr1) * wc0;
        v = v + (p11 * wr1) * wc1;
        out[t] = (unsigned char)(int)v;                                    // astype('uint8')
    }
}

}  // namespace

namespace {
// MaxOverFramesAndFrameskipEnvWrapper.step's `np.max(self.observations_stack, axis=0)`
// (environments/gym_environment.py:148-175): element-wise maximum of the K newest raw frames of every env.
// in: u8 [n_env][K][frame_bytes] (the host repeats a frame when an episode ended before K frames were
// collected: max with a duplicate changes nothing), out: u8 [n_env][frame_bytes].  HBM-bound, 16-byte accesses.
__device__ __forceinline__ uint32_t max_u8x4(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        const uint32_t x = (a >> s) & 0xffu, y = (b >> s) & 0xffu;
        r |= (x > y ? x : y) << s;
    }
    return r;
}
__global__ void max_over_frames_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                       int K, long long frame_bytes) {
    const int e = blockIdx.y;
    const long long words = frame_bytes >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)e * K * frame_bytes);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)e * frame_bytes);
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < words;
         w += (long long)gridDim.x * blockDim.x) {
        uint4 m = src[w];
        for (int k = 1; k < K; ++k) {
            const uint4 v = src[(size_t)k * words + w];
            m.x = max_u8x4(m.x, v.x); m.y = max_u8x4(m.y, v.y); m.z = max_u8x4(m.z, v.z); m.w = max_u8x4(m.w, v.w);
        }
        dst[w] = m;
    }
}
}  // namespace

extern "C" {

int rlx_rgb_to_y_u8(const unsigned char *rgb, unsigned char *out, long long n_pixels,
                    double input_low, double input_high, void *stream) {
    RLX_REQUIRE(rgb && out, "rlx_rgb_to_y_u8: null pointer");
    RLX_REQUIRE(n_pixels > 0, "rlx_rgb_to_y_u8: empty input");
    RLX_REQUIRE(input_high > input_low,
                "The input observation space high values can be less or equal to the input "
                "observation space low values");   // reference message (to_uint8 :40-41)
    RLX_REQUIRE(((uintptr_t)rgb & 3) == 0 && ((uintptr_t)out & 3) == 0,
                "rlx_rgb_to_y_u8: buffers must be 4-byte aligned");
    RLX_LAUNCH((rgb_to_y_u8_kernel), rlx::grid_for(n_pixels / 4 + 1, kBlock), kBlock, 0,
                         rlx::as_stream(stream), rgb, out, n_pixels, input_low, input_high);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_running_stats_push(const void *samples, int samples_are_f64, long long n, int dim,
                           double *sum, double *sum_squares, double *count, double *mean,
                           double *std, double epsilon, void *stream) {
    RLX_REQUIRE(samples && sum && sum_squares && count && mean && std,
                "rlx_running_stats_push: null pointer");
    RLX_REQUIRE(n > 0 && dim > 0, "RunningStats input shape mismatch (n=%lld dim=%d)", n, dim);
    hipStream_t s = rlx::as_stream(stream);
    int grid = (dim + 63) / 64;
    if (samples_are_f64)
        RLX_LAUNCH((running_stats_push_kernel<double>), grid, 64, 0, s, static_cast<const double *>(samples), n, dim, sum, sum_squares, count, mean, std, epsilon);
    else
        RLX_LAUNCH((running_stats_push_kernel<float>), grid, 64, 0, s, static_cast<const float *>(samples), n, dim, sum, sum_squares, count, mean, std, epsilon);
    RLX_LAUNCH_CHECK();
    RLX_LAUNCH((running_stats_count_kernel), 1, 1, 0, s, count, (double)n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_running_stats_merge(const double *delta, int dim, double *sum, double *sum_squares,
                            double *count, double *mean, double *std, double epsilon, void *stream) {
    RLX_REQUIRE(delta && sum && sum_squares && count && mean && std, "rlx_running_stats_merge: null pointer");
    RLX_REQUIRE(dim > 0, "rlx_running_stats_merge: dim=%d", dim);
    RLX_LAUNCH((running_stats_merge_kernel), 1, 256, 0, rlx::as_stream(stream), delta, dim, sum, sum_squares, count,
                                                                      mean, std, epsilon);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_running_stats_normalize(const void *x, int x_is_f64, long long n, int dim,
                                const double *mean, const double *std, double clip_low,
                                double clip_high, float *out32, double *out64, void *stream) {
    RLX_REQUIRE(x && mean && std && (out32 || out64), "rlx_running_stats_normalize: null pointer");
    RLX_REQUIRE(n > 0 && dim > 0, "rlx_running_stats_normalize: empty input");
    const long long total = n * dim;
    hipStream_t s = rlx::as_stream(stream);
    if (x_is_f64)
        RLX_LAUNCH((running_stats_normalize_kernel<double>), rlx::grid_for(total, kBlock), kBlock, 0, s, static_cast<const double *>(x), total, dim, mean, std, clip_low, clip_high, out32, out64);
    else
        RLX_LAUNCH((running_stats_normalize_kernel<float>), rlx::grid_for(total, kBlock), kBlock, 0, s, static_cast<const float *>(x), total, dim, mean, std, clip_low, clip_high, out32, out64);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_max_over_frames_u8(const unsigned char *frames, unsigned char *out, int n_env, int n_frames,
                           long long frame_bytes, void *stream) {
    RLX_REQUIRE(frames && out, "rlx_max_over_frames_u8: null pointer");
    RLX_REQUIRE(n_env > 0 && n_frames >= 1 && frame_bytes > 0 && frame_bytes % 16 == 0 &&
                    (((uintptr_t)frames | (uintptr_t)out) & 15) == 0,
                "rlx_max_over_frames_u8: frames must be 16-byte aligned multiples of 16 bytes (got %lld)", frame_bytes);
    dim3 grid(rlx::grid_for(frame_bytes / 16, 256, 64), n_env);
    RLX_LAUNCH((max_over_frames_kernel), grid, 256, 0, rlx::as_stream(stream), frames, out, n_frames, frame_bytes);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_reward_filter(const float *rewards, float *out, long long n, double rescale_factor,
                      int has_clip, double clipping_low, double clipping_high, void *stream) {
    RLX_REQUIRE(rewards && out, "rlx_reward_filter: null pointer");
    RLX_REQUIRE(n > 0, "rlx_reward_filter: empty input");
    RLX_REQUIRE(rescale_factor != 0.0, "The reward rescale value can not be set to 0");
    RLX_REQUIRE(!has_clip || clipping_low <= clipping_high,
                "The reward clipping low must be lower than the reward clipping max");
    // `if self.clipping_high:` / `if self.clipping_low:` — a bound equal to 0 is not applied.
    const int use_hi = has_clip && clipping_high != 0.0;
    const int use_lo = has_clip && clipping_low != 0.0;
    RLX_LAUNCH((reward_filter_kernel), rlx::grid_for(n, kBlock), kBlock, 0, rlx::as_stream(stream), rewards, out, n, rescale_factor, use_hi, clipping_high, use_lo, clipping_low);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_resize_bilinear_u8(const unsigned char *in, unsigned char *out, int n, int H, int W, int C,
                           int OH, int OW, void *stream) {
    RLX_REQUIRE(in && out, "rlx_resize_bilinear_u8: null pointer");
    RLX_REQUIRE(n > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "rlx_resize_bilinear_u8: bad shape");
    const long long total = (long long)n * OH * OW * C;
    RLX_LAUNCH((resize_bilinear_u8_kernel), rlx::grid_for(total, kBlock), kBlock, 0, rlx::as_stream(stream), in, out, n, H, W, C, OH, OW);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
