// Narrow dense layers (N <= 16 output units): the value / policy / Q heads and their gradients.
//
// Replaces tf.layers.dense and tf.gradients for the network heads
// (rl_coach/architectures/tensorflow_components/layers.py:168-185; heads/v_head.py:43-48,
// heads/ppo_head.py:100-116, heads/q_head.py, heads/ddpg_actor_head.py:48-56, td3_v_head.py:40-60).
//
// A head has K = 256..512 inputs and 1..17 outputs on a minibatch of 32..256 rows: 0.05-2 MFLOP.
// As an MFMA GEMM that is one 32x32 tile with 26 of 32 columns wasted plus a split-K reduce launch;
// here it is plain fp32 FMA work laid out for coalescing:
//   forward : one workgroup per row m, a quarter of K per wave; lanes stride over k (x[m][k]
//             coalesced, W[k][0..N) from L1/L2), N accumulators per lane, wave64 butterfly reduction,
//             the four wave partials combined through LDS.
//   backward: ONE launch produces dW = x^T dz, db = 1^T dz and dx = dz W^T.  A workgroup owns 64
//             input features k (64 lanes) x 4 row groups; dz (with this layer's activation
//             derivative applied on the way in) sits in LDS, x[m][k] is read exactly once and used
//             for both dW and the activation derivative of the layer below (dx is written as the
//             lower layer's dz), the 4 row-group partials of dW are combined in a fixed order.
// Latency-bound by construction (a few hundred KB); the point is 2 launches per head instead of ~8.
#include "dense_small_body.hpp"

namespace {
using namespace rlx_small;

template <int NN>
__global__ void __launch_bounds__(256) dense_small_fwd_kernel(const SmallDense p) {
    __shared__ float part[4][NN];
    dense_small_fwd_row<NN>(p, blockIdx.x, blockIdx.y, part);
}

template <int NN>
__global__ void __launch_bounds__(256) dense_small_fwd_multi_kernel(const MultiFwd m) {
    const SmallDense &p = m.p[blockIdx.z];
    if ((int)blockIdx.x >= p.M || (int)blockIdx.y >= m.towers[blockIdx.z]) return;   // whole workgroup exits
    __shared__ float part[4][NN];
    dense_small_fwd_row<NN>(p, blockIdx.x, blockIdx.y, part);
}

// 16 features x 16 row groups per workgroup (see dense_small_bwd_body)
template <int NN>
__global__ void __launch_bounds__(256) dense_small_bwd_wide_kernel(const SmallDenseBwd p) {
    extern __shared__ float smem[];
    dense_small_bwd_body<NN, 16, 16>(p, blockIdx.x, blockIdx.y, smem);
}

template <int NN>
__global__ void __launch_bounds__(256) dense_small_bwd_multi_kernel(const MultiBwd m) {
    extern __shared__ float smem[];
    const SmallDenseBwd &p = m.p[blockIdx.z];
    if ((int)blockIdx.x * kKL >= p.K || (int)blockIdx.y >= m.towers[blockIdx.z]) return;
    dense_small_bwd_body<NN>(p, blockIdx.x, blockIdx.y, smem);
}

}  // namespace

extern "C" {

int rlx_dense_small_forward(const float *x, long long x_tower_stride, const float *w,
                            long long w_tower_stride, const float *bias, long long bias_tower_stride,
                            float *y, long long y_tower_stride, int towers, int M, int K, int N,
                            int activation, void *stream) {
    RLX_REQUIRE(x && w && y, "rlx_dense_small_forward: null pointer");
    RLX_REQUIRE(towers > 0 && M > 0 && K > 0 && N > 0 && N <= kMaxN,
                "rlx_dense_small_forward: need 1 <= N <= %d (got M=%d K=%d N=%d)", kMaxN, M, K, N);
    RLX_REQUIRE(activation >= 0 && activation <= 2, "rlx_dense_small_forward: unknown activation");
    SmallDense p{x, x_tower_stride, w, w_tower_stride, bias, bias_tower_stride, y, y_tower_stride,
                 M, K, N, activation};
    dim3 grid(M, towers);
    hipStream_t s = rlx::as_stream(stream);
    if (N <= 1) RLX_LAUNCH((dense_small_fwd_kernel<1>), grid, 256, 0, s, p);
    else if (N <= 4) RLX_LAUNCH((dense_small_fwd_kernel<4>), grid, 256, 0, s, p);
    else if (N <= 8) RLX_LAUNCH((dense_small_fwd_kernel<8>), grid, 256, 0, s, p);
    else RLX_LAUNCH((dense_small_fwd_kernel<16>), grid, 256, 0, s, p);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dense_small_backward(const float *x, long long x_tower_stride, const float *w,
                             long long w_tower_stride, const float *dy, long long dy_tower_stride,
                             const float *y, long long y_tower_stride, float *dw,
                             long long dw_tower_stride, float *db, long long db_tower_stride, float *dx,
                             long long dx_tower_stride, int towers, int M, int K, int N, int activation,
                             int lower_activation, void *stream) {
    RLX_REQUIRE(x && w && dy, "rlx_dense_small_backward: null pointer");
    RLX_REQUIRE(dw || dx, "rlx_dense_small_backward: nothing to produce");
    RLX_REQUIRE(towers > 0 && M > 0 && K > 0 && N > 0 && N <= kMaxN,
                "rlx_dense_small_backward: need 1 <= N <= %d (got M=%d K=%d N=%d)", kMaxN, M, K, N);
    RLX_REQUIRE(activation >= 0 && activation <= 2 && lower_activation >= 0 && lower_activation <= 2,
                "rlx_dense_small_backward: unknown activation");
    RLX_REQUIRE(activation == 0 || y, "rlx_dense_small_backward: the activation derivative needs y");
    const int NN = N <= 1 ? 1 : N <= 4 ? 4 : N <= 8 ? 8 : 16;
    const size_t smem = ((size_t)M * N + kRG * kKL * NN) * sizeof(float);
    RLX_REQUIRE(smem <= 64 * 1024, "rlx_dense_small_backward: batch %d x %d outputs exceeds the LDS budget", M, N);
    SmallDenseBwd p{x, x_tower_stride, w, w_tower_stride, dy, dy_tower_stride, y, y_tower_stride,
                    dw, dw_tower_stride, db, db_tower_stride, dx, dx_tower_stride,
                    M, K, N, activation, lower_activation};
    hipStream_t s = rlx::as_stream(stream);
    // 16 features x 16 row groups per workgroup: twice the workgroups of the 32 x 8 split the multi-problem kernel uses
    // and, up to 128 rows, every x load of a thread in one chunk (same-box A/B, profiles/r03_ab_candidates.txt: C4 +6 %,
    // C5 +4 %, C1 unchanged)
    dim3 wgrid((K + 15) / 16, towers);
    if (NN == 1) RLX_LAUNCH((dense_small_bwd_wide_kernel<1>), wgrid, 256, smem, s, p);
    else if (NN == 4) RLX_LAUNCH((dense_small_bwd_wide_kernel<4>), wgrid, 256, smem, s, p);
    else if (NN == 8) RLX_LAUNCH((dense_small_bwd_wide_kernel<8>), wgrid, 256, smem, s, p);
    else RLX_LAUNCH((dense_small_bwd_wide_kernel<16>), wgrid, 256, smem, s, p);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int nn_class(int N) { return N <= 1 ? 1 : N <= 4 ? 4 : N <= 8 ? 8 : 16; }

int rlx_dense_small_forward_multi(const rlx_small_dense_problem *problems_host, int n_problems, void *stream) {
    RLX_REQUIRE(problems_host && n_problems >= 1 && n_problems <= kMaxProblems,
                "rlx_dense_small_forward_multi: need 1..%d problems", kMaxProblems);
    MultiFwd m;
    m.n = n_problems;
    int max_m = 0, max_t = 0, nn = 1;
    for (int i = 0; i < n_problems; ++i) {
        const rlx_small_dense_problem &q = problems_host[i];
        RLX_REQUIRE(q.x && q.w && q.y, "rlx_dense_small_forward_multi: null pointer in problem %d", i);
        RLX_REQUIRE(q.towers > 0 && q.M > 0 && q.K > 0 && q.N > 0 && q.N <= kMaxN,
                    "rlx_dense_small_forward_multi: problem %d needs 1 <= N <= %d", i, kMaxN);
        RLX_REQUIRE(q.activation >= 0 && q.activation <= 2, "rlx_dense_small_forward_multi: unknown activation");
        m.p[i] = SmallDense{q.x, q.x_tower_stride, q.w, q.w_tower_stride, q.bias, q.bias_tower_stride,
                            q.y, q.y_tower_stride, q.M, q.K, q.N, q.activation};
        m.towers[i] = q.towers;
        if (q.M > max_m) max_m = q.M;
        if (q.towers > max_t) max_t = q.towers;
        if (nn_class(q.N) > nn) nn = nn_class(q.N);
    }
    dim3 grid(max_m, max_t, n_problems);
    hipStream_t s = rlx::as_stream(stream);
    if (nn == 1) RLX_LAUNCH((dense_small_fwd_multi_kernel<1>), grid, 256, 0, s, m);
    else if (nn == 4) RLX_LAUNCH((dense_small_fwd_multi_kernel<4>), grid, 256, 0, s, m);
    else if (nn == 8) RLX_LAUNCH((dense_small_fwd_multi_kernel<8>), grid, 256, 0, s, m);
    else RLX_LAUNCH((dense_small_fwd_multi_kernel<16>), grid, 256, 0, s, m);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rlx_dense_small_backward_multi(const rlx_small_dense_problem *problems_host, int n_problems, void *stream) {
    RLX_REQUIRE(problems_host && n_problems >= 1 && n_problems <= kMaxProblems,
                "rlx_dense_small_backward_multi: need 1..%d problems", kMaxProblems);
    MultiBwd m;
    m.n = n_problems;
    int max_kb = 0, max_t = 0, nn = 1;
    size_t smem = 0;
    for (int i = 0; i < n_problems; ++i) {
        const rlx_small_dense_problem &q = problems_host[i];
        RLX_REQUIRE(q.x && q.w && q.dy && (q.dw || q.dx),
                    "rlx_dense_small_backward_multi: null pointer in problem %d", i);
        RLX_REQUIRE(q.towers > 0 && q.M > 0 && q.K > 0 && q.N > 0 && q.N <= kMaxN,
                    "rlx_dense_small_backward_multi: problem %d needs 1 <= N <= %d", i, kMaxN);
        RLX_REQUIRE(q.activation >= 0 && q.activation <= 2 && q.lower_activation >= 0 && q.lower_activation <= 2,
                    "rlx_dense_small_backward_multi: unknown activation");
        RLX_REQUIRE(q.activation == 0 || q.y, "rlx_dense_small_backward_multi: the activation derivative needs y");
        m.p[i] = SmallDenseBwd{q.x, q.x_tower_stride, q.w, q.w_tower_stride, q.dy, q.dy_tower_stride,
                               q.activation ? q.y : nullptr, q.y_tower_stride, q.dw, q.dw_tower_stride, q.db,
                               q.db_tower_stride, q.dx, q.dx_tower_stride, q.M, q.K, q.N, q.activation,
                               q.lower_activation};
        m.towers[i] = q.towers;
        const int kb = (q.K + kKL - 1) / kKL;
        if (kb > max_kb) max_kb = kb;
        if (q.towers > max_t) max_t = q.towers;
        if (nn_class(q.N) > nn) nn = nn_class(q.N);
    }
    for (int i = 0; i < n_problems; ++i) {
        const size_t need = ((size_t)problems_host[i].M * problems_host[i].N + (size_t)kRG * kKL * nn) * sizeof(float);
        if (need > smem) smem = need;
    }
    RLX_REQUIRE(smem <= 64 * 1024, "rlx_dense_small_backward_multi: batch x outputs exceeds the LDS budget");
    dim3 grid(max_kb, max_t, n_problems);
    hipStream_t s = rlx::as_stream(stream);
    if (nn == 1) RLX_LAUNCH((dense_small_bwd_multi_kernel<1>), grid, 256, smem, s, m);
    else if (nn == 4) RLX_LAUNCH((dense_small_bwd_multi_kernel<4>), grid, 256, smem, s, m);
    else if (nn == 8) RLX_LAUNCH((dense_small_bwd_multi_kernel<8>), grid, 256, smem, s, m);
    else RLX_LAUNCH((dense_small_bwd_multi_kernel<16>), grid, 256, smem, s, m);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
