// Clipped-PPO head losses AND the backward pass of both heads in ONE launch (rlx_ppo_heads_loss_backward).
//
// Replaces, for the discrete Clipped-PPO network (value head N = 1, policy head N = A <= 16):
//   heads/ppo_head.py:52-116 + heads/v_head.py:43-52 + head.py:143-186   losses and tf.gradients to the head outputs
//   architecture.py:312-385                                                 the heads' share of accumulate_gradients
// i.e. rlx_ppo_discrete_value_losses followed by rlx_dense_small_backward_multi: the loss kernel is 1-3 us of work
// behind a launch boundary, and what it produces — dV [B] and dlogits [B, A] — is exactly what every workgroup of the
// heads' backward stages into LDS first.  Here each workgroup computes those rows itself (one row per thread, the
// per-sample arithmetic of losses_body.hpp: one definition, one rounding), so no grid-wide dependency appears; the
// workgroup that owns feature block 0 of a head also writes the gradients to memory, the head's loss scalars (same
// reduction tree as the stand-alone loss kernel: bit-identical scalars) and the likelihood ratios.
// Bound: latency (tens of workgroups, a few KB); one launch instead of two in a chain of ~20 per update.
#include "dense_small_body.hpp"
#include "losses_body.hpp"
#include <algorithm>

namespace {
using namespace rlx_small;
using namespace rlx_losses;

struct PpoHeadsBwdArgs {
    SmallDenseBwd value, policy;                 // .dy receives the gradients (written by feature block 0)
    const float *logits; long long ld;
    const int *actions;
    const float *advantages;
    const float *old_probs; long long ld_old;
    const float *v; const float *v_target;
    const float *clip_scale;                     // device scalar or null
    float *scalars;                              // [5]: surrogate, entropy, KL, policy total, value loss
    float *ratio_out, *clipped_out;              // [B] or null
    int *status;
    int batch, n_actions, red_threads;
    float clip_eps, beta, grad_scale;
};

template <int NN>
__global__ void __launch_bounds__(256) ppo_heads_loss_bwd_kernel(const PpoHeadsBwdArgs a) {
    extern __shared__ float smem[];
    __shared__ float red[3][256];
    const bool is_policy = blockIdx.z == 1;
    const SmallDenseBwd &p = is_policy ? a.policy : a.value;
    if ((int)blockIdx.x * kKL >= p.K) return;
    const bool writer = blockIdx.x == 0;
    const int b = threadIdx.x, B = a.batch;
    float *dz = smem;                                         // [B][N] of this head
    // the head losses and dz: runs inside dense_small_bwd_body, behind the issue of that body's weight and input loads
    auto stage = [&]() {
        if (is_policy) {
            const float clip = a.clip_scale ? a.clip_eps * *a.clip_scale : a.clip_eps;
            float l_sur = 0.f, l_ent = 0.f, l_kl = 0.f;
            if (b < B) {
                PpoRowTerms t;
                if (!ppo_discrete_row(a.logits + (size_t)b * a.ld, a.old_probs + (size_t)b * a.ld_old, a.actions[b],
                                      a.n_actions, a.advantages[b], clip, a.beta, a.grad_scale, B,
                                      dz + (size_t)b * a.n_actions, (writer && a.ratio_out) ? a.ratio_out + b : nullptr,
                                      (writer && a.clipped_out) ? a.clipped_out + b : nullptr, t)) {
                    if (writer) atomicOr(a.status, 1);
                    for (int j = 0; j < a.n_actions; ++j) dz[(size_t)b * a.n_actions + j] = 0.f;
                } else {
                    l_sur = t.sur; l_ent = t.ent; l_kl = t.kl;
                }
            }
            if (writer) {
                block_sum3_first(l_sur, l_ent, l_kl, red, a.red_threads);
                if (threadIdx.x == 0) ppo_discrete_scalars(l_sur, l_ent, l_kl, a.beta, B, a.scalars);
            }
        } else {
            float local = 0.f;
            if (b < B) {                                       // VHead: MSE(target, V), loss weight 1 (head.py:172-181)
                const float w = 1.f;
                const float e = a.v[b] - a.v_target[b];
                float l, g;
                regression_terms(e, 0, l, g);
                dz[b] = regression_grad(a.grad_scale, w, g, B);
                local = w * l;
            }
            if (writer) {
                const float s = block_sum_first(local, red[0], a.red_threads);
                if (threadIdx.x == 0) a.scalars[4] = s / (float)B;
            }
        }
        __syncthreads();
        if (writer && p.dy) {                                  // the head-output gradients, for whoever reads y.grad
            float *dy = const_cast<float *>(p.dy);
            for (int i = threadIdx.x; i < B * p.N; i += 256) dy[i] = dz[i];
        }
    };
    dense_small_bwd_body<NN, kKL, kRG, true>(p, blockIdx.x, 0, smem, stage);
}

}  // namespace

extern "C" {

int rlx_ppo_heads_loss_backward(const rlx_small_dense_problem *value_head, const rlx_small_dense_problem *policy_head,
                                const float *values, const float *value_targets, const float *logits,
                                const int *actions, const float *advantages, const float *old_probs, long long ld_old,
                                int batch, float clip_epsilon, const float *clip_scale, float beta_entropy,
                                float grad_scale, float *scalars, float *likelihood_ratio,
                                float *clipped_likelihood_ratio, int *status, void *stream) {
    RLX_REQUIRE(value_head && policy_head && values && value_targets && logits && actions && advantages &&
                old_probs && scalars && status, "rlx_ppo_heads_loss_backward: null pointer");
    const rlx_small_dense_problem *q[2] = {value_head, policy_head};
    SmallDenseBwd d[2];
    for (int i = 0; i < 2; ++i) {
        RLX_REQUIRE(q[i]->x && q[i]->w && (q[i]->dw || q[i]->dx), "rlx_ppo_heads_loss_backward: null pointer in head %d", i);
        RLX_REQUIRE(q[i]->towers == 1 && q[i]->M == batch && q[i]->K > 0 && q[i]->activation == 0,
                    "rlx_ppo_heads_loss_backward: head %d must be one linear tower over the %d rows of the batch", i, batch);
        RLX_REQUIRE(q[i]->lower_activation >= 0 && q[i]->lower_activation <= 2, "rlx_ppo_heads_loss_backward: unknown activation");
        d[i] = SmallDenseBwd{q[i]->x, q[i]->x_tower_stride, q[i]->w, q[i]->w_tower_stride, q[i]->dy, q[i]->dy_tower_stride,
                             nullptr, 0, q[i]->dw, q[i]->dw_tower_stride, q[i]->db, q[i]->db_tower_stride, q[i]->dx,
                             q[i]->dx_tower_stride, q[i]->M, q[i]->K, q[i]->N, 0, q[i]->lower_activation};
    }
    RLX_REQUIRE(value_head->N == 1 && policy_head->N >= 2 && policy_head->N <= kMaxN,
                "rlx_ppo_heads_loss_backward: value head N = 1, policy head 2 <= N <= %d", kMaxN);
    RLX_REQUIRE(batch >= 1 && batch <= 256, "rlx_ppo_heads_loss_backward: 1 <= batch <= 256 (one row per thread)");
    int red_threads = 64;
    while (red_threads < batch) red_threads <<= 1;       // block_for() of the stand-alone loss kernels
    const int nn = policy_head->N <= 4 ? 4 : (policy_head->N <= 8 ? 8 : 16);
    const int kb = (std::max(value_head->K, policy_head->K) + kKL - 1) / kKL;
    const size_t smem = ((size_t)batch * policy_head->N + (size_t)kRG * kKL * nn) * sizeof(float);
    RLX_REQUIRE(smem <= 64 * 1024, "rlx_ppo_heads_loss_backward: batch x actions exceeds the LDS budget");
    PpoHeadsBwdArgs a{d[0], d[1], logits, policy_head->N, actions, advantages, old_probs, ld_old, values, value_targets,
                      clip_scale, scalars, likelihood_ratio, clipped_likelihood_ratio, status, batch, policy_head->N,
                      red_threads, clip_epsilon, beta_entropy, grad_scale};
    dim3 grid(kb, 1, 2);
    hipStream_t s = rlx::as_stream(stream);
    if (nn == 4) RLX_LAUNCH((ppo_heads_loss_bwd_kernel<4>), grid, 256, smem, s, a);
    else if (nn == 8) RLX_LAUNCH((ppo_heads_loss_bwd_kernel<8>), grid, 256, smem, s, a);
    else RLX_LAUNCH((ppo_heads_loss_bwd_kernel<16>), grid, 256, smem, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // extern "C"
