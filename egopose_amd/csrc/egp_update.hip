// egp_update.hip -- the element-wise tail of a PPO epoch: both losses with their gradients w.r.t. the nets' outputs in one
// launch, and gradient-norm clip + Adam over flat parameter buffers in two.
//
//   k_ppo_loss   agents/agent_pg.py:19-26 (critic MSE), agents/agent_ppo.py:58-65 (clipped surrogate) over
//                core/distributions.py:6-25 / utils/math.py:14-17 (diagonal Gaussian log-density), plus what autograd would
//                send back to `values_pred` and `action_mean`
//   k_sqnorm     torch.nn.utils.clip_grad_norm_'s total norm (agents/agent_ppo.py:53-56)
//   k_adam       torch.optim.Adam.step for every parameter group of both optimizers (agent_ppo.py:24-30)
//
// All three are one pass over their operands (HBM-bound, a few MB): what they replace is ~45 launch-bound library kernels per
// epoch. Reductions are float64 in a fixed order: per-workgroup partials, summed in index order by a one-workgroup launch.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "egp_internal.hpp"

namespace {

inline int after_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        egp::set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EGP_E_HIP;
    }
    return EGP_OK;
}

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_GROUP = 16;                 // lanes per policy row
constexpr int LOSS_MAX_BLOCKS = 1024;
constexpr int LOSS_MAX_ACT = 256;              // action width limit (d_log_std accumulators per lane: act_dim / 16)

struct LossArgs {
    int n, n_pol, act_dim;
    const long long *rows;
    const float *pred, *returns, *mean, *actions, *log_std, *adv;
    long ld_mean, ld_act, ld_dmean;
    float *fixed_logp;
    int write_fixed;
    double clip_eps, inv_n_val, inv_n_exp;
    float *d_pred, *d_mean, *d_log_std;
    double *losses;
    double *part;              // [gridDim.x][2 + act_dim]: value-loss sum, surrogate sum, d_log_std sums
    unsigned *counter;
    int pol_blocks;            // blocks [0, pol_blocks) take policy rows, the rest value elements
};

__device__ __forceinline__ double block_sum(double v, double *s_red) {
    // fixed tree over the 256 threads: lane butterflies inside a wave, then the four wave sums in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(LOSS_BLOCK) void k_ppo_loss(LossArgs a) {
    __shared__ double s_red[4];
    __shared__ float s_grp[LOSS_BLOCK / LOSS_GROUP][LOSS_MAX_ACT];        // d_log_std sums of the block's 16 row groups
    const int t = threadIdx.x;
    double v_sum = 0.0, s_sum = 0.0;
    const bool want_dls = a.d_log_std != nullptr;
    if (want_dls) {
        for (int j = t; j < (LOSS_BLOCK / LOSS_GROUP) * LOSS_MAX_ACT; j += LOSS_BLOCK) (&s_grp[0][0])[j] = 0.f;
        __syncthreads();
    }
    if ((int)blockIdx.x >= a.pol_blocks) {
        // ---- critic: (pred - returns)^2, d_pred = 2 (pred - returns) / n_val
        const int vb = blockIdx.x - a.pol_blocks, nvb = gridDim.x - a.pol_blocks;
        const float two_inv = (float)(2.0 * a.inv_n_val);
        for (long i = (long)vb * LOSS_BLOCK + t; i < a.n; i += (long)nvb * LOSS_BLOCK) {
            const float d = a.pred[i] - a.returns[i];
            v_sum += (double)d * (double)d;
            if (a.d_pred) a.d_pred[i] = two_inv * d;
        }
    } else {
        // ---- actor: a 16-lane group per row, lane l takes action dimensions l, l + 16, ...
        const int grp = t / LOSS_GROUP, l = t % LOSS_GROUP;
        constexpr int GPB = LOSS_BLOCK / LOSS_GROUP;
        constexpr int MAXD = LOSS_MAX_ACT / LOSS_GROUP;
        float dls[MAXD];
#pragma unroll
        for (int k = 0; k < MAXD; ++k) dls[k] = 0.f;
        const float lo = (float)(1.0 - a.clip_eps), hi = (float)(1.0 + a.clip_eps);
        const float neg_inv = (float)(-a.inv_n_exp);
        const float half_log_2pi = 0.91893853320467274178f;
        for (long r0 = (long)blockIdx.x * GPB; r0 < a.n_pol; r0 += (long)a.pol_blocks * GPB) {
            const long i = r0 + grp;
            const bool ok = i < a.n_pol;
            const long s = ok ? (a.rows ? (long)a.rows[i] : i) : 0;
            float z[MAXD], istd[MAXD];
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < MAXD; ++k) {
                const int j = l + LOSS_GROUP * k;
                z[k] = 0.f; istd[k] = 0.f;
                if (ok && j < a.act_dim) {
                    const float ls = a.log_std[j];
                    istd[k] = expf(-ls);
                    z[k] = (a.actions[s * a.ld_act + j] - a.mean[i * a.ld_mean + j]) * istd[k];
                    // normal_log_density per element: -z^2 / 2 - 0.5 log(2 pi) - log_std
                    acc += -0.5f * z[k] * z[k] - half_log_2pi - ls;
                }
            }
#pragma unroll
            for (int off = LOSS_GROUP / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, LOSS_GROUP);
            const float logp = acc;
            float fixed = logp;
            if (ok) {
                if (a.write_fixed) { if (l == 0) a.fixed_logp[i] = logp; }
                else fixed = a.fixed_logp[i];
            }
            const float adv = ok ? a.adv[s] : 0.f;
            const float ratio = expf(logp - fixed);
            const float clamped = fminf(fmaxf(ratio, lo), hi);
            const float surr1 = ratio * adv, surr2 = clamped * adv;
            const float surr = fminf(surr1, surr2);
            // d surr / d ratio with torch.min's tie rule (half each way) and clamp's sub-gradient (1 inside [lo, hi], ends included)
            const float inside = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
            float g_ratio;
            if (surr1 < surr2) g_ratio = adv;
            else if (surr1 > surr2) g_ratio = adv * inside;
            else g_ratio = 0.5f * adv + 0.5f * adv * inside;
            // loss = -inv_n_exp * sum surr;  d loss / d logp = -inv_n_exp * g_ratio * ratio
            const float g_logp = ok ? neg_inv * g_ratio * ratio : 0.f;
            if (ok && l == 0) s_sum += (double)surr;
#pragma unroll
            for (int k = 0; k < MAXD; ++k) {
                const int j = l + LOSS_GROUP * k;
                if (ok && j < a.act_dim) {
                    // d logp / d mean_j = z_j / std_j;  d logp / d log_std_j = z_j^2 - 1
                    a.d_mean[i * a.ld_dmean + j] = g_logp * z[k] * istd[k];
                    dls[k] += g_logp * (z[k] * z[k] - 1.f);
                }
            }
        }
        if (want_dls) {
#pragma unroll
            for (int k = 0; k < MAXD; ++k) {
                const int j = l + LOSS_GROUP * k;
                if (j < a.act_dim) s_grp[grp][j] = dls[k];
            }
        }
    }
    const double vs = block_sum(v_sum, s_red);
    const double ss = block_sum(s_sum, s_red);
    double *mine = a.part + (long)blockIdx.x * (2 + a.act_dim);
    if (t == 0) { mine[0] = vs; mine[1] = ss; }
    if (want_dls) {
        __syncthreads();
        for (int j = t; j < a.act_dim; j += LOSS_BLOCK) {        // the 16 row groups in index order
            double d = 0.0;
            for (int g = 0; g < LOSS_BLOCK / LOSS_GROUP; ++g) d += (double)s_grp[g][j];
            mine[2 + j] = d;
        }
    }
}

// The partials -> losses[2] (and d_log_std): one workgroup, index order. A launch of its own rather than "whoever finishes
// last": a device-scope release / acquire per workgroup writes back and invalidates the XCD's whole L2 on this chip -- the
// loss kernel took 170 us with that pattern and 1 100 workgroups, against the ~30 us its 84 MB of traffic need.
__global__ __launch_bounds__(LOSS_BLOCK) void k_ppo_loss_final(LossArgs a, int nb) {
    __shared__ double s_red[4];
    const int t = threadIdx.x, stride = 2 + a.act_dim;
    double v = 0.0, s = 0.0;
    for (int b = t; b < nb; b += LOSS_BLOCK) {               // 256 interleaved slices, combined by block_sum's fixed tree
        const double *p = a.part + (long)b * stride;
        v += p[0];
        s += p[1];
    }
    v = block_sum(v, s_red);
    s = block_sum(s, s_red);
    if (t == 0) {
        a.losses[0] = v * a.inv_n_val;
        a.losses[1] = -s * a.inv_n_exp;
    }
    if (a.d_log_std) {
        for (int j = t; j < a.act_dim; j += LOSS_BLOCK) {
            double d = 0.0;
            for (int b = 0; b < a.pol_blocks; ++b) d += a.part[(long)b * stride + 2 + j];
            a.d_log_std[j] = (float)d;
        }
    }
}

// ------------------------------------------------------------------------------------------------ clip + Adam
constexpr int ADAM_BLOCK = 256;
constexpr int NORM_BLOCKS = 256;               // partials per clip group

struct AdamSeg {
    long begin, end;
    double lr, beta1, beta2, eps, wd, bias1, bias2, max_norm;
    int clip_group;
};
struct AdamArgs {
    int n_seg;
    AdamSeg seg[EGP_ADAM_MAX_SEGMENTS];
    double *part;              // [EGP_ADAM_MAX_SEGMENTS + 1][NORM_BLOCKS] squared-norm partials per clip group
    double *norms_out;         // [EGP_ADAM_MAX_SEGMENTS + 1] or null
};

// squared 2-norm of the gradient of every clip group: blockIdx.y = group (1-based), NORM_BLOCKS partials each
template <typename TG>
__global__ __launch_bounds__(ADAM_BLOCK) void k_sqnorm(AdamArgs a, const TG *__restrict__ grad) {
    __shared__ double s_red[4];
    const int group = blockIdx.y + 1;
    double acc = 0.0;
    for (int s = 0; s < a.n_seg; ++s) {
        if (a.seg[s].clip_group != group) continue;
        for (long i = a.seg[s].begin + (long)blockIdx.x * ADAM_BLOCK + threadIdx.x; i < a.seg[s].end; i += (long)NORM_BLOCKS * ADAM_BLOCK) {
            const double g = (double)grad[i];
            acc += g * g;
        }
    }
    const double tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) a.part[(long)group * NORM_BLOCKS + blockIdx.x] = tot;
}

template <typename TP, typename TG>
__global__ __launch_bounds__(ADAM_BLOCK) void k_adam(AdamArgs a, const TG *__restrict__ grad, TP *__restrict__ param, TP *__restrict__ m,
                                                     TP *__restrict__ v, float *__restrict__ shadow, long total) {
    __shared__ double s_red[4];
    __shared__ double s_coef[EGP_ADAM_MAX_SEGMENTS + 1];
    // every block derives the clip coefficients itself from the partials (the same operations in the same order everywhere)
    for (int g = 1; g <= EGP_ADAM_MAX_SEGMENTS; ++g) {
        bool used = false;
        for (int s = 0; s < a.n_seg; ++s) used |= a.seg[s].clip_group == g;
        if (!used) { if (threadIdx.x == 0) s_coef[g] = 1.0; continue; }       // (uniform branch: `used` depends on kernel arguments only)
        double max_norm = 0.0;
        for (int s = 0; s < a.n_seg; ++s) if (a.seg[s].clip_group == g) max_norm = a.seg[s].max_norm;
        const double p = threadIdx.x < NORM_BLOCKS ? a.part[(long)g * NORM_BLOCKS + threadIdx.x] : 0.0;
        const double norm = sqrt(block_sum(p, s_red));
        if (threadIdx.x == 0) {
            s_coef[g] = fmin(1.0, max_norm / (norm + 1e-6));                  // clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max = 1)
            if (a.norms_out && blockIdx.x == 0) a.norms_out[g] = norm;
        }
    }
    if (threadIdx.x == 0) s_coef[0] = 1.0;
    __syncthreads();
    for (long i = (long)blockIdx.x * ADAM_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * ADAM_BLOCK) {
        int s = 0;
        while (s < a.n_seg && !(i >= a.seg[s].begin && i < a.seg[s].end)) ++s;
        if (s == a.n_seg) continue;                                           // an element outside every stepped segment
        const AdamSeg &sg = a.seg[s];
        const TP coef = (TP)s_coef[sg.clip_group];
        const TP p = param[i];
        TP g = (TP)grad[i] * coef;
        if (sg.wd != 0.0) g += (TP)sg.wd * p;
        const TP b1 = (TP)sg.beta1, b2 = (TP)sg.beta2;
        const TP mm = m[i] + (g - m[i]) * ((TP)1 - b1);                        // lerp(exp_avg, grad, 1 - beta1)
        const TP vv = b2 * v[i] + ((TP)1 - b2) * g * g;
        const TP step_size = (TP)(sg.lr / sg.bias1);
        const TP denom = (TP)sqrt((double)vv) / (TP)sqrt(sg.bias2) + (TP)sg.eps;
        const TP pn = p - step_size * (mm / denom);
        m[i] = mm;
        v[i] = vv;
        param[i] = pn;
        if (shadow) shadow[i] = (float)pn;
    }
}

template <typename TP, typename TG>
int adam_launch(int n_seg, const egp_adam_segment *seg, const TG *grad, TP *param, TP *m, TP *v, float *shadow, void *workspace,
                double *norms_out, void *stream) {
    EGP_REQUIRE(n_seg >= 0 && n_seg <= EGP_ADAM_MAX_SEGMENTS, "n_seg out of range");
    if (n_seg == 0) return EGP_OK;
    EGP_REQUIRE(seg && grad && param && m && v && workspace, "NULL pointer");
    AdamArgs a{};
    a.n_seg = n_seg;
    a.part = (double *)workspace;
    a.norms_out = norms_out;
    long lo = -1, hi = 0;
    int max_group = 0;
    for (int s = 0; s < n_seg; ++s) {
        const egp_adam_segment &q = seg[s];
        EGP_REQUIRE(q.begin >= 0 && q.end >= q.begin, "bad segment range");
        EGP_REQUIRE(q.clip_group >= 0 && q.clip_group <= EGP_ADAM_MAX_SEGMENTS, "clip_group out of range");
        EGP_REQUIRE(q.clip_group == 0 || q.max_norm > 0.0, "a clipped segment needs max_norm > 0");
        EGP_REQUIRE(q.bias1 > 0.0 && q.bias2 > 0.0, "bias corrections must be positive (step >= 1)");
        a.seg[s] = AdamSeg{(long)q.begin, (long)q.end, q.lr, q.beta1, q.beta2, q.eps, q.weight_decay, q.bias1, q.bias2, q.max_norm, q.clip_group};
        if (q.end > q.begin) {
            lo = lo < 0 ? (long)q.begin : (q.begin < lo ? (long)q.begin : lo);
            hi = q.end > hi ? (long)q.end : hi;
        }
        max_group = q.clip_group > max_group ? q.clip_group : max_group;
    }
    if (hi <= 0 || lo < 0) return EGP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (max_group > 0) {
        k_sqnorm<TG><<<dim3(NORM_BLOCKS, max_group), dim3(ADAM_BLOCK), 0, st>>>(a, grad);
        int rc = after_launch("k_sqnorm");
        if (rc != EGP_OK) return rc;
    }
    const long blocks = std::min<long>((hi + ADAM_BLOCK - 1) / ADAM_BLOCK, 2048);
    k_adam<TP, TG><<<dim3((unsigned)blocks), dim3(ADAM_BLOCK), 0, st>>>(a, grad, param, m, v, shadow, hi);
    return after_launch("k_adam");
}

}  // namespace

extern "C" {

int64_t egp_ppo_loss_workspace_bytes(int32_t n, int32_t n_pol, int32_t act_dim) {
    (void)n; (void)n_pol;
    return (int64_t)(2 * LOSS_MAX_BLOCKS) * (2 + (act_dim > 0 ? act_dim : 0)) * (int64_t)sizeof(double) + 64;
}

int egp_ppo_loss_f32(const egp_ppo_loss_desc *d, void *stream) {
    EGP_REQUIRE(d, "descriptor is NULL");
    EGP_REQUIRE(d->n >= 0 && d->n_pol >= 0 && d->act_dim >= 0 && d->act_dim <= LOSS_MAX_ACT, "bad sizes (act_dim <= 256)");
    EGP_REQUIRE(d->losses && d->workspace, "NULL losses / workspace");
    EGP_REQUIRE(d->n == 0 || (d->pred && d->returns), "NULL critic operand");
    EGP_REQUIRE(d->n_pol == 0 || (d->mean && d->actions && d->log_std && d->adv && d->fixed_logp && d->d_mean), "NULL actor operand");
    LossArgs a{};
    a.n = d->n; a.n_pol = d->n_pol; a.act_dim = d->act_dim;
    a.rows = (const long long *)d->rows;
    a.pred = d->pred; a.returns = d->returns; a.mean = d->mean; a.actions = d->actions; a.log_std = d->log_std; a.adv = d->adv;
    a.ld_mean = d->ld_mean; a.ld_act = d->ld_act; a.ld_dmean = d->ld_dmean;
    a.fixed_logp = d->fixed_logp; a.write_fixed = d->write_fixed;
    a.clip_eps = d->clip_eps; a.inv_n_val = d->inv_n_val; a.inv_n_exp = d->inv_n_exp;
    a.d_pred = d->d_pred; a.d_mean = d->d_mean; a.d_log_std = d->d_log_std;
    a.losses = d->losses;
    a.counter = nullptr;
    a.part = (double *)d->workspace;                        // [workgroups][2 + act_dim] partial sums
    constexpr int GPB = LOSS_BLOCK / LOSS_GROUP;
    long pb = ((long)d->n_pol + 4 * GPB - 1) / (4 * GPB);          // ~4 passes of 16 rows per block
    pb = pb < 1 ? (d->n_pol > 0 ? 1 : 0) : (pb > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : pb);
    long vb = ((long)d->n + 4 * LOSS_BLOCK - 1) / (4 * LOSS_BLOCK);
    vb = vb < 1 ? (d->n > 0 ? 1 : 0) : (vb > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : vb);
    if (pb + vb == 0) {
        // nothing to sum: both losses are zero
        hipError_t e = hipMemsetAsync(d->losses, 0, 2 * sizeof(double), (hipStream_t)stream);
        if (e != hipSuccess) { egp::set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
        return EGP_OK;
    }
    a.pol_blocks = (int)pb;
    k_ppo_loss<<<dim3((unsigned)(pb + vb)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream>>>(a);
    int rc = after_launch("k_ppo_loss");
    if (rc != EGP_OK) return rc;
    k_ppo_loss_final<<<dim3(1), dim3(LOSS_BLOCK), 0, (hipStream_t)stream>>>(a, (int)(pb + vb));
    return after_launch("k_ppo_loss_final");
}

int64_t egp_adam_workspace_bytes(void) { return (int64_t)(EGP_ADAM_MAX_SEGMENTS + 1) * NORM_BLOCKS * (int64_t)sizeof(double); }

int egp_adam_step_f32(int32_t n_seg, const egp_adam_segment *seg, const float *grad, float *param, float *exp_avg, float *exp_avg_sq,
                      void *workspace, double *norms_out, void *stream) {
    return adam_launch<float, float>(n_seg, seg, grad, param, exp_avg, exp_avg_sq, nullptr, workspace, norms_out, stream);
}
int egp_adam_step_f64(int32_t n_seg, const egp_adam_segment *seg, const float *grad, double *param, double *exp_avg, double *exp_avg_sq,
                      float *shadow, void *workspace, double *norms_out, void *stream) {
    return adam_launch<double, float>(n_seg, seg, grad, param, exp_avg, exp_avg_sq, shadow, workspace, norms_out, stream);
}
int egp_adam_step_f64g(int32_t n_seg, const egp_adam_segment *seg, const double *grad, double *param, double *exp_avg, double *exp_avg_sq,
                       void *workspace, double *norms_out, void *stream) {
    return adam_launch<double, double>(n_seg, seg, grad, param, exp_avg, exp_avg_sq, nullptr, workspace, norms_out, stream);
}

}  // extern "C"
