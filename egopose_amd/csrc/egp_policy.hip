// Rollout-time policy step in ONE launch:
//   x = [video context row (fp32) | filtered observation (fp64 -> fp32)]  ->  MLP (hidden layers + activation)
//   -> action_mean (Linear) -> action = mean + exp(log_std) * noise  (float32 arithmetic, stored as float64 for
//   the engine), i.e. PolicyGaussian.select_action of models/policy_gaussian.py:19-27 over models/mlp.py:5-25 with
//   the VideoStateNet concatenation of models/video_state_net.py:37-43, for all envs of a group at once.
// During a rollout this chain is ~13 launch-bound torch ops per tick (3 GEMMs of 512 rows, gather, cat, casts);
// here a 4-row tile walks the layers through LDS while the transposed weights stream from L2.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include "egp_internal.hpp"
#include "egp_filter_dev.hpp"

namespace {

// The filter's apply pass folded into the policy step (egp_policy_gaussian_filter_f32): the observation rows of the tile are
// formed, normalised with the running statistics merged from the tick's tile partials (exactly k_zf_apply's arithmetic: same
// merge order, same expression, float64) and written to next_states[k] / states[k + 1] on their way into the MLP's input -- one
// launch and its dependent memory round trips less on the chain filter -> policy -> env-step of every tick without a reset.
struct PolFilter {
    egp::ZfSrc<double> src;          // rows of the group: observation from (qpos, qvel)
    const double *st_in; double *st_out;
    const double *ws; int n_tiles;
    double clip;
    double *y, *y2;                  // [n][dim] each (y2 may be NULL)
};

constexpr int POL_ROWS = 4;          // rows per workgroup
constexpr int POL_MAX_LAYERS = 8;

struct PolLayers {
    const float *wt[POL_MAX_LAYERS];     // W^T, [in][round_up(out, 4)] row-major, 16-byte aligned
    const float *bias[POL_MAX_LAYERS];
    int in_dim[POL_MAX_LAYERS], out_dim[POL_MAX_LAYERS];
    int n;                                // hidden layers + the output layer
};

__device__ __forceinline__ float pol_act(float v, int kind) {
    if (kind == 1) return fmaxf(v, 0.0f);                 // relu
    if (kind == 2) return 1.0f / (1.0f + expf(-v));       // sigmoid
    return tanhf(v);                                      // tanh
}

// Thread t of a layer pass owns 4 consecutive outputs (one 16-byte weight load per input feature) for one slice of
// the input features; the slices' partial sums meet in LDS. With 512 threads: 300 outputs -> 75 columns x 6 slices.
// `stage_src` / `stage_dst` (optional): the tick's flag slab. The rollout stages the integer flags and context-row indices of a
// tick in pinned host memory; instead of a copy-engine transfer in front of this kernel (one more dependent operation, ~6 us,
// on the chain filter -> policy -> env-step of every tick) the workgroups copy the slab to its device copy themselves -- the
// kernels that run after the env-step (reward, filter) read it there -- and take their own rows' indices (`t_idx`, which then
// points into the pinned slab) with ONE load per row.
template <bool FILTER>
__global__ void k_policy_gaussian(const float *__restrict__ ctx_rows, long ctx_row_stride, int ctx_dim,
                                  const long long *__restrict__ t_idx, const double *__restrict__ state, int state_dim, int n,
                                  PolLayers L, int act_kind, int kmax, int part_elems, const float *__restrict__ log_std,
                                  const float *__restrict__ noise, double *__restrict__ action, float *__restrict__ mean_out,
                                  const unsigned *__restrict__ stage_src, unsigned *__restrict__ stage_dst, int stage_words,
                                  PolFilter F) {
    extern __shared__ float4 s_act[];     // cur[kmax] | nxt[kmax] | part[part_elems]   (one float per row of the tile) [| mean, inv: 2 dim doubles]
    __shared__ long long s_ti[POL_ROWS];
    double *s_ms = reinterpret_cast<double *>(s_act + 2 * kmax + part_elems);
    if constexpr (FILTER) {               // k_zf_apply's first phase: the merged statistics, every workgroup for itself
        const int dim = state_dim;
        for (int c = threadIdx.x; c < dim; c += blockDim.x) {
            double cnt, mean, S;
            egp::zf_merge_column(dim, F.n_tiles, F.ws, F.st_in, c, cnt, mean, S);
            if (blockIdx.x == 0) {
                F.st_out[1 + c] = mean;
                F.st_out[1 + dim + c] = S;
                if (c == 0) F.st_out[0] = cnt;
            }
            const double var = cnt > 1.0 ? S / (cnt - 1.0) : mean * mean;
            s_ms[c] = mean;
            s_ms[dim + c] = 1.0 / (sqrt(var) + 1e-8);
        }
    }
    float4 *cur = s_act, *nxt = s_act + kmax, *part = s_act + 2 * kmax;
    const int r0 = blockIdx.x * POL_ROWS;
    const int in0 = ctx_dim + state_dim;
    if (threadIdx.x < POL_ROWS) s_ti[threadIdx.x] = r0 + (int)threadIdx.x < n ? t_idx[r0 + threadIdx.x] : 0;
    if (stage_src)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stage_words; i += gridDim.x * blockDim.x) stage_dst[i] = stage_src[i];
    __syncthreads();
    for (int k = threadIdx.x; k < in0; k += blockDim.x) {
        float v[POL_ROWS];
#pragma unroll
        for (int r = 0; r < POL_ROWS; ++r) {
            const int row = r0 + r;
            if (row >= n) { v[r] = 0.0f; continue; }
            if (k < ctx_dim) {
                v[r] = ctx_rows[(long)row * ctx_row_stride + (long)s_ti[r] * ctx_dim + k];
            } else if constexpr (FILTER) {            // k_zf_apply's second phase for this element
                const int c = k - ctx_dim;
                double x = ((double)F.src.at(row, c) - s_ms[c]) * s_ms[state_dim + c];
                if (F.clip > 0.0) x = fmin(fmax(x, -F.clip), F.clip);
                const long e = (long)row * state_dim + c;
                F.y[e] = x;
                if (F.y2) F.y2[e] = x;
                v[r] = (float)x;
            } else {
                v[r] = (float)state[(long)row * state_dim + (k - ctx_dim)];
            }
        }
        cur[k] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int l = 0; l < L.n; ++l) {
        const int in = L.in_dim[l], out = L.out_dim[l];
        const int ldw = (out + 3) & ~3;                      // weight rows are padded to 4 floats
        const int ncol = ldw >> 2;
        const float4 *__restrict__ wt = reinterpret_cast<const float4 *>(L.wt[l]);
        const bool last = l == L.n - 1;
        const int cols = ncol < (int)blockDim.x ? ncol : (int)blockDim.x;
        const int G = blockDim.x / cols;                     // input-feature slices
        const int kc = (in + G - 1) / G;
        const int col_l = threadIdx.x % cols, grp = threadIdx.x / cols;
        for (int c0 = 0; c0 < ncol; c0 += cols) {
            const int col = c0 + col_l;
            if (grp < G && col < ncol) {
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;      // a_o = output o of the column, .xyzw = rows
                const int k0 = grp * kc, k1 = min(in, k0 + kc);
                int k = k0;
#define POL_FMA(W, X)                                                                                                     \
    a0.x = fmaf(W.x, X.x, a0.x); a0.y = fmaf(W.x, X.y, a0.y); a0.z = fmaf(W.x, X.z, a0.z); a0.w = fmaf(W.x, X.w, a0.w);   \
    a1.x = fmaf(W.y, X.x, a1.x); a1.y = fmaf(W.y, X.y, a1.y); a1.z = fmaf(W.y, X.z, a1.z); a1.w = fmaf(W.y, X.w, a1.w);   \
    a2.x = fmaf(W.z, X.x, a2.x); a2.y = fmaf(W.z, X.y, a2.y); a2.z = fmaf(W.z, X.z, a2.z); a2.w = fmaf(W.z, X.w, a2.w);   \
    a3.x = fmaf(W.w, X.x, a3.x); a3.y = fmaf(W.w, X.y, a3.y); a3.z = fmaf(W.w, X.z, a3.z); a3.w = fmaf(W.w, X.w, a3.w);
                // eight weight rows in flight per round (sixteen: no further gain): the pass is a chain of L2 round trips (one workgroup streams every
                // layer's weights), so the depth of each round is what its time is made of; same summation order as before
                for (; k + 8 <= k1; k += 8) {
                    float4 w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = wt[(long)(k + u) * ncol + col];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float4 x = cur[k + u];
                        POL_FMA(w[u], x)
                    }
                }
                for (; k + 4 <= k1; k += 4) {
                    const float4 w0 = wt[(long)(k + 0) * ncol + col], w1 = wt[(long)(k + 1) * ncol + col];
                    const float4 w2 = wt[(long)(k + 2) * ncol + col], w3 = wt[(long)(k + 3) * ncol + col];
                    const float4 x0 = cur[k], x1 = cur[k + 1], x2 = cur[k + 2], x3 = cur[k + 3];
                    POL_FMA(w0, x0) POL_FMA(w1, x1) POL_FMA(w2, x2) POL_FMA(w3, x3)
                }
                for (; k < k1; ++k) {
                    const float4 w = wt[(long)k * ncol + col];
                    const float4 x = cur[k];
                    POL_FMA(w, x)
                }
#undef POL_FMA
                float4 *pp = part + ((long)grp * ncol + col) * 4;
                pp[0] = a0; pp[1] = a1; pp[2] = a2; pp[3] = a3;
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < out; j += blockDim.x) {
            const float b = L.bias[l][j];
            float4 acc = make_float4(b, b, b, b);
            for (int g = 0; g < G; ++g) {                   // fixed order: deterministic
                const float4 p = part[((long)g * ncol + (j >> 2)) * 4 + (j & 3)];
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            if (!last) {
                nxt[j] = make_float4(pol_act(acc.x, act_kind), pol_act(acc.y, act_kind), pol_act(acc.z, act_kind), pol_act(acc.w, act_kind));
            } else {
                const float m[POL_ROWS] = {acc.x, acc.y, acc.z, acc.w};
                const float sd = noise ? expf(log_std[j]) : 0.0f;
#pragma unroll
                for (int r = 0; r < POL_ROWS; ++r) {
                    const int row = r0 + r;
                    if (row >= n) continue;
                    const float a = noise ? fmaf(sd, noise[(long)row * out + j], m[r]) : m[r];
                    action[(long)row * out + j] = (double)a;
                    if (mean_out) mean_out[(long)row * out + j] = m[r];
                }
            }
        }
        __syncthreads();
        float4 *t = cur; cur = nxt; nxt = t;
    }
}

}  // namespace

static int policy_launch(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                         const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                         int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                         double *action, float *mean_out, const void *stage_src, void *stage_dst, int64_t stage_bytes, void *stream,
                         const PolFilter *flt = nullptr) {
    EGP_REQUIRE(n >= 0, "n < 0");
    EGP_REQUIRE(stage_bytes >= 0 && stage_bytes % 4 == 0 && stage_bytes < (1ll << 31) && (stage_bytes == 0 || (stage_src && stage_dst)), "bad staging slab");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(ctx_rows && t_idx && (state || flt) && layers && action, "NULL pointer");
    EGP_REQUIRE(!noise || log_std, "noise needs log_std");
    EGP_REQUIRE(n_layers >= 1 && n_layers <= POL_MAX_LAYERS, "1..8 layers (hidden layers + output layer)");
    EGP_REQUIRE(activation >= 0 && activation <= 2, "activation: 0 tanh, 1 relu, 2 sigmoid");
    EGP_REQUIRE(ctx_dim >= 0 && state_dim >= 0 && ctx_dim + state_dim > 0, "bad input dims");
    PolLayers L;
    int kmax = ctx_dim + state_dim, prev = ctx_dim + state_dim;
    for (int l = 0; l < n_layers; ++l) {
        EGP_REQUIRE(layers[l].wt && layers[l].bias, "NULL layer");
        EGP_REQUIRE(layers[l].in_dim == prev && layers[l].out_dim > 0, "layer dims do not chain");
        L.wt[l] = layers[l].wt; L.bias[l] = layers[l].bias;
        L.in_dim[l] = layers[l].in_dim; L.out_dim[l] = layers[l].out_dim;
        prev = layers[l].out_dim;
        if (prev > kmax) kmax = prev;
    }
    L.n = n_layers;
    EGP_REQUIRE(kmax <= 2048, "layer wider than 2048");
    const int threads = 512;            // (320 .. 960 measured in the rollout: 20.9 / 19.2 (512) / 19.8 (640) / 21.1 us)
    int part_elems = 0;                 // float4 slots for the partial sums: slices x padded outputs
    for (int l = 0; l < n_layers; ++l) {
        const int ncol = (layers[l].out_dim + 3) / 4;
        const int cols = ncol < threads ? ncol : threads;
        const int G = threads / cols;
        const int e = G * ncol * 4;
        if (e > part_elems) part_elems = e;
    }
    const size_t lds = ((size_t)2 * kmax + part_elems) * sizeof(float4) + (flt ? (size_t)2 * state_dim * sizeof(double) : 0);
    EGP_REQUIRE(lds <= 150 * 1024, "layers too wide for the LDS tile");
    const dim3 grid((n + POL_ROWS - 1) / POL_ROWS), block(threads);
    if (flt)
        k_policy_gaussian<true><<<grid, block, lds, (hipStream_t)stream>>>(
            ctx_rows, (long)ctx_row_stride, ctx_dim, (const long long *)t_idx, state, state_dim, n, L, activation, kmax, part_elems,
            log_std, noise, action, mean_out, stage_bytes ? (const unsigned *)stage_src : nullptr, (unsigned *)stage_dst, (int)(stage_bytes / 4), *flt);
    else
        k_policy_gaussian<false><<<grid, block, lds, (hipStream_t)stream>>>(
            ctx_rows, (long)ctx_row_stride, ctx_dim, (const long long *)t_idx, state, state_dim, n, L, activation, kmax, part_elems,
            log_std, noise, action, mean_out, stage_bytes ? (const unsigned *)stage_src : nullptr, (unsigned *)stage_dst, (int)(stage_bytes / 4), PolFilter{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("k_policy_gaussian launch failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}

extern "C" int egp_policy_gaussian_f32(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                       const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                                       int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                                       double *action, float *mean_out, void *stream) {
    return policy_launch(ctx_rows, ctx_row_stride, ctx_dim, t_idx, state, state_dim, n, layers, n_layers, activation, log_std, noise, action,
                         mean_out, nullptr, nullptr, 0, stream);
}

extern "C" int egp_policy_gaussian_staged_f32(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                              const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                                              int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                                              double *action, float *mean_out, const void *stage_src, void *stage_dst, int64_t stage_bytes,
                                              void *stream) {
    return policy_launch(ctx_rows, ctx_row_stride, ctx_dim, t_idx, state, state_dim, n, layers, n_layers, activation, log_std, noise, action,
                         mean_out, stage_src, stage_dst, stage_bytes, stream);
}

// egp_policy_gaussian_staged_f32 with the filter's apply pass in front (see PolFilter): the policy input's state columns are the
// filtered observations of (qpos, qvel) -- the group's n rows -- normalised with `zf_in` merged with the tile statistics that
// egp_obs_zfilter_stats_f64 left in `zf_workspace`; they are also written to y (and y2), and the merged statistics to zf_out:
// together exactly what egp_obs_zfilter_apply_f64 followed by egp_policy_gaussian_staged_f32 on y2 computes, in one launch.
extern "C" int egp_policy_gaussian_filter_f32(egp_ctx *ctx, const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                              const double *qpos, const double *qvel, int32_t n, const double *zf_in, double *zf_out,
                                              double clip, double *y, double *y2, const void *zf_workspace,
                                              const egp_mlp_layer *layers, int32_t n_layers, int32_t activation, const float *log_std,
                                              const float *noise, double *action, float *mean_out, const void *stage_src, void *stage_dst,
                                              int64_t stage_bytes, void *stream) {
    EGP_REQUIRE(ctx && qpos && qvel && zf_in && zf_out && zf_in != zf_out && y && zf_workspace, "NULL pointer / zf_out must differ from zf_in");
    EGP_REQUIRE(n > 0 && n <= 64 * egp::ZF_FUSED_TILES, "1 .. egp_obs_zfilter_split_max_rows() rows");
    const int dim = ctx->dm.obs_dim;
    int rpt, nt;
    egp::zf_tiling(n, &rpt, &nt);
    PolFilter f;
    f.src = egp::ZfSrc<double>{nullptr, qpos, qvel, ctx->dm.nq, ctx->dm.nv, dim, egp::obs_opt_of(ctx->dm)};
    f.st_in = zf_in; f.st_out = zf_out; f.ws = (const double *)zf_workspace; f.n_tiles = nt; f.clip = clip; f.y = y; f.y2 = y2;
    return policy_launch(ctx_rows, ctx_row_stride, ctx_dim, t_idx, nullptr, dim, n, layers, n_layers, activation, log_std, noise, action, mean_out,
                         stage_src, stage_dst, stage_bytes, stream, &f);
}
