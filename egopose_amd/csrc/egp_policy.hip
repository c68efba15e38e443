// Rollout-time policy step in ONE launch:
//   x = [video context row (fp32) | filtered observation (fp64 -> fp32)]  ->  MLP (hidden layers + activation)
//   -> action_mean (Linear) -> action = mean + exp(log_std) * noise  (float32 arithmetic, stored as float64 for
//   the engine), i.e. PolicyGaussian.select_action of models/policy_gaussian.py:19-27 over models/mlp.py:5-25 with
//   the VideoStateNet concatenation of models/video_state_net.py:37-43, for all envs of a group at once.
// During a rollout this chain is ~13 launch-bound torch ops per tick (3 GEMMs of 512 rows, gather, cat, casts);
// here a 4-row tile walks the layers through LDS while the transposed weights stream from L2.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "egp_internal.hpp"

namespace {

constexpr int POL_ROWS = 4;          // rows per workgroup
constexpr int POL_MAX_LAYERS = 8;

struct PolLayers {
    const float *wt[POL_MAX_LAYERS];     // W^T, [in][out] row-major
    const float *bias[POL_MAX_LAYERS];
    int in_dim[POL_MAX_LAYERS], out_dim[POL_MAX_LAYERS];
    int n;                                // hidden layers + the output layer
};

__device__ __forceinline__ float pol_act(float v, int kind) {
    if (kind == 1) return fmaxf(v, 0.0f);                 // relu
    if (kind == 2) return 1.0f / (1.0f + expf(-v));       // sigmoid
    return tanhf(v);                                      // tanh
}

__global__ void k_policy_gaussian(const float *__restrict__ ctx_rows, long ctx_row_stride, int ctx_dim,
                                  const long long *__restrict__ t_idx, const double *__restrict__ state, int state_dim, int n,
                                  PolLayers L, int act_kind, int kmax, const float *__restrict__ log_std,
                                  const float *__restrict__ noise, double *__restrict__ action, float *__restrict__ mean_out) {
    extern __shared__ float4 s_act[];     // two buffers of kmax float4 (one float per row of the tile)
    float4 *cur = s_act, *nxt = s_act + kmax;
    const int r0 = blockIdx.x * POL_ROWS;
    const int in0 = ctx_dim + state_dim;
    for (int k = threadIdx.x; k < in0; k += blockDim.x) {
        float v[POL_ROWS];
#pragma unroll
        for (int r = 0; r < POL_ROWS; ++r) {
            const int row = r0 + r;
            if (row >= n) { v[r] = 0.0f; continue; }
            v[r] = k < ctx_dim ? ctx_rows[(long)row * ctx_row_stride + (long)t_idx[row] * ctx_dim + k]
                               : (float)state[(long)row * state_dim + (k - ctx_dim)];
        }
        cur[k] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int l = 0; l < L.n; ++l) {
        const int in = L.in_dim[l], out = L.out_dim[l];
        const float *__restrict__ wt = L.wt[l];
        const bool last = l == L.n - 1;
        for (int j = threadIdx.x; j < out; j += blockDim.x) {
            const float b = L.bias[l][j];
            float a0 = b, a1 = b, a2 = b, a3 = b;
            int k = 0;
            for (; k + 4 <= in; k += 4) {           // 4 weight loads in flight per thread
                const float w0 = wt[(long)(k + 0) * out + j], w1 = wt[(long)(k + 1) * out + j];
                const float w2 = wt[(long)(k + 2) * out + j], w3 = wt[(long)(k + 3) * out + j];
                const float4 x0 = cur[k], x1 = cur[k + 1], x2 = cur[k + 2], x3 = cur[k + 3];
                a0 = fmaf(w0, x0.x, a0); a1 = fmaf(w0, x0.y, a1); a2 = fmaf(w0, x0.z, a2); a3 = fmaf(w0, x0.w, a3);
                a0 = fmaf(w1, x1.x, a0); a1 = fmaf(w1, x1.y, a1); a2 = fmaf(w1, x1.z, a2); a3 = fmaf(w1, x1.w, a3);
                a0 = fmaf(w2, x2.x, a0); a1 = fmaf(w2, x2.y, a1); a2 = fmaf(w2, x2.z, a2); a3 = fmaf(w2, x2.w, a3);
                a0 = fmaf(w3, x3.x, a0); a1 = fmaf(w3, x3.y, a1); a2 = fmaf(w3, x3.z, a2); a3 = fmaf(w3, x3.w, a3);
            }
            for (; k < in; ++k) {
                const float w = wt[(long)k * out + j];
                const float4 x = cur[k];
                a0 = fmaf(w, x.x, a0); a1 = fmaf(w, x.y, a1); a2 = fmaf(w, x.z, a2); a3 = fmaf(w, x.w, a3);
            }
            if (!last) {
                nxt[j] = make_float4(pol_act(a0, act_kind), pol_act(a1, act_kind), pol_act(a2, act_kind), pol_act(a3, act_kind));
            } else {
                const float m[POL_ROWS] = {a0, a1, a2, a3};
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

extern "C" int egp_policy_gaussian_f32(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                       const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                                       int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                                       double *action, float *mean_out, void *stream) {
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(ctx_rows && t_idx && state && layers && action, "NULL pointer");
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
    int widest = 64;
    for (int l = 0; l < n_layers; ++l) widest = layers[l].out_dim > widest ? layers[l].out_dim : widest;
    int threads = ((widest + 63) / 64) * 64;
    if (threads > 512) threads = 512;
    const size_t lds = (size_t)2 * kmax * sizeof(float4);
    k_policy_gaussian<<<dim3((n + POL_ROWS - 1) / POL_ROWS), dim3(threads), lds, (hipStream_t)stream>>>(
        ctx_rows, (long)ctx_row_stride, ctx_dim, (const long long *)t_idx, state, state_dim, n, L, activation, kmax, log_std, noise,
        action, mean_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("k_policy_gaussian launch failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}
