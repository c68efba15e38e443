// Persistent recurrent kernels for the LSTMs of the video / state front ends (float32; hidden 64 = one direction of
// ego_mimic's bi-LSTM, hidden 128 = the causal video net and the state net of ego_forecast).
//
// The reference evaluates the bi-LSTM with 2*T nn.LSTMCell calls (models/rnn.py:45-61) and every PPO epoch
// re-runs it forward+backward over the padded episode contexts (models/video_state_net.py:65-69), which on
// the GPU becomes ~900 launch-bound steps of tiny GEMM + pointwise kernels per net and epoch. Here one launch
// walks all T timesteps: a 256-thread workgroup owns a tile of sequences, thread c keeps row c of W_hh
// (4H x H) in 64 VGPRs, the tile's hidden state lives in LDS and is read with broadcast ds_read_b128, and the
// input projection x_t W_ih^T + b (one large MFMA GEMM done by rocBLAS beforehand) is streamed from HBM with
// the next step's tile prefetched into registers.
//
//   forward : gates_x [T][B][4H] (+ biases), W_hh [4H][H]  ->  h_out [T][B][H], and for training the
//             activated gates [T][B][4H] and cell states [T][B][H]
//   backward: d h_out, saved gates / cells  ->  d gates_pre [T][B][4H] (what the weight-gradient GEMMs and
//             the bias reduction consume); the recurrent term dgates W_hh stays in the kernel
// Gate order i, f, g, o as torch.nn.LSTMCell. Sequences start from zero state (as the reference does).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "egp_internal.hpp"

namespace egp {

constexpr int KC = 64;          // hidden-state columns multiplied per register chunk

// v_exp_f32 + v_rcp_f32 (1 ulp each): ~1e-7 relative, far inside the float32 parity budget
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// TILE sequences per workgroup of 4*LH threads (thread c = gate column c); small tiles put several workgroups on a CU.
template <int TILE, int LH>
__global__ __launch_bounds__(4 * LH) void k_lstm_fwd(const float *__restrict__ gx, const float *__restrict__ w_hh, int T, int B,
                                                     int reverse, float *__restrict__ h_out, float *__restrict__ gates_out,
                                                     float *__restrict__ c_out) {
    constexpr int LG = 4 * LH, NT = 4 * LH;
    constexpr int NP = (TILE * LH + NT - 1) / NT;   // (row, unit) pairs per thread in the pointwise phase
    static_assert(LH % KC == 0, "hidden size");
    __shared__ __attribute__((aligned(16))) float s_h[2][TILE][LH];   // ping-pong: written for step+1 while step reads
    __shared__ float s_g[2][TILE][LG + 1];
    const int c = threadIdx.x;
    const int r0 = blockIdx.x * TILE;
    float w[LH];
#pragma unroll
    for (int k = 0; k < LH; ++k) w[k] = w_hh[c * LH + k];
    float cst[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) cst[q] = 0.f;
    for (int i = threadIdx.x; i < 2 * TILE * LH; i += NT) (&s_h[0][0][0])[i] = 0.f;
    // software prefetch: the input-projection tile of the NEXT step is in flight while this step computes
    float nxt[TILE];
    {
        const int t0 = reverse ? T - 1 : 0;
        const float *g0 = gx + ((long)t0 * B + r0) * LG;
#pragma unroll
        for (int r = 0; r < TILE; ++r) nxt[r] = (r0 + r < B) ? g0[(long)r * LG + c] : 0.f;
    }
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = reverse ? T - 1 - step : step;
        float acc[TILE];
#pragma unroll
        for (int r = 0; r < TILE; ++r) acc[r] = nxt[r];
        if (step + 1 < T) {
            const int tn = reverse ? t - 1 : t + 1;
            const float *gn = gx + ((long)tn * B + r0) * LG;
#pragma unroll
            for (int r = 0; r < TILE; ++r) nxt[r] = (r0 + r < B) ? gn[(long)r * LG + c] : 0.f;
        }
        // h W_hh^T row by row: the 16 broadcast ds_read_b128 of a 64-column chunk are issued back to back into distinct
        // registers (one exposed LDS latency per chunk, not per read) and feed 4 independent FMA chains
#pragma unroll
        for (int r = 0; r < TILE; ++r) {
            float a0 = acc[r], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int kc = 0; kc < LH; kc += KC) {
                float4 hv[KC / 4];
#pragma unroll
                for (int k4 = 0; k4 < KC / 4; ++k4) hv[k4] = reinterpret_cast<const float4 *>(&s_h[step & 1][r][kc])[k4];
#pragma unroll
                for (int k4 = 0; k4 < KC / 4; ++k4) {
                    a0 = fmaf(w[kc + 4 * k4 + 0], hv[k4].x, a0);
                    a1 = fmaf(w[kc + 4 * k4 + 1], hv[k4].y, a1);
                    a2 = fmaf(w[kc + 4 * k4 + 2], hv[k4].z, a2);
                    a3 = fmaf(w[kc + 4 * k4 + 3], hv[k4].w, a3);
                }
            }
            acc[r] = (a0 + a1) + (a2 + a3);
        }
#pragma unroll
        for (int r = 0; r < TILE; ++r) s_g[step & 1][r][c] = acc[r];
        __syncthreads();       // gate pre-activations of every column are in s_g[step&1]
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int p = threadIdx.x + NT * q, r = p / LH, j = p % LH;
            if (TILE * LH < NT && p >= TILE * LH) continue;          // tiles smaller than the workgroup (TILE < 4)
            const float *sg = &s_g[step & 1][r][0];
            const float ig = sigmoidf_(sg[j]), fg = sigmoidf_(sg[LH + j]);
            const float gg = tanhf_(sg[2 * LH + j]), og = sigmoidf_(sg[3 * LH + j]);
            const float cn = fg * cst[q] + ig * gg;
            const float hn = og * tanhf_(cn);
            cst[q] = cn;
            s_h[(step + 1) & 1][r][j] = hn;
            if (r0 + r < B) {
                const long row = (long)t * B + r0 + r;
                h_out[row * LH + j] = hn;
                if (gates_out) {
                    float *go = gates_out + row * LG;
                    go[j] = ig; go[LH + j] = fg; go[2 * LH + j] = gg; go[3 * LH + j] = og;
                    c_out[row * LH + j] = cn;
                }
            }
        }
        __syncthreads();       // every wave needs the whole new hidden tile before the next step's products
    }
}

// backward through time for one direction. Thread (q = tid/LH, k = tid%LH) keeps W_hh[q*LH .. q*LH+LH-1][k]
// so that dh_rec[r][k] = sum_c dpre[r][c] W_hh[c][k] is a 4-way partial sum reduced through LDS.
template <int TILE, int LH>
__global__ __launch_bounds__(4 * LH) void k_lstm_bwd(const float *__restrict__ dh_out, const float *__restrict__ gates,
                                                  const float *__restrict__ cells, const float *__restrict__ w_hh, int T, int B,
                                                  int reverse, float *__restrict__ dpre) {
    constexpr int LG = 4 * LH, NT = 4 * LH;
    constexpr int NP = (TILE * LH + NT - 1) / NT;
    constexpr bool PARTIAL = TILE * LH < NT;        // fewer (row, unit) pairs than threads
    __shared__ __attribute__((aligned(16))) float s_d[TILE][LG];        // d(pre-activation gates) of this step
    __shared__ float s_part[4][TILE][LH + 1];
    const int q = threadIdx.x / LH, k = threadIdx.x % LH;
    const int r0 = blockIdx.x * TILE;
    float w[LH];
#pragma unroll
    for (int cc = 0; cc < LH; ++cc) w[cc] = w_hh[(q * LH + cc) * LH + k];
    float dc_next[NP], dh_rec[NP];
    // operands of the step being processed, fetched one step ahead: activated gates (4), cell, previous cell, dh
    float pg[NP][4], pc[NP], pcp[NP], pdh[NP];
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) { dc_next[qq] = 0.f; dh_rec[qq] = 0.f; }

#define EGP_LSTM_FETCH(STEP)                                                                         \
    {                                                                                                \
        const int f_t = reverse ? T - 1 - (STEP) : (STEP);                                           \
        const int f_tp = reverse ? f_t + 1 : f_t - 1;                                                \
        _Pragma("unroll") for (int qq = 0; qq < NP; ++qq) {                                          \
            const int p = threadIdx.x + NT * qq, r = p / LH, j = p % LH;                             \
            if ((!PARTIAL || p < TILE * LH) && r0 + r < B) {                                         \
                const long row = (long)f_t * B + r0 + r;                                             \
                const float *g = gates + row * LG;                                                   \
                pg[qq][0] = g[j]; pg[qq][1] = g[LH + j]; pg[qq][2] = g[2 * LH + j]; pg[qq][3] = g[3 * LH + j]; \
                pc[qq] = cells[row * LH + j];                                                        \
                pcp[qq] = (STEP) > 0 ? cells[((long)f_tp * B + r0 + r) * LH + j] : 0.f;              \
                pdh[qq] = dh_out[row * LH + j];                                                      \
            } else {                                                                                 \
                pg[qq][0] = pg[qq][1] = pg[qq][2] = pg[qq][3] = 0.f;                                 \
                pc[qq] = pcp[qq] = pdh[qq] = 0.f;                                                    \
            }                                                                                        \
        }                                                                                            \
    }

    EGP_LSTM_FETCH(T - 1)
    for (int step = T - 1; step >= 0; --step) {
        const int t = reverse ? T - 1 - step : step;
#pragma unroll
        for (int qq = 0; qq < NP; ++qq) {
            const int p = threadIdx.x + NT * qq, r = p / LH, j = p % LH;
            if (PARTIAL && p >= TILE * LH) continue;
            const float ig = pg[qq][0], fg = pg[qq][1], gg = pg[qq][2], og = pg[qq][3];
            const float tc = tanhf_(pc[qq]);
            const float dh = pdh[qq] + dh_rec[qq];
            const float dc = dh * og * (1.f - tc * tc) + dc_next[qq];
            const float d_o = dh * tc * og * (1.f - og);
            const float di = dc * gg * ig * (1.f - ig);
            const float df = dc * pcp[qq] * fg * (1.f - fg);
            const float dg = dc * ig * (1.f - gg * gg);
            dc_next[qq] = dc * fg;
            if (r0 + r < B) {
                float *dp = dpre + ((long)t * B + r0 + r) * LG;
                dp[j] = di; dp[LH + j] = df; dp[2 * LH + j] = dg; dp[3 * LH + j] = d_o;
            }
            s_d[r][j] = di; s_d[r][LH + j] = df; s_d[r][2 * LH + j] = dg; s_d[r][3 * LH + j] = d_o;
        }
        if (step > 0) EGP_LSTM_FETCH(step - 1)
        __syncthreads();
        float acc[TILE];
#pragma unroll
        for (int r = 0; r < TILE; ++r) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int kc = 0; kc < LH; kc += KC) {
                float4 dv[KC / 4];
#pragma unroll
                for (int c4 = 0; c4 < KC / 4; ++c4) dv[c4] = reinterpret_cast<const float4 *>(&s_d[r][q * LH + kc])[c4];
#pragma unroll
                for (int c4 = 0; c4 < KC / 4; ++c4) {
                    a0 = fmaf(w[kc + 4 * c4 + 0], dv[c4].x, a0);
                    a1 = fmaf(w[kc + 4 * c4 + 1], dv[c4].y, a1);
                    a2 = fmaf(w[kc + 4 * c4 + 2], dv[c4].z, a2);
                    a3 = fmaf(w[kc + 4 * c4 + 3], dv[c4].w, a3);
                }
            }
            acc[r] = (a0 + a1) + (a2 + a3);
        }
#pragma unroll
        for (int r = 0; r < TILE; ++r) s_part[q][r][k] = acc[r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NP; ++qq) {
            const int p = threadIdx.x + NT * qq, r = p / LH, j = p % LH;
            if (PARTIAL && p >= TILE * LH) continue;
            dh_rec[qq] = s_part[0][r][j] + s_part[1][r][j] + s_part[2][r][j] + s_part[3][r][j];
        }
        __syncthreads();
    }
#undef EGP_LSTM_FETCH
}

}  // namespace egp

using namespace egp;

// sequences per workgroup: small tiles put several workgroups on a CU so their LDS / barrier latencies overlap
static int lstm_tile(int B) {
    const char *e = getenv("EGP_LSTM_TILE");
    if (e) { const int t = atoi(e); if (t == 2 || t == 4 || t == 8 || t == 16) return t; }
    return B >= 16384 ? 16 : (B >= 8192 ? 8 : 4);
}

static int lstm_launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EGP_E_HIP;
    }
    return EGP_OK;
}

template <int LH>
static void launch_fwd(int tile, const float *gates_x, const float *w_hh, int T, int B, int reverse, float *h_out, float *gates_save,
                       float *cells_save, hipStream_t s) {
    if (tile == 16)
        k_lstm_fwd<16, LH><<<dim3((B + 15) / 16), dim3(4 * LH), 0, s>>>(gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save);
    else if (tile == 8)
        k_lstm_fwd<8, LH><<<dim3((B + 7) / 8), dim3(4 * LH), 0, s>>>(gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save);
    else if (tile == 4)
        k_lstm_fwd<4, LH><<<dim3((B + 3) / 4), dim3(4 * LH), 0, s>>>(gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save);
    else
        k_lstm_fwd<2, LH><<<dim3((B + 1) / 2), dim3(4 * LH), 0, s>>>(gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save);
}

template <int LH>
static void launch_bwd(int tile, const float *dh_out, const float *gates_save, const float *cells_save, const float *w_hh, int T, int B,
                       int reverse, float *d_pre, hipStream_t s) {
    if (tile == 16)
        k_lstm_bwd<16, LH><<<dim3((B + 15) / 16), dim3(4 * LH), 0, s>>>(dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre);
    else if (tile == 8)
        k_lstm_bwd<8, LH><<<dim3((B + 7) / 8), dim3(4 * LH), 0, s>>>(dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre);
    else if (tile == 4)
        k_lstm_bwd<4, LH><<<dim3((B + 3) / 4), dim3(4 * LH), 0, s>>>(dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre);
    else
        k_lstm_bwd<2, LH><<<dim3((B + 1) / 2), dim3(4 * LH), 0, s>>>(dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre);
}

extern "C" {

int egp_lstm_fwd_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t reverse,
                     float *h_out, float *gates_save, float *cells_save, void *stream) {
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(gates_x && w_hh && h_out, "NULL pointer");
    EGP_REQUIRE((gates_save == nullptr) == (cells_save == nullptr), "gates_save and cells_save go together");
    hipStream_t s = (hipStream_t)stream;
    if (hidden == 64) launch_fwd<64>(lstm_tile(B), gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save, s);
    else launch_fwd<128>(4, gates_x, w_hh, T, B, reverse, h_out, gates_save, cells_save, s);
    return lstm_launch_check("k_lstm_fwd");
}

int egp_lstm_bwd_f32(const float *dh_out, const float *gates_save, const float *cells_save, const float *w_hh, int32_t T, int32_t B,
                     int32_t hidden, int32_t reverse, float *d_pre, void *stream) {
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(dh_out && gates_save && cells_save && w_hh && d_pre, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hidden == 64) launch_bwd<64>(lstm_tile(B), dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre, s);
    else launch_bwd<128>(4, dh_out, gates_save, cells_save, w_hh, T, B, reverse, d_pre, s);
    return lstm_launch_check("k_lstm_bwd");
}

}  // extern "C"
