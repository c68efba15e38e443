// Rollout-time policy step in ONE launch:
//   x = [video context row (fp32) | filtered observation (fp64 -> fp32)]  ->  MLP (hidden layers + activation)
//   -> action_mean (Linear) -> action = mean + exp(log_std) * noise  (float32 arithmetic, stored as float64 for
//   the engine), i.e. PolicyGaussian.select_action of models/policy_gaussian.py:19-27 over models/mlp.py:5-25 with
//   the VideoStateNet concatenation of models/video_state_net.py:37-43, for all envs of a group at once.
// During a rollout this chain is ~13 launch-bound torch ops per tick (3 GEMMs of 512 rows, gather, cat, casts).
//
// Round 4: the layer products run on the matrix cores in exact float32 (v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4
// outer products per instruction = 4 batch rows x 64 output columns x one input feature; an fmaf chain, bit for bit).
// A workgroup owns R = 4 or 8 batch rows and walks the layers with the activations in LDS. Its NW waves split a layer's
// input features (k) between them; every wave streams its k range of the PACKED weights (egp_mlp_pack_f32: for each
// group of 64 output columns and each quad of input features one 1-KiB block, lane l's 16 bytes = column l at the four
// k of the quad -> one coalesced global_load_dwordx4 per wave and block, PF blocks in flight per column group) and
// reads the rows' activations as broadcast ds_read_b128; the per-wave partial sums meet in LDS, where bias and
// activation are applied in a fixed order (deterministic, independent of the row's position in the batch).
// The weights of the next pass are requested BEFORE the partial sums are written and reduced (they do not depend on the
// activations), and the first layer's before the input rows are gathered: the dependent global round trips of a tick
// (context-row index -> context row, tile statistics -> filtered observation) overlap the weight stream.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "egp_internal.hpp"
#include "egp_filter_dev.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// EGP_POLICY_TRACE (tools/probes/policy_trace.py builds the library with it): wall_clock64 stamps (100 MHz) of thread 0 of one workgroup
#ifdef EGP_POLICY_TRACE
__device__ long long g_pol_trace[64];
#define POL_TR(i) do { if (blockIdx.x == EGP_POLICY_TRACE && threadIdx.x == 0) g_pol_trace[i] = wall_clock64(); } while (0)
// (the prologue of a wave that holds state columns: stamps 32..39 of thread 128)
#define POL_TR_S(i) do { if (blockIdx.x == EGP_POLICY_TRACE && threadIdx.x == 128) g_pol_trace[32 + (i)] = wall_clock64(); } while (0)
#else
#define POL_TR(i) do { } while (0)
#define POL_TR_S(i) do { } while (0)
#endif

constexpr int POL_MAX_LAYERS = 8;
constexpr int POL_GC = 5;            // column groups (64 outputs each) a pass accumulates in registers: 320 >= the 300-wide layer

struct PolLayers {
    const float *wp[POL_MAX_LAYERS];     // packed weights (egp_mlp_pack_f32)
    const float *bias[POL_MAX_LAYERS];
    int in_dim[POL_MAX_LAYERS], out_dim[POL_MAX_LAYERS];
    int n;                                // hidden layers + the output layer
    int sum_out4;                         // sum of the output widths, each rounded up to 4 (the LDS copy of the biases)
    int warm_lines;                       // 128-byte lines of the packed weights when the layers' buffers follow each other in memory, else 0
};

__device__ __forceinline__ float pol_act(float v, int kind) {
    if (kind == 1) return fmaxf(v, 0.0f);                 // relu
    if (kind == 2) return 1.0f / (1.0f + expf(-v));       // sigmoid
    return tanhf(v);                                      // tanh
}

// nn.Linear weight W[out][in] (row stride ldw) -> packed[(g * nkq + kq) * 64 + lane][kk] = W[64 g + lane][4 kq + kk], zero outside
__global__ void k_mlp_pack(const float *__restrict__ W, long ldw, int in_dim, int out_dim, float *__restrict__ dst, long total) {
    const int nkq = (in_dim + 3) >> 2;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(e & 3), lane = (int)((e >> 2) & 63);
        const long blk = e >> 8;
        const int kq = (int)(blk % nkq), g = (int)(blk / nkq);
        const int o = 64 * g + lane, i = 4 * kq + kk;
        dst[e] = (o < out_dim && i < in_dim) ? W[(long)o * ldw + i] : 0.0f;
    }
}

// The wave's weight blocks of one pass: PF quads x up to POL_GC column groups in flight.
template <int PF>
struct PolStage {
    f32x4 w[PF][POL_GC];
};

// request the blocks of quads kq .. kq + PF - 1 (clamped into the wave's range) of column groups g0 .. g0 + ng - 1 (clamped)
// (a pass over ONE column group -- the output layer -- uses the POL_GC slots of a stage for POL_GC different quads: PF x POL_GC quads
//  in flight instead of PF; with PF of them the 50 quads of the 200 -> 52 layer were a chain of L2 round trips, 1.8 of the step's 15 us)
template <int PF>
__device__ __forceinline__ void pol_preload(PolStage<PF> &st, const float *__restrict__ wl, int nkq, int g0, int ng, int kq0, int kq1, int lane) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
        for (int g = 0; g < POL_GC; ++g) {
            const int k = min(ng == 1 ? kq0 + s * POL_GC + g : kq0 + s, kq1 - 1);
            const int gg = g0 + min(g, ng - 1);
            st.w[s][g] = *reinterpret_cast<const f32x4 *>(wl + (((long)gg * nkq + k) * 64 + lane) * 4);
        }
    }
}

// One pass: acc[g][h] += W[cols of group g0 + g][k range of the wave] * x[rows 4 h .. 4 h + 3][k range]. Straight-line loop body
// (every load is unconditional with a clamped address, quads past the wave's range multiply zeros), so the compiler's
// counted waits keep PF blocks per group in flight.
template <int NG, int R, int PF>
__device__ __forceinline__ void pol_pass(PolStage<PF> &st, const float *__restrict__ wl, int nkq, int g0, int kq0, int kq1, const float *x, int xs,
                                         int lane, f32x4 (&acc)[POL_GC][R / 4]) {
    const float *xr = x + (lane & 3) * xs;
    const float *wb = wl + (((long)g0 * nkq) * 64 + lane) * 4;
    const long gstride = (long)nkq * 256;
    if constexpr (NG == 1) {                             // one column group: the stage's slots hold PF x POL_GC consecutive quads
        constexpr int D = PF * POL_GC;
        for (int kq = kq0; kq < kq1; kq += D) {
#pragma unroll
            for (int s = 0; s < PF; ++s)
#pragma unroll
                for (int g = 0; g < POL_GC; ++g) {
                    const int k = kq + s * POL_GC + g;
                    const bool live = k < kq1;
                    const int kc = live ? k : kq1 - 1;
                    f32x4 b[R / 4];
#pragma unroll
                    for (int h = 0; h < R / 4; ++h) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * h * xs + 4 * kc);
                        b[h][0] = live ? v[0] : 0.0f; b[h][1] = live ? v[1] : 0.0f; b[h][2] = live ? v[2] : 0.0f; b[h][3] = live ? v[3] : 0.0f;
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int h = 0; h < R / 4; ++h)
                            acc[0][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(st.w[s][g][kk], b[h][kk], acc[0][h], 0, 0, 0);
                    const int kn = min(k + D, kq1 - 1);
                    st.w[s][g] = *reinterpret_cast<const f32x4 *>(wb + (long)kn * 256);
                }
        }
        return;
    }
    for (int kq = kq0; kq < kq1; kq += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int k = kq + s;
            const bool live = k < kq1;
            const int kc = live ? k : kq1 - 1;
            f32x4 b[R / 4];
#pragma unroll
            for (int h = 0; h < R / 4; ++h) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * h * xs + 4 * kc);
                b[h][0] = live ? v[0] : 0.0f; b[h][1] = live ? v[1] : 0.0f; b[h][2] = live ? v[2] : 0.0f; b[h][3] = live ? v[3] : 0.0f;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int h = 0; h < R / 4; ++h)
                        acc[g][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(st.w[s][g][kk], b[h][kk], acc[g][h], 0, 0, 0);
            const int kn = min(k + PF, kq1 - 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) st.w[s][g] = *reinterpret_cast<const f32x4 *>(wb + g * gstride + (long)kn * 256);
        }
    }
}

// `stage_src` / `stage_dst` (optional): the tick's flag slab. The rollout stages the integer flags and context-row indices of a
// tick in pinned host memory; instead of a copy-engine transfer in front of this kernel (one more dependent operation, ~6 us,
// on the chain filter -> policy -> env-step of every tick) the workgroups copy the slab to its device copy themselves -- the
// kernels that run after the env-step (reward, filter) read it there -- and take their own rows' indices (`t_idx`, which then
// points into the pinned slab) with ONE load per row.
template <int R, int NW, int PF, bool FILTER>
__device__ __forceinline__ void policy_body(const float *__restrict__ ctx_rows, long ctx_row_stride, int ctx_dim,
                                  const long long *__restrict__ t_idx, const double *__restrict__ state, int state_dim, int n,
                                  const PolLayers &L, int act_kind, int xs, const float *__restrict__ log_std,
                                  const float *__restrict__ noise, double *__restrict__ action, float *__restrict__ mean_out,
                                  const unsigned *__restrict__ stage_src, unsigned *__restrict__ stage_dst, int stage_words,
                                  const PolFilter &F) {
    constexpr int T = NW * 64;
    constexpr int PS = POL_GC * 64 + 4;              // row stride of a wave's partial sums (floats)
    extern __shared__ __attribute__((aligned(16))) float s_f[];     // cur[R][xs] | nxt[R][xs] | part[NW][R][PS] [| mean, 1/std: 2 dim doubles]
    float *cur = s_f, *nxt = s_f + R * xs, *part = s_f + 2 * R * xs;
    // small operands of the epilogues, fetched once in the prologue (a global load in an epilogue is a cold round trip on the chain):
    // every layer's bias | exp(log_std) | the rows' noise
    float *s_bias = part + NW * R * PS;
    const int out_last = L.out_dim[L.n - 1];
    float *s_sd = s_bias + L.sum_out4, *s_noise = s_sd + ((out_last + 3) & ~3);
    double *s_ms = reinterpret_cast<double *>(s_noise + R * ((out_last + 3) & ~3) + ((R * ((out_last + 3) & ~3)) & 1));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the k loops' bounds live in SGPRs)
    const int r0 = blockIdx.x * R;
    const int in0 = ctx_dim + state_dim;

    POL_TR(0);
    // the rows' context-row indices first: the longest dependent chain of the prologue (index -> context row) starts here
    long tix[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tix[r] = (long)t_idx[min(r0 + r, n - 1)];
    // One input column per thread (in0 <= T, the shipped 128 + 115 on 256 threads): the thread that normalises state column c
    // merges that column's statistics itself -- no LDS hop and no barrier between the filter's two phases, and the waves that
    // gather the context columns do not wait for the merge. (Waves hold context columns, state columns or straddle the
    // boundary: `state_wave` is wave-uniform.)
    const int in0p = (in0 + 3) & ~3;
    const bool own_col = in0p <= T;
    const bool state_wave = own_col && wave * 64 + 63 >= ctx_dim && wave * 64 < in0;
    const bool mine = tid >= ctx_dim && tid < in0;
    const int c_own = min(max(tid - ctx_dim, 0), state_dim - 1);
    // ... and the filter's loads next, ahead of the weight traffic below in the memory queue (loads return in order): the
    // statistics' tile partials and the rows' raw state, which do not depend on each other -- one cold round trip for both
    egp::ZfWaveMerge M;
    egp::ZfSrc<double>::Rows<R> pend;
    if constexpr (FILTER) {
        if (state_wave) {
            M.begin(state_dim, F.st_in, c_own);
            M.load(state_dim, F.n_tiles, F.ws, c_own, 0);
            pend = F.src.template load_rows<R>(r0, n, c_own);
        }
    }
    // L2 warm-up. A kernel starts with a cold L2 on every XCD (the per-XCD L2s are invalidated at kernel boundaries), so the
    // first touch of a weight line costs a trip to the memory side (~0.7 us) and a wave's PF x POL_GC KiB in flight turn the
    // weight stream into a chain of such trips. The workgroups of one XCD (round-robin dispatch: blockIdx & 7) therefore each
    // touch a DIFFERENT slice of the packed weights right away -- one dword per 128-byte line, all requests in flight at once --
    // while the input rows are gathered; by the time the k loops start, the XCD's L2 holds every layer and the stream runs at
    // L2-hit latency. (Pure prefetch: results are never used; up to POL_WARM x T lines per workgroup; needs the layers' packed
    // buffers back to back in memory -- FusedGaussianPolicy allocates them so -- else warm_lines = 0 and one line is touched.)
    constexpr int POL_WARM = 8;
    float warm[POL_WARM];
    {
        const int per_xcd = (gridDim.x + 7) >> 3, me = blockIdx.x >> 3;
        const int lo = (int)((long)L.warm_lines * me / per_xcd), hi = (int)((long)L.warm_lines * (me + 1) / per_xcd);
#pragma unroll
        for (int u = 0; u < POL_WARM; ++u) {
            const int i = min(lo + tid + u * T, max(hi - 1, 0));
            warm[u] = L.wp[0][(long)i * 32];
        }
    }
    // the first layer's own first blocks
    PolStage<PF> st;
    int nkq = (L.in_dim[0] + 3) >> 2;
    int kq0 = wave * nkq / NW, kq1 = (wave + 1) * nkq / NW;
    {
        const int ng_all = (L.out_dim[0] + 63) >> 6;
        pol_preload<PF>(st, L.wp[0], nkq, 0, min(ng_all, POL_GC), kq0, max(kq1, kq0 + 1), lane);
    }
    // the context columns (waiting for the indices requested first; everything above stays in flight)
    long ctx_off[R];                      // every thread resolves its rows' context offsets itself: no LDS hop + barrier between the two dependent loads
#pragma unroll
    for (int r = 0; r < R; ++r) ctx_off[r] = (long)min(r0 + r, n - 1) * ctx_row_stride + tix[r] * ctx_dim;
    float v_in[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v_in[r] = 0.0f;
    if (own_col && wave * 64 < ctx_dim) {
        if (tid < ctx_dim) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r0 + r < n) v_in[r] = ctx_rows[ctx_off[r] + tid];
        }
    }
    // The small operands of the epilogues (every layer's bias | exp(log_std) | the rows' noise: one LDS range from s_bias on):
    // requested here into registers and parked in LDS behind the inputs' barrier -- a load followed by its LDS store right
    // here would stall every wave until the weight loads queued in front of it have landed.
    constexpr int POL_SMALL = 4;
    const int osd4 = (out_last + 3) & ~3;
    const int small_n = L.sum_out4 + (noise ? osd4 + R * out_last : 0);
    // (where LDS float f of the range comes from, without touching memory: 0 padding, 1 value, 2 exp(value), 3 zero)
    auto small_source = [&](int f, const float *&src) -> int {
        src = L.bias[0];
        if (f >= small_n) return 0;
        if (f < L.sum_out4) {
            int off = 0, kind = 0;
            for (int l = 0; l < L.n; ++l) {
                const int j = f - off;
                const bool hit = j >= 0 && j < L.out_dim[l];
                src = hit ? L.bias[l] + j : src;
                kind = hit ? 1 : kind;
                off += (L.out_dim[l] + 3) & ~3;
            }
            return kind;
        }
        const int j = f - L.sum_out4;
        if (j < osd4) {
            src = j < out_last ? log_std + j : src;
            return j < out_last ? 2 : 0;
        }
        const int e = j - osd4, r = e / out_last;
        const bool live = r0 + r < n;
        src = live ? noise + (long)(r0 + r) * out_last + (e - r * out_last) : src;
        return live ? 1 : 3;
    };
    float small_v[POL_SMALL];
    int small_kind[POL_SMALL];
#pragma unroll
    for (int u = 0; u < POL_SMALL; ++u) {
        const float *src;
        small_kind[u] = small_source(tid + u * T, src);
        small_v[u] = *src;                        // (unconditional: a load under a branch is waited for at the branch's end)
    }

    if constexpr (FILTER) if (!own_col) {   // k_zf_apply's first phase: the merged statistics, every workgroup for itself
        const int dim = state_dim;
        for (int c = tid; c < dim; c += T) {
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
    POL_TR(1);
    if (stage_src)
        for (int i = blockIdx.x * T + tid; i < stage_words; i += gridDim.x * T) stage_dst[i] = stage_src[i];
    if (own_col) {
        const int k = tid;
        float (&v)[R] = v_in;
        if (state_wave) {
            const int c = c_own;
            if constexpr (FILTER) {
                POL_TR_S(0);
                M.merge(F.n_tiles, 0);
                POL_TR_S(1);
                for (int q0 = 8; q0 < F.n_tiles; q0 += 8) {
                    M.load(state_dim, F.n_tiles, F.ws, c, q0);
                    M.merge(F.n_tiles, q0);
                }
                const double cnt = M.cnt, mean = M.mean, S = M.S;
                if (blockIdx.x == 0 && mine) {
                    F.st_out[1 + c] = mean;
                    F.st_out[1 + state_dim + c] = S;
                    if (c == 0) F.st_out[0] = cnt;
                }
                const double var = cnt > 1.0 ? S / (cnt - 1.0) : mean * mean;
                const double istd = 1.0 / (sqrt(var) + 1e-8);
                POL_TR_S(2);
                F.src.template finish_rows<R>(pend, r0, n, c);
                const double (&raw)[R] = pend.own;
                POL_TR_S(3);
                if (mine) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = r0 + r;
                        if (row < n) {
                            double x = (raw[r] - mean) * istd;
                            if (F.clip > 0.0) x = fmin(fmax(x, -F.clip), F.clip);
                            const long e = (long)row * state_dim + c;
                            F.y[e] = x;
                            if (F.y2) F.y2[e] = x;
                            v[r] = (float)x;
                        }
                    }
                }
            } else if (mine) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r0 + r < n) v[r] = (float)state[(long)(r0 + r) * state_dim + c];
            }
        }
        POL_TR_S(4);
        if (k < in0p) {
#pragma unroll
            for (int r = 0; r < R; ++r) cur[r * xs + k] = v[r];
        }
        POL_TR_S(5);
    } else {
    if constexpr (FILTER) __syncthreads();            // the merged statistics (s_ms) are read by the staging loop below
    for (int k = tid; k < in0p; k += T) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = r0 + r;
            float v = 0.0f;
            if (row < n && k < in0) {
                if (k < ctx_dim) {
                    v = ctx_rows[ctx_off[r] + k];
                } else if constexpr (FILTER) {            // k_zf_apply's second phase for this element
                    const int c = k - ctx_dim;
                    double x = ((double)F.src.at(row, c) - s_ms[c]) * s_ms[state_dim + c];
                    if (F.clip > 0.0) x = fmin(fmax(x, -F.clip), F.clip);
                    const long e = (long)row * state_dim + c;
                    F.y[e] = x;
                    if (F.y2) F.y2[e] = x;
                    v = (float)x;
                } else {
                    v = (float)state[(long)row * state_dim + (k - ctx_dim)];
                }
            }
            cur[r * xs + k] = v;
        }
    }
    }
    // the small operands into LDS (their loads have landed with the inputs'); the rest of a range beyond POL_SMALL x T floats directly
#pragma unroll
    for (int u = 0; u < POL_SMALL; ++u)
        if (small_kind[u]) s_bias[tid + u * T] = small_kind[u] == 2 ? expf(small_v[u]) : (small_kind[u] == 3 ? 0.0f : small_v[u]);
    for (int f = tid + POL_SMALL * T; f < small_n; f += T) {
        const float *src;
        const int kind = small_source(f, src);
        const float val = *src;
        if (kind) s_bias[f] = kind == 2 ? expf(val) : (kind == 3 ? 0.0f : val);
    }
    // (the barrier for the staged inputs in LDS only: __syncthreads() would also wait for the acknowledgement of the filtered
    //  rows' global stores above -- a trip to the memory side on the critical path; nothing in this kernel reads them back)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {   // the warm-up loads have long landed; their registers are free from here on
        float sink = 0.0f;
#pragma unroll
        for (int u = 0; u < POL_WARM; ++u) sink += warm[u];
        asm volatile("" ::"v"(sink));
    }
    POL_TR(2);

    int bias_off = 0;
    for (int l = 0; l < L.n; ++l) {
        const int out = L.out_dim[l];
        const int ng_all = (out + 63) >> 6;
        const bool last = l == L.n - 1;
        const float *wl = L.wp[l];
        for (int g0 = 0; g0 < ng_all; g0 += POL_GC) {
            const int ng = min(POL_GC, ng_all - g0);
            f32x4 acc[POL_GC][R / 4];
#pragma unroll
            for (int g = 0; g < POL_GC; ++g)
#pragma unroll
                for (int h = 0; h < R / 4; ++h) acc[g][h] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kq1 > kq0) {
                switch (ng) {
                    case 1: pol_pass<1, R, PF>(st, wl, nkq, g0, kq0, kq1, cur, xs, lane, acc); break;
                    case 2: pol_pass<2, R, PF>(st, wl, nkq, g0, kq0, kq1, cur, xs, lane, acc); break;
                    case 3: pol_pass<3, R, PF>(st, wl, nkq, g0, kq0, kq1, cur, xs, lane, acc); break;
                    case 4: pol_pass<4, R, PF>(st, wl, nkq, g0, kq0, kq1, cur, xs, lane, acc); break;
                    default: pol_pass<5, R, PF>(st, wl, nkq, g0, kq0, kq1, cur, xs, lane, acc); break;
                }
            }
            POL_TR(3 + 4 * l);
            // the next pass's first blocks: same layer's next column chunk, or the next layer
            {
                int nl = l, ng0 = g0 + POL_GC;
                if (ng0 >= ng_all) { nl = l + 1; ng0 = 0; }
                if (nl < L.n) {
                    const int nkq_n = (L.in_dim[nl] + 3) >> 2;
                    const int a = wave * nkq_n / NW, b = (wave + 1) * nkq_n / NW;
                    const int ng_n = min(POL_GC, ((L.out_dim[nl] + 63) >> 6) - ng0);
                    pol_preload<PF>(st, L.wp[nl], nkq_n, ng0, ng_n, a, max(b, a + 1), lane);
                }
            }
            // partial sums -> LDS: register i of lane 4 b + j = out[row 4 h + j][column 64 g + 4 b + i]
#pragma unroll
            for (int g = 0; g < POL_GC; ++g)
                if (g < ng) {
#pragma unroll
                    for (int h = 0; h < R / 4; ++h)
                        *reinterpret_cast<f32x4 *>(part + ((wave * R) + 4 * h + (lane & 3)) * PS + 64 * g + (lane & ~3)) = acc[g][h];
                }
            POL_TR(4 + 4 * l);
            __syncthreads();
            POL_TR(5 + 4 * l);
            const int c_base = 64 * g0;
            const int cw = min(out - c_base, POL_GC * 64);               // real columns of this chunk
            const int cw4 = last ? cw : min((cw + 3) & ~3, ng * 64);      // hidden layers: the pad columns of the last quad become zeros
            for (int c = tid; c < cw4; c += T) {
                const int col = c_base + c;
                const bool real = c < cw;
                const float bv = real ? s_bias[bias_off + col] : 0.0f;
                float v[R];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = bv;
#pragma unroll
                for (int w = 0; w < NW; ++w)                         // fixed order: deterministic
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] += part[(w * R + r) * PS + c];
                if (!last) {
#pragma unroll
                    for (int r = 0; r < R; ++r) nxt[r * xs + col] = real ? pol_act(v[r], act_kind) : 0.0f;
                } else {
                    const float sd = noise ? s_sd[col] : 0.0f;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = r0 + r;
                        if (row >= n) continue;
                        const float a = noise ? fmaf(sd, s_noise[r * out + col], v[r]) : v[r];
                        action[(long)row * out + col] = (double)a;
                        if (mean_out) mean_out[(long)row * out + col] = v[r];
                    }
                }
            }
            __syncthreads();
            POL_TR(6 + 4 * l);
        }
        bias_off += (out + 3) & ~3;
        if (!last) {
            nkq = (L.in_dim[l + 1] + 3) >> 2;
            kq0 = wave * nkq / NW; kq1 = (wave + 1) * nkq / NW;
            float *t = cur; cur = nxt; nxt = t;
        }
    }
}

// Registers: no cap. A resident K1 workgroup keeps one 346-register wave on every SIMD of its CU for the length of an env-step
// (160 left); with 2 groups the OTHER group's K1 occupies half the CUs while this group's policy step runs. A policy workgroup
// that needs more than 160 registers per SIMD cannot be placed beside a K1 wave and goes to the K1-free CUs -- which is where it
// runs fastest anyway (no polling waves competing for issue slots), so the prefetch depth is chosen for speed, not for fitting.
#define POL_KERNEL_ARGS                                                                                                          \
    const float *__restrict__ ctx_rows, long ctx_row_stride, int ctx_dim, const long long *__restrict__ t_idx,                     \
        const double *__restrict__ state, int state_dim, int n, PolLayers L, int act_kind, int xs, const float *__restrict__ log_std, \
        const float *__restrict__ noise, double *__restrict__ action, float *__restrict__ mean_out,                               \
        const unsigned *__restrict__ stage_src, unsigned *__restrict__ stage_dst, int stage_words, PolFilter F
#define POL_KERNEL_PASS ctx_rows, ctx_row_stride, ctx_dim, t_idx, state, state_dim, n, L, act_kind, xs, log_std, noise, action, mean_out, stage_src, stage_dst, stage_words, F
template <int R, int PF, bool FILTER>
__global__ __launch_bounds__(256) void k_policy_gaussian_w4(POL_KERNEL_ARGS) {
    policy_body<R, 4, PF, FILTER>(POL_KERNEL_PASS);
}
template <int R, int PF, bool FILTER>
__global__ __launch_bounds__(512) void k_policy_gaussian_w8(POL_KERNEL_ARGS) {
    policy_body<R, 8, PF, FILTER>(POL_KERNEL_PASS);
}

}  // namespace

#ifdef EGP_POLICY_TRACE
extern "C" int egp_policy_trace_read(long long *out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_pol_trace), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// Packed form of an nn.Linear weight for the policy step: egp_mlp_pack_floats(in, out) floats.
extern "C" int64_t egp_mlp_pack_floats(int32_t in_dim, int32_t out_dim) {
    if (in_dim <= 0 || out_dim <= 0) return 0;
    return (int64_t)((out_dim + 63) / 64) * ((in_dim + 3) / 4) * 256;
}

extern "C" int egp_mlp_pack_f32(const float *weight, int64_t ldw, int32_t in_dim, int32_t out_dim, float *packed, void *stream) {
    EGP_REQUIRE(weight && packed && in_dim > 0 && out_dim > 0 && ldw >= in_dim, "bad weight");
    const long total = (long)egp_mlp_pack_floats(in_dim, out_dim);
    const int threads = 256;
    const long blocks = (total + threads - 1) / threads;
    k_mlp_pack<<<dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(threads), 0, (hipStream_t)stream>>>(weight, (long)ldw, in_dim, out_dim, packed, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("k_mlp_pack launch failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}

// tile of the policy step: rows per workgroup x waves x weight blocks in flight per column group.
// Round-4 sweep on the MI355X (tools/probes/policy_tile_sweep.sh; 512 rows, bench workload): 4x4x2 15.2 us per launch and the
// lowest / least scattered T_sample; 4x8x2 14.9 us; 8-row tiles 19-21 us (twice the MFMAs per workgroup on half the CUs).
static void policy_tile(int *R, int *NW, int *PF) { *R = 4; *NW = 4; *PF = 2; }

template <int R, int NW, int PF>
static void policy_launch_t(bool flt, dim3 grid, size_t lds, hipStream_t s, const float *ctx_rows, long ctx_row_stride, int ctx_dim,
                            const long long *t_idx, const double *state, int state_dim, int n, const PolLayers &L, int act, int xs,
                            const float *log_std, const float *noise, double *action, float *mean_out, const unsigned *ssrc, unsigned *sdst,
                            int swords, const PolFilter &F) {
#define POL_GO(KERN, FLT) KERN<R, PF, FLT><<<grid, dim3(NW * 64), lds, s>>>(ctx_rows, ctx_row_stride, ctx_dim, t_idx, state, state_dim, n, L, act, xs, \
                                                                            log_std, noise, action, mean_out, ssrc, sdst, swords, F)
    if constexpr (NW == 4) { if (flt) POL_GO(k_policy_gaussian_w4, true); else POL_GO(k_policy_gaussian_w4, false); }
    else { if (flt) POL_GO(k_policy_gaussian_w8, true); else POL_GO(k_policy_gaussian_w8, false); }
#undef POL_GO
}

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
        EGP_REQUIRE(((uintptr_t)layers[l].wt & 15) == 0, "packed weights must be 16-byte aligned");
        EGP_REQUIRE(layers[l].in_dim == prev && layers[l].out_dim > 0, "layer dims do not chain");
        L.wp[l] = layers[l].wt; L.bias[l] = layers[l].bias;
        L.in_dim[l] = layers[l].in_dim; L.out_dim[l] = layers[l].out_dim;
        prev = layers[l].out_dim;
        if (prev > kmax) kmax = prev;
    }
    L.n = n_layers;
    EGP_REQUIRE(kmax <= 2048, "layer wider than 2048");
    L.sum_out4 = 0;
    long lines = 0;
    bool contiguous = true;
    for (int l = 0; l < n_layers; ++l) {
        L.sum_out4 += (layers[l].out_dim + 3) & ~3;
        if (l > 0 && layers[l].wt != layers[l - 1].wt + egp_mlp_pack_floats(layers[l - 1].in_dim, layers[l - 1].out_dim)) contiguous = false;
        lines += egp_mlp_pack_floats(layers[l].in_dim, layers[l].out_dim) / 32;
    }
    L.warm_lines = contiguous && lines < (1l << 30) ? (int)lines : 0;
    int R, NW, PF;
    policy_tile(&R, &NW, &PF);
    const int xs = ((kmax + 31) & ~31) + 4;          // activation row stride: rows 0..3 of a broadcast read sit on different banks
    size_t small = 0;                                 // biases | exp(log_std) | noise rows (floats, see the kernel's carve-up)
    for (int l = 0; l < n_layers; ++l) small += (size_t)((layers[l].out_dim + 3) & ~3);
    const size_t out4 = (size_t)((layers[n_layers - 1].out_dim + 3) & ~3);
    small += out4 + (size_t)R * out4 + 1;
    const size_t lds = ((size_t)2 * R * xs + (size_t)NW * R * (POL_GC * 64 + 4) + small) * sizeof(float) + (flt ? (size_t)2 * state_dim * sizeof(double) : 0);
    EGP_REQUIRE(lds <= 150 * 1024, "layers too wide for the LDS tile");
    const dim3 grid((n + R - 1) / R);
    const PolFilter F = flt ? *flt : PolFilter{};
    const unsigned *ssrc = stage_bytes ? (const unsigned *)stage_src : nullptr;
#define POL_CASE(r, w, p)                                                                                                              \
    if (R == r && NW == w && PF == p) {                                                                                                \
        policy_launch_t<r, w, p>(flt != nullptr, grid, lds, (hipStream_t)stream, ctx_rows, (long)ctx_row_stride, ctx_dim, (const long long *)t_idx, \
                                 state, state_dim, n, L, activation, xs, log_std, noise, action, mean_out, ssrc, (unsigned *)stage_dst,  \
                                 (int)(stage_bytes / 4), F);                                                                            \
    } else
    POL_CASE(4, 4, 2)
    { egp::set_error("policy tile %dx%dx%d is not built", R, NW, PF); return EGP_E_INVALID; }
#undef POL_CASE
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
                                              const double *qpos, const double *qvel, const int32_t *phase_t, int32_t n, const double *zf_in, double *zf_out,
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
    EGP_REQUIRE(!ctx->dm.obs_phase || phase_t, "the model has obs_phase: phase_t (the rows' cur_t) is required");
    f.src = egp::ZfSrc<double>{nullptr, qpos, qvel, ctx->dm.nq, ctx->dm.nv, dim, egp::obs_opt_of(ctx->dm), phase_t};
    f.st_in = zf_in; f.st_out = zf_out; f.ws = (const double *)zf_workspace; f.n_tiles = nt; f.clip = clip; f.y = y; f.y2 = y2;
    return policy_launch(ctx_rows, ctx_row_stride, ctx_dim, t_idx, nullptr, dim, n, layers, n_layers, activation, log_std, noise, action, mean_out,
                         stage_src, stage_dst, stage_bytes, stream, &f);
}
