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

// ------------------------------------------------------------------------------------------------------------------
// MFMA recurrences. (Round 2's FMA kernels -- removed in round 5 -- read the hidden tile with broadcast ds_read_b128 -- every lane receives
// the same 16 bytes, and the LDS->VGPR return path (128 B/clk per CU) delivers 1 KiB per read: 4 waves x 16 reads x
// TILE rows x 8 clk = 2-4 k clk per timestep, more than the FMAs themselves (measured 1.76 us per step at B = 1 280).)
// v_mfma_f32_4x4x1_16b_f32 takes the same product from per-lane operands instead: 16 independent 4x4 outer products
// per instruction, block b = lanes 4b..4b+3; A[i] and B[j] come from lane 4b+i / 4b+j, D[i][j] lands in register i of
// lane 4b+j. Exact f32 (an fmaf chain), 8 clk per instruction.
//
//   forward : block b of wave w = hidden unit u = 16w + b; A[i] = W_hh[gate i][u][k], B[j] = h[row j][k]; after the k
//             loop lane (b, j) holds the four gate pre-activations of (row j, unit u) in its four accumulator
//             registers, so the pointwise cell update stays in registers (no gate tile in LDS, one barrier per step).
//   backward: wave (q, uh) multiplies a quarter of the d-gate columns with its 64 units' slice of W_hh:
//             A[i] = W_hh[col c][64uh + 4b + i], B[j] = dpre[row j][col c]; the four partial sums meet in LDS.
//
// Gate layout (EGP_LSTM_GATES_UNIT_MAJOR): gates_x, the saved gates and d_pre are [T][B][unit][gate] -- the four gates
// of a unit are one 16-byte access of the lane that owns them (1 load + 3 stores per step instead of 4 + 6; the stores
// were a third of the step). The caller permutes the rows of W_ih / its bias once (256 x D) and un-permutes dW.
// What bounds a step now (ablations on the MI355X, one workgroup per CU): LDS latency of the hidden-tile reads and the
// barrier ~0.25 us, stores ~0.1 us, the 64 MFMAs hide behind both. NQ = row quads per workgroup.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The save-set (activated gates, cell states: written once by the forward sweep, read once by the backward sweep a few
// milliseconds later, 1.3 GB per grouped sweep of the update) and d_pre go past the caches with the non-temporal hint
// (EGP_LSTM_NT bit 0: save-set stores and loads, bit 1: d_pre stores; default both. A/B/C inside one lease, tools/probes/abn.sh +
// update_time.py, three rounds: T_update 44.85 / 44.55 / 44.84 ms without, 44.29 / 45.17 / 44.86 with bit 0, 44.04 / 44.31 / 44.56 with both.)
#ifndef EGP_LSTM_NT
#define EGP_LSTM_NT 3
#endif
template <typename V> __device__ __forceinline__ void st_save(V *p, V v) {
    if constexpr ((EGP_LSTM_NT & 1) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}
template <typename V> __device__ __forceinline__ V ld_save(const V *p) {
    if constexpr ((EGP_LSTM_NT & 1) != 0) return __builtin_nontemporal_load(p); else return *p;
}
template <typename V> __device__ __forceinline__ void st_dpre(V *p, V v) {
    if constexpr ((EGP_LSTM_NT & 2) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}

// Several sweeps over the same (T, B) in one launch (blockIdx.y = problem): the directions of a bi-LSTM, or the LSTMs of
// the critic's and the actor's video nets -- a sweep is latency-bound and 320 workgroups leave 64 CUs with double
// duty; 4 x 320 = 5 per CU. Problems share one gate buffer [T*B][P*4H] (columns p*4H.. belong to problem p: what ONE
// input-projection GEMM against the stacked W_ih produces) and differ in W_hh, direction and output slabs.
struct LstmGroup {
    int ld_g;               // row stride of gates_x / gates_save / d_pre (floats): P * 4H
    int rev_mask;           // bit p: problem p runs t = T-1 .. 0
    int ld_h;               // row stride of h_out (floats): H, or 2H when two directions share one [T][B][2H] buffer
    int ld_dh;              // ... of d h_out
    long c_stride;          // floats between the cells_save slabs of consecutive problems
    float *h[4];            // forward: h_out per problem (row (t, b) at h[p] + (t*B + b) * ld_h)
    const float *dh[4];     // backward: d h_out per problem
    float *db;              // backward, optional: [P][4H] accumulates sum_t,b d_pre (atomics; zeroed by the caller)
    // optional (egp_lstm_group_*_len_f32): ragged batches. order[p] = sequence (column b of the [T][B] layout) at position p,
    // workgroups take ROWS consecutive positions; steps[p] = time steps the sequence at position p needs FROM THE START
    // (its outputs at t >= steps[p] are never read). A problem that runs forward in time stops at the longest of its rows
    // (h_out / d_pre of the skipped steps are written as zeros); problems that run backward in time go through all T
    // steps -- their state at the last needed step depends on everything after it. Sorting the positions by steps makes the
    // rows of a workgroup alike.
    const int *order;
    const int *steps;
    int leave_skipped;     // the skipped steps' h_out / d_pre stay unwritten (the caller never reads those rows)
    // optional (forward): the input projection as a table over the dataset's FRAMES instead of over the (t, b) rows of the padded
    // windows. A sequence is a window of consecutive frames, so x[t][b] = frame (seq_base[b] + t) and its projection row is
    // gates_x[seq_base[b] + t]: the projection GEMM runs over the unique frames once (16 k rows for the bench's dataset) instead
    // of over every window row (154 k), and the sweeps read an L2-resident table instead of streaming 630 MB.
    const int *seq_base;
};

// steps a forward-running workgroup has to make: the longest of its rows
template <int ROWS>
__device__ __forceinline__ int lstm_group_steps(const LstmGroup &grp, int reverse, int r0, int B, int T) {
    if (reverse || !grp.steps) return T;
    int mx = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
        if (r0 + i < B) mx = max(mx, grp.steps[r0 + i]);
    return min(T, mx);
}

// FULL = every row of every workgroup exists (B % ROWS == 0), TRAIN = gates / cells are saved. Both are template
// parameters so that the timestep body is straight-line code: with a branch around any load or store the compiler's
// s_waitcnt pass no longer knows how many memory operations are outstanding and waits for ALL of them (vmcnt(0))
// before touching a prefetched register -- i.e. for the acknowledgement of the stores it has just issued, every step.
// tools/probes/lstm_trace.hip compiles this file with EGP_LSTM_TRACE = a workgroup index: per-phase cycle sums of wave 0 of that
// workgroup of the LAST problem of a grouped forward launch (a problem that runs backward in time: every step; the kernels
// walk the problems from the last one down, prob = gridDim.y - 1 - blockIdx.y, so that is blockIdx.y == 0)
#ifdef EGP_LSTM_TRACE
__device__ long long g_lstm_trace[8];
#define EGP_LT_DECL long long lt_acc[6] = {0, 0, 0, 0, 0, 0}, lt_prev = 0; const bool lt_on = blockIdx.x == EGP_LSTM_TRACE && blockIdx.y == 0 && threadIdx.x == 0
#define EGP_LT_START if (lt_on) lt_prev = (long long)__builtin_readcyclecounter()
#define EGP_LT(i) if (lt_on) { const long long lt_now = (long long)__builtin_readcyclecounter(); lt_acc[i] += lt_now - lt_prev; lt_prev = lt_now; }
#define EGP_LT_END(steps) if (lt_on) { for (int i = 0; i < 6; ++i) g_lstm_trace[i] = lt_acc[i]; g_lstm_trace[6] = (steps); }
#else
#define EGP_LT_DECL do { } while (0)
#define EGP_LT_START do { } while (0)
#define EGP_LT(i) do { } while (0)
#define EGP_LT_END(steps) do { } while (0)
#endif
#ifndef EGP_LSTM_PD1
#define EGP_LSTM_PD1 (NQ == 1 ? 4 : 4)
#endif
#ifndef EGP_LSTM_KCH
#define EGP_LSTM_KCH (NQ == 1 && LH == 64 ? 32 : 64)
#endif
#ifndef EGP_LSTM_BPD
#define EGP_LSTM_BPD ((NQ == 1 && LH == 64) ? 4 : 2)
#endif
#ifndef EGP_LSTM_BKCH
#define EGP_LSTM_BKCH 64
#endif
template <int NQ, int LH, bool FULL, bool TRAIN>
__global__ __launch_bounds__(4 * LH) void k_lstm_fwd_mfma(const float *__restrict__ gx, const float *__restrict__ w_hh, int T, int B,
                                                          LstmGroup grp, float *__restrict__ gates_out, float *__restrict__ c_out) {
    constexpr int LG = 4 * LH, ROWS = 4 * NQ;
    // Workgroups are dispatched x first, then y, and a grouped launch is a few rounds of the chip (1 100 workgroups on 768 / 512
    // slots at the update's shapes): the LAST problems' workgroups form the tail. The callers put the forward-running problems
    // (ragged: short) first and the backward-running ones (every step) last, so the y index is walked downwards -- the long
    // workgroups start first and the short ones fill the tail (longest-processing-time-first): 0.508 -> 0.493 ms forward,
    // 0.573 -> 0.555 ms backward per grouped sweep of the update
    const int prob = (int)(gridDim.y - 1 - blockIdx.y), reverse = (grp.rev_mask >> prob) & 1, ld = grp.ld_g, ld_h = grp.ld_h;
    float *__restrict__ h_out = grp.h[prob];
    gx += prob * LG; w_hh += (long)prob * LG * LH;
    if (TRAIN) { gates_out += prob * LG; c_out += prob * grp.c_stride; }
    constexpr int PD = LH == 128 ? 2 : EGP_LSTM_PD1;   // steps per unrolled iteration = input-projection tiles in flight (even)
    __shared__ __attribute__((aligned(16))) float s_h[2][ROWS][LH + 4];     // +4: the 4 rows of a read hit distinct banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = 16 * wave + (lane >> 2), sub = lane & 3;                   // sub = gate index as A, row-in-quad as B / D
    const int r0 = blockIdx.x * ROWS;
    float w[LH];
#pragma unroll
    for (int k = 0; k < LH; ++k) w[k] = w_hh[(long)(sub * LH + u) * LH + k];
    float cst[NQ];
    f32x4 pre[PD][NQ];
    bool live[NQ];
    int rowc[NQ];               // row to read: a missing row of a ragged last tile reads the last real row instead
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        cst[q] = 0.f;
        live[q] = FULL || r0 + 4 * q + sub < B;
        const int pos = live[q] ? r0 + 4 * q + sub : B - 1;
        rowc[q] = grp.order ? grp.order[pos] : pos;
    }
    // row of the projection a step reads: f_t * gmul + gadd[q] -- (t, b) rows of a dense buffer, or frames of a table
    const long gmul = grp.seq_base ? 1 : B;
    long gadd[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) gadd[q] = grp.seq_base ? (long)grp.seq_base[rowc[q]] : (long)rowc[q];
    const int Tp = lstm_group_steps<ROWS>(grp, reverse, r0, B, T);       // (wave-uniform)
    for (int i = threadIdx.x; i < 2 * ROWS * (LH + 4); i += 4 * LH) (&s_h[0][0][0])[i] = 0.f;

#define EGP_LSTM_FETCH_GX(STEP, DST)                                                                 \
    {                                                                                                \
        const int f_s = (STEP) < T ? (STEP) : T - 1;                                                 \
        const int f_t = reverse ? T - 1 - f_s : f_s;                                                 \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                               \
            DST[q] = *reinterpret_cast<const f32x4 *>(gx + ((long)f_t * gmul + gadd[q]) * ld + 4 * u); \
    }

    // one timestep: products, cell update in registers, stores, barrier
    EGP_LT_DECL;
    auto do_step = [&](const int step, const int par, const f32x4 (&init)[NQ]) {
        const int t = reverse ? T - 1 - step : step;
        EGP_LT(0);                                           // since the end of the previous step: loop overhead, prefetch issue, waits on the tile
        // the whole hidden tile of this lane's rows first (16 reads per quad in flight, one exposed LDS latency) ...
        // (64 hidden columns at a time: with LH = 128 the registers do not hold the whole row next to W_hh)
        f32x4 acc[NQ], acc2[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { acc[q] = init[q]; acc2[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        constexpr int KCH = EGP_LSTM_KCH;
#pragma unroll
        for (int kc = 0; kc < LH; kc += KCH) {
            float4 hv[NQ][KCH / 4];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int k4 = 0; k4 < KCH / 4; ++k4) hv[q][k4] = *reinterpret_cast<const float4 *>(&s_h[par][4 * q + sub][kc + 4 * k4]);
            __builtin_amdgcn_sched_barrier(0);
            EGP_LT(1);                                       // hidden-tile reads issued (not yet waited for)
            // ... then the products; two accumulator chains per quad cover the dependent-issue latency (four: no gain)
#pragma unroll
            for (int k4 = 0; k4 < KCH / 4; ++k4)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * k4 + 0], hv[q][k4].x, acc[q], 0, 0, 0);
                    acc2[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * k4 + 1], hv[q][k4].y, acc2[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * k4 + 2], hv[q][k4].z, acc[q], 0, 0, 0);
                    acc2[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * k4 + 3], hv[q][k4].w, acc2[q], 0, 0, 0);
                }
        }
        EGP_LT(2);                                           // products issued
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 a = acc[q] + acc2[q];
            const float ig = sigmoidf_(a[0]), fg = sigmoidf_(a[1]), gg = tanhf_(a[2]), og = sigmoidf_(a[3]);
            const float cn = fg * cst[q] + ig * gg;
            const float hn = og * tanhf_(cn);
            cst[q] = cn;
            s_h[par ^ 1][4 * q + sub][u] = hn;
            if (FULL || live[q]) {
                const long row = (long)t * B + rowc[q];
                h_out[row * ld_h + u] = hn;
                if (TRAIN) {
                    st_save(reinterpret_cast<f32x4 *>(gates_out + row * ld + 4 * u), f32x4{ig, fg, gg, og});
                    st_save(c_out + row * LH + u, cn);
                }
            }
        }
        EGP_LT(3);                                           // cell update, LDS / global stores issued
        __syncthreads();       // the whole new hidden tile before the next step's products (s_h is double-buffered)
        EGP_LT(4);                                           // barrier
    };

#pragma unroll
    for (int d = 0; d < PD; ++d) EGP_LSTM_FETCH_GX(d, pre[d])
    __syncthreads();
    EGP_LT_START;
    int s0 = 0;
    for (; s0 + PD <= Tp; s0 += PD) {
        // this iteration's tiles move to `cur`, then ALL loads of the next iteration are issued before the first step:
        // when they are needed (next iteration) only operations issued after them -- this iteration's stores -- may
        // still be outstanding, so the in-order vmcnt wait never stalls on a young load or store. No exit from the
        // middle of the iteration either (the remainder of T / PD runs in the plain loop below): every path into the
        // wait must have issued the same operations.
        f32x4 cur[PD][NQ];
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int q = 0; q < NQ; ++q) cur[d][q] = pre[d][q];
#pragma unroll
        for (int d = 0; d < PD; ++d) EGP_LSTM_FETCH_GX(s0 + PD + d, pre[d])
        // (past the end the fetch repeats the last step: the count of loads per iteration stays fixed.
        //  In training gx doubles as gates_out: a lane re-reads only what it wrote itself, or has not reached yet)
#pragma unroll
        for (int d = 0; d < PD; ++d) do_step(s0 + d, d & 1, cur[d]);
    }
    for (int d = 0; s0 + d < Tp; ++d) {         // Tp % PD last steps: their tiles are pre[0..] (fetched, not clamped)
        f32x4 last[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) last[q] = d == 0 ? pre[0][q] : d == 1 ? pre[1 % PD][q] : d == 2 ? pre[2 % PD][q] : d == 3 ? pre[3 % PD][q]
                                             : d == 4 ? pre[4 % PD][q] : d == 5 ? pre[5 % PD][q] : pre[6 % PD][q];
        do_step(s0 + d, d & 1, last);
    }
#undef EGP_LSTM_FETCH_GX
    EGP_LT_END(Tp);
    // the steps a ragged forward-running workgroup skipped: zeros (a weight gradient multiplies these rows with a zero d_pre,
    // and 0 * whatever an uninitialised buffer holds may be NaN)
    // (leave_skipped: zeros only up to the longest sequence of the aligned group of 8 positions -- the granularity of the
    //  caller's row lists, which must not meet an unwritten row --, nothing beyond)
    const int Tz = grp.leave_skipped ? lstm_group_steps<8>(grp, reverse, r0 & ~7, B, T) : T;
    for (int t = Tp; t < Tz; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (FULL || live[q]) h_out[((long)t * B + rowc[q]) * ld_h + u] = 0.f;
}

template <int NQ, int LH, bool FULL>
__global__ __launch_bounds__(4 * LH) void k_lstm_bwd_mfma(const float *__restrict__ gates, const float *__restrict__ cells,
                                                          const float *__restrict__ w_hh, int T, int B, LstmGroup grp,
                                                          float *__restrict__ dpre) {
    constexpr int LG = 4 * LH, NT = 4 * LH, ROWS = 4 * NQ, UH = LH / 64;
    const int prob = (int)(gridDim.y - 1 - blockIdx.y), reverse = (grp.rev_mask >> prob) & 1, ld = grp.ld_g, ld_dh = grp.ld_dh;     // (as in the forward kernel)
    const float *__restrict__ dh_out = grp.dh[prob];
    gates += prob * LG; dpre += prob * LG; cells += prob * grp.c_stride; w_hh += (long)prob * LG * LH;
    constexpr int NP = (ROWS * LH + NT - 1) / NT;     // = NQ: (row, unit) pairs per thread in the pointwise phase
    constexpr int PD = EGP_LSTM_BPD;   // steps per unrolled iteration = operand sets in flight
    __shared__ __attribute__((aligned(16))) float s_d[ROWS][LG + 4];     // d-gates of the step, [row][unit][gate]
    __shared__ float s_part[4][ROWS][LH + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = wave / UH, uh = wave % UH, sub = lane & 3;
    const int r0 = blockIdx.x * ROWS;
    // wave (q, uh): d-gate columns [q*LH, (q+1)*LH) of the unit-major layout = units q*LH/4 .. , all four gates;
    // column n = 4*unit + gate is row gate*LH + unit of W_hh
    float w[LH];
#pragma unroll
    for (int cc = 0; cc < LH; ++cc) {
        const int n = q * LH + cc;
        w[cc] = w_hh[(long)((n & 3) * LH + (n >> 2)) * LH + 64 * uh + lane];
    }
    float dc_next[NP], dh_rec[NP];
    f32x4 dsum[NP];                    // sum over time of this thread's d-gates: the bias gradient, reduced at the end
    f32x4 pg[PD][NP];
    float pc[PD][NP], pdh[PD][NP];     // activated gates, cell state, dh of the staged steps
    bool live[NP];
    int rowc[NP];
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) {
        dc_next[qq] = 0.f; dh_rec[qq] = 0.f; dsum[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int r = (threadIdx.x + NT * qq) / LH;
        live[qq] = FULL || r0 + r < B;
        const int pos = live[qq] ? r0 + r : B - 1;
        rowc[qq] = grp.order ? grp.order[pos] : pos;
    }
    // ragged forward-running problems: the steps the forward sweep skipped carry no gradient (their d_pre is written as zeros)
    const int Tp = lstm_group_steps<ROWS>(grp, reverse, r0, B, T);
    const int Tz = grp.leave_skipped ? lstm_group_steps<8>(grp, reverse, r0 & ~7, B, T) : T;     // (as in the forward kernel)
    for (int t = Tp; t < Tz; ++t)
#pragma unroll
        for (int qq = 0; qq < NP; ++qq)
            if (FULL || live[qq])
                *reinterpret_cast<f32x4 *>(dpre + ((long)t * B + rowc[qq]) * ld + 4 * ((threadIdx.x + NT * qq) % LH)) = f32x4{0.f, 0.f, 0.f, 0.f};

#define EGP_LSTM_FETCH(STEP, D)                                                                      \
    {                                                                                                \
        const int f_s = (STEP) > 0 ? (STEP) : 0;                                                     \
        const int f_t = reverse ? T - 1 - f_s : f_s;                                                 \
        _Pragma("unroll") for (int qq = 0; qq < NP; ++qq) {                                          \
            const int j = (threadIdx.x + NT * qq) % LH;                                              \
            const long row = (long)f_t * B + rowc[qq];                                               \
            pg[D][qq] = ld_save(reinterpret_cast<const f32x4 *>(gates + row * ld + 4 * j));         \
            pc[D][qq] = ld_save(cells + row * LH + j);                                               \
            pdh[D][qq] = dh_out[row * ld_dh + j];                                                    \
        }                                                                                            \
    }

    // one timestep: d-gates of the step (pointwise), then the recurrent product dgates W_hh through the matrix cores
    auto do_step = [&](const int step, const f32x4 (&g4)[NP], const float (&cc_)[NP], const float (&cdh)[NP], const float (&cprev)[NP]) {
        const int t = reverse ? T - 1 - step : step;
#pragma unroll
        for (int qq = 0; qq < NP; ++qq) {
            const int p = threadIdx.x + NT * qq, r = p / LH, j = p % LH;
            const float ig = g4[qq][0], fg = g4[qq][1], gg = g4[qq][2], og = g4[qq][3];
            const float c_prev = step > 0 ? cprev[qq] : 0.f;
            const float tc = tanhf_(cc_[qq]);
            const float dh = (FULL || live[qq] ? cdh[qq] : 0.f) + dh_rec[qq];
            const float dc = dh * og * (1.f - tc * tc) + dc_next[qq];
            const float d_o = dh * tc * og * (1.f - og);
            const float di = dc * gg * ig * (1.f - ig);
            const float df = dc * c_prev * fg * (1.f - fg);
            const float dg = dc * ig * (1.f - gg * gg);
            dc_next[qq] = dc * fg;
            const f32x4 d4 = f32x4{di, df, dg, d_o};
            if (FULL || live[qq]) { st_dpre(reinterpret_cast<f32x4 *>(dpre + ((long)t * B + rowc[qq]) * ld + 4 * j), d4); dsum[qq] += d4; }
            *reinterpret_cast<f32x4 *>(&s_d[r][4 * j]) = d4;
        }
        __syncthreads();
        f32x4 acc[NQ], acc2[NQ];
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) acc[qd] = acc2[qd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < LH; kc += EGP_LSTM_BKCH) {
            float4 dv[NQ][EGP_LSTM_BKCH / 4];
#pragma unroll
            for (int qd = 0; qd < NQ; ++qd)
#pragma unroll
                for (int c4 = 0; c4 < EGP_LSTM_BKCH / 4; ++c4) dv[qd][c4] = *reinterpret_cast<const float4 *>(&s_d[4 * qd + sub][q * LH + kc + 4 * c4]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c4 = 0; c4 < EGP_LSTM_BKCH / 4; ++c4)
#pragma unroll
                for (int qd = 0; qd < NQ; ++qd) {
                    acc[qd] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * c4 + 0], dv[qd][c4].x, acc[qd], 0, 0, 0);
                    acc2[qd] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * c4 + 1], dv[qd][c4].y, acc2[qd], 0, 0, 0);
                    acc[qd] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * c4 + 2], dv[qd][c4].z, acc[qd], 0, 0, 0);
                    acc2[qd] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kc + 4 * c4 + 3], dv[qd][c4].w, acc2[qd], 0, 0, 0);
                }
        }
        // D register i of lane (b, j) = partial dh_rec of (row j, unit 64uh + 4b + i)
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            const f32x4 a = acc[qd] + acc2[qd];
#pragma unroll
            for (int i = 0; i < 4; ++i) s_part[q][4 * qd + sub][64 * uh + (lane & ~3) + i] = a[i];
        }
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NP; ++qq) {
            const int p = threadIdx.x + NT * qq, r = p / LH, j = p % LH;
            dh_rec[qq] = s_part[0][r][j] + s_part[1][r][j] + s_part[2][r][j] + s_part[3][r][j];
        }
        // no third barrier: s_d is rewritten only after every wave passed the barrier above (its reads are done),
        // s_part only after the next step's first barrier, which a wave reaches after these reads
    };

#pragma unroll
    for (int d = 0; d < PD; ++d) EGP_LSTM_FETCH(Tp - 1 - d, d)
    int s0 = Tp - 1;
    for (; s0 - PD + 1 >= 0; s0 -= PD) {
        // as in the forward kernel: this iteration's operands move to `c*`, all loads of the next iteration go out
        // first, and nothing leaves the iteration half way
        f32x4 cg[PD][NP];
        float cc_[PD][NP], cdh[PD][NP];
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int qq = 0; qq < NP; ++qq) {
                cg[d][qq] = pg[d][qq];
                cc_[d][qq] = pc[d][qq];
                cdh[d][qq] = pdh[d][qq];
            }
#pragma unroll
        for (int d = 0; d < PD; ++d) EGP_LSTM_FETCH(s0 - PD - d, d)      // (clamped at step 0: unused, fixed load count)
#pragma unroll
        for (int d = 0; d < PD; ++d)      // cell state of step - 1: the next stage of this iteration, or the first of the next
            do_step(s0 - d, cg[d], cc_[d], cdh[d], d + 1 < PD ? cc_[d + 1 < PD ? d + 1 : 0] : pc[0]);
    }
#pragma unroll
    for (int d = 0; d < PD - 1; ++d)      // T % PD last steps, staged in p*[0..]
        if (s0 - d >= 0) do_step(s0 - d, pg[d], pc[d], pdh[d], pc[d + 1]);
#undef EGP_LSTM_FETCH
    if (grp.db) {                         // db[4*unit + gate] += sum over this workgroup's rows (s_d is free again)
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NP; ++qq) {
            const int p = threadIdx.x + NT * qq;
            *reinterpret_cast<f32x4 *>(&s_d[p / LH][4 * (p % LH)]) = dsum[qq];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < LG; c += NT) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) a += s_d[r][c];
            atomicAdd(grp.db + prob * LG + c, a);
        }
    }
}

}  // namespace egp

using namespace egp;

// row quads per workgroup of the hidden-64 kernels (4-row workgroups below 4 096 sequences: several per CU, their LDS / barrier
// latencies overlap)
static int lstm_quads(int B) { return B >= 4096 ? 2 : 1; }

static int lstm_launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EGP_E_HIP;
    }
    return EGP_OK;
}

template <int NQ, int LH>
static void launch_fwd_mfma_t(bool full, bool train, int P, const float *gx, const float *w_hh, int T, int B, const LstmGroup &g,
                              float *gates_save, float *cells_save, hipStream_t s) {
    const dim3 grid((B + 4 * NQ - 1) / (4 * NQ), P), block(4 * LH);
    if (full && train) k_lstm_fwd_mfma<NQ, LH, true, true><<<grid, block, 0, s>>>(gx, w_hh, T, B, g, gates_save, cells_save);
    else if (full) k_lstm_fwd_mfma<NQ, LH, true, false><<<grid, block, 0, s>>>(gx, w_hh, T, B, g, gates_save, cells_save);
    else if (train) k_lstm_fwd_mfma<NQ, LH, false, true><<<grid, block, 0, s>>>(gx, w_hh, T, B, g, gates_save, cells_save);
    else k_lstm_fwd_mfma<NQ, LH, false, false><<<grid, block, 0, s>>>(gx, w_hh, T, B, g, gates_save, cells_save);
}
static void launch_fwd_mfma(int hidden, int P, const float *gx, const float *w_hh, int T, int B, const LstmGroup &g,
                            float *gates_save, float *cells_save, hipStream_t s) {
    const int nq = hidden == 64 ? lstm_quads(B) : 1;      // (4-row workgroups also win for grouped launches: measured)
    const bool full = B % (4 * nq) == 0, train = gates_save != nullptr;
    if (hidden == 128) launch_fwd_mfma_t<1, 128>(full, train, P, gx, w_hh, T, B, g, gates_save, cells_save, s);
    else if (nq >= 2) launch_fwd_mfma_t<2, 64>(full, train, P, gx, w_hh, T, B, g, gates_save, cells_save, s);
    else launch_fwd_mfma_t<1, 64>(full, train, P, gx, w_hh, T, B, g, gates_save, cells_save, s);
}

template <int NQ, int LH>
static void launch_bwd_mfma_t(bool full, int P, const float *gates_save, const float *cells_save, const float *w_hh, int T, int B,
                              const LstmGroup &g, float *d_pre, hipStream_t s) {
    const dim3 grid((B + 4 * NQ - 1) / (4 * NQ), P), block(4 * LH);
    if (full) k_lstm_bwd_mfma<NQ, LH, true><<<grid, block, 0, s>>>(gates_save, cells_save, w_hh, T, B, g, d_pre);
    else k_lstm_bwd_mfma<NQ, LH, false><<<grid, block, 0, s>>>(gates_save, cells_save, w_hh, T, B, g, d_pre);
}
static void launch_bwd_mfma(int hidden, int P, const float *gates_save, const float *cells_save, const float *w_hh, int T, int B,
                            const LstmGroup &g, float *d_pre, hipStream_t s) {
    const int nq = hidden == 64 ? lstm_quads(B) : 1;      // (4-row workgroups also win for grouped launches: measured)
    const bool full = B % (4 * nq) == 0;
    if (hidden == 128) launch_bwd_mfma_t<1, 128>(full, P, gates_save, cells_save, w_hh, T, B, g, d_pre, s);
    else if (nq >= 2) launch_bwd_mfma_t<2, 64>(full, P, gates_save, cells_save, w_hh, T, B, g, d_pre, s);
    else launch_bwd_mfma_t<1, 64>(full, P, gates_save, cells_save, w_hh, T, B, g, d_pre, s);
}

extern "C" {

int32_t egp_lstm_gate_layout(void) { return EGP_LSTM_GATES_UNIT_MAJOR; }

int egp_lstm_fwd_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t reverse,
                     float *h_out, float *gates_save, float *cells_save, void *stream) {
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(gates_x && w_hh && h_out, "NULL pointer");
    EGP_REQUIRE((gates_save == nullptr) == (cells_save == nullptr), "gates_save and cells_save go together");
    hipStream_t s = (hipStream_t)stream;
    LstmGroup g = {};
    g.ld_g = 4 * hidden; g.rev_mask = reverse ? 1 : 0; g.ld_h = hidden; g.h[0] = h_out;
    launch_fwd_mfma(hidden, 1, gates_x, w_hh, T, B, g, gates_save, cells_save, s);
    return lstm_launch_check("k_lstm_fwd");
}

int egp_lstm_bwd_f32(const float *dh_out, const float *gates_save, const float *cells_save, const float *w_hh, int32_t T, int32_t B,
                     int32_t hidden, int32_t reverse, float *d_pre, void *stream) {
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(dh_out && gates_save && cells_save && w_hh && d_pre, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    LstmGroup g = {};
    g.ld_g = 4 * hidden; g.rev_mask = reverse ? 1 : 0; g.ld_dh = hidden; g.dh[0] = dh_out;
    launch_bwd_mfma(hidden, 1, gates_save, cells_save, w_hh, T, B, g, d_pre, s);
    return lstm_launch_check("k_lstm_bwd");
}

int egp_lstm_group_fwd_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t n_problems,
                           int32_t reverse_mask, float *const *h_out, int32_t ld_h, float *gates_save, float *cells_save, void *stream) {
    return egp_lstm_group_fwd_len_f32(gates_x, w_hh, T, B, hidden, n_problems, reverse_mask, h_out, ld_h, gates_save, cells_save, nullptr, nullptr, 0,
                                      nullptr, stream);
}

int egp_lstm_group_fwd_len_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t n_problems,
                               int32_t reverse_mask, float *const *h_out, int32_t ld_h, float *gates_save, float *cells_save,
                               const int32_t *seq_order, const int32_t *seq_steps, int32_t leave_skipped, const int32_t *seq_base,
                               void *stream) {
    EGP_REQUIRE((seq_order == nullptr) == (seq_steps == nullptr), "seq_order and seq_steps go together");
    EGP_REQUIRE(!seq_base || !gates_save || gates_save != gates_x, "a frame table cannot double as the saved gates (rows are shared between sequences)");
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(n_problems >= 1 && n_problems <= 4, "1..4 problems per group");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(gates_x && w_hh && h_out && ld_h >= hidden, "NULL pointer / h_out row stride below the hidden size");
    EGP_REQUIRE((gates_save == nullptr) == (cells_save == nullptr), "gates_save and cells_save go together");
    LstmGroup g = {};
    g.ld_g = 4 * hidden * n_problems; g.rev_mask = reverse_mask; g.ld_h = ld_h; g.c_stride = (long)T * B * hidden;
    for (int p = 0; p < n_problems; ++p) {
        EGP_REQUIRE(h_out[p], "NULL h_out");
        g.h[p] = h_out[p];
    }
    g.order = seq_order; g.steps = seq_steps; g.leave_skipped = seq_steps && leave_skipped; g.seq_base = seq_base;
    launch_fwd_mfma(hidden, n_problems, gates_x, w_hh, T, B, g, gates_save, cells_save, (hipStream_t)stream);
    return lstm_launch_check("k_lstm_fwd_mfma (group)");
}

int egp_lstm_group_bwd_f32(const float *const *dh_out, int32_t ld_dh, const float *gates_save, const float *cells_save, const float *w_hh,
                           int32_t T, int32_t B, int32_t hidden, int32_t n_problems, int32_t reverse_mask, float *d_pre, float *d_bias,
                           void *stream) {
    return egp_lstm_group_bwd_len_f32(dh_out, ld_dh, gates_save, cells_save, w_hh, T, B, hidden, n_problems, reverse_mask, d_pre, d_bias, nullptr,
                                      nullptr, 0, stream);
}

int egp_lstm_group_bwd_len_f32(const float *const *dh_out, int32_t ld_dh, const float *gates_save, const float *cells_save, const float *w_hh,
                               int32_t T, int32_t B, int32_t hidden, int32_t n_problems, int32_t reverse_mask, float *d_pre, float *d_bias,
                               const int32_t *seq_order, const int32_t *seq_steps, int32_t leave_skipped, void *stream) {
    EGP_REQUIRE((seq_order == nullptr) == (seq_steps == nullptr), "seq_order and seq_steps go together");
    EGP_REQUIRE(hidden == 64 || hidden == 128, "egp_lstm kernels are built for hidden size 64 and 128");
    EGP_REQUIRE(n_problems >= 1 && n_problems <= 4, "1..4 problems per group");
    EGP_REQUIRE(T >= 0 && B >= 0, "negative size");
    if (T == 0 || B == 0) return EGP_OK;
    EGP_REQUIRE(dh_out && gates_save && cells_save && w_hh && d_pre && ld_dh >= hidden, "NULL pointer / d h_out row stride below the hidden size");
    LstmGroup g = {};
    g.ld_g = 4 * hidden * n_problems; g.rev_mask = reverse_mask; g.ld_dh = ld_dh; g.c_stride = (long)T * B * hidden; g.db = d_bias;
    for (int p = 0; p < n_problems; ++p) {
        EGP_REQUIRE(dh_out[p], "NULL d h_out");
        g.dh[p] = dh_out[p];
    }
    g.order = seq_order; g.steps = seq_steps; g.leave_skipped = seq_steps && leave_skipped;
    launch_bwd_mfma(hidden, n_problems, gates_save, cells_save, w_hh, T, B, g, d_pre, (hipStream_t)stream);
    return lstm_launch_check("k_lstm_bwd_mfma (group)");
}

}  // extern "C"
