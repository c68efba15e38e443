// egp_gemm.hip -- float32 GEMMs of the PPO update on the bf16 matrix cores (gfx950), float32-class accuracy.
//
// The update's products (core/policy_gaussian.py:19-24 + models/mlp.py:22-25 evaluated for the whole batch, the LSTM
// input projection of models/rnn.py:45-61, and their gradients) are "one huge, two small" shapes: 134 k x 243 x 300,
// 150 k x 128 x 1024, ... In float32 they run at the float32 MFMA rate, 1/16 of the bf16 rate, and gfx950 has no
// xf32 / TF32 mode. Here every float32 operand is split on its way into LDS into a bf16 head and a bf16 tail
// (x = hi + lo + O(2^-17 |x|)) and a product is three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in float32: about 16
// mantissa bits per product (the dropped lo*lo term is 2^-18), i.e. float32-class results at a third of the bf16 rate,
// five times the float32 rate. `terms = 1` keeps only hi*hi (plain bf16 inputs).
//
//   C[M][N] = A[M][K] * B[K][N]      A given as [m][k] (k contiguous) or [k][m] (m contiguous),
//                                    B given as [n][k] (k contiguous) or [k][n] (n contiguous)
//   epilogue: + bias[n], ReLU, * (mask[m][n] > 0)       (bias + activation forward, dReLU in a data gradient)
//   split-K:  partial sums go to a workspace, a second launch reduces them in a fixed order (deterministic); used by
//             the weight gradients (K = batch rows). A virtual column of ones appended to B yields the bias gradient
//             (column sums of A^T) from the same launch.
//
// Tiling: 256 threads = 4 waves, tile 128 x BN x 32 (BN = 128: waves 2 x 2, each 64 x 64 = 2 x 2 MFMA blocks of
// v_mfma_f32_32x32x16_bf16; BN = 64: waves 4 x 1, each 32 x 64). Operand tiles live in LDS as four k-panels [row][8 bf16]
// (fragment reads and staging stores are contiguous 16-byte accesses, no padding). Global -> register -> LDS staging; for
// operands whose memory is not k-contiguous the transpose is in the register naming (a wave reads 64 consecutive rows of
// one k-row per load). The loads of tile s+2 are issued before tile s is multiplied; one barrier per k-tile.
#include "egp_internal.hpp"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte load, 4-byte aligned
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline int after_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        egp::set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EGP_E_HIP;
    }
    return EGP_OK;
}

constexpr int BM = 128, BK = 32;

// tools/probes/gemm_trace.hip compiles this file with EGP_GEMM_TRACE: cycle stamps of one wave of one workgroup
#ifdef EGP_GEMM_TRACE
__device__ long long g_gemm_trace[1024];
__device__ int g_gemm_trace_n;
#define EGP_TR(tag)                                                                         \
    do {                                                                                    \
        if (blockIdx.x == EGP_GEMM_TRACE && blockIdx.z == 0 && threadIdx.x == 0 && g_gemm_trace_n < 510) { \
            g_gemm_trace[2 * g_gemm_trace_n] = (tag);                                       \
            g_gemm_trace[2 * g_gemm_trace_n + 1] = (long long)__builtin_readcyclecounter(); \
            ++g_gemm_trace_n;                                                               \
        }                                                                                   \
    } while (0)
// the same for the two roles of k_gemm_ws: who = 0 consumer (thread 0), 1 producer (thread 256); separate halves of the
// buffer, the entry counter lives in a register (a counter in memory costs a load round trip per stamp)
__device__ int g_gemm_trace_n2[2];
#define EGP_TRW_DECL int egp_trn = 0
#define EGP_TRW(who, tag)                                                                   \
    do {                                                                                    \
        if (blockIdx.x == (EGP_GEMM_TRACE & 255) && threadIdx.x == (who) * 256 && egp_trn < 250) { \
            g_gemm_trace[(who) * 512 + 2 * egp_trn] = (tag);                                \
            g_gemm_trace[(who) * 512 + 2 * egp_trn + 1] = (long long)__builtin_readcyclecounter(); \
            g_gemm_trace_n2[who] = ++egp_trn;                                               \
        }                                                                                   \
    } while (0)
#else
#define EGP_TR(tag) do { } while (0)
#define EGP_TRW_DECL do { } while (0)
#define EGP_TRW(who, tag) do { } while (0)
#endif
__host__ __device__ constexpr int panel_el(int R) { return R * 8 + 32; }      // bf16 elements per k-panel of an R-row tile
__host__ __device__ constexpr int tile_el(int R) { return 4 * panel_el(R); }

struct GemmArgs {
    int M, N, K;                              // N counts the real columns of B (the virtual ones column is extra)
    const float *A; long lda; int a_kc;       // a_kc: A[m * lda + k], else A[k * lda + m]
    const float *B; long ldb; int b_kc;       // b_kc: B[n * ldb + k], else B[k * ldb + n]
    float *C; long ldc;
    const float *bias; int relu;
    const float *mask; long ldmask;
    int ones_col;                             // B has a virtual column N of ones (its result: column N of the workspace)
    int k_per_split;                          // multiple of BK; gridDim.z splits
    float *ws; int ldws;                      // [splits][M][ldws] when there are k splits or a ones column; ldws = N + ones_col rounded up to 4
    int tiles_m, tiles_n, xcd_order;
    int split_xcd;        // k_gemm_ws, split-K launches: the tiles of one k range go to ONE XCD (they read the same k rows of A and B)
    // fused gather / scatter (egp_gemm_desc: a_rows ... c_rows), k_gemm_ws only
    const long long *a_rows; const float *A2; long lda2; int a_split;
    const long long *a_krows;                 // A given as [k][m]: k-row k lives at row a_krows[k]
    const long long *b_krows; const float *B2; long ldb2; int b_split;
    const long long *c_rows;
    int partial;                              // results go to the workspace (k splits and / or the ones column), reduced by k_gemm_reduce
    int grid_tiles, n_items, zs;              // k_gemm_ws: tile slots (the XCD order pads tiles_m to a multiple of 8), slots x k splits, k splits
};

// LDS image of an operand tile (R rows x 32 k, bf16): four k-panels of [R][8] (+ 64 bytes between panels), element
// (row, k) at (k >> 3) * PANEL(R) + row * 8 + (k & 7). An MFMA fragment read (32 consecutive rows x 8 k per half-wave) is
// 512 contiguous bytes; staging stores are 8- or 16-byte writes of consecutive k of one row, and the panel pad keeps the
// four panels a k-contiguous row is scattered over on different banks.
//
// Staging registers: R / 8 floats per thread and operand, in the shape its memory form loads best.
// All loads are branch-free (the compiler keeps them in flight across the multiply with counted waits): row indices past
// the operand are clamped -- such rows only feed output rows / columns that are never stored. The one k-tile that may
// reach past the end of the k range is handled after the pipelined loop (`ktail`: element-wise loads, zero select).
// k-contiguous memory (P[row * ld + k]): thread t -> k quad t & 7, rows (t >> 3) + 32 u: eight lanes read the 128 bytes one
// row contributes to the tile with one 16-byte load each. Registers: v[4 u + j] = element (row_u, k0 + 4 (t & 7) + j).
template <int R, bool ktail>
__device__ __forceinline__ void load_kc(const float *__restrict__ P, long ld, int rows, int r0, int k0, int kend, float (&v)[R / 8],
                                        const int t = threadIdx.x) {
    const int k = k0 + (t & 7) * 4, rb = t >> 3;
#pragma unroll
    for (int u = 0; u < R / 32; ++u) {
        const int row = min(r0 + rb + 32 * u, rows - 1);
        const float *src = P + (long)row * ld;
        if constexpr (!ktail) {
            const f32x4u x = *(const f32x4u *)(src + k);
            v[4 * u] = x[0]; v[4 * u + 1] = x[1]; v[4 * u + 2] = x[2]; v[4 * u + 3] = x[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = src[min(k + j, kend - 1)];
                v[4 * u + j] = k + j < kend ? x : 0.f;
            }
        }
    }
}

// row-contiguous memory (P[k * ld + row]): lane l -> rows l + 64 u, wave w -> panel w (k0 + 8 w + j): every load instruction
// reads 64 consecutive floats of one k-row; the transpose happens in the register naming.
// `ones_row` >= 0: a virtual row of ones at that index (the bias-gradient column of a weight gradient).
template <int R, bool ktail>
__device__ __forceinline__ void load_rc(const float *__restrict__ P, long ld, int rows, int r0, int k0, int kend, int ones_row,
                                        float (&v)[R / 8], const int t = threadIdx.x) {
    const int l = t & 63, kb = k0 + (t >> 6) * 8;
#pragma unroll
    for (int u = 0; u < R / 64; ++u) {
        const int row = r0 + l + 64 * u;
        const bool one = row == ones_row;
        const float *src = P + min(row, rows - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb + j;
            float x = src[(long)(ktail ? min(k, kend - 1) : k) * ld];
            x = one ? 1.f : x;
            v[8 * u + j] = (!ktail || k < kend) ? x : 0.f;
        }
    }
}

// registers -> LDS: NIMG bf16 images of the tile (head, tail, second tail), `img` elements apart
template <int NIMG>
__device__ __forceinline__ void split_to(float x, __bf16 (&piece)[3]) {
    const __bf16 h = (__bf16)x;                       // v_cvt_pk_bf16_f32: round to nearest even
    piece[0] = h;
    if constexpr (NIMG > 1) {
        const float r1 = x - (float)h;                // exact
        const __bf16 m = (__bf16)r1;
        piece[1] = m;
        if constexpr (NIMG > 2) piece[2] = (__bf16)(r1 - (float)m);
    }
}

// k-contiguous staging: 4 consecutive k of row (t >> 3) + 32 u -> one 8-byte store per image
template <int R, int NIMG>
__device__ __forceinline__ void store_kc(const float (&v)[R / 8], __bf16 *dst, int img, const int t = threadIdx.x) {
    const int kq = t & 7, rb = t >> 3;
#pragma unroll
    for (int u = 0; u < R / 32; ++u) {
        bf16x4 q[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 pc[3];
            split_to<NIMG>(v[4 * u + j], pc);
#pragma unroll
            for (int c = 0; c < NIMG; ++c) q[c][j] = pc[c];
        }
        const int off = (kq >> 1) * panel_el(R) + (rb + 32 * u) * 8 + (kq & 1) * 4;
#pragma unroll
        for (int c = 0; c < NIMG; ++c) *(bf16x4 *)(dst + c * img + off) = q[c];
    }
}

// row-contiguous staging: 8 consecutive k (panel = wave) of row lane + 64 u -> one 16-byte store per image
template <int R, int NIMG>
__device__ __forceinline__ void store_rc(const float (&v)[R / 8], __bf16 *dst, int img, const int t = threadIdx.x) {
    const int l64 = t & 63, panel = t >> 6;
#pragma unroll
    for (int u = 0; u < R / 64; ++u) {
        bf16x8 q[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 pc[3];
            split_to<NIMG>(v[8 * u + j], pc);
#pragma unroll
            for (int c = 0; c < NIMG; ++c) q[c][j] = pc[c];
        }
        const int off = panel * panel_el(R) + (l64 + 64 * u) * 8;
#pragma unroll
        for (int c = 0; c < NIMG; ++c) *(bf16x8 *)(dst + c * img + off) = q[c];
    }
}

// Tile order. Workgroups are dealt round-robin to the 8 XCDs by their linear id. With many m-tiles (activations x
// weights) the n-tiles of one m-tile read the same rows of A: they go to the same XCD (one L2) back to back. With few
// tiles (weight gradients: the parallelism is in the k splits) the plain order spreads them over all XCDs.
__device__ __forceinline__ bool tile_of(const GemmArgs &g, int L, int &tm, int &tn) {
    if (g.xcd_order) {
        const int xcd = L & 7, q = L >> 3;
        tm = (q / g.tiles_n) * 8 + xcd;
        tn = q % g.tiles_n;
    } else {
        tm = L / g.tiles_n;
        tn = L % g.tiles_n;
    }
    return tm < g.tiles_m;
}

template <int BN, int TERMS, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16x(GemmArgs g) {
    // TERMS = 1: bf16 x bf16;  3: two-piece operands, hi*hi + hi*lo + lo*hi (~16 mantissa bits per product);
    // 6: three-piece operands, every cross term down to 2^-16 (float32-class products). The three images of TERMS = 6
    // take one LDS buffer (two barriers per k-tile) so that two workgroups still fit a CU.
    constexpr int NIMG = TERMS == 1 ? 1 : (TERMS == 3 ? 2 : 3);
    constexpr int NBUF = TERMS == 6 ? 1 : 2;
    constexpr int WN = BN == 128 ? 2 : 1;                 // waves along n
    constexpr int MI = BN == 128 ? 2 : 1, NJ = 2;         // MFMA blocks per wave: rows x cols
    constexpr int WROWS = 32 * MI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per buffer: A hi | A lo | B hi | B lo
    constexpr int A_EL = tile_el(BM), B_EL = tile_el(BN);
    constexpr int BUF_EL = NIMG * (A_EL + B_EL);
    __bf16 *base = (__bf16 *)smem;

    int tm, tn;
    if (!tile_of(g, blockIdx.x, tm, tn)) return;
    EGP_TR(1);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nst = (kend - kbeg + BK - 1) / BK;
    const int ones_row = g.ones_col ? g.N : -1;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: the loads of tile s + 2 are issued before tile s is multiplied, so a load has two full
    // iterations to land (the k loops are short and only two workgroups share a CU: latency has to be hidden here).
    // The steady-state loop has no conditional load, so every wait in it is a counted vmcnt.
    float va0[BM / 8], vb0[BN / 8], va1[BM / 8], vb1[BN / 8];
    auto gload = [&](int s, float (&va)[BM / 8], float (&vb)[BN / 8], auto ktail) {
        constexpr bool KT = decltype(ktail)::value;
        const int k0 = kbeg + s * BK;
        if constexpr (A_KC) load_kc<BM, KT>(g.A, g.lda, g.M, m0, k0, kend, va);
        else load_rc<BM, KT>(g.A, g.lda, g.M, m0, k0, kend, -1, va);
        if constexpr (B_KC) load_kc<BN, KT>(g.B, g.ldb, g.N, n0, k0, kend, vb);
        else load_rc<BN, KT>(g.B, g.ldb, g.N, n0, k0, kend, ones_row, vb);
    };
    constexpr std::false_type FULL{};
    constexpr std::true_type TAIL{};
    auto sstore = [&](int buf, const float (&va)[BM / 8], const float (&vb)[BN / 8]) {
        if constexpr (NBUF == 1) __syncthreads();         // the single buffer is still being multiplied
        __bf16 *pa = base + (buf & (NBUF - 1)) * BUF_EL, *pb = pa + NIMG * A_EL;
        if constexpr (A_KC) store_kc<BM, NIMG>(va, pa, A_EL); else store_rc<BM, NIMG>(va, pa, A_EL);
        if constexpr (B_KC) store_kc<BN, NIMG>(vb, pb, B_EL); else store_rc<BN, NIMG>(vb, pb, B_EL);
    };
    const int frow = lane & 31, fkh = lane >> 5;
    auto compute = [&](int buf) {
        const __bf16 *pa = base + (buf & (NBUF - 1)) * BUF_EL, *pb = pa + NIMG * A_EL;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[NIMG][MI], fb[NIMG][NJ];
#pragma unroll
            for (int c = 0; c < NIMG; ++c) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[c][i] = *(const bf16x8 *)(pa + c * A_EL + (2 * ks + fkh) * panel_el(BM) + (wm * WROWS + 32 * i + frow) * 8);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    fb[c][j] = *(const bf16x8 *)(pb + c * B_EL + (2 * ks + fkh) * panel_el(BN) + (wn * 64 + 32 * j + frow) * 8);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x16 a = acc[i][j];                 // smallest terms first
                    if constexpr (NIMG == 3) {
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], a, 0, 0, 0);
                    }
                    if constexpr (NIMG >= 2) {
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], a, 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], a, 0, 0, 0);
                }
        }
    };

    const int nfull = (kend - kbeg) / BK;
    const bool has_tail = (kend - kbeg) % BK != 0;
    int last = 1;                                         // LDS buffer multiplied last (1: none yet, the tail takes buffer 0)
    if (nfull == 1) {
        gload(0, va0, vb0, FULL);
        sstore(0, va0, vb0);
        __syncthreads();
        compute(0);
        last = 0;
    } else if (nfull >= 2) {
        gload(0, va0, vb0, FULL);
        gload(1, va1, vb1, FULL);
        sstore(0, va0, vb0);
        __syncthreads();
        int s = 0;
        while (s + 3 < nfull) {                           // buffer 0 holds tile s, set 1 holds tile s + 1 (in flight)
            // (sched_barrier: the compiler must not hoist the conversions of the in-flight set -- and with them the
            //  wait for its loads -- above the multiply that is there to cover their latency)
            EGP_TR(10);
            gload(s + 2, va0, vb0, FULL);
            __builtin_amdgcn_sched_barrier(0);
            EGP_TR(11);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            EGP_TR(12);
            sstore(1, va1, vb1);
            EGP_TR(13);
            __syncthreads();
            EGP_TR(14);
            gload(s + 3, va1, vb1, FULL);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            EGP_TR(15);
            sstore(0, va0, vb0);
            __syncthreads();
            EGP_TR(16);
            s += 2;
        }
        const bool three = nfull - s == 3;                // two or three tiles left
        if (three) gload(s + 2, va0, vb0, FULL);
        compute(0);
        sstore(1, va1, vb1);
        __syncthreads();
        compute(1);
        last = 1;
        if (three) {
            sstore(0, va0, vb0);
            __syncthreads();
            compute(0);
            last = 0;
        }
    }
    if (has_tail) {                                       // buffer last ^ 1 was read before the previous barrier
        gload(nfull, va0, vb0, TAIL);
        sstore(last ^ 1, va0, vb0);
        __syncthreads();
        compute(last ^ 1);
    }

    EGP_TR(20);
    // ---- epilogue. acc[i][j][r]: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31 of the 32 x 32 block.
    // Branch-free per element: mask values of a block are fetched together (clamped addresses), stores of interior tiles
    // carry no predicate.
    const bool partial = gridDim.z > 1 || g.ones_col;
    const int n_out = g.N + (g.ones_col ? 1 : 0);
    const bool interior = m0 + BM <= g.M && n0 + BN <= n_out;
    const float floor_v = g.relu ? 0.f : -__builtin_inff();
    float *dst = partial ? g.ws + (long)blockIdx.z * g.M * g.ldws : g.C;
    const long ldd = partial ? g.ldws : g.ldc;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn * 64 + 32 * j + (lane & 31);
            const int rbase = m0 + wm * WROWS + 32 * i + 4 * (lane >> 5);
            const int colc = min(col, n_out - 1);
            const float bv = (!partial && g.bias) ? g.bias[min(colc, g.N - 1)] : 0.f;
            float keep[16];
            if (!partial && g.mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                    keep[r] = g.mask[(long)row * g.ldmask + min(colc, g.N - 1)];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) keep[r] = 1.f;
            }
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    float x = partial ? acc[i][j][r] : fmaxf(acc[i][j][r] + bv, floor_v);
                    x = keep[r] > 0.f ? x : 0.f;
                    dst[(long)row * ldd + col] = x;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    float x = partial ? acc[i][j][r] : fmaxf(acc[i][j][r] + bv, floor_v);
                    x = keep[r] > 0.f ? x : 0.f;
                    if (row < g.M && col < n_out) dst[(long)row * ldd + col] = x;
                }
            }
        }
    EGP_TR(21);
}

// ---------------------------------------------------------------------------------------------------------------------
// Three-piece products, warp-specialised and persistent (the default for terms = 6).
//
// In k_gemm_bf16x every wave loads, splits, stores, waits at two barriers and multiplies in turn; with the short k loops
// of the update (7-10 k-tiles) and two workgroups per CU a k-tile takes ~4 k cycles of which the matrix cores are busy
// for 1.5 k (cycle stamps: tools/probes/gemm_trace.hip). Here a workgroup is 8 waves with fixed roles:
//   waves 4-7 (producers): global loads of the operand tiles four k-tiles ahead (four register sets), split into the
//                          three bf16 images, stores into one of TWO LDS buffers
//   waves 0-3 (consumers): fragment reads + 48 MFMAs per k-tile, the epilogue
// One barrier per k-tile: while the consumers multiply buffer p & 1 the producers fill the other one. A SIMD holds one
// wave of each kind, so the conversions (VALU) and the matrix pipe overlap by hardware scheduling rather than by the
// compiler's instruction order. The workgroup is persistent: it walks its work items (tile slot x k split, dealt
// round-robin over the grid) as ONE stream of k-tiles, so the loads of the next tile are in flight while the consumers
// write the previous tile's result -- no pipeline fill per tile.
// A ragged last k-tile is loaded as the last 32 k of the range (in bounds, branch-free) with its already-multiplied head
// zeroed on the way into LDS; ranges shorter than one k-tile take k_gemm_bf16x.
struct WsCursor {                  // position in a workgroup's stream of k-tiles; wave-uniform
    int w, s, nst;                 // work item, k-tile within it, k-tiles of the item
    int m0, n0, z, kbeg, kend;
};

template <int BN>
__device__ __forceinline__ bool ws_item(const GemmArgs &g, int w, WsCursor &c) {       // c is written only for a real tile
    const int L = w % g.grid_tiles, z = w / g.grid_tiles;
    int tm, tn;
    if (!tile_of(g, L, tm, tn)) return false;
    c.w = w; c.z = z; c.m0 = tm * BM; c.n0 = tn * BN;
    c.kbeg = z * g.k_per_split;
    c.kend = z == g.zs - 1 ? g.K : c.kbeg + g.k_per_split;        // (the last split may carry up to one k-tile more, see the launcher)
    c.nst = (c.kend - c.kbeg + BK - 1) / BK;
    return true;
}

// next k-tile; past the end the cursor stays on the last one (the producers' look-ahead then reloads it, harmlessly)
template <int BN>
__device__ __forceinline__ void ws_next(const GemmArgs &g, WsCursor &c, int &w_scan) {
    if (++c.s < c.nst) return;
    for (w_scan += gridDim.x; w_scan < g.n_items; w_scan += gridDim.x)
        if (ws_item<BN>(g, w_scan, c)) { c.s = 0; return; }
    c.s = c.nst - 1;
}

// Producer-side staging of k_gemm_ws. The per-thread part of a load address (row offset, clamped) is a 32-bit byte offset
// computed once per work item; the k-tile's part is wave-uniform and goes into the scalar base of the load
// (global_load ... v_off, s[base:base+1]): a k-tile's loads cost no vector address arithmetic.
template <int R, bool KC>
struct WsStage {
    // k-contiguous memory: thread t -> k octet t & 3 (= LDS k-panel), rows (t >> 2) + 64 u: two 16-byte loads per row, and the
    // eight values are exactly one [row][8 bf16] panel entry (one 16-byte LDS store per image, no bank conflicts).
    // row-contiguous memory, 128-row tiles: lane -> the row PAIR (2 lane, 2 lane + 1), wave -> k-panel: one 8-byte load per k
    // (a wave reads 128 consecutive floats of a k-row per instruction). A pair that would start on the operand's last row
    // (odd row counts: 243 input features) is read one row earlier and shifted, so no load leaves the operand.
    // row-contiguous memory, 64-row tiles: lane -> row, wave -> k-panel, scalar loads (as load_rc).
    static constexpr bool PAIR = !KC && R == 128;
    static constexpr int ROW0_MUL = PAIR ? 2 : 1, ROW_STEP = PAIR ? 1 : 64;   // rows of a thread: ROW0_MUL * first + ROW_STEP * u
    unsigned off[PAIR ? 1 : R / 64];           // byte offset of the thread's rows
    unsigned off2[KC ? R / 64 : 1];            // k-contiguous form with a second source (columns k >= split): its row offsets
    bool one[KC ? 1 : 2];                      // row-contiguous forms: the thread's row u is the virtual row of ones
    bool shift;                                // PAIR: the pair was read one row early
    // `gather` (k-contiguous form): the operand's row m lives at row gather[m] of P; `ld2`: row stride of the second source
    __device__ __forceinline__ void bind(long ld, int rows, int r0, int ones_row, int t, const long long *gather = nullptr, long ld2 = 0) {
        shift = false;
        off2[0] = 0;
        if constexpr (KC) {
#pragma unroll
            for (int u = 0; u < R / 64; ++u) {
                const long row = min(r0 + (t >> 2) + 64 * u, rows - 1);
                const long src = gather ? gather[row] : row;
                off[u] = (unsigned)((src * ld + (t & 3) * 8) * 4);
                off2[u] = (unsigned)((row * ld2 + (t & 3) * 8) * 4);
            }
            one[0] = false;
        } else if constexpr (PAIR) {
            const int row = r0 + 2 * (t & 63);
            one[0] = row == ones_row;
            one[1] = row + 1 == ones_row;
            shift = row == rows - 1;
            off[0] = (unsigned)max(min(row, rows - 2), 0) * 4u;
        } else {
            one[1] = false;
#pragma unroll
            for (int u = 0; u < R / 64; ++u) {
                const int row = r0 + (t & 63) + 64 * u;
                one[u] = row == ones_row;
                off[u] = (unsigned)min(row, rows - 1) * 4u;
            }
        }
    }
    // The loads are inline asm and the wait for them is explicit (ws_wait): the compiler's own wait insertion loses count
    // of loads across the unrolled, branching producer loop and falls back to "everything older than the last few", which
    // collapses the four-deep prefetch into one. Staged registers: k-contiguous -> 16-byte pieces q[2 u], q[2 u + 1] =
    // k 0..3, 4..7 of row u; row pairs -> p[j] = (row 0, row 1) at k j; scalar form -> f[8 u + j].
    static constexpr int NLOAD = KC ? R / 32 : (PAIR ? 8 : R / 8);            // load instructions per k-tile and thread
    using Regs = std::conditional_t<KC, f32x4[R / 32], std::conditional_t<PAIR, f32x2[8], float[R / 8]>>;
    // k0: first k of the tile; `panel`: the wave's k-panel (wave-uniform), used by the row-contiguous forms;
    // `second` (k-contiguous form): the rows of the second source; `krows` (row-contiguous forms): the 8 k-rows of the wave's panel
    // live at rows krows[0..7] of P (a k-gather; the producer loop resolves the indices one stage early)
    __device__ __forceinline__ void load(const float *__restrict__ P, long ld, int k0, int panel, Regs &r, bool second, bool kgather,
                                         const int (&krows)[8]) const {
        if constexpr (KC) {
            const char *base = (const char *)(P + k0);
#pragma unroll
            for (int u = 0; u < R / 64; ++u) {
                const unsigned o = second ? off2[u] : off[u];
                // (the non-temporal hint on these loads was measured in round 6 and is wrong here: the three n-tiles' re-reads of an activation
                //  tile and every m-tile's re-reads of the weights must hit L2 -- forward products 183 -> 256 us with it, the row-contiguous
                //  forms of the weight gradients 183 -> 204 us)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[2 * u]) : "v"(o), "s"(base));
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(r[2 * u + 1]) : "v"(o), "s"(base));
            }
        } else {
            int krow[8];                               // wave-uniform (scalar address arithmetic); `krows`: gathered k-rows, resolved by the caller
#pragma unroll
            for (int j = 0; j < 8; ++j) krow[j] = __builtin_amdgcn_readfirstlane(kgather ? krows[j] : k0 + 8 * panel + j);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const char *base = (const char *)(P + (long)krow[j] * ld);
                if constexpr (PAIR) {
                    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r[j]) : "v"(off[0]), "s"(base));
                } else {
#pragma unroll
                    for (int u = 0; u < R / 64; ++u) asm volatile("global_load_dword %0, %1, %2" : "=v"(r[8 * u + j]) : "v"(off[u]), "s"(base));
                }
            }
        }
    }
    // after ws_wait: v[8 u + j] = element (row u of the thread, k = 8 panel + j) of the tile in all forms
    __device__ __forceinline__ void unpack(const Regs &r, float (&v)[R / 8]) const {
        if constexpr (KC) {
#pragma unroll
            for (int u = 0; u < R / 64; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[8 * u + e] = r[2 * u][e]; v[8 * u + 4 + e] = r[2 * u + 1][e]; }
        } else if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = shift ? r[j][1] : r[j][0];
                v[j] = one[0] ? 1.f : a;
                v[8 + j] = one[1] ? 1.f : r[j][1];
            }
        } else {
#pragma unroll
            for (int u = 0; u < R / 64; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * u + j] = one[u] ? 1.f : r[8 * u + j];
        }
    }
};

// s_waitcnt vmcnt(N) tied to the registers of one staged set: no use (or copy) of them can move above the wait
template <int N> __device__ __forceinline__ void ws_wait(f32x4 (&r)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws_wait(f32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws_wait(f32x2 (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws_wait(float (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws_wait(float (&r)[16]) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                   "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N));
}

// exact three-way split by truncation: x = h + m + l with 8 significant bits each (the float32 mantissa is 24 bits), the
// bf16 pieces are the upper halves of h, m and l. Two floats at a time: the pieces of a pair pack with one v_perm each.
// (Round 3, measured and NOT adopted -- A/B runs of whole-library builds inside one GPU lease, tools/probes/ab_libs.sh +
//  update_time.py: the two subtractions of a pair as v_pk_add_f32 (9 vector instructions per pair instead of 11) make the
//  update 1.5-2 ms SLOWER (51.9 -> 53.5 ms); two producer sets per workgroup (12 waves, half a tile each, which needs the
//  consumers' fragments loaded per k-step to fit 168 VGPRs) are neutral (51.1 vs 50.9 ms). Why neither helps:
//  tools/probes/mfma_valu_overlap.hip -- a SIMD's vector ALU makes no progress while its matrix pipe runs back to back, so a
//  k-tile costs MFMA time PLUS conversion time whoever issues the conversion; this kernel is at 82 % of that serial bound.)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &ph, unsigned &pm, unsigned &pl) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float h0 = __uint_as_float(u0 & 0xffff0000u), h1 = __uint_as_float(u1 & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;                                   // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float m0 = __uint_as_float(v0 & 0xffff0000u), m1 = __uint_as_float(v1 & 0xffff0000u);
    const float l0 = r0 - m0, l1 = r1 - m1;                                   // exact, at most 8 significant bits
    ph = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                          // { hi16(x1), hi16(x0) }
    pm = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    pl = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

// staged registers -> the three LDS images: v[8 u + j] = (row r_first + ROW_STEP u, k = 8 panel + j), one 16-byte store per image
template <int R, int ROW_STEP, bool FAKE = false>
__device__ __forceinline__ void ws_store(const float (&v)[R / 8], __bf16 *dst, int img, int panel, int r_first) {
#pragma unroll
    for (int u = 0; u < R / 64; ++u) {
        unsigned q[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (FAKE) {       // timing probe only (-DEGP_FAKE_SPLIT=1: B, 2: A and B; wrong numerics): what pre-split operands would
                                        // save. Round 3, in-lease A/B/C (tools/probes/ab3.sh): B's conversion gone = T_update 50.8 -> 49.5 ms, i.e.
                                        // keeping the weights as three bf16 planes is worth 1.3 ms -- not built
                q[0][j] = __builtin_amdgcn_perm(__float_as_uint(v[8 * u + 2 * j + 1]), __float_as_uint(v[8 * u + 2 * j]), 0x07060302u);
                q[1][j] = q[0][j] ^ 0x00010001u; q[2][j] = q[0][j] ^ 0x00020002u;
            } else {
                split3_pair(v[8 * u + 2 * j], v[8 * u + 2 * j + 1], q[0][j], q[1][j], q[2][j]);
            }
        }
        const int off = panel * panel_el(R) + (r_first + ROW_STEP * u) * 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) *(uint4 *)(dst + c * img + off) = make_uint4(q[c][0], q[c][1], q[c][2], q[c][3]);
    }
}
// zero the elements of a staged tile whose k (relative to the tile's first k) lies before `zrel`
template <int R>
__device__ __forceinline__ void ws_zero_head(float (&v)[R / 8], int zrel, int panel) {
#pragma unroll
    for (int u = 0; u < R / 64; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * u + j] = 8 * panel + j >= zrel ? v[8 * u + j] : 0.f;
}

template <int BN, bool A_KC, bool B_KC, bool FUSED>      // FUSED: the gather / scatter operands of egp_gemm_desc (a_rows ... c_rows) are compiled in
__global__ __launch_bounds__(512) void k_gemm_ws(GemmArgs g) {
    constexpr int NIMG = 3, NSET = 4;
    constexpr int WN = BN == 128 ? 2 : 1;
    constexpr int MI = BN == 128 ? 2 : 1, NJ = 2;
    constexpr int WROWS = 32 * MI;
    constexpr int A_EL = tile_el(BM), B_EL = tile_el(BN);
    constexpr int BUF_EL = NIMG * (A_EL + B_EL);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *base = (__bf16 *)smem;
    const int t = threadIdx.x, pt = t & 255, lane = t & 63;
    const bool producer = __builtin_amdgcn_readfirstlane(t) >= 256;         // wave-uniform, and the compiler knows it
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6) & 3;
    const int ones_row = g.ones_col ? g.N : -1;

    // this workgroup's stream: P k-tiles over its items
    // Workgroups are dealt to the 8 XCDs round-robin by their id. In a split-K launch the few output tiles of one k range read
    // the same k rows of both operands: with consecutive ids they would sit on different XCDs and every one of them
    // would fetch those rows from HBM for itself (2.4 x the operands' bytes for a 300 x 243 weight gradient). The
    // transposed numbering puts consecutive work items on consecutive workgroups of ONE XCD, so a k row crosses from
    // HBM once and the other tiles hit that XCD's L2.
    int vb = blockIdx.x;
    if (g.split_xcd && (gridDim.x & 7) == 0) vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    int P = 0, w_first = -1;
    WsCursor cur{};
    for (int w = vb; w < g.n_items; w += gridDim.x) {
        WsCursor c{};
        if (ws_item<BN>(g, w, c)) {
            P += c.nst;
            if (w_first < 0) { w_first = w; cur = c; }
        }
    }
    if (P == 0) return;
    cur.s = 0;
    int w_scan = w_first;
    EGP_TRW_DECL;

    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, 2> S2{};
    constexpr std::integral_constant<int, 3> S3{};

    // iteration p: the consumers multiply k-tile p (buffer p & 1); the producers stage k-tile p + 1 from register set
    // (p + 1) & 3 into the other buffer and request k-tile p + 5 into the freed set. Both roles run the same number of
    // barriers, each in its own loop (separate loops keep the producers' staging registers and the consumers' accumulators
    // out of each other's live ranges).
    if (producer) {
        using SA = WsStage<BM, A_KC>;
        using SB = WsStage<BN, B_KC>;
        typename SA::Regs ra[NSET];
        typename SB::Regs rb[NSET];
        int zrel[NSET] = {0, 0, 0, 0};
        SA sa;
        SB sb;
        int bound = -1;
        // loads of the younger sets that may still be in flight when a set is staged: three sets, or as many as the 6-bit
        // counter can express (row-contiguous operands take 16 scalar loads each)
        constexpr int PER_SET = SA::NLOAD + SB::NLOAD;
        constexpr int YOUNGER = 3 * PER_SET <= 63 ? 3 * PER_SET : (2 * PER_SET <= 63 ? 2 * PER_SET : PER_SET);
        // k-gathers (a_krows / b_krows): the index entries of the k-tile under the cursor, fetched (scalar loads) BEFORE the
        // staging work of the iteration so that they have landed when `issue` forms the load addresses from them
        int ka[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto prefetch = [&]() __attribute__((always_inline)) {
            if constexpr (FUSED && (!A_KC || !B_KC)) {
                const bool last = cur.s == cur.nst - 1;
                const int k0 = (last ? cur.kend - BK : cur.kbeg + cur.s * BK) + 8 * wave;
                if (!A_KC && g.a_krows) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ka[j] = __builtin_amdgcn_readfirstlane((int)g.a_krows[k0 + j]);
                }
                if (!B_KC && g.b_krows) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) kb[j] = __builtin_amdgcn_readfirstlane((int)g.b_krows[k0 + j]);
                }
            }
        };
        auto issue = [&](auto setc) __attribute__((always_inline)) {      // loads of the k-tile under the cursor; cursor moves on
            constexpr int SET = decltype(setc)::value;
            // B given as [k][n] with a second source: the tile's columns come from one of them (b_split is a multiple of BN)
            const bool b_second = FUSED && !B_KC && g.B2 && cur.n0 >= g.b_split;
            if (cur.w != bound) {
                if constexpr (FUSED) {
                    sa.bind(g.lda, g.M, cur.m0, -1, pt, A_KC ? g.a_rows : nullptr, g.lda2);
                    if (b_second) sb.bind(g.ldb2, g.N - g.b_split, cur.n0 - g.b_split, ones_row - g.b_split, pt);
                    else sb.bind(g.ldb, (!B_KC && g.B2) ? g.b_split : g.N, cur.n0, (!B_KC && g.B2) ? -1 : ones_row, pt);
                } else {
                    sa.bind(g.lda, g.M, cur.m0, -1, pt);
                    sb.bind(g.ldb, g.N, cur.n0, ones_row, pt);
                }
                bound = cur.w;
            }
            const bool last = cur.s == cur.nst - 1;
            const int kz = cur.kbeg + cur.s * BK;
            const int k0 = last ? cur.kend - BK : kz;
            zrel[SET] = kz - k0;
#if defined(EGP_WS_SKIP) && EGP_WS_SKIP == 3      // timing probe: no operand loads at all (the staged registers hold whatever they held)
            if (g.M < 0)
#endif
            if constexpr (FUSED) {
                const bool a_second = A_KC && g.A2 && k0 >= g.a_split;        // (a_split is a multiple of BK: a k-tile has one source)
                sa.load(a_second ? g.A2 : g.A, a_second ? g.lda2 : g.lda, a_second ? k0 - g.a_split : k0, wave, ra[SET], a_second,
                        !A_KC && g.a_krows != nullptr, ka);
                sb.load(b_second ? g.B2 : g.B, b_second ? g.ldb2 : g.ldb, k0, wave, rb[SET], false, !B_KC && !b_second && g.b_krows != nullptr, kb);
            } else {
                sa.load(g.A, g.lda, k0, wave, ra[SET], false, false, ka);
                sb.load(g.B, g.ldb, k0, wave, rb[SET], false, false, kb);
            }
            ws_next<BN>(g, cur, w_scan);
        };
        auto stage = [&](auto setc, int buf) __attribute__((always_inline)) {   // register set -> LDS buffer `buf`
            constexpr int SET = decltype(setc)::value;
            // (panel, first row) of the thread: k-contiguous -> (t & 3, t >> 2), row-contiguous -> (wave, lane or 2 lane)
            const int pa_panel = A_KC ? (pt & 3) : wave, pa_row = A_KC ? (pt >> 2) : SA::ROW0_MUL * lane;
            const int pb_panel = B_KC ? (pt & 3) : wave, pb_row = B_KC ? (pt >> 2) : SB::ROW0_MUL * lane;
            ws_wait<YOUNGER>(ra[SET]);
            ws_wait<YOUNGER>(rb[SET]);
            EGP_TRW(1, 33);
            float va[BM / 8], vb[BN / 8];
            sa.unpack(ra[SET], va);
            sb.unpack(rb[SET], vb);
            if (zrel[SET] > 0) {
                ws_zero_head<BM>(va, zrel[SET], pa_panel);
                ws_zero_head<BN>(vb, zrel[SET], pb_panel);
            }
            __bf16 *pa = base + buf * BUF_EL, *pb = pa + NIMG * A_EL;
#ifndef EGP_FAKE_SPLIT
#define EGP_FAKE_SPLIT 0
#endif
            ws_store<BM, SA::ROW_STEP, (EGP_FAKE_SPLIT >= 2)>(va, pa, A_EL, pa_panel, pa_row);
            EGP_TRW(1, 34);
            ws_store<BN, SB::ROW_STEP, (EGP_FAKE_SPLIT >= 1)>(vb, pb, B_EL, pb_panel, pb_row);
        };
        prefetch(); issue(S0); prefetch(); issue(S1); prefetch(); issue(S2); prefetch(); issue(S3);      // k-tiles 0..3 in flight
        prefetch();
        stage(S0, 0);
        issue(S0);                                       // k-tile 4
        __syncthreads();
        auto iteration = [&](auto setc, int p) __attribute__((always_inline)) {
            EGP_TRW(1, 30);
            if (p + 1 < P) {
                prefetch();
#if !defined(EGP_WS_SKIP) || EGP_WS_SKIP != 2      // (timing probes, garbage numerics: 1 consumers idle, 3 no operand loads; 2 -- no staging --
                                                   //  ends in a memory access fault since round 6, do not run it)
                stage(setc, (p + 1) & 1);
#endif
                EGP_TRW(1, 31);
                issue(setc);
            }
            EGP_TRW(1, 32);
            __syncthreads();
        };
        for (int p = 0; p < P; p += 4) {
            iteration(S1, p);
            if (p + 1 < P) iteration(S2, p + 1);
            if (p + 2 < P) iteration(S3, p + 2);
            if (p + 3 < P) iteration(S0, p + 3);
        }
        return;
    }

    // ---- consumers
    // (s_setprio 3 for these waves -- the matrix pipe's wave first when both want to issue -- was measured: no effect)
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fkh = lane >> 5;
    f32x16 acc[MI][NJ];
    auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // The product is formed transposed (B fragment as the first MFMA operand): acc[i][j][r] is
    //   C[m = 32 i + (lane & 31)][n = 32 j + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)]     (within the wave's 64 x 64 / 32 x 64 part)
    // so that a lane holds four consecutive n of one row: 16-byte stores (and mask loads).
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const __bf16 *pa = base + buf * BUF_EL, *pb = pa + NIMG * A_EL;
        // every fragment of the k-tile is requested before the first MFMA (the matrix pipe then runs the 48 products
        // back to back instead of idling through an LDS round trip in the middle)
        bf16x8 fa[BK / 16][NIMG][MI], fb[BK / 16][NIMG][NJ];
#ifndef EGP_GEMM_FRAG_ORDER
#define EGP_GEMM_FRAG_ORDER 0
#endif
#define EGP_LD_FA(ks, c, i) fa[ks][c][i] = *(const bf16x8 *)(pa + (c) * A_EL + (2 * (ks) + fkh) * panel_el(BM) + (wm * WROWS + 32 * (i) + frow) * 8)
#define EGP_LD_FB(ks, c, j) fb[ks][c][j] = *(const bf16x8 *)(pb + (c) * B_EL + (2 * (ks) + fkh) * panel_el(BN) + (wn * 64 + 32 * (j) + frow) * 8)
#if EGP_GEMM_FRAG_ORDER
        // the fragments in the order the products below consume them: the first MFMA waits for two reads, not for ten
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            constexpr int CB[3] = {0, 2, 1}, CA[3] = {2, 0, 1};
#pragma unroll
            for (int q = 0; q < 3; ++q) { EGP_LD_FB(ks, CB[q], 0); EGP_LD_FA(ks, CA[q], 0); }
#pragma unroll
            for (int j = 1; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) EGP_LD_FB(ks, CB[q], j);
#pragma unroll
            for (int i = 1; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) EGP_LD_FA(ks, CA[q], i);
        }
#else
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
            for (int c = 0; c < NIMG; ++c) {
#pragma unroll
                for (int i = 0; i < MI; ++i) EGP_LD_FA(ks, c, i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) EGP_LD_FB(ks, c, j);
            }
#endif
#undef EGP_LD_FA
#undef EGP_LD_FB
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x16 a = acc[i][j];                 // smallest terms first
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][0][j], fa[ks][2][i], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][2][j], fa[ks][0][i], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][1][j], fa[ks][1][i], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][0][j], fa[ks][1][i], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][1][j], fa[ks][0][i], a, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][0][j], fa[ks][0][i], a, 0, 0, 0);
                }
    };
    // Epilogue: a lane holds 4 consecutive n of row m = lane & 31 -- 16 bytes, but 32 different rows per store instruction.
    // The wave's part of the tile therefore goes through a wave-private LDS patch (32 rows x 64 columns at a time, written
    // as 16-byte pieces, read back row-major): every global store instruction then writes four 256-byte row segments.
    constexpr int EP_LD = 64 + 4;                         // floats per patch row (+4: the 16-byte writes of 16 lanes hit 64 banks)
    float *patch = (float *)(smem + (size_t)2 * BUF_EL * sizeof(__bf16)) + wave * 32 * EP_LD;
    const bool partial = g.partial != 0;
    // (the launcher sends only outputs with N a multiple of 4 here; the workspace rows of split / ones-column launches are
    //  padded to a multiple of 4 floats)
    const int ncols = partial ? g.ldws : g.N;
    const long ldd = partial ? g.ldws : g.ldc;
    const float floor_v = g.relu ? 0.f : -__builtin_inff();
    // (the flags are template constants of the body: inside it there is no branch, so the compiler counts its waits --
    //  all mask rows are requested first, and no store is ever waited for)
    auto epilogue_body = [&](auto partial_c, auto mask_c) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_c)::value, MASK = decltype(mask_c)::value;
        float *dst = PARTIAL ? g.ws + (long)cur.z * g.M * g.ldws : g.C;
        const int col = cur.n0 + wn * 64 + 4 * (lane & 15);
        const int colc = min(col, ncols - 4);
        const bool col_ok = col < ncols;
        const int row0 = cur.m0 + wm * WROWS + (lane >> 4);
        long drow[MI][8];                                 // destination rows (scatter: c_rows)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = row0 + 32 * i + 4 * it;
                drow[i][it] = (FUSED && !PARTIAL && g.c_rows) ? (long)g.c_rows[min(row, g.M - 1)] : (long)row;
            }
        // (16-byte accesses on 4-byte-aligned addresses throughout: leading dimensions like 243 are welcome)
        f32x4u bv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!PARTIAL) bv = *(const f32x4u *)((g.bias ? g.bias : g.C) + colc);      // (no bias: any readable address, the value is dropped)
        if (PARTIAL || !g.bias) bv = f32x4u{0.f, 0.f, 0.f, 0.f};
        f32x4u keep[MI][8];
        if constexpr (MASK) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    keep[i][it] = *(const f32x4u *)(g.mask + (long)min(row0 + 32 * i + 4 * it, g.M - 1) * g.ldmask + colc);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *(float4 *)(patch + (lane & 31) * EP_LD + 32 * j + 8 * rq + 4 * (lane >> 5)) =
                        make_float4(acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = row0 + 32 * i + 4 * it;
                float4 x = *(const float4 *)(patch + ((lane >> 4) + 4 * it) * EP_LD + 4 * (lane & 15));
                if constexpr (!PARTIAL) {
                    x.x = fmaxf(x.x + bv[0], floor_v); x.y = fmaxf(x.y + bv[1], floor_v);
                    x.z = fmaxf(x.z + bv[2], floor_v); x.w = fmaxf(x.w + bv[3], floor_v);
                }
                if constexpr (MASK) {
                    x.x = keep[i][it][0] > 0.f ? x.x : 0.f; x.y = keep[i][it][1] > 0.f ? x.y : 0.f;
                    x.z = keep[i][it][2] > 0.f ? x.z : 0.f; x.w = keep[i][it][3] > 0.f ? x.w : 0.f;
                }
                if (row < g.M && col_ok) *(f32x4u *)(dst + drow[i][it] * ldd + col) = f32x4u{x.x, x.y, x.z, x.w};
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        if (partial) epilogue_body(std::true_type{}, std::false_type{});
        else if (g.mask) epilogue_body(std::false_type{}, std::true_type{});
        else epilogue_body(std::false_type{}, std::false_type{});
    };
    clear();
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        EGP_TRW(0, 40);
#if !defined(EGP_WS_SKIP) || EGP_WS_SKIP != 1
        compute(p & 1);
#endif
        EGP_TRW(0, 41);
        if (cur.s == cur.nst - 1) {
            epilogue();
            EGP_TRW(0, 43);
            clear();
            EGP_TRW(0, 42);
        }
        ws_next<BN>(g, cur, w_scan);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Thin products. A value function's output layer makes three of them per epoch -- y = X w (N = 1), dX = dy w^T under the
// ReLU mask (K = 1) and dw = dy^T X (M = 1) -- that are one pass over the activations each: as 128-wide matrix-core tiles
// they took 40 / 72 / 96 us for 105 MB, as plain float32 streaming kernels they are bound by those bytes. Products and
// sums in float32 (fmaf), fixed order: the same class as the split-operand products.
__global__ __launch_bounds__(256) void k_gemv_rows(GemmArgs g) {         // N == 1, A k-contiguous: C[m] = A[m][:] . B[:]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long bs = g.b_kc ? 1 : g.ldb;
    const float floor_v = g.relu ? 0.f : -__builtin_inff();
    for (long m = (long)blockIdx.x * 4 + wave; m < g.M; m += (long)gridDim.x * 4) {
        const float *__restrict__ a = g.A + m * g.lda;
        float acc = 0.f;
        for (int k = lane; k < g.K; k += 64) acc = fmaf(a[k], g.B[k * bs], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) {
            float x = fmaxf(acc + (g.bias ? g.bias[0] : 0.f), floor_v);
            if (g.mask && !(g.mask[m * g.ldmask] > 0.f)) x = 0.f;
            g.C[m * g.ldc] = x;
        }
    }
}

__global__ __launch_bounds__(256) void k_rank1(GemmArgs g) {             // K == 1: C[m][n] = A[m] * B[n] (+ bias, ReLU, mask)
    const long as = g.a_kc ? g.lda : 1, bs = g.b_kc ? g.ldb : 1;
    const float floor_v = g.relu ? 0.f : -__builtin_inff();
    const long total = (long)g.M * g.N;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const long m = id / g.N;
        const int n = (int)(id - m * g.N);
        float x = fmaxf(fmaf(g.A[m * as], g.B[n * bs], g.bias ? g.bias[n] : 0.f), floor_v);
        if (g.mask && !(g.mask[m * g.ldmask + n] > 0.f)) x = 0.f;
        g.C[m * g.ldc + n] = x;
    }
}

// M == 1, B given as [k][n], split-K / ones-column launch: workgroup z sums its k range of a[k] * B[k][:] (four waves take
// every fourth row, 8 rows in flight each) into row z of the workspace, the ones column = sum of a[k]; k_gemm_reduce follows
__global__ __launch_bounds__(256) void k_colsum(GemmArgs g) {
    __shared__ float s_part[4][256 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, z = blockIdx.x;
    const long as = g.a_kc ? 1 : g.lda;
    const int kbeg = z * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    const int n_out = g.N + (g.ones_col ? 1 : 0);
    float *__restrict__ dst = g.ws + (long)z * g.ldws;                   // (M = 1: a split's block is one row)
    for (int n0 = 0; n0 < g.N; n0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};                             // columns n0 + lane + 64 j
        for (int k = kbeg + wave; k < kend; k += 32) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = k + 4 * u < kend ? g.A[(long)(k + 4 * u) * as] : 0.f;
            float b[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *__restrict__ row = g.B + (long)min(k + 4 * u, kend - 1) * g.ldb + n0;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[u][j] = n0 + lane + 64 * j < g.N ? row[lane + 64 * j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(a[u], b[u][j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_part[wave][lane + 64 * j] = acc[j];
        __syncthreads();
        if (wave == 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + lane + 64 * j;
                if (n < g.N) dst[n] = (s_part[0][lane + 64 * j] + s_part[1][lane + 64 * j]) + (s_part[2][lane + 64 * j] + s_part[3][lane + 64 * j]);
            }
        __syncthreads();
    }
    if (g.ones_col) {                                                    // bias gradient: sum of a over the k range
        float sa = 0.f;
        for (int k = kbeg + threadIdx.x; k < kend; k += 256) sa += g.A[(long)k * as];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sa += __shfl_down(sa, off, 64);
        if (lane == 0) s_part[wave][0] = sa;
        __syncthreads();
        if (threadIdx.x == 0) dst[n_out - 1] = (s_part[0][0] + s_part[1][0]) + (s_part[2][0] + s_part[3][0]);
    }
}

// partial sums -> C (+ the ones column -> bias_grad), splits added in index order
__global__ __launch_bounds__(256) void k_gemm_reduce(const float *__restrict__ ws_, int ldws, int splits, int M, int N, int n_out, float *__restrict__ C,
                                                     long ldc, float *__restrict__ bias_grad, int accumulate) {
    const long id0 = (long)blockIdx.x * 256 + threadIdx.x;
    if (id0 >= (long)M * n_out) return;
    const int row = (int)(id0 / n_out), col = (int)(id0 % n_out);
    const long idx = (long)row * ldws + col;
    const float *__restrict__ ws = ws_;
    // eight independent chains (the loads of a chain are 0.5 us apart otherwise); the order is fixed, so results repeat
    const long stride = (long)M * ldws;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] += ws[(k + j) * stride + idx];
    }
    for (; k < splits; ++k) p[k & 7] += ws[k * stride + idx];
    const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (col < N) {
        float *dst = C + (long)row * ldc + col;
        *dst = accumulate ? *dst + s : s;
    } else if (bias_grad) {
        bias_grad[row] = accumulate ? bias_grad[row] + s : s;
    }
}

// ---- the policy / value input of the update: out[i] = [ ctx[idx[i]][0:H] | x[i][0:S] ]  (models/video_state_net.py:65-69:
// the gather of the per-sample video context and the concatenation with the state in one pass), and its adjoint for the
// context rows (every context row is gathered at most once, so the scatter needs no atomics; rows nobody gathered keep
// the zeros they were initialised with).
__global__ __launch_bounds__(256) void k_gather_concat(const float *__restrict__ ctx, long ld_ctx, const long long *__restrict__ idx,
                                                       const float *__restrict__ x, long ldx, int H, int S, float *__restrict__ out, long ldo) {
    const long i = blockIdx.x;
    const float *c = ctx + idx[i] * ld_ctx, *xs = x + i * ldx;
    float *o = out + i * ldo;
    for (int j = threadIdx.x; j < H + S; j += 256) o[j] = j < H ? c[j] : xs[j - H];
}

__global__ __launch_bounds__(256) void k_scatter_rows(const float *__restrict__ dout, long ldd, const long long *__restrict__ idx, int H,
                                                      float *__restrict__ dctx, long ld_ctx) {
    const long i = blockIdx.x;
    const float *d = dout + i * ldd;
    float *c = dctx + idx[i] * ld_ctx;
    for (int j = threadIdx.x; j < H; j += 256) c[j] = d[j];
}

template <int BN, int TERMS>
int launch_variant(const GemmArgs &g, dim3 grid, size_t lds, hipStream_t s) {
#define EGP_GEMM_LAUNCH(AK, BKC)                                                                                      \
    do {                                                                                                              \
        auto kern = k_gemm_bf16x<BN, TERMS, AK, BKC>;                                                                 \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            EGP_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        kern<<<grid, dim3(256), lds, s>>>(g);                                                                         \
    } while (0)
    if (g.a_kc && g.b_kc) EGP_GEMM_LAUNCH(true, true);
    else if (g.a_kc && !g.b_kc) EGP_GEMM_LAUNCH(true, false);
    else if (!g.a_kc && g.b_kc) EGP_GEMM_LAUNCH(false, true);
    else EGP_GEMM_LAUNCH(false, false);
#undef EGP_GEMM_LAUNCH
    return after_launch("k_gemm_bf16x");
}

template <int BN, bool FUSED>
int launch_ws(const GemmArgs &g, hipStream_t s) {
    static int n_cu = 0;                        // one persistent workgroup per CU the process really gets (egp_device_usable_cus: under a
    if (!n_cu) {                                // CU mask the attribute over-reports, and the workgroups without a CU would be a second round)
        int dev = 0;
        EGP_HIP_CHECK(hipGetDevice(&dev));
        n_cu = egp_device_usable_cus(dev);
        if (n_cu <= 0) EGP_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const size_t lds = (size_t)2 * 3 * (tile_el(BM) + tile_el(BN)) * sizeof(__bf16) + 4 * 32 * (64 + 4) * sizeof(float);   // + the epilogue patches
    const dim3 grid((unsigned)(g.n_items < n_cu ? g.n_items : n_cu));
#define EGP_GEMM_WS_LAUNCH(AK, BKC)                                                                                   \
    do {                                                                                                              \
        auto kern = k_gemm_ws<BN, AK, BKC, FUSED>;                                                                    \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            EGP_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        kern<<<grid, dim3(512), lds, s>>>(g);                                                                         \
    } while (0)
    if (g.a_kc && g.b_kc) EGP_GEMM_WS_LAUNCH(true, true);
    else if (g.a_kc && !g.b_kc) EGP_GEMM_WS_LAUNCH(true, false);
    else if (!g.a_kc && g.b_kc) EGP_GEMM_WS_LAUNCH(false, true);
    else EGP_GEMM_WS_LAUNCH(false, false);
#undef EGP_GEMM_WS_LAUNCH
    return after_launch("k_gemm_ws");
}

}  // namespace

extern "C" {

int egp_gather_concat_f32(const float *ctx, int64_t ld_ctx, const int64_t *idx, const float *x, int64_t ldx, int32_t n, int32_t H, int32_t S,
                          float *out, int64_t ldo, void *stream) {
    EGP_REQUIRE(n >= 0 && H >= 0 && S >= 0, "negative size");
    if (n == 0 || H + S == 0) return EGP_OK;
    EGP_REQUIRE(ctx && idx && out && (x || S == 0), "NULL pointer");
    k_gather_concat<<<dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream>>>(ctx, ld_ctx, (const long long *)idx, x, ldx, H, S, out, ldo);
    return after_launch("k_gather_concat");
}

int egp_scatter_rows_f32(const float *dout, int64_t ldd, const int64_t *idx, int32_t n, int32_t H, float *dctx, int64_t ld_ctx, void *stream) {
    EGP_REQUIRE(n >= 0 && H >= 0, "negative size");
    if (n == 0 || H == 0) return EGP_OK;
    EGP_REQUIRE(dout && idx && dctx, "NULL pointer");
    k_scatter_rows<<<dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream>>>(dout, ldd, (const long long *)idx, H, dctx, ld_ctx);
    return after_launch("k_scatter_rows");
}

int64_t egp_gemm_workspace_floats(int32_t M, int32_t N, int32_t ones_col, int32_t splits) {
    if (splits <= 1 && !ones_col) return 0;
    return (int64_t)(splits < 1 ? 1 : splits) * M * ((N + (ones_col ? 1 : 0) + 3) & ~3);      // rows padded to 16 bytes
}

int egp_gemm_f32(const egp_gemm_desc *d, void *stream) {
    EGP_REQUIRE(d, "descriptor is NULL");
    EGP_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "negative size");
    EGP_REQUIRE(d->terms == 1 || d->terms == 3 || d->terms == 6, "terms must be 1 (bf16), 3 (two-piece split) or 6 (three-piece split)");
    if (d->M == 0 || d->N + (d->bias_grad ? 1 : 0) == 0) return EGP_OK;
    EGP_REQUIRE(d->A && d->B && d->C, "NULL operand");
    const int ones = d->bias_grad ? 1 : 0;
    const int splits = d->splits < 1 ? 1 : d->splits;
    const bool partial = splits > 1 || ones;
    EGP_REQUIRE(!partial || d->workspace, "split-K / bias-gradient launches need a workspace (egp_gemm_workspace_floats)");
    EGP_REQUIRE(!partial || (!d->bias && !d->relu && !d->mask), "no epilogue on split-K launches");
    EGP_REQUIRE(!ones || !d->b_kcontig, "the ones column (bias gradient) goes with B given as [k][n]");
    // (Round 3, measured and not kept: launching the column remainder of N = 300 as a 64-column-tile product of its own
    //  -- [0, 256) on 128-column tiles + [256, 300) on 64-column tiles, 2.58 instead of 3 tile units -- makes the update 1 ms
    //  SLOWER in in-lease A/B runs (51.6 vs 50.6 ms): the narrow launch converts the whole A operand again for 44 columns.)
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.A = d->A; g.lda = d->lda; g.a_kc = d->a_kcontig;
    g.B = d->B; g.ldb = d->ldb; g.b_kc = d->b_kcontig;
    g.C = d->C; g.ldc = d->ldc;
    g.bias = d->bias; g.relu = d->relu; g.mask = d->mask; g.ldmask = d->ldmask;
    g.ones_col = ones; g.ws = d->workspace; g.ldws = (d->N + ones + 3) & ~3; g.partial = partial ? 1 : 0;
    g.a_rows = (const long long *)d->a_rows; g.A2 = d->A2; g.lda2 = d->lda2; g.a_split = d->a_split;
    g.b_krows = (const long long *)d->b_krows; g.B2 = d->B2; g.ldb2 = d->ldb2; g.b_split = d->b_split;
    g.c_rows = (const long long *)d->c_rows;
    g.a_krows = (const long long *)d->a_krows;
    const bool fused_io = d->a_rows || d->A2 || d->b_krows || d->B2 || d->c_rows || d->a_krows;
    if (fused_io) {
        EGP_REQUIRE(!(d->a_rows || d->A2) || d->a_kcontig, "a_rows / A2 go with a k-contiguous A");
        EGP_REQUIRE(!(d->b_krows || d->B2) || !d->b_kcontig, "b_krows / B2 go with B given as [k][n]");
        EGP_REQUIRE(!d->a_krows || !d->a_kcontig, "a_krows goes with A given as [k][m]");
        EGP_REQUIRE(!d->A2 || (d->a_split > 0 && d->a_split % BK == 0 && d->K - d->a_split >= BK),
                    "a_split must be a multiple of 32 and leave at least 32 columns to A2 (a k-tile reads one source)");
        EGP_REQUIRE(!d->A2 || splits == 1, "A2 does not go with split-K");
        EGP_REQUIRE(!d->B2 || (d->b_split > 0 && d->b_split % 128 == 0 && d->b_split < d->N + ones), "b_split must be a multiple of 128 inside (0, N)");
        EGP_REQUIRE(!d->c_rows || !partial, "c_rows (scatter) does not go with split-K / bias-gradient launches");
        EGP_REQUIRE(!d->a_rows || d->a_src_rows > 0, "a_rows needs a_src_rows (rows of the gathered source)");
        EGP_REQUIRE(!d->b_krows || d->b_src_rows > 0, "b_krows needs b_src_rows (rows of the gathered source)");
    }
    // thin products (see k_gemv_rows / k_rank1 / k_colsum): plain float32 streaming kernels, no tiles
    if (!fused_io) {
        if (d->N == 1 && !partial && d->a_kcontig && d->K >= 1) {
            const long blocks = std::min<long>(((long)d->M + 3) / 4, 256 * 16);
            k_gemv_rows<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g);
            return after_launch("k_gemv_rows");
        }
        if (d->K == 1 && !partial) {
            const long blocks = std::min<long>(((long)d->M * d->N + 255) / 256, 256 * 32);
            k_rank1<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g);
            return after_launch("k_rank1");
        }
        if (d->M == 1 && partial && !d->b_kcontig && d->K >= 1) {
            const int per_rows = (d->K + splits - 1) / splits;
            g.k_per_split = per_rows < 1 ? 1 : per_rows;
            const int zs1 = (d->K + g.k_per_split - 1) / g.k_per_split;
            k_colsum<<<dim3((unsigned)zs1), dim3(256), 0, s>>>(g);
            int rc1 = after_launch("k_colsum");
            if (rc1 != EGP_OK) return rc1;
            const int n_out1 = d->N + ones;
            k_gemm_reduce<<<dim3((unsigned)((n_out1 + 255) / 256)), dim3(256), 0, s>>>(d->workspace, g.ldws, zs1, 1, d->N, n_out1, d->C, d->ldc, d->bias_grad,
                                                                                    d->accumulate);
            return after_launch("k_gemm_reduce");
        }
    }
    const int n_out = d->N + ones;
    const bool bn64 = n_out <= 64;                 // narrow outputs: 64-column tiles
    const int BNv = bn64 ? 64 : 128;
    g.tiles_m = (d->M + BM - 1) / BM;
    g.tiles_n = (n_out + BNv - 1) / BNv;
    const int kt = (d->K + BK - 1) / BK;
    const int per = (kt + splits - 1) / splits;
    g.k_per_split = (per < 1 ? 1 : per) * BK;
    const int zs = d->K == 0 ? 1 : (d->K + g.k_per_split - 1) / g.k_per_split;
    g.xcd_order = g.tiles_m >= 64;
    dim3 grid((g.xcd_order ? ((g.tiles_m + 7) / 8) * 8 : g.tiles_m) * g.tiles_n, 1, zs);
    const int nimg = d->terms == 1 ? 1 : (d->terms == 3 ? 2 : 3), nbuf = d->terms == 6 ? 1 : 2;
    const size_t lds = (size_t)nbuf * nimg * (tile_el(BM) + tile_el(BNv)) * sizeof(__bf16);
    g.grid_tiles = (int)grid.x;
    // k_gemm_ws wants k ranges of at least one k-tile: a shorter remainder rides with the split before it
    const int rem = d->K - (zs - 1) * g.k_per_split;
    const int zs_ws = (zs > 1 && rem < BK) ? zs - 1 : zs;
    g.zs = zs_ws;
    g.n_items = g.grid_tiles * zs_ws;
    g.split_xcd = zs_ws > 1 && !g.xcd_order && g.grid_tiles > 1;
    // three-piece products: the warp-specialised persistent kernel, unless a k range is shorter than one k-tile
    const char *ws_env = getenv("EGP_GEMM_WS");            // EGP_GEMM_WS=0: k_gemm_bf16x for everything (read per call: tests switch it)
    const bool ws_on = !(ws_env && atoi(ws_env) == 0);
    const int last_len = d->K - (zs_ws - 1) * g.k_per_split;
    int rc;
    const bool small32 = (size_t)(d->a_rows ? d->a_src_rows : d->M) * (size_t)(d->a_kcontig ? d->lda : 1) < (1u << 30) &&
                         (size_t)d->N * (size_t)(d->b_kcontig ? d->ldb : 1) < (1u << 30) && (!d->A2 || (size_t)d->M * (size_t)d->lda2 < (1u << 30));
    const bool wide = (partial || d->N % 4 == 0) && (d->a_kcontig || d->M >= 2) && (d->b_kcontig || d->N >= 2);
    int n_splits_written = zs;
    if (d->terms == 6 && ws_on && last_len >= BK && small32 && wide) {
        if (fused_io) rc = bn64 ? launch_ws<64, true>(g, s) : launch_ws<128, true>(g, s);
        else rc = bn64 ? launch_ws<64, false>(g, s) : launch_ws<128, false>(g, s);
        n_splits_written = zs_ws;
    } else if (fused_io) {
        egp::set_error("invalid argument: %s", "gather / scatter operands need the persistent three-piece kernel (terms = 6, EGP_GEMM_WS != 0, k ranges >= 32, N %% 4 == 0)");
        return EGP_E_INVALID;
    } else
    if (bn64) rc = d->terms == 6 ? launch_variant<64, 6>(g, grid, lds, s) : d->terms == 3 ? launch_variant<64, 3>(g, grid, lds, s) : launch_variant<64, 1>(g, grid, lds, s);
    else rc = d->terms == 6 ? launch_variant<128, 6>(g, grid, lds, s) : d->terms == 3 ? launch_variant<128, 3>(g, grid, lds, s) : launch_variant<128, 1>(g, grid, lds, s);
    if (rc != EGP_OK) return rc;
    if (partial) {
        const long total = (long)d->M * n_out;
        k_gemm_reduce<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(d->workspace, g.ldws, n_splits_written, d->M, d->N, n_out, d->C, d->ldc,
                                                                                d->bias_grad, d->accumulate);
        return after_launch("k_gemm_reduce");
    }
    return EGP_OK;
}

}  // extern "C"
