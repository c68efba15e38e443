// egp_chain.hip -- the policy / value MLP of the PPO update as ONE launch per direction: a 128-row tile walks
//   forward :  [ctx[idx] | state] (243) -> 300 -> 200 -> 52 | 1     (models/mlp.py:22-25 + core/policy_gaussian.py:19-24 / core/critic.py:15-18)
//   backward:  d out -> d z2 (200) -> d z1 (300) -> d ctx (128)      (the data-gradient chain of the same three layers)
// with the intermediate activations in REGISTERS. Round 3 ran every layer as its own GEMM launch: each launch converted its
// float32 operands to three bf16 pieces on the way into LDS (both operands, once per output tile), wrote a float32 activation
// and the next launch read and converted it again; the matrix cores were busy 15-39 % of those launches.
//
// How the chain stays on chip. Everything is computed TRANSPOSED: out^T[feature][row] = W[feature][k] * in^T[k][row], with the
// weights as the MFMA's A operand and the batch rows as its columns. v_mfma_f32_32x32x16_bf16 leaves C[i][j] in lane j (+ 32 for
// the upper half of each 8-row group): a lane OWNS one batch row and holds 16 of the block's 32 output features -- which is
// exactly the shape of the next product's B operand (lane = column, 8 consecutive k per lane) up to the ORDER of k inside a
// k-step. The order of a dot product's terms is free as long as both operands agree, so the next layer's weights are packed in
// the order the accumulators already have (egp_mlp_chain_pack_f32, `chained`): registers 8t..8t+7 of block b ARE the B
// fragment of k-step 2b + t. Bias, ReLU (or the ReLU mask of the backward pass) and the split into three bf16 pieces happen on
// those registers; an activation is converted once, by the lane that owns it, and never written for the next layer.
// A wave owns 32 batch rows and needs all of a layer's weights: the workgroup's four waves (128 rows) share them through LDS.
// The weights are packed per layer as a stream of k-steps, each k-step as [i-block][piece][64 lanes x 16 bytes] = the A
// fragments in the order they are read; the waves copy the next-but-one k-step into a three-slot LDS ring with
// global_load_lds_dwordx4 (no registers, no conversion: the pieces were split when the weights were packed, once per epoch)
// while they multiply the current one, and the stream simply continues into the next layer and the next tile.
// Products are the six-term three-piece products of egp_gemm.hip (same pieces, same term order): float32-class.
//
// What still goes through HBM: the weight gradients contract over the batch, whose accumulators (573 kB per workgroup) fit
// no CU, so the forward chain saves x^T, h1^T, h2^T and the backward chain d z^T -- [feature][row], every store and load a
// coalesced 128-byte row segment -- and the weight gradients are the k-contiguous products egp_gemm_f32 already has.
#include "egp_internal.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));       // 16-byte load, 4-byte aligned

// EGP_CHAIN_TRACE=<workgroup> (tools/probes/chain_trace.py): wall_clock64 stamps (100 MHz) of lane 0 of wave 0 on its first tile, and the
// cycles it spent at stage boundaries
#ifdef EGP_CHAIN_TRACE
__device__ long long g_chain_trace[32];
#define CH_TR(i) do { if (blockIdx.x == EGP_CHAIN_TRACE && threadIdx.x == 0 && t == (int)blockIdx.x) g_chain_trace[i] = wall_clock64(); } while (0)
#define CH_TRW0 const long long trw0 = __builtin_readcyclecounter()
#define CH_TRW1 do { if (blockIdx.x == EGP_CHAIN_TRACE && threadIdx.x == 0) g_chain_trace[16] += __builtin_readcyclecounter() - trw0; } while (0)
#else
#define CH_TR(i) do { } while (0)
#define CH_TRW0 do { } while (0)
#define CH_TRW1 do { } while (0)
#endif

constexpr int CH_ROWS = 128;                 // batch rows per tile: 4 waves x 32
constexpr int CH_MAXB = 10;                  // i-blocks (32 output features each) of the widest layer
constexpr int CH_SLOT = CH_MAXB * 3 * 1024;  // bytes of one ring slot: one k-step of the widest layer
constexpr int CH_NSLOT = 3;

struct ChainArgs {
    int n, n_tiles;
    // first operand rows: columns [0, c1) from src1 (row gather[r] if gather, else r), columns [c1, c1 + c2) from src2 (row r)
    const float *src1; long ld1; const long long *gather; int c1;
    const float *src2; long ld2; int c2;
    const unsigned char *packed[3];          // fragment streams (egp_mlp_chain_pack_f32)
    const float *bias[3];                    // forward only
    int n_out[3];                            // real output features of the three products
    int ks[3];                               // k-steps (16 k each) of the three products
    const float *mask1, *mask2;              // backward: saved activations [feature][ldT] whose sign masks products 1 and 2
    float *inT, *o1T, *o2T; long ldT;        // [feature][ldT] saves (may be NULL): the input, and the outputs of products 1 and 2
    float *out; long ld_out;                 // product 3: row r -> out[(scatter ? scatter[r] : r) * ld_out + f]
    const long long *scatter;
};

// exact three-way split by truncation (egp_gemm.hip: split3_pair): the pieces of two floats pack with one v_perm each
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &ph, unsigned &pm, unsigned &pl) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float h0 = __uint_as_float(u0 & 0xffff0000u), h1 = __uint_as_float(u1 & 0xffff0000u);
    const float r0 = x0 - h0, r1 = x1 - h1;
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float m0 = __uint_as_float(v0 & 0xffff0000u), m1 = __uint_as_float(v1 & 0xffff0000u);
    const float l0 = r0 - m0, l1 = r1 - m1;
    ph = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    pm = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    pl = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

// eight consecutive k of one column -> the three B fragments
__device__ __forceinline__ void split_frag(const float (&v)[8], bf16x8 (&img)[3]) {
    unsigned q[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3_pair(v[2 * j], v[2 * j + 1], q[0][j], q[1][j], q[2][j]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const u32x4 w = {q[c][0], q[c][1], q[c][2], q[c][3]};
        img[c] = __builtin_bit_cast(bf16x8, w);
    }
}

// LDS-DMA of one fragment (64 lanes x 16 bytes): each lane's 16 bytes land at lds_dst + 16 lane. M0 is written in the statement
// that reads it (the compiler does not preserve it); the load is invisible to the compiler's wait counting -- see ch_wait_vm.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// wait until at most n memory operations of this wave are outstanding (n wave-uniform, rounded down to a built constant: never
// waits for less than asked). Memory operations complete in order, so whatever the compiler has in flight besides the DMA
// pieces only makes the wait stricter.
__device__ __forceinline__ void ch_wait_vm(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one k-step of a product: acc[b] += A(b) * B for the NB i-blocks; A fragments from the ring slot, six terms smallest first.
// The fragments of the next block PAIR are read while the current pair multiplies (an LDS read issued right in front of the
// MFMA that needs it leaves the matrix pipe idle for the LDS round trip: four waves read 84-120 kB per k-step), and the two
// blocks of a pair alternate, so that consecutive MFMAs never wait for each other's accumulator.
struct ChFrag { bf16x8 a[3]; };
__device__ __forceinline__ ChFrag ch_frag(const unsigned char *slot, int lane, int b) {
    ChFrag f;
#pragma unroll
    for (int c = 0; c < 3; ++c) f.a[c] = *reinterpret_cast<const bf16x8 *>(slot + ((b * 3 + c) * 64 + lane) * 16);
    return f;
}
#ifdef EGP_CHAIN_NO_MFMA        // (timing experiment: everything but the products)
#define CH_MFMA(A, B, C) ((C)[0] += (float)(A)[0] + (float)(B)[0], (C))
#else
#define CH_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif
template <int NB>
__device__ __forceinline__ void ch_kstep(const unsigned char *slot, int lane, const bf16x8 (&bi)[3], f32x16 (&acc)[NB]) {
    ChFrag cur0 = ch_frag(slot, lane, 0), cur1 = ch_frag(slot, lane, NB > 1 ? 1 : 0);
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
        ChFrag nxt0 = cur0, nxt1 = cur1;
        if (b + 2 < NB) nxt0 = ch_frag(slot, lane, b + 2);
        if (b + 3 < NB) nxt1 = ch_frag(slot, lane, b + 3);
        f32x16 c0 = acc[b], c1 = acc[b + 1 < NB ? b + 1 : b];
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // (A piece, B piece) of the six terms
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            c0 = CH_MFMA(cur0.a[TA[t]], bi[TB[t]], c0);
            if (b + 1 < NB) c1 = CH_MFMA(cur1.a[TA[t]], bi[TB[t]], c1);
        }
        acc[b] = c0;
        if (b + 1 < NB) acc[b + 1] = c1;
        cur0 = nxt0; cur1 = nxt1;
    }
}

// The weight stream of a workgroup: the k-steps of products 0, 1, 2 of its first tile, then of its next tile, ... -- one stage
// per k-step, stage q in ring slot q mod 3. Constants here; the running state is a handful of wave-uniform ints the kernel
// keeps in scalar registers (passed by reference, everything inlined).
struct ChConst {
    const unsigned char *b0, *b1, *b2;       // fragment streams of the three products
    int ks0, ks1, ks2, nf0, nf1, nf2;        // k-steps, fragments per k-step
    unsigned lds0;                           // LDS byte address of ring slot 0
    int wave, lane;
};
// request the next stage (this wave's share: fragments wave, wave + 4, ...); returns how many fragments that was
__device__ __forceinline__ int ch_request(const ChConst c, int &left, int &ip, int &is, int &islot) {
    if (left == 0) return 0;
    const unsigned char *bp = ip == 0 ? c.b0 : (ip == 1 ? c.b1 : c.b2);
    const int nf = ip == 0 ? c.nf0 : (ip == 1 ? c.nf1 : c.nf2);
    const int kp = ip == 0 ? c.ks0 : (ip == 1 ? c.ks1 : c.ks2);
    const unsigned char *src = bp + (long)is * nf * 1024 + c.lane * 16;
    const unsigned dst = c.lds0 + (unsigned)islot * CH_SLOT;
    int cnt = 0;
#ifndef EGP_CHAIN_NO_GLDS       // (timing experiments only: EGP_CHAIN_NO_GLDS multiplies stale LDS, EGP_CHAIN_NO_MFMA skips the products)
    for (int f = c.wave; f < nf; f += 4) { glds16(src + f * 1024, __builtin_amdgcn_readfirstlane(dst + f * 1024)); ++cnt; }
#endif
    is = is + 1;
    if (is == kp) { is = 0; ip = ip == 2 ? 0 : ip + 1; }
    islot = islot == CH_NSLOT - 1 ? 0 : islot + 1;
    left = left - 1;
    return cnt;
}
// A stage boundary: my pieces of the stage multiplied next have landed (the ones of the stage after it may still fly),
// everybody's have after the barrier -- and everybody has finished reading the previous stage, whose slot the stage requested
// now takes. Returns the slot to multiply.
__device__ __forceinline__ const unsigned char *ch_stage(const ChConst c, const unsigned char *ring, int &left, int &ip, int &is, int &islot,
                                                         int &cslot, int &m_next) {
    __builtin_amdgcn_sched_barrier(0);                 // (nothing of the next k-step -- conversions, fragment reads -- is hoisted above the boundary)
    CH_TRW0;
    ch_wait_vm(m_next);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    CH_TRW1;
    m_next = ch_request(c, left, ip, is, islot);
    const unsigned char *slot = ring + cslot * CH_SLOT;
    cslot = cslot == CH_NSLOT - 1 ? 0 : cslot + 1;
    __builtin_amdgcn_sched_barrier(0);
    return slot;
}

// feature of accumulator register r of i-block b in the lane's half h: 32 b + 8 (r >> 2) + 4 h + (r & 3)
__device__ __forceinline__ constexpr int ch_feat(int b, int r, int half) { return 32 * b + 8 * (r >> 2) + 4 * half + (r & 3); }

// epilogue of products 1 and 2 (N real output features, compile time): forward -> relu(acc + bias); backward -> acc where the
// saved activation is positive, else 0. The result stays in the accumulators (the next product's B operand) and is saved as
// [feature][row]. `bias`: padded to whole 32-feature blocks. Branch-free arithmetic; the stores sit under ONE row test.
template <int NB, int N, bool BWD>
__device__ __forceinline__ void ch_epilogue(f32x16 (&acc)[NB], const float *__restrict__ bias, const float *__restrict__ mask,
                                            float *__restrict__ saveT, long ldT, int row, int rc, bool valid, int half) {
    // [feature][row] addresses walk down the features with a running pointer (ldT is opaque to the optimiser in the tile loop:
    // 300 loop-invariant row offsets hoisted into registers were what made round 4's first build of this kernel spill)
    const float *mp = BWD ? mask + (long)(4 * half) * ldT + rc : nullptr;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        __builtin_amdgcn_sched_barrier(0);              // one block at a time: the epilogue must not cost registers
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int f0 = 32 * b + 8 * g4 + 4 * half;           // the lane's four consecutive features of this group
            float add[4] = {0.f, 0.f, 0.f, 0.f}, m[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (!BWD) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + f0);
                add[0] = bv.x; add[1] = bv.y; add[2] = bv.z; add[3] = bv.w;
            } else {
                const float *me = mp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (32 * b + 8 * g4 + 4 + e < N) m[e] = *me;                            // both halves inside the matrix
                    else if (32 * b + 8 * g4 + e < N) m[e] = f0 + e < N ? *me : 0.0f;        // (reads row f0 + e <= N + 3: the saves are padded)
                    me += ldT;
                }
                mp += 8 * ldT;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                float v = acc[b][r];
                if constexpr (BWD) v = m[e] > 0.0f ? v : 0.0f;
                else v = fmaxf(v + add[e], 0.0f);
                if (32 * b + 8 * g4 + 4 + e >= N) v = (f0 + e < N) ? v : 0.0f;      // (only the block that straddles N pays the test)
                acc[b][r] = v;
            }
        }
    }
    if (saveT && valid) {
        float *sp = saveT + (long)(4 * half) * ldT + row;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float *se = sp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    if (32 * b + 8 * g4 + 4 + e < N) *se = acc[b][r];
                    else if (32 * b + 8 * g4 + e < N) { if (32 * b + 8 * g4 + 4 * half + e < N) *se = acc[b][r]; }
                    se += ldT;
                }
                sp += 8 * ldT;
            }
    }
}

template <int N1, int N2, int N3, bool BWD>
__global__ __launch_bounds__(256) void k_mlp_chain(ChainArgs g) {
    constexpr int NB1 = (N1 + 31) / 32, NB2 = (N2 + 31) / 32, NB3 = (N3 + 31) / 32;
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < g.n_tiles; t += gridDim.x) ++my_tiles;
    if (my_tiles == 0) return;
    ChConst sc;
    sc.b0 = g.packed[0]; sc.b1 = g.packed[1]; sc.b2 = g.packed[2];
    sc.ks0 = g.ks[0]; sc.ks1 = g.ks[1]; sc.ks2 = g.ks[2];
    sc.nf0 = 3 * NB1; sc.nf1 = 3 * NB2; sc.nf2 = 3 * NB3;
    sc.lds0 = (unsigned)(size_t)ring; sc.wave = wave; sc.lane = lane;
    const int ks0 = sc.ks0, ks1 = sc.ks1, ks2 = sc.ks2;
    int left = my_tiles * (ks0 + ks1 + ks2);     // stages not yet requested
    int ip = 0, is = 0, islot = 0;               // product / k-step / ring slot of the next stage to request
    int cslot = 0;                               // ring slot of the stage multiplied next
    ch_request(sc, left, ip, is, islot);
    int m_next = ch_request(sc, left, ip, is, islot);      // fragments this wave requested for the stage AFTER the one multiplied next
#define CH_STAGE() ch_stage(sc, ring, left, ip, is, islot, cslot, m_next)
    const int kin = g.c1 + g.c2;
    for (int t = blockIdx.x; t < g.n_tiles; t += gridDim.x) {
        long ldT = g.ldT;
        asm volatile("" : "+s"(ldT));                   // not loop-invariant as far as the optimiser knows (see ch_epilogue)
        const int row = t * CH_ROWS + 32 * wave + (lane & 31);
        const bool valid = row < g.n;
        const int rc = valid ? row : g.n - 1;
        const float *p1 = g.src1 + (g.gather ? (long)g.gather[rc] : (long)rc) * g.ld1;
        const float *p2 = g.src2 ? g.src2 + (long)rc * g.ld2 - g.c1 : p1;          // indexed by the global column
        // the lane's eight input columns of k-step s: 16 s + 8 half + j (zeros past the last column). Two 16-byte loads where the
        // eight columns lie inside one source (4-byte aligned is enough for a global load), element loads at the seams.
        auto load8 = [&](int s, float (&v)[8]) {
            const int c0 = 16 * s + 8 * half;
#ifdef EGP_CHAIN_NO_X           // (timing experiment: no input loads)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.001f * (c0 + j);
            return;
#endif
            const float *src = nullptr;
            if (16 * s + 16 <= g.c1) src = p1 + c0;                               // wave-uniform tests
            else if (16 * s >= g.c1 && 16 * s + 16 <= kin) src = p2 + c0;
            if (src) {
                const f32x4u lo = *reinterpret_cast<const f32x4u *>(src), hi = *reinterpret_cast<const f32x4u *>(src + 4);
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cc = min(c0 + j, kin - 1);
                    const float x = (cc < g.c1 ? p1 : p2)[cc];
                    v[j] = c0 + j < kin ? x : 0.0f;
                }
            }
        };
        CH_TR(0);
        // ---- product 1: B from memory, requested two k-steps ahead
        f32x16 acc1[NB1];
#pragma unroll
        for (int b = 0; b < NB1; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[b][r] = 0.0f;
        float xa[8], xb[8];
        load8(0, xa);
        load8(ks0 > 1 ? 1 : 0, xb);
        auto step1 = [&](int s, float (&xv)[8]) {
            bf16x8 bi[3];
            split_frag(xv, bi);
            if (g.inT && valid) {
                float *ip_ = g.inT + (long)(16 * s + 8 * half) * ldT + row;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (16 * s + 8 * half + j < kin) *ip_ = xv[j];
                    ip_ += ldT;
                }
            }
            const unsigned char *slot = CH_STAGE();
            load8(s + 2 < ks0 ? s + 2 : ks0 - 1, xv);            // (past the end: a harmless reload)
            ch_kstep<NB1>(slot, lane, bi, acc1);
        };
        for (int s = 0; s < ks0; s += 2) {
            step1(s, xa);
            if (s + 1 < ks0) step1(s + 1, xb);
        }
        CH_TR(1);
        ch_epilogue<NB1, N1, BWD>(acc1, g.bias[0], g.mask1, g.o1T, ldT, row, rc, valid, half);
        CH_TR(2);
        // ---- product 2
        f32x16 acc2[NB2];
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[b][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2 * NB1; ++s) {              // chained: the previous product's accumulators are the B operand
            if (s < ks1) {                               // (ks1 <= 2 NB1; wave-uniform)
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc1[s >> 1][8 * (s & 1) + j];
                bf16x8 bi[3];
                split_frag(v, bi);
                ch_kstep<NB2>(CH_STAGE(), lane, bi, acc2);
            }
        }
        CH_TR(3);
        ch_epilogue<NB2, N2, BWD>(acc2, g.bias[1], g.mask2, g.o2T, ldT, row, rc, valid, half);
        CH_TR(4);
        // ---- product 3
        f32x16 acc3[NB3];
#pragma unroll
        for (int b = 0; b < NB3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[b][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2 * NB2; ++s) {
            if (s < ks2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc2[s >> 1][8 * (s & 1) + j];
                bf16x8 bi[3];
                split_frag(v, bi);
                ch_kstep<NB3>(CH_STAGE(), lane, bi, acc3);
            }
        }
        CH_TR(5);
        if (valid) {
            float *o = g.out + (g.scatter ? (long)g.scatter[row] : (long)row) * g.ld_out;
#pragma unroll
            for (int b = 0; b < NB3; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = ch_feat(b, r, half);
                    if (ch_feat(b, r, 0) < N3 && f < N3) o[f] = acc3[b][r] + (BWD ? 0.0f : g.bias[2][f]);
                }
        }
        CH_TR(6);
    }
#undef CH_STAGE
}

// nn.Linear weight W[n_out][k_in] (transpose: the operand is W^T, i.e. element (i, k) = W[k][i]) -> fragment stream:
// [k-step s][i-block b][piece c][lane][8 bf16], lane = (i & 31) + 32 kg; the lane's eight k of the k-step are
//   natural order (a product fed from memory): 16 s + 8 kg + j
//   chained order (fed from accumulators)    : 32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 kg + (j & 3)
// pieces by truncation, as split3_pair. Zeros outside the matrix.
__global__ __launch_bounds__(256) void k_chain_pack(const float *__restrict__ W, long ldw, int n_rows_a, int n_k, int transpose, int chained,
                                                    int nb, int ks, unsigned short *__restrict__ dst) {
    const long total = (long)ks * nb * 64 * 8;           // (stage, block, lane, j): the three pieces are written together
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const long sb = e >> 9;
        const int b = (int)(sb % nb), s = (int)(sb / nb);
        const int i = 32 * b + (lane & 31), kg = lane >> 5;
        const int k = chained ? 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * kg + (j & 3) : 16 * s + 8 * kg + j;
        float x = 0.0f;
        if (i < n_rows_a && k < n_k) x = transpose ? W[(long)k * ldw + i] : W[(long)i * ldw + k];
        const unsigned u = __float_as_uint(x);
        const float h = __uint_as_float(u & 0xffff0000u);
        const float r1 = x - h;
        const unsigned v = __float_as_uint(r1);
        const float m = __uint_as_float(v & 0xffff0000u);
        const float l = r1 - m;
        const long o = (((long)(s * nb + b) * 3) * 64 + lane) * 8 + j;
        dst[o] = (unsigned short)(u >> 16);
        dst[o + 64 * 8] = (unsigned short)(v >> 16);
        dst[o + 2 * 64 * 8] = (unsigned short)(__float_as_uint(l) >> 16);
    }
}

int after_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}

template <int N1, int N2, int N3, bool BWD>
int launch_chain(const ChainArgs &g, hipStream_t s) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        EGP_HIP_CHECK(hipGetDevice(&dev));
        EGP_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    auto kern = k_mlp_chain<N1, N2, N3, BWD>;
    static bool attr_set = false;
    if (!attr_set) {
        EGP_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CH_NSLOT * CH_SLOT));
        attr_set = true;
    }
    const int grid = g.n_tiles < n_cu ? g.n_tiles : n_cu;
    kern<<<dim3((unsigned)grid), dim3(256), CH_NSLOT * CH_SLOT, s>>>(g);
    return after_launch("k_mlp_chain");
}

}  // namespace

extern "C" {

#ifdef EGP_CHAIN_TRACE
int egp_chain_trace_read(long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_chain_trace), sizeof(long long) * 32) != hipSuccess) return -1;
    if (reset) { long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

int32_t egp_mlp_chain_ksteps(int32_t k_in) { return k_in <= 0 ? 0 : (k_in + 15) / 16; }
int64_t egp_mlp_chain_pack_bytes(int32_t n_out, int32_t k_in) {
    if (n_out <= 0 || k_in <= 0) return 0;
    return (int64_t)egp_mlp_chain_ksteps(k_in) * ((n_out + 31) / 32) * 3 * 1024;
}

int egp_mlp_chain_pack_f32(const float *weight, int64_t ldw, int32_t n_out, int32_t k_in, int32_t transpose, int32_t chained, void *packed,
                           void *stream) {
    EGP_REQUIRE(weight && packed && n_out > 0 && k_in > 0, "bad weight");
    EGP_REQUIRE(((uintptr_t)packed & 15) == 0, "the fragment stream must be 16-byte aligned");
    const int nb = (n_out + 31) / 32, ks = egp_mlp_chain_ksteps(k_in);
    EGP_REQUIRE(nb <= CH_MAXB, "more than 320 output features");
    const long total = (long)ks * nb * 512;
    const long blocks = (total + 255) / 256;
    k_chain_pack<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream>>>(weight, (long)ldw, n_out, k_in, transpose, chained, nb,
                                                                                                        ks, (unsigned short *)packed);
    return after_launch("k_chain_pack");
}

int egp_mlp_chain_f32(const egp_mlp_chain_desc *d, void *stream) {
    EGP_REQUIRE(d, "descriptor is NULL");
    EGP_REQUIRE(d->n >= 0, "n < 0");
    if (d->n == 0) return EGP_OK;
    EGP_REQUIRE(d->src1 && d->c1 > 0 && d->c2 >= 0 && (d->c2 == 0 || d->src2) && d->out, "NULL operand");
    for (int p = 0; p < 3; ++p) {
        EGP_REQUIRE(d->packed[p] && d->dims[p] > 0 && d->dims[p + 1] > 0, "NULL / empty layer");
        EGP_REQUIRE(d->backward || d->bias[p], "the forward chain needs every bias");
    }
    EGP_REQUIRE(d->c1 + d->c2 == d->dims[0], "input columns do not add up to dims[0]");
    EGP_REQUIRE(!d->backward || (d->mask1 && d->mask2), "the backward chain needs the saved activations of both hidden layers");
    EGP_REQUIRE(!(d->inT || d->o1T || d->o2T || d->backward) || d->ldT >= d->n, "ldT < n");
    ChainArgs g{};
    g.n = d->n; g.n_tiles = (d->n + CH_ROWS - 1) / CH_ROWS;
    g.src1 = d->src1; g.ld1 = d->ld1; g.gather = (const long long *)d->gather; g.c1 = d->c1;
    g.src2 = d->src2; g.ld2 = d->ld2; g.c2 = d->c2;
    for (int p = 0; p < 3; ++p) {
        g.packed[p] = (const unsigned char *)d->packed[p];
        g.bias[p] = d->bias[p];
        g.n_out[p] = d->dims[p + 1];
        g.ks[p] = egp_mlp_chain_ksteps(d->dims[p]);
    }
    g.mask1 = d->mask1; g.mask2 = d->mask2;
    g.inT = d->inT; g.o1T = d->o1T; g.o2T = d->o2T; g.ldT = d->ldT;
    g.out = d->out; g.ld_out = d->ld_out; g.scatter = (const long long *)d->scatter;
    const int nb1 = (d->dims[1] + 31) / 32, nb2 = (d->dims[2] + 31) / 32;
    // chained k-steps must fit the previous product's blocks (two k-steps per block)
    EGP_REQUIRE(g.ks[1] <= 2 * nb1 && g.ks[2] <= 2 * nb2, "layer widths do not chain");
    hipStream_t s = (hipStream_t)stream;
#define CH_CASE(A, B, C, W)                                                                     \
    if (d->dims[1] == A && d->dims[2] == B && d->dims[3] == C && (d->backward != 0) == W) return launch_chain<A, B, C, W>(g, s);
    CH_CASE(300, 200, 52, false) CH_CASE(300, 200, 1, false)       // 243 -> 300 -> 200 -> 52 | 1
    CH_CASE(200, 300, 128, true)                                   // 52 | 1 -> 200 -> 300 -> 128
#undef CH_CASE
    egp::set_error("invalid argument: no chain kernel is built for %d -> %d -> %d -> %d (%s)", d->dims[0], d->dims[1], d->dims[2], d->dims[3],
                   d->backward ? "backward" : "forward");
    return EGP_E_INVALID;
}

}  // extern "C"
