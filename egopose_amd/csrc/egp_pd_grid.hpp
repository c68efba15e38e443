// K1 on a lane GRID: the stable-PD solve (compute_torque / compute_desired_accel, ego_pose/envs/humanoid_v1.py:130-156)
// of one env by one wavefront whose 64 lanes form 4 rows x 16 columns of the matrix instead of 64 matrix rows.
//
// Why: with "lane i owns row i" (k_pd_torque_tree58) a pivot's update touches one COLUMN per instruction, only the
// pivot's ancestor rows do useful work (<= 28 of 64 lanes) and every column costs two v_readlane for the broadcast of
// the pivot row's entry: 852 column updates x 3 VALU issues. On the grid, element (i, j) of the augmented matrix
// [ rhs | M + Kd dt ] lives in lane (r = i % 4, c = pos(j) % 16), register A[i / 4][pos(j) / 16]; one instruction
// updates a 4 x 16 tile (rows 4 ih .. 4 ih + 3, column positions 16 jh .. 16 jh + 15):
//     A[ih][jh] += bcast_c(A[ih][jh_k]) * nrow[jh]          v_fmac_f64_dpp ... row_newbcast:c_k
// where the multiplier M[i][k] of each row is broadcast inside its 16-lane row by the DPP operand itself (gfx90a+
// allows row_newbcast on the 64-bit VOP2 forms) and nrow = -(row k) / pivot sits column-wise in every 16-lane row (one
// ds_bpermute pair per 16 columns fetches it from the 16-lane row that owns matrix row k). A pivot therefore costs
// |row groups| x |column groups| instructions -- 852 for the whole humanoid tree, the same count as before, but with
// no broadcast instructions next to them.
//
// Same algorithm as the row kernels: Gauss-Jordan without pivoting (SPD), pivots ordered leaves -> root so that a pivot
// row is non-zero only on its ancestor columns (MuJoCo's L^T D L sparsity, no fill-in); the dof tree is the compile-time
// table of egp_tree58.inc. The products are associated as M[i][k] * (M[k][j] / d) instead of (M[i][k] / d) * M[k][j]:
// results agree with the row kernels to round-off, not bit for bit.
//
// Included by egp_kernels.hip (needs Tree58, fast_rcp, DevModel, PdLd, PdDone).
#pragma once

struct Grid58 {
    static constexpr int NV = Tree58::NV;
    static constexpr int IHN = 15, JHN = 4;       // A[ih][jh]: rows 4 ih + r, column positions 16 jh + c; lane = 16 r + c
    // column positions: 0..3 (bank 0 of every 16-lane row) receive 1 / pivot of the lane's row, 4 = right-hand side,
    // 5 + j = dof j, 63 unused
    static constexpr int RHS_POS = 4, COL0 = 5;
    // LDS image of one env, in doubles: the sparse inertia (nM = 910), the right-hand side, one zero
    static constexpr int IMG_B = 912, IMG_ZERO = 970, IMG_N = 976;

    static constexpr bool is_anc(int a, int k) {        // a is a proper ancestor of k
        for (int p = Tree58::PARENT[k]; p >= 0; p = Tree58::PARENT[p]) if (p == a) return true;
        return false;
    }
    // groups of four rows holding a row the pivot changes: its ancestors and its descendants
    static constexpr unsigned row_groups(int k) {
        unsigned m = 0;
        for (int i = 0; i < NV; ++i) if (is_anc(i, k) || is_anc(k, i)) m |= 1u << (i / 4);
        return m;
    }
    // groups of sixteen column positions the pivot changes: its ancestors and the right-hand side
    static constexpr unsigned col_groups(int k) {
        unsigned m = 1u << (RHS_POS / 16);
        for (int j = 0; j < NV; ++j) if (is_anc(j, k)) m |= 1u << ((COL0 + j) / 16);
        return m;
    }
    // elimination order: the five limbs round-robin (consecutive pivots are independent, so the fetch of the next pivot
    // row overlaps the current update), then the trunk
    struct Order { int k[NV]; };
    static constexpr Order make_order() {
        Order o{};
        int n = 0;
        // the limbs of humanoid_1205_v1 (dof ranges, each a chain): arms 24..33 / 34..43, 18..23, legs 44..50 / 51..57
        int limb_top[5] = {33, 43, 23, 50, 57}, limb_bot[5] = {24, 34, 18, 44, 51};
        bool any = true;
        while (any) {
            any = false;
            for (int l = 0; l < 5; ++l)
                if (limb_top[l] >= limb_bot[l]) { o.k[n++] = limb_top[l]--; any = true; }
        }
        for (int k = 17; k >= 0; --k) o.k[n++] = k;
        return o;
    }
    // a valid order eliminates every dof exactly once and all descendants of a dof before the dof itself
    static constexpr bool order_ok(const Order &o) {
        bool seen[NV] = {};
        for (int t = 0; t < NV; ++t) {
            const int k = o.k[t];
            if (k < 0 || k >= NV || seen[k]) return false;
            for (int i = 0; i < NV; ++i) if (is_anc(k, i) && !seen[i]) return false;
            seen[k] = true;
        }
        return true;
    }
};
inline constexpr Grid58::Order GRID58_ORDER = Grid58::make_order();
static_assert(Grid58::order_ok(GRID58_ORDER), "grid58: elimination order is not leaves -> root for the compiled-in dof tree");

// 1 / x: v_rcp_f64 (2^-25 relative on gfx950) and ONE third-order step r (1 + e + e^2), e = 1 - x r: three FMAs instead of the
// four of two Newton steps (fast_rcp), <= 1 ulp for normal x (tools/probes/rcp_probe.hip)
__device__ __forceinline__ double grid_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// acc += bcast(src, lane C of every 16-lane row) * mul, written only in the 16-lane rows of RM
template <int C, int RM>
__device__ __forceinline__ void grid_fmac(double &acc, const double &src, const double &mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C), "n"(RM));
}
// the same with the two wait states a DPP read needs after a VALU write of its source (the compiler does not see into
// the asm): for the places where the source may have been written by the instruction just before
template <int C, int RM>
__device__ __forceinline__ void grid_fmac_safe(double &acc, const double &src, const double &mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C), "n"(RM));
}
template <int C>
__device__ __forceinline__ double grid_bcast(const double &src) {
    double d;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "n"(C));
    return d;
}
// dst[lanes 0..3 of the 16-lane rows in RM] = v (v is uniform over the lanes)
template <int RM>
__device__ __forceinline__ void grid_put_bank0(double &dst, const double &v) {
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:%2 bank_mask:0x1" : "+v"(dst) : "v"(v), "n"(RM));
}
// value of lane `byte_addr / 4` (ds_bpermute: a cross-lane read through the LDS crossbar, no LDS storage involved)
__device__ __forceinline__ double grid_fetch(const double &v, int byte_addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int K, int IH, int JH>
__device__ __forceinline__ void grid_tile_update(double (&A)[Grid58::IHN][Grid58::JHN], const double (&nrow)[Grid58::JHN]) {
    constexpr int pos = Grid58::COL0 + K, jhk = pos / 16, ck = pos % 16;
    constexpr int RM = IH == K / 4 ? (0xf & ~(1 << (K % 4))) : 0xf;          // the pivot's own row stays as it is
    if constexpr ((Grid58::col_groups(K) >> JH) & 1) grid_fmac<ck, RM>(A[IH][JH], A[IH][jhk], nrow[JH]);
}

template <int K, int IH>
__device__ __forceinline__ void grid_rows_update(double (&A)[Grid58::IHN][Grid58::JHN], const double (&nrow)[Grid58::JHN], const double &inv) {
    if constexpr ((Grid58::row_groups(K) >> IH) & 1) {
        constexpr int jhk = (Grid58::COL0 + K) / 16;
        // the tile that holds column k goes last: it overwrites the multipliers the other tiles of the row group read
        if constexpr (jhk != 0) grid_tile_update<K, IH, 0>(A, nrow);
        if constexpr (jhk != 1) grid_tile_update<K, IH, 1>(A, nrow);
        if constexpr (jhk != 2) grid_tile_update<K, IH, 2>(A, nrow);
        if constexpr (jhk != 3) grid_tile_update<K, IH, 3>(A, nrow);
        grid_tile_update<K, IH, jhk>(A, nrow);
    }
    // 1 / pivot into bank 0 of the pivot's own row, as soon as its row group is through (a later pivot of the same
    // group then does not wait for the rest of this one)
    if constexpr (IH == K / 4) grid_put_bank0<(1 << (K % 4))>(A[IH][0], inv);
}

template <int K, int FIRST, int... IH>
__device__ __forceinline__ void grid_all_rows(double (&A)[Grid58::IHN][Grid58::JHN], const double (&nrow)[Grid58::JHN],
                                              const double &inv, std::integer_sequence<int, IH...>) {
    // the row group of the NEXT pivot first: its fetch can then start while the other groups are still being updated
    grid_rows_update<K, FIRST>(A, nrow, inv);
    ((IH != FIRST ? grid_rows_update<K, IH>(A, nrow, inv) : (void)0), ...);
}

// one pivot of the elimination. xa[r] = ds_bpermute address of "my column in 16-lane row r".
template <int T>
__device__ __forceinline__ void grid_pivot(double (&A)[Grid58::IHN][Grid58::JHN], const int (&xa)[4]) {
    constexpr int K = GRID58_ORDER.k[T];
    constexpr int NEXT = T + 1 < Grid58::NV ? GRID58_ORDER.k[T + 1] : 0;
    constexpr int ihk = K / 4, rk = K % 4, pos = Grid58::COL0 + K, jhk = pos / 16, ck = pos % 16;
    constexpr unsigned CG = Grid58::col_groups(K);
    double p[Grid58::JHN] = {0.0, 0.0, 0.0, 0.0}, nrow[Grid58::JHN] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jh = 0; jh < Grid58::JHN; ++jh)
        if (((CG | (1u << jhk)) >> jh) & 1) p[jh] = grid_fetch(A[ihk][jh], xa[rk]);
    const double d = grid_bcast<ck>(p[jhk]);
    const double inv = grid_rcp(d);
    const double ninv = -inv;
#pragma unroll
    for (int jh = 0; jh < Grid58::JHN; ++jh)
        if ((CG >> jh) & 1) nrow[jh] = p[jh] * ninv;
    grid_all_rows<K, NEXT / 4>(A, nrow, inv, std::make_integer_sequence<int, Grid58::IHN>{});
    if constexpr (T + 1 < Grid58::NV) grid_pivot<T + 1>(A, xa);
}

template <typename TIO>
__global__ __launch_bounds__(256) void k_pd_torque_grid58(DevModel m, PdLd ld, const unsigned short *__restrict__ grid_off,
                                                          const TIO *__restrict__ qpos, const TIO *__restrict__ qvel,
                                                          const TIO *__restrict__ action, const TIO *__restrict__ qM,
                                                          const TIO *__restrict__ C, int n, TIO *__restrict__ torque,
                                                          TIO *__restrict__ torque_raw, PdDone done) {
    using G = Grid58;
    constexpr int NREG = G::IHN * G::JHN;
    __shared__ unsigned short s_off[NREG * 64];      // [register][lane] -> byte offset into the env's LDS image
    __shared__ double s_img[4][G::IMG_N];
    __shared__ double s_x[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    const bool valid = env < n;
    const int row = lane < PD_NV ? lane : PD_NV - 1;
    const int act = row >= 6 ? row - 6 : 0;
    TIO r_q = TIO(0), r_v = TIO(0), r_c = TIO(0), r_a = TIO(0);
    if (valid) {
        r_q = qpos[env * ld.qpos + 7 + act];
        r_v = qvel[env * ld.qvel + row];
        r_c = C[env * ld.bias + row];
        r_a = action[env * ld.action + act];
    }
    const double c_kp = m.jkp[act], c_kd = m.jkd[act], c_ref = m.a_ref[act], c_scale = m.a_scale[act], c_lim = m.torque_lim[act];
    const int diag_id = m.m_map[row * PD_NV + row];
    {
        constexpr int OFF_IT = NREG * 64 / 2 / 256 + 1;          // the table as 32-bit words
        unsigned t_off[OFF_IT];
        const unsigned *off32 = reinterpret_cast<const unsigned *>(grid_off);
#pragma unroll
        for (int k = 0; k < OFF_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            t_off[k] = i < NREG * 32 ? off32[i] : 0u;
        }
        constexpr int QM_IT = PD_NM_MAX / 64;
        TIO t_qM[QM_IT];
        const TIO *src = qM + (valid ? env : 0) * ld.qM;
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) {
            const int i = lane + 64 * k;
            t_qM[k] = (valid && i < m.nM) ? src[i] : TIO(0);
        }
        unsigned *s_off32 = reinterpret_cast<unsigned *>(s_off);
#pragma unroll
        for (int k = 0; k < OFF_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < NREG * 32) s_off32[i] = t_off[k];
        }
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) s_img[wave][lane + 64 * k] = (double)t_qM[k];       // 960 <= IMG_N: covers the inertia
    }
    double kp = 0.0, kd = 0.0, eq = 0.0;
    if (row >= 6) {
        kp = c_kp;
        kd = c_kd;
        const double target = c_ref + (double)r_a * c_scale;
        eq = (double)r_q - target;
    }
    const double qv = (double)r_v;
    __syncthreads();
    // right-hand side, Kd dt on the diagonal, the zero every structurally empty element points at
    if (lane < PD_NV) {
        s_img[wave][G::IMG_B + lane] = -(double)r_c - kp * eq - kd * qv;
        s_img[wave][diag_id] += kd * m.sub_dt;
    } else {
        s_img[wave][G::IMG_B + lane] = 0.0;            // 970..975: the zero and the tail of the image
    }
    __syncthreads();
    if (valid) {
        double A[G::IHN][G::JHN];
        const char *img = reinterpret_cast<const char *>(s_img[wave]);
#pragma unroll
        for (int t = 0; t < NREG; ++t) A[t / G::JHN][t % G::JHN] = *reinterpret_cast<const double *>(img + s_off[t * 64 + lane]);
        const int xc = (lane & 15) * 4;
        const int xa[4] = {xc, xc + 64, xc + 128, xc + 192};
        grid_pivot<0>(A, xa);
        // lane (r, RHS_POS) of A[ih][0] holds the eliminated right-hand side of row 4 ih + r, lanes (r, 0..3) 1 / pivot
#pragma unroll
        for (int ih = 0; ih < G::IHN; ++ih) {
            double x = 0.0;
            grid_fmac_safe<0, 0xf>(x, A[ih][0], A[ih][0]);
            if ((lane & 15) == G::RHS_POS) s_x[wave][4 * ih + (lane >> 4)] = x;
        }
        __builtin_amdgcn_wave_barrier();
        const double qacc = s_x[wave][row];
        if (lane < PD_NV && row >= 6) {
            const double ev = qv + qacc * m.sub_dt;
            const double tau = -kp * eq - kd * ev;
            const double tc = fmin(fmax(tau, -c_lim), c_lim);
            torque[env * m.nu + act] = (TIO)tc;
            if (torque_raw) torque_raw[env * m.nu + act] = (TIO)tau;
        }
    }
    if (done.counter) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            const unsigned prev = atomicAdd(done.counter, 1u);
            if (prev == gridDim.x - 1) {
                *done.counter = 0u;
                __threadfence_system();
                __hip_atomic_store(done.host_flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// host side: the [register][lane] gather table of the kernel above from the dense (i, j) -> sparse-index map
static inline std::vector<unsigned short> grid58_offsets(const std::vector<short> &mmap) {
    using G = Grid58;
    std::vector<unsigned short> off((size_t)G::IHN * G::JHN * 64);
    for (int ih = 0; ih < G::IHN; ++ih)
        for (int jh = 0; jh < G::JHN; ++jh)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = 4 * ih + lane / 16, pos = 16 * jh + lane % 16;
                int word = G::IMG_ZERO;
                if (i < G::NV) {
                    if (pos == G::RHS_POS) word = G::IMG_B + i;
                    else if (pos >= G::COL0 && pos < G::COL0 + G::NV) {
                        const int id = mmap[(size_t)i * G::NV + (pos - G::COL0)];
                        if (id >= 0) word = id;
                    }
                }
                off[((size_t)ih * G::JHN + jh) * 64 + lane] = (unsigned short)(word * 8);
            }
    return off;
}
