// Device side of K8 (see egp_dynamics.hip): the per-env dynamics pass as a wave-level function, shared by the stand-alone
// kernel k_dynamics and the resident stable-PD kernel (egp_kernels.hip: k_pd_server_tree58<true>), which runs it on the
// state rows it has just read instead of taking qM / qfrc_bias from the host.
#pragma once
#include <hip/hip_runtime.h>

// EGP_DYN_TRACE (tools/probes/dyn_trace.py): wall_clock64 stamps (100 MHz) of lane 0 of the first wave of workgroup 0 after each phase
#ifdef EGP_DYN_TRACE
__device__ long long g_dyn_trace[16];
#define DY_TR(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dyn_trace[i] = wall_clock64(); } while (0)
#else
#define DY_TR(i) do { } while (0)
#endif

namespace egp_dyn {

constexpr int DY_MAXB = 24;      // bodies
constexpr int DY_MAXJ = 64;      // hinges
constexpr int DY_MAXC = 4;       // children per body
constexpr int DY_MAXV = 64;      // dofs

struct DynTables {               // device copy of the tree, laid out as the kernel stages it
    int nb, nj, nv, max_level;
    int parent[DY_MAXB], level[DY_MAXB], nchild[DY_MAXB], child[DY_MAXB][DY_MAXC], first_j[DY_MAXB], ndof[DY_MAXB];
    int dof_parent[DY_MAXV], dof_madr[DY_MAXV], dof_body[DY_MAXV];
    int last_dof[DY_MAXB];       // deepest dof of the body's chain (its own last hinge, or the nearest ancestor's)
    signed char dof_anc[6][DY_MAXV];   // dof_anc[k][d]: the 2^k-th ancestor of dof d in the dof tree (-1: none) -- pointer jumping
    int scan_rounds;             // ceil(log2(longest dof chain))
    int nM;                      // entries of the sparse inertia row
    unsigned char ent_row[1024], ent_col[1024];   // entry e of the row = M[ent_row[e]][ent_col[e]] (MuJoCo's order: dof d, then its ancestors)
    int subtree_end[DY_MAXB];    // bodies [b, subtree_end[b]) form b's subtree (depth-first body order)
    double off[DY_MAXB][3];      // body_pos - body_pos[parent]   (zero pose, global)
    double com_l[DY_MAXB][3];    // body_com - body_pos
    double I_l[DY_MAXB][6];      // inertia about the COM, axes of the zero pose: xx, yy, zz, xy, xz, yz
    double mass[DY_MAXB];
    double axis[DY_MAXJ][3], anc[DY_MAXJ][3];   // hinge axis, anchor - body_pos[body]
    double armature, g[3];
};

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct M3 { double m[9]; };      // row major
__device__ __forceinline__ V3 mul(const M3 &R, V3 v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
// Rodrigues: I + sin K + (1 - cos) K^2
__device__ __forceinline__ M3 axis_angle(V3 a, double s, double c1) {
    M3 R;
    R.m[0] = 1.0 + c1 * (-a.y * a.y - a.z * a.z); R.m[1] = -s * a.z + c1 * a.x * a.y;          R.m[2] = s * a.y + c1 * a.x * a.z;
    R.m[3] = s * a.z + c1 * a.x * a.y;           R.m[4] = 1.0 + c1 * (-a.x * a.x - a.z * a.z); R.m[5] = -s * a.x + c1 * a.y * a.z;
    R.m[6] = -s * a.y + c1 * a.x * a.z;          R.m[7] = s * a.x + c1 * a.y * a.z;           R.m[8] = 1.0 + c1 * (-a.x * a.x - a.y * a.y);
    return R;
}

// spatial vectors: [angular; linear-at-origin] motion, [moment-about-origin; force] force
struct Sp { V3 w, v; };
__device__ __forceinline__ Sp operator+(Sp a, Sp b) { return {a.w + b.w, a.v + b.v}; }
__device__ __forceinline__ Sp operator*(double s, Sp a) { return {s * a.w, s * a.v}; }
__device__ __forceinline__ Sp cross_m(Sp a, Sp b) { return {cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)}; }
__device__ __forceinline__ Sp cross_f(Sp a, Sp f) { return {cross(a.w, f.w) + cross(a.v, f.v), cross(a.w, f.v)}; }
__device__ __forceinline__ double sdot(Sp a, Sp f) { return dot(a.w, f.w) + dot(a.v, f.v); }
// spatial inertia about the world origin: mass, first moment h = m c, rotational inertia I (xx, yy, zz, xy, xz, yz)
__device__ __forceinline__ Sp inertia_apply(const double *in10, Sp s) {
    const double m = in10[0];
    const V3 h = {in10[1], in10[2], in10[3]};
    const double *I = in10 + 4;
    const V3 Iw = {I[0] * s.w.x + I[3] * s.w.y + I[4] * s.w.z, I[3] * s.w.x + I[1] * s.w.y + I[5] * s.w.z, I[4] * s.w.x + I[5] * s.w.y + I[2] * s.w.z};
    return {Iw + cross(h, s.v), m * s.v + cross(s.w, h)};
}

__device__ __forceinline__ void st_sp(double *p, Sp s) { p[0] = s.w.x; p[1] = s.w.y; p[2] = s.w.z; p[3] = s.v.x; p[4] = s.v.y; p[5] = s.v.z; }
__device__ __forceinline__ Sp ld_sp(const double *p) { return {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}}; }

// per-env LDS (doubles), sized by the model (nb bodies, nj hinges, nv dofs):
//   R1  [nb][12] local transforms + [nj][6] local hinge axis / anchor (phases A-C), reused from phase D on as
//       [nb][10] own spatial inertias + [nb][6] body forces  (16 nb <= 12 nb + 6 nj whenever nj >= nb: every body but the root has a hinge)
//   sW  [nb][12] world frames,  sS [nv][6] joint motion vectors,  sQ [nv] the env's qvel
// (round 3: 16.9 kB -> 9.8 kB per env for the humanoid -- the number of envs a CU holds at once is what bounds this kernel at
//  large batches -- by the aliasing, by sizing to the model instead of the table maxima, and by dropping the S_d * qvel_d
//  copy: phase D forms the product. Keeping the copy (12.1 kB, 12 envs per CU) measured 674 us at 65 536 envs against 653.)
// Row strides of the per-env LDS arrays, in doubles: ODD, so that the lanes of a half-wave (lane = body / hinge / dof reading its own
// row) fall into different banks. With the natural strides 12 / 6 / 10 (96 / 48 / 80 bytes against 256 bytes of banks) every 8th /
// 16th / 16th row shares its banks: at 65 536 envs the LDS was busy 86 % of the kernel's time (SQ_LDS_IDX_ACTIVE), a third of that
// in bank-conflict cycles (SQ_LDS_BANK_CONFLICT) -- K8 is LDS-bound, not latency-bound as rounds 3-5 assumed.
constexpr int DY_LW = 13;        // frames: R (9), p (3)  [sLoc, sW]
constexpr int DY_LJ = 7;         // hinge axis (3), anchor (3)  [sJl]
constexpr int DY_LS = 7;         // spatial vectors  [sS, sX, sF]
constexpr int DY_LI = 11;        // spatial inertias (10)  [sIb]
__host__ __device__ inline int dy_env_doubles(int nb, int nj, int nv) {
    const int r1a = nb * DY_LW + nj * DY_LJ, r1b = nb * (DY_LI + DY_LS), r1c = nv * DY_LS;
    const int r1 = r1a > r1b ? (r1a > r1c ? r1a : r1c) : (r1b > r1c ? r1b : r1c);
    return r1 + nb * DY_LW + nv * DY_LS + nv;
}
constexpr int DY_ENV_DOUBLES = DY_MAXB * DY_LW + DY_MAXJ * DY_LJ + DY_MAXB * DY_LW + DY_MAXV * DY_LS + DY_MAXV;      // the same for the table maxima

// Every env is owned by ONE wavefront, so the phases only need the wave's own LDS writes to have landed before its
// next reads: LDS operations of a wave complete in order; the fences keep the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void ld_m3(const double *p, M3 &R) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R.m[i] = p[i];
}
__device__ __forceinline__ void st_m3(double *p, const M3 &R) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = R.m[i];
}
__device__ __forceinline__ V3 ld_v3(const double *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st_v3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// No level-by-level sweep: every lane walks its own (short) ancestor chain, so the tree costs four wave-local
// synchronisations instead of two per depth level, and no lane waits for lanes of other levels.
//   A  lane = body : chain of the body's own hinges in the PARENT frame -> local transform (Rl, tl) and the hinges' local
//                    axis / anchor (parent-frame coordinates)
//   B  lane = body : world frame = root frame o local transforms along the ancestor path (walked upwards)
//   C  lane = dof  : joint motion vector S_d in world coordinates, and S_d * qvel_d
//   D  lane = body : spatial velocity and bias acceleration from the dof chain (walked upwards with suffix sums), own
//                    spatial inertia, body force
//   E  lane = body : composite inertia / subtree force = sum over the body's contiguous (depth-first) subtree range
//   F  lane = dof  : column of M up the ancestor chain, bias entry
// `tb`: the tree tables (LDS copy); `base`: DY_ENV_DOUBLES doubles of wave-private LDS; q / qd: the env's qpos / qvel (any
// memory the wave can read); outputs may be NULL: qM_out = the env's sparse inertia row (MuJoCo order), bias_out [nv],
// xpos_out [nb][3]. `valid` = false makes the wave compute on its inputs and write nothing.
__device__ __forceinline__ void dynamics_wave(const DynTables &tb, double *base, const double *q, const double *qd, int lane, bool valid,
                                              double *qM_out, double *bias_out, double *xpos_out) {
    const int nb = tb.nb, nv = tb.nv, nj = tb.nj;
    double *sLoc = base;                                 // [nb][12]  local transform in the parent frame: Rl (9), tl (3)       (A, B)
    double *sJl = sLoc + nb * DY_LW;                        // [nj][6]   hinge axis (3) and anchor (3) in the parent frame of its body (A, C)
    double *sIb = base;                                  // [nb][10]  own spatial inertia about the world origin        (D, F: over sLoc / sJl)
    double *sF = sIb + nb * DY_LI;                          // [nb][6]   body force                                        (D, F)
    const int r1 = dy_env_doubles(nb, nj, nv) - (nb * DY_LW + nv * DY_LS + nv);
    double *sW = base + r1;                              // [nb][12]  world frame: R (9), p (3)
    double *sS = sW + nb * DY_LW;                           // [nv][6]   joint motion vectors (world)
    double *sQ = sS + nv * DY_LS;                            // [nv]      qvel
    const int b = lane;
    DY_TR(0);
    if (lane < nv) sQ[lane] = qd[lane];
    // ---- A0: lane = hinge: sin and 1 - cos of every hinge angle at once (a body's up to three hinges used to take their
    //          double-precision sincos one after the other); parked in sS, which phase C fills later
    if (lane < nj) {
        double sn, cs;
        sincos(q[7 + lane], &sn, &cs);
        sS[2 * lane] = sn;
        sS[2 * lane + 1] = 1.0 - cs;
    }
    wave_sync();
    // ---- A: own hinge chain in the parent frame
    if (b < nb) {
        M3 Rl;
        V3 tl;
        if (b == 0) {
            double qw = q[3], qx = q[4], qy = q[5], qz = q[6];
            const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw *= inv; qx *= inv; qy *= inv; qz *= inv;
            Rl.m[0] = 1 - 2 * (qy * qy + qz * qz); Rl.m[1] = 2 * (qx * qy - qz * qw);     Rl.m[2] = 2 * (qx * qz + qy * qw);
            Rl.m[3] = 2 * (qx * qy + qz * qw);     Rl.m[4] = 1 - 2 * (qx * qx + qz * qz); Rl.m[5] = 2 * (qy * qz - qx * qw);
            Rl.m[6] = 2 * (qx * qz - qy * qw);     Rl.m[7] = 2 * (qy * qz + qx * qw);     Rl.m[8] = 1 - 2 * (qx * qx + qy * qy);
            tl = {q[0], q[1], q[2]};
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) Rl.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
            tl = {tb.off[b][0], tb.off[b][1], tb.off[b][2]};
            const int j0 = tb.first_j[b];
            for (int k = 0; k < tb.ndof[b]; ++k) {
                const int j = j0 + k;
                const V3 a_loc = {tb.axis[j][0], tb.axis[j][1], tb.axis[j][2]};
                const V3 anc_loc = {tb.anc[j][0], tb.anc[j][1], tb.anc[j][2]};
                const V3 a_p = mul(Rl, a_loc);                   // axis / anchor before this hinge turns, parent-frame coordinates
                const V3 r_p = tl + mul(Rl, anc_loc);
                st_v3(sJl + j * DY_LJ, a_p);
                st_v3(sJl + j * DY_LJ + 3, r_p);
                Rl = mul(Rl, axis_angle(a_loc, sS[2 * j], sS[2 * j + 1]));
                tl = r_p - mul(Rl, anc_loc);
            }
        }
        st_m3(sLoc + b * DY_LW, Rl);
        st_v3(sLoc + b * DY_LW + 9, tl);
    }
    wave_sync();
    DY_TR(1);
    // ---- B: world frames (compose upwards: T_world = T_root o ... o T_parent o T_b)
    if (b < nb) {
        M3 R;
        ld_m3(sLoc + b * DY_LW, R);
        V3 t = ld_v3(sLoc + b * DY_LW + 9);
        for (int c = tb.parent[b]; c >= 0; c = tb.parent[c]) {
            M3 Rc;
            ld_m3(sLoc + c * DY_LW, Rc);
            t = ld_v3(sLoc + c * DY_LW + 9) + mul(Rc, t);
            R = mul(Rc, R);
        }
        st_m3(sW + b * DY_LW, R);
        st_v3(sW + b * DY_LW + 9, t);
        if (valid && xpos_out) st_v3(xpos_out + b * 3, t);
    }
    wave_sync();
    DY_TR(2);
    // ---- C: joint motion vectors
    if (lane < nv) {
        const int d = lane;
        Sp S;
        if (d < 3) {
            S = {{0, 0, 0}, {d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0}};
        } else if (d < 6) {      // rotation about the root's own axis d-3 through the root origin
            const V3 a = {sW[d - 3], sW[3 + d - 3], sW[6 + d - 3]};
            S = {a, cross(ld_v3(sW + 9), a)};
        } else {
            const int j = d - 6, par = tb.parent[tb.dof_body[d]];
            M3 Rp;
            ld_m3(sW + par * DY_LW, Rp);
            const V3 a_w = mul(Rp, ld_v3(sJl + j * DY_LJ));
            const V3 r_w = ld_v3(sW + par * DY_LW + 9) + mul(Rp, ld_v3(sJl + j * DY_LJ + 3));
            S = {a_w, cross(r_w, a_w)};
        }
        st_sp(sS + d * DY_LS, S);
    }
    wave_sync();
    DY_TR(3);
    // ---- D: velocity, bias acceleration, own inertia, body force.
    // v_b = sum of S_d qd_d over the body's dof chain; a_b = a0 + sum over pairs u above l of the chain of (S_u qd_u) x (S_l qd_l),
    // except pairs inside the root's three rotational dofs (their axes ride on the root itself: d/dt S = v_root x S, whose rot-rot
    // part cancels). Round 3 walked the chain per body (up to 28 dependent steps of LDS loads and cross products: 7 of the pass's
    // 18 us). Both are sums over the ancestors of a dof: with P(d) = sum of S q over d and its ancestors,
    //     a_b - a0 = sum over the chain's l of P'(parent l) x (S_l qd_l),    P' = P, but P(dof 2) for the root's rotational dofs,
    // so two ancestor-prefix sums by pointer jumping (lane = dof, ceil(log2 28) = 5 rounds each) replace the walks.
    double *sX = base;                                   // [nv][6] scan buffer: over sLoc / sJl (dead after C), under sIb / sF (written at the end of D)
    Sp sq = {{0, 0, 0}, {0, 0, 0}};
    if (lane < nv) sq = sQ[lane] * ld_sp(sS + lane * DY_LS);
    auto ancestor_sums = [&](Sp x) -> Sp {              // inclusive sum of x over the lane's dof and its ancestors; leaves the sums in sX
        if (lane < nv) st_sp(sX + lane * DY_LS, x);
        wave_sync();
        for (int k = 0; k < tb.scan_rounds; ++k) {
            const int an = lane < nv ? tb.dof_anc[k][lane] : -1;
            Sp up = {{0, 0, 0}, {0, 0, 0}};
            if (an >= 0) up = ld_sp(sX + an * DY_LS);
            wave_sync();                                 // every lane has read round k's values
            x = x + up;
            if (an >= 0) st_sp(sX + lane * DY_LS, x);
            wave_sync();
        }
        return x;
    };
    ancestor_sums(sq);
    Sp v = {{0, 0, 0}, {0, 0, 0}};
    if (b < nb) v = ld_sp(sX + tb.last_dof[b] * DY_LS);
    Sp cterm = {{0, 0, 0}, {0, 0, 0}};
    if (lane < nv) {
        const int par = (lane >= 3 && lane < 6) ? 2 : tb.dof_parent[lane];
        if (par >= 0) cterm = cross_m(ld_sp(sX + par * DY_LS), sq);
    }
    wave_sync();                                         // P has been read; the buffer is reused
    ancestor_sums(cterm);
    Sp a = {{0, 0, 0}, {-tb.g[0], -tb.g[1], -tb.g[2]}};
    if (b < nb) a = a + ld_sp(sX + tb.last_dof[b] * DY_LS);
    wave_sync();                                         // ... and again, by sIb / sF below
    if (b < nb) {
        M3 R;
        ld_m3(sW + b * DY_LW, R);
        const V3 p = ld_v3(sW + b * DY_LW + 9);
        const V3 c = p + mul(R, V3{tb.com_l[b][0], tb.com_l[b][1], tb.com_l[b][2]});
        const double *Il = tb.I_l[b];
        M3 I0;
        I0.m[0] = Il[0]; I0.m[1] = Il[3]; I0.m[2] = Il[4];
        I0.m[3] = Il[3]; I0.m[4] = Il[1]; I0.m[5] = Il[5];
        I0.m[6] = Il[4]; I0.m[7] = Il[5]; I0.m[8] = Il[2];
        M3 Rt;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) Rt.m[3 * i + jj] = R.m[3 * jj + i];
        const M3 Iw = mul(mul(R, I0), Rt);
        const double m = tb.mass[b], cc = dot(c, c);
        double in10[10];
        in10[0] = m; in10[1] = m * c.x; in10[2] = m * c.y; in10[3] = m * c.z;
        in10[4] = Iw.m[0] + m * (cc - c.x * c.x); in10[5] = Iw.m[4] + m * (cc - c.y * c.y); in10[6] = Iw.m[8] + m * (cc - c.z * c.z);
        in10[7] = Iw.m[1] - m * c.x * c.y;        in10[8] = Iw.m[2] - m * c.x * c.z;        in10[9] = Iw.m[5] - m * c.y * c.z;
#pragma unroll
        for (int i = 0; i < 10; ++i) sIb[b * DY_LI + i] = in10[i];
        st_sp(sF + b * DY_LS, inertia_apply(in10, a) + cross_f(v, inertia_apply(in10, v)));
    }
    wave_sync();
    DY_TR(4);
    // ---- E: composite inertia / subtree force of every body, in place. The bodies are listed depth first, so a subtree is the
    //         contiguous range [b, subtree_end[b]): a suffix sum over the body index (lane = body, ceil(log2 nb) doubling rounds
    //         through LDS) gives T[b] = sum of bodies >= b, and the subtree's sum is T[b] - T[subtree_end[b]]. (Round 3 let every
    //         dof lane add up its body's range itself -- 21 bodies x 16 doubles for the root's dofs. The subtraction costs a few
    //         ulps of the WHOLE tree's inertia about the world origin, which is what every entry of M is formed from anyway.)
    {
        double t16[16];
        if (b < nb) {
#pragma unroll
            for (int i = 0; i < 10; ++i) t16[i] = sIb[b * DY_LI + i];
#pragma unroll
            for (int i = 0; i < 6; ++i) t16[10 + i] = sF[b * DY_LS + i];
        }
        for (int step = 1; step < nb; step <<= 1) {
            const bool has = b < nb && b + step < nb;
            double up[16];
            if (has) {
#pragma unroll
                for (int i = 0; i < 10; ++i) up[i] = sIb[(b + step) * DY_LI + i];
#pragma unroll
                for (int i = 0; i < 6; ++i) up[10 + i] = sF[(b + step) * DY_LS + i];
            }
            wave_sync();
            if (has) {
#pragma unroll
                for (int i = 0; i < 16; ++i) t16[i] += up[i];
#pragma unroll
                for (int i = 0; i < 10; ++i) sIb[b * DY_LI + i] = t16[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) sF[b * DY_LS + i] = t16[10 + i];
            }
            wave_sync();
        }
        const int end = b < nb ? tb.subtree_end[b] : nb;
        double sub[16];
        if (end < nb) {
#pragma unroll
            for (int i = 0; i < 10; ++i) sub[i] = sIb[end * DY_LI + i];
#pragma unroll
            for (int i = 0; i < 6; ++i) sub[10 + i] = sF[end * DY_LS + i];
        }
        wave_sync();
        if (end < nb) {
#pragma unroll
            for (int i = 0; i < 10; ++i) sIb[b * DY_LI + i] = t16[i] - sub[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) sF[b * DY_LS + i] = t16[10 + i] - sub[10 + i];
        }
        wave_sync();
    }
    DY_TR(6);
    // ---- F: lane = dof: F_d = Ic(body of d) S_d, bias entry; then lane = ENTRY of the sparse row: M[d][i] = S_i . F_d for the 910
    //         (dof, ancestor) pairs, 15 independent products per lane (round 3: every dof lane walked its own chain, up to 28
    //         dependent steps of two LDS round trips each)
    Sp Fd = {{0, 0, 0}, {0, 0, 0}};
    if (lane < nv) {
        const int d = lane, bd = tb.dof_body[d];
        double ic[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) ic[i] = sIb[bd * DY_LI + i];
        const Sp fc = ld_sp(sF + bd * DY_LS);
        const Sp S = ld_sp(sS + d * DY_LS);
        Fd = inertia_apply(ic, S);
        if (valid && bias_out) bias_out[d] = sdot(S, fc);
    }
    wave_sync();                                         // the composites have been read: their place takes F
    if (lane < nv) st_sp(sX + lane * DY_LS, Fd);
    wave_sync();
    if (valid && qM_out) {
#pragma unroll 4
        for (int e = lane; e < tb.nM; e += 64) {
            const int d = tb.ent_row[e], i = tb.ent_col[e];
            double v = sdot(ld_sp(sS + i * DY_LS), ld_sp(sX + d * DY_LS));
            if (i == d && d >= 6) v += tb.armature;
            qM_out[e] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    DY_TR(5);
}

}  // namespace egp_dyn
