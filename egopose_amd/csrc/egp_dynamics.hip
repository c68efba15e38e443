// K8: rigid-body dynamics terms of the humanoid on the GPU: from (qpos, qvel) per env
//   * forward kinematics of the MJCF tree           -> body frame positions  (mjData.xpos)
//   * composite-rigid-body algorithm                -> joint-space inertia in MuJoCo's legacy sparse order (mjData.qM)
//   * recursive Newton-Euler at zero acceleration   -> Coriolis + centrifugal + gravity force (mjData.qfrc_bias)
// i.e. the three mjData fields the stable-PD controller of the reference consumes
// (/root/reference/ego_pose/envs/humanoid_v1.py:130-144: mj_fullM(model, M, data.qM), data.qfrc_bias; :98-111: data.body_xpos),
// so a host physics backend only has to hand over qpos / qvel (117 doubles per env-substep instead of 1085).
//
// Mapping: one 64-lane wavefront per env, 4 envs per 256-thread workgroup. The skeleton tree (parents, levels, child
// lists, joint axes / anchors, masses, inertias: 5 kB) is staged in LDS once per workgroup; per-env intermediates
// (body frames, joint motion vectors, spatial inertias, velocities, forces: 11 kB) live in LDS. Lane = body in the two
// tree sweeps (root -> leaves: frames, motion vectors, velocities, bias accelerations, body forces; leaves -> root:
// composite inertias and accumulated forces), lane = dof in the final pass (one column of M walked up its ancestor
// chain, one bias entry). All spatial vectors are expressed in world coordinates about the world origin, so parent /
// child transforms are identities and a sweep level is a handful of cross products.
// Conventions are MuJoCo's: qvel[0:3] root linear velocity (world), qvel[3:6] root angular velocity in the BODY frame,
// MJCF coordinate="global" (all body frames axis aligned at the zero pose), armature on the hinge diagonal.
// Parity: MuJoCo is not available to pin against; tests/test_dynamics_gpu.py checks the kernel against
// oracle/dynamics.py (Jacobian-sum M, finite-difference Newton-Euler bias, kinetic-energy identity).
#include <hip/hip_runtime.h>

#include <math.h>
#include <string.h>

#include <vector>

#include "egp_internal.hpp"

namespace {

constexpr int DY_MAXB = 24;      // bodies
constexpr int DY_MAXJ = 64;      // hinges
constexpr int DY_MAXC = 4;       // children per body
constexpr int DY_MAXV = 64;      // dofs

struct DynTables {               // device copy of the tree, laid out as the kernel stages it
    int nb, nj, nv, max_level;
    int parent[DY_MAXB], level[DY_MAXB], nchild[DY_MAXB], child[DY_MAXB][DY_MAXC], first_j[DY_MAXB], ndof[DY_MAXB];
    int dof_parent[DY_MAXV], dof_madr[DY_MAXV], dof_body[DY_MAXV];
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
__device__ __forceinline__ M3 axis_angle(V3 a, double ang) {
    const double s = sin(ang), c1 = 1.0 - cos(ang);
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

constexpr int DY_ENV_DOUBLES = DY_MAXB * 12 + DY_MAXV * 6 + DY_MAXB * 10 * 2 + DY_MAXB * 6 * 3;

__global__ __launch_bounds__(256) void k_dynamics(const DynTables *__restrict__ tab_g, const double *__restrict__ qpos,
                                                  const double *__restrict__ qvel, int n, int nq, double *__restrict__ qM, long ld_m,
                                                  double *__restrict__ bias, double *__restrict__ xpos) {
    __shared__ DynTables tb;
    extern __shared__ double s_env[];            // 4 x DY_ENV_DOUBLES
    {
        const int words = sizeof(DynTables) / 4;
        const int *src = reinterpret_cast<const int *>(tab_g);
        int *dst = reinterpret_cast<int *>(&tb);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    const bool valid = env < n;
    const long e = valid ? env : 0;              // out-of-range waves shadow env 0 and write nothing
    double *base = s_env + wave * DY_ENV_DOUBLES;
    double *sR = base;                                   // [nb][9]
    double *sP = sR + DY_MAXB * 9;                       // [nb][3]
    double *sS = sP + DY_MAXB * 3;                       // [nv][6]   joint motion vectors
    double *sIb = sS + DY_MAXV * 6;                      // [nb][10]  own spatial inertia
    double *sIc = sIb + DY_MAXB * 10;                    // [nb][10]  composite
    double *sV = sIc + DY_MAXB * 10;                     // [nb][6]   spatial velocity
    double *sA = sV + DY_MAXB * 6;                       // [nb][6]   bias acceleration (gravity as base acceleration)
    double *sF = sA + DY_MAXB * 6;                       // [nb][6]   body force -> subtree force
    const double *q = qpos + e * nq;
    const double *qd = qvel + e * (nq - 1);
    __syncthreads();
    const int nb = tb.nb, nv = tb.nv;
    const int b = lane;
    // ---- sweep root -> leaves
    for (int L = 0; L <= tb.max_level; ++L) {
        if (b < nb && tb.level[b] == L) {
            M3 R;
            V3 p;
            Sp v_run, a_run;
            if (b == 0) {
                double qw = q[3], qx = q[4], qy = q[5], qz = q[6];
                const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
                qw *= inv; qx *= inv; qy *= inv; qz *= inv;
                R.m[0] = 1 - 2 * (qy * qy + qz * qz); R.m[1] = 2 * (qx * qy - qz * qw);     R.m[2] = 2 * (qx * qz + qy * qw);
                R.m[3] = 2 * (qx * qy + qz * qw);     R.m[4] = 1 - 2 * (qx * qx + qz * qz); R.m[5] = 2 * (qy * qz - qx * qw);
                R.m[6] = 2 * (qx * qz - qy * qw);     R.m[7] = 2 * (qy * qz + qx * qw);     R.m[8] = 1 - 2 * (qx * qx + qy * qy);
                p = {q[0], q[1], q[2]};
                // free joint: three translations along the world axes, three rotations about the body axes through p
                Sp v = {{0, 0, 0}, {qd[0], qd[1], qd[2]}};
                Sp srot[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    Sp st = {{0, 0, 0}, {d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0}};
                    st_sp(sS + d * 6, st);
                    const V3 a = {R.m[d], R.m[3 + d], R.m[6 + d]};         // column d of R
                    srot[d] = {a, cross(p, a)};
                    st_sp(sS + (3 + d) * 6, srot[d]);
                    v = v + qd[3 + d] * srot[d];
                }
                // the rotational axes are fixed in the root itself: dS/dt = v_root x S
                Sp a = {{0, 0, 0}, {-tb.g[0], -tb.g[1], -tb.g[2]}};
#pragma unroll
                for (int d = 0; d < 3; ++d) a = a + qd[3 + d] * cross_m(v, srot[d]);
                v_run = v;
                a_run = a;
            } else {
                const int par = tb.parent[b];
#pragma unroll
                for (int i = 0; i < 9; ++i) R.m[i] = sR[par * 9 + i];
                const V3 pp = {sP[par * 3], sP[par * 3 + 1], sP[par * 3 + 2]};
                p = pp + mul(R, V3{tb.off[b][0], tb.off[b][1], tb.off[b][2]});
                v_run = ld_sp(sV + par * 6);
                a_run = ld_sp(sA + par * 6);
                const int j0 = tb.first_j[b];
                for (int k = 0; k < tb.ndof[b]; ++k) {
                    const int j = j0 + k;
                    const V3 a_loc = {tb.axis[j][0], tb.axis[j][1], tb.axis[j][2]};
                    const V3 anc_loc = {tb.anc[j][0], tb.anc[j][1], tb.anc[j][2]};
                    const V3 a_w = mul(R, a_loc);
                    const V3 r_w = p + mul(R, anc_loc);
                    const Sp S = {a_w, cross(r_w, a_w)};
                    st_sp(sS + (6 + j) * 6, S);
                    const double qdj = qd[6 + j];
                    a_run = a_run + qdj * cross_m(v_run, S);       // the axis is fixed in the frame before this hinge
                    v_run = v_run + qdj * S;
                    R = mul(R, axis_angle(a_loc, q[7 + j]));
                    p = r_w - mul(R, anc_loc);
                }
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) sR[b * 9 + i] = R.m[i];
            sP[b * 3] = p.x; sP[b * 3 + 1] = p.y; sP[b * 3 + 2] = p.z;
            st_sp(sV + b * 6, v_run);
            st_sp(sA + b * 6, a_run);
            // own spatial inertia about the world origin
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
            for (int i = 0; i < 10; ++i) { sIb[b * 10 + i] = in10[i]; sIc[b * 10 + i] = in10[i]; }
            // body force at zero joint acceleration: I a + v x* (I v)
            const Sp f = inertia_apply(in10, a_run) + cross_f(v_run, inertia_apply(in10, v_run));
            st_sp(sF + b * 6, f);
            if (valid && xpos) { double *xp = xpos + (env * nb + b) * 3; xp[0] = p.x; xp[1] = p.y; xp[2] = p.z; }
        }
        __syncthreads();
    }
    // ---- sweep leaves -> root: composite inertias and subtree forces (a parent gathers its children)
    for (int L = tb.max_level - 1; L >= 0; --L) {
        if (b < nb && tb.level[b] == L) {
            for (int k = 0; k < tb.nchild[b]; ++k) {
                const int ch = tb.child[b][k];
#pragma unroll
                for (int i = 0; i < 10; ++i) sIc[b * 10 + i] += sIc[ch * 10 + i];
#pragma unroll
                for (int i = 0; i < 6; ++i) sF[b * 6 + i] += sF[ch * 6 + i];
            }
        }
        __syncthreads();
    }
    // ---- lane = dof: column of M up the ancestor chain (MuJoCo's legacy sparse order) and the bias entry
    if (valid && lane < nv) {
        const int d = lane, bd = tb.dof_body[d];
        const Sp S = ld_sp(sS + d * 6);
        if (qM) {
            const Sp F = inertia_apply(sIc + bd * 10, S);
            double *out = qM + env * ld_m + tb.dof_madr[d];
            int i = d, k = 0;
            while (i >= 0) {
                double v = sdot(ld_sp(sS + i * 6), F);
                if (i == d && d >= 6) v += tb.armature;
                out[k++] = v;
                i = tb.dof_parent[i];
            }
        }
        if (bias) bias[env * nv + d] = sdot(S, ld_sp(sF + bd * 6));
    }
}

}  // namespace

extern "C" {

int egp_set_dynamics_model(egp_ctx *ctx, const egp_dynamics_desc *d) {
    EGP_REQUIRE(ctx && d, "NULL pointer");
    EGP_REQUIRE(d->nbody >= 1 && d->nbody <= DY_MAXB && d->njoint >= 0 && d->njoint <= DY_MAXJ - 6, "tree too large");
    EGP_REQUIRE(d->body_parent && d->body_pos && d->body_com && d->body_inertia && d->body_mass && d->body_ndof && d->joint_axis && d->joint_anchor,
                "NULL table");
    EGP_REQUIRE(6 + d->njoint == ctx->dm.nv && d->nbody == ctx->dm.nbody, "dynamics tree does not match the context's model");
    DynTables t;
    memset(&t, 0, sizeof(t));
    t.nb = d->nbody; t.nj = d->njoint; t.nv = 6 + d->njoint;
    t.armature = d->armature;
    for (int i = 0; i < 3; ++i) t.g[i] = d->gravity[i];
    int j = 0;
    std::vector<int> last_dof(t.nb, -1);
    for (int dd = 0; dd < 6; ++dd) { t.dof_parent[dd] = dd - 1; t.dof_body[dd] = 0; }
    last_dof[0] = 5;
    for (int b = 0; b < t.nb; ++b) {
        const int par = d->body_parent[b];
        EGP_REQUIRE(b == 0 ? par < 0 : (par >= 0 && par < b), "bodies must be listed parents first, body 0 = root");
        t.parent[b] = par;
        t.level[b] = b == 0 ? 0 : t.level[par] + 1;
        if (t.level[b] > t.max_level) t.max_level = t.level[b];
        if (b > 0) {
            EGP_REQUIRE(t.nchild[par] < DY_MAXC, "more than 4 children on one body");
            t.child[par][t.nchild[par]++] = b;
        }
        const int nd = b == 0 ? 0 : d->body_ndof[b];
        t.first_j[b] = j;
        t.ndof[b] = nd;
        for (int c = 0; c < 3; ++c) {
            t.off[b][c] = b == 0 ? 0.0 : d->body_pos[b * 3 + c] - d->body_pos[par * 3 + c];
            t.com_l[b][c] = d->body_com[b * 3 + c] - d->body_pos[b * 3 + c];
        }
        const double *I = d->body_inertia + (size_t)b * 9;
        t.I_l[b][0] = I[0]; t.I_l[b][1] = I[4]; t.I_l[b][2] = I[8]; t.I_l[b][3] = I[1]; t.I_l[b][4] = I[2]; t.I_l[b][5] = I[5];
        t.mass[b] = d->body_mass[b];
        for (int k = 0; k < nd; ++k, ++j) {
            EGP_REQUIRE(j < d->njoint, "body_ndof sums past njoint");
            for (int c = 0; c < 3; ++c) {
                t.axis[j][c] = d->joint_axis[j * 3 + c];
                t.anc[j][c] = d->joint_anchor[j * 3 + c] - d->body_pos[b * 3 + c];
            }
            t.dof_parent[6 + j] = k == 0 ? last_dof[par] : 6 + j - 1;
            t.dof_body[6 + j] = b;
        }
        last_dof[b] = nd > 0 ? 6 + j - 1 : (b == 0 ? 5 : last_dof[par]);
    }
    EGP_REQUIRE(j == d->njoint, "body_ndof does not sum to njoint");
    int adr = 0;
    for (int i = 0; i < t.nv; ++i) {
        t.dof_madr[i] = adr;
        for (int k = i; k >= 0; k = t.dof_parent[k]) ++adr;
    }
    EGP_REQUIRE(adr == ctx->dm.nM, "sparse inertia size differs from the context's model");
    EGP_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->dyn_tables) {
        void *p = nullptr;
        EGP_HIP_CHECK(hipMalloc(&p, sizeof(DynTables)));
        ctx->allocs.push_back(p);
        ctx->dyn_tables = p;
    }
    EGP_HIP_CHECK(hipMemcpy(ctx->dyn_tables, &t, sizeof(DynTables), hipMemcpyHostToDevice));
    return EGP_OK;
}

int egp_dynamics_f64(egp_ctx *ctx, const double *qpos, const double *qvel, int32_t n, double *qM, int64_t ld_m, double *qfrc_bias,
                     double *xpos, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    if (!ctx->dyn_tables) { egp::set_error("egp_set_dynamics_model must be called before egp_dynamics"); return EGP_E_STATE; }
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(qpos && qvel, "NULL pointer");
    EGP_REQUIRE(!qM || ld_m >= ctx->dm.nM, "ld_m smaller than nM");
    const size_t lds = (size_t)4 * DY_ENV_DOUBLES * sizeof(double);
    k_dynamics<<<dim3((n + 3) / 4), dim3(256), lds, (hipStream_t)stream>>>((const DynTables *)ctx->dyn_tables, qpos, qvel, n, ctx->dm.nq, qM, (long)ld_m, qfrc_bias, xpos);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("k_dynamics launch failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}

}  // extern "C"
