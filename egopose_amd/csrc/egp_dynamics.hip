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

#include "egp_dynamics_dev.hpp"

namespace {

using namespace egp_dyn;

// One wavefront per env and the env's intermediates in LDS. Round 6's counters (65 536 envs): SQ_LDS_IDX_ACTIVE = 86 % of the
// kernel's CU-cycles, a third of them bank-conflict cycles -- the kernel is LDS-bound (496 LDS instructions per env), so the rows of
// the per-env arrays got odd strides (egp_dynamics_dev.hpp) at the price of 11.0 instead of 9.8 kB per env: 6 envs per workgroup
// (6 x 11.0 kB + the 9.9 kB of tree tables = 75.8 kB, two workgroups per CU = 12 envs; rounds 3-5: 7 x 9.8 + 9.9 = 78.3 kB, 14 envs).
// Small batches (a rollout group) keep 4 envs per workgroup: spread over more CUs, one wave per SIMD (1 024 envs: 22 us against 27).

// `list` (optional): the launch covers the envs list[k * list_stride] instead of 0 .. n - 1 (the engine's reset: rows of the envs
// being reset, = the reference's sim.forward() after set_state); an entry whose list[k * list_stride + 2] is non-zero writes to the
// `_alt` outputs (the per-substep engine form keeps two generations of (qM, bias) rows, see egp_engine.hip).
template <int DYN_ENVS_PER_BLOCK>
__global__ __launch_bounds__(DYN_ENVS_PER_BLOCK * 64) void k_dynamics(const DynTables *__restrict__ tab_g, const double *__restrict__ qpos,
                                                  const double *__restrict__ qvel, int n, long ld_q, long ld_v, double *__restrict__ qM,
                                                  long ld_m, double *__restrict__ bias, long ld_b, double *__restrict__ xpos, int env_doubles,
                                                  const int *__restrict__ list, int list_stride, double *__restrict__ qM_alt,
                                                  double *__restrict__ bias_alt) {
    __shared__ DynTables tb;
    extern __shared__ double s_env[];            // (blockDim / 64) x env_doubles
    {
        const int words = sizeof(DynTables) / 4;
        const int *src = reinterpret_cast<const int *>(tab_g);
        int *dst = reinterpret_cast<int *>(&tb);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long k = (long)blockIdx.x * DYN_ENVS_PER_BLOCK + wave;
    const bool valid = k < n;
    const long env = !valid ? 0 : (list ? (long)list[k * list_stride] : k);    // out-of-range waves shadow env 0 and write nothing
    if (valid && list && list[k * list_stride + 2] != 0) { qM = qM_alt; bias = bias_alt; }
    __syncthreads();
    dynamics_wave(tb, s_env + wave * env_doubles, qpos + env * ld_q, qvel + env * ld_v, lane, valid, qM ? qM + env * ld_m : nullptr,
                  bias ? bias + env * ld_b : nullptr, xpos ? xpos + env * tb.nb * 3 : nullptr);
}

}  // namespace

// engine entry: state rows / outputs with arbitrary row strides (doubles); `list` / `_alt`: see k_dynamics
int egp_launch_dynamics_strided(egp_ctx *ctx, const double *qpos, long ld_q, const double *qvel, long ld_v, int32_t n, double *qM,
                                long ld_m, double *bias, long ld_b, double *xpos, hipStream_t stream, const int *list, int list_stride,
                                double *qM_alt, double *bias_alt) {
    if (!ctx->dyn_tables) { egp::set_error("egp_set_dynamics_model must be called before egp_dynamics"); return EGP_E_STATE; }
    const int env_doubles = dy_env_doubles(ctx->dm.nbody, ctx->dm.nv - 6, ctx->dm.nv);
    static const hipError_t attr = hipFuncSetAttribute((const void *)k_dynamics<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const bool wide = n >= 4096;
    const size_t lds = (size_t)(wide ? 6 : 4) * env_doubles * sizeof(double);
    if (attr != hipSuccess || lds > (wide ? 100 : 64) * 1024) { egp::set_error("k_dynamics: LDS budget (%zu bytes)", lds); return EGP_E_HIP; }
    const DynTables *tab = (const DynTables *)ctx->dyn_tables;
    if (wide)
        k_dynamics<6><<<dim3((n + 5) / 6), dim3(384), lds, stream>>>(tab, qpos, qvel, n, ld_q, ld_v, qM, ld_m, bias, ld_b, xpos, env_doubles,
                                                                     list, list_stride, qM_alt, bias_alt);
    else
        k_dynamics<4><<<dim3((n + 3) / 4), dim3(256), lds, stream>>>(tab, qpos, qvel, n, ld_q, ld_v, qM, ld_m, bias, ld_b, xpos, env_doubles,
                                                                     list, list_stride, qM_alt, bias_alt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { egp::set_error("k_dynamics launch failed: %s", hipGetErrorString(e)); return EGP_E_HIP; }
    return EGP_OK;
}

extern "C" {

#ifdef EGP_DYN_TRACE
int egp_dyn_trace_read(long long *out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dyn_trace), sizeof(long long) * 16) == hipSuccess ? 0 : -1; }
#endif

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
        t.last_dof[b] = last_dof[b];
    }
    // subtrees must be contiguous (MJCF depth-first order): body c belongs to b's subtree iff b is on c's ancestor path
    for (int b = 0; b < t.nb; ++b) {
        int end = b + 1;
        for (int c = b + 1; c < t.nb; ++c) {
            bool under = false;
            for (int a = t.parent[c]; a >= 0; a = t.parent[a]) under = under || a == b;
            if (under) { EGP_REQUIRE(c == end, "bodies must be listed in depth-first order"); end = c + 1; }
        }
        t.subtree_end[b] = end;
    }
    EGP_REQUIRE(j == d->njoint, "body_ndof does not sum to njoint");
    {   // ancestor jump tables of the dof tree (phase D's prefix sums) and the rounds the longest chain needs
        int longest = 1;
        for (int i = 0; i < t.nv; ++i) {
            t.dof_anc[0][i] = (signed char)t.dof_parent[i];
            int len = 0;
            for (int k = i; k >= 0; k = t.dof_parent[k]) ++len;
            if (len > longest) longest = len;
        }
        for (int k = 1; k < 6; ++k)
            for (int i = 0; i < t.nv; ++i) {
                const int h = t.dof_anc[k - 1][i];
                t.dof_anc[k][i] = h >= 0 ? t.dof_anc[k - 1][h] : (signed char)-1;
            }
        t.scan_rounds = 0;
        while ((1 << t.scan_rounds) < longest) ++t.scan_rounds;
        EGP_REQUIRE(t.scan_rounds <= 6, "dof chain longer than 64");
    }
    int adr = 0;
    for (int i = 0; i < t.nv; ++i) {
        t.dof_madr[i] = adr;
        for (int k = i; k >= 0; k = t.dof_parent[k]) ++adr;
    }
    EGP_REQUIRE(adr == ctx->dm.nM, "sparse inertia size differs from the context's model");
    EGP_REQUIRE(adr <= 1024, "sparse inertia row longer than 1024 entries");
    t.nM = adr;
    for (int i = 0; i < t.nv; ++i) {                 // entry table of the row: dof i, then its ancestors (MuJoCo's legacy order)
        int e = t.dof_madr[i];
        for (int k = i; k >= 0; k = t.dof_parent[k], ++e) { t.ent_row[e] = (unsigned char)i; t.ent_col[e] = (unsigned char)k; }
    }
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
    return egp_launch_dynamics_strided(ctx, qpos, ctx->dm.nq, qvel, ctx->dm.nv, n, qM, ld_m, qfrc_bias, ctx->dm.nv, xpos,
                                       (hipStream_t)stream);
}

}  // extern "C"
