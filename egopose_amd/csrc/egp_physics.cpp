// Host physics boundary + the deterministic surrogate backend.
//
// The reference steps MuJoCo through mujoco_py inside HumanoidEnv (envs/common/mujoco_env.py:84-105,
// ego_pose/envs/humanoid_v1.py:158-177): set_state+forward on reset, `data.ctrl[:] = torque; sim.step()`
// per substep, then reads qpos/qvel/qM/qfrc_bias/body_xpos. MuJoCo is an un-vendored dependency and is
// not present in this image, so the boundary is a vtable of C callbacks (egp_physics_vtable) and the
// built-in backend is a surrogate with the same data contract:
//   semi-implicit Euler on  qacc = M0^-1 (tau - C(q, v)),   M0 fixed tree-sparse SPD (zero-pose CRBA),
//   C = viscous joint damping + a vertical root support spring/damper standing in for contacts,
//   free-joint integration as MuJoCo does it (world-frame linear velocity, body-frame angular velocity),
//   forward kinematics of the MJCF tree for body_xpos.
// Like mj_step, `drain` after `step` returns the NEW qpos/qvel together with the inertia/bias that were
// computed at the PREVIOUS state (compute_torque sees one-substep-stale M and C).
// This is NOT MuJoCo: physics parity is unpinned.
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <stdio.h>
#include <string>
#include <vector>

#include "egp_internal.hpp"

struct egp_physics {
    egp_physics_vtable vt{};
    int32_t n_env = 0;
    bool owns_user = false;
};

namespace {

struct Surrogate {
    int nq, nv, nu, nbody, nM, njoint;
    std::vector<double> qM0, Minv0, body_pos, joint_axis, joint_anchor;
    std::vector<int> body_parent, body_ndof;
    double dt, damping, support_k, support_c;
    // per-env state
    std::vector<double> qpos, qvel, bias, zref;
    int n_env;
    // EGP_SURROGATE_SUBSTEP_US: every step() takes at least this long (busy wait), to measure the pipeline at the per-substep
    // cost of a real simulator (an mj_step of this humanoid is tens of microseconds; the surrogate's own step is ~0.3)
    long long min_step_ns = 0;
    std::string name = "surrogate-euler-M0";
};

inline void quat_mul(const double *a, const double *b, double *o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

inline void quat_to_mat(const double *q, double *R) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

inline void mat_vec(const double *R, const double *v, double *o) {
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

inline void mat_mul(const double *A, const double *B, double *O) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

inline void axis_angle_mat(const double *a, double ang, double *R) {
    double s, c;
    sincos(ang, &s, &c);               // (glibc computes both from one argument reduction; same values as sin() / cos())
    const double t = 1 - c;
    R[0] = c + a[0] * a[0] * t;        R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
    R[3] = a[1] * a[0] * t + a[2] * s; R[4] = c + a[1] * a[1] * t;        R[5] = a[1] * a[2] * t - a[0] * s;
    R[6] = a[2] * a[0] * t - a[1] * s; R[7] = a[2] * a[1] * t + a[0] * s; R[8] = c + a[2] * a[2] * t;
}

// body frame positions for one qpos: MJCF coordinate="global" tree, hinges about anchors (x->y->z per body)
void forward_kinematics(const Surrogate &S, const double *qpos, double *xpos) {
    double R[EGP_MAX_BODY * 9];        // (no heap allocation on the per-env-step path)
    quat_to_mat(qpos + 3, R);
    xpos[0] = qpos[0]; xpos[1] = qpos[1]; xpos[2] = qpos[2];
    int j = 0;
    for (int b = 1; b < S.nbody; ++b) {
        const int p = S.body_parent[b];
        double Rb[9], pb[3], off[3], tmp[3];
        memcpy(Rb, &R[p * 9], sizeof(Rb));
        for (int k = 0; k < 3; ++k) off[k] = S.body_pos[b * 3 + k] - S.body_pos[p * 3 + k];
        mat_vec(&R[p * 9], off, tmp);
        for (int k = 0; k < 3; ++k) pb[k] = xpos[p * 3 + k] + tmp[k];
        for (int d = 0; d < S.body_ndof[b]; ++d, ++j) {
            double al[3], aw[3], Rj[9], Rn[9];
            for (int k = 0; k < 3; ++k) al[k] = S.joint_anchor[j * 3 + k] - S.body_pos[b * 3 + k];
            mat_vec(Rb, al, tmp);
            for (int k = 0; k < 3; ++k) aw[k] = pb[k] + tmp[k];
            axis_angle_mat(&S.joint_axis[j * 3], qpos[7 + j], Rj);
            mat_mul(Rb, Rj, Rn);
            memcpy(Rb, Rn, sizeof(Rb));
            mat_vec(Rb, al, tmp);
            for (int k = 0; k < 3; ++k) pb[k] = aw[k] - tmp[k];
        }
        memcpy(&R[b * 9], Rb, sizeof(Rb));
        for (int k = 0; k < 3; ++k) xpos[b * 3 + k] = pb[k];
    }
}

#if defined(__HIP_DEVICE_COMPILE__)      // (the file is also parsed for the GPU: no x86 code there)
static inline void matvec_axpy(const double *, const double *, double *, int) {}
#else
// acc[i] = sum_j M[j][i] * f[j] with one fused multiply-add per term in ascending j (every variant below produces the
// same bits). This product is most of a surrogate step; the plain loop keeps acc[] in memory (a load, an FMA and a store
// per term: ~1 800 core cycles for nv = 58), the vector variants keep the accumulators in registers across j.
static void matvec_axpy_plain(const double *M, const double *f, double *acc, int nv) {
    for (int i = 0; i < nv; ++i) acc[i] = 0.0;
    for (int j = 0; j < nv; ++j) {
        const double *__restrict row = M + (size_t)j * nv;
        const double fj = f[j];
        for (int i = 0; i < nv; ++i) acc[i] = __builtin_fma(row[i], fj, acc[i]);
    }
}

// (fixed unrolling: accumulators indexed by a run-time loop live on the stack and the variant is no faster than the plain one)
__attribute__((target("avx2,fma"))) static void matvec_axpy_avx2(const double *M, const double *f, double *acc, int nv) {
    // two passes of 8 ymm accumulators = 32 columns each; columns past nv are masked off (masked lanes are never touched)
    for (int blk = 0; blk < 64 && blk < nv; blk += 32) {
        __m256i mk[8];
        for (int k = 0; k < 8; ++k) {
            const int left = nv - blk - 4 * k;
            mk[k] = _mm256_set_epi64x(left > 3 ? -1 : 0, left > 2 ? -1 : 0, left > 1 ? -1 : 0, left > 0 ? -1 : 0);
        }
        const __m256i m0 = mk[0], m1 = mk[1], m2 = mk[2], m3 = mk[3], m4 = mk[4], m5 = mk[5], m6 = mk[6], m7 = mk[7];
        __m256d a0 = _mm256_setzero_pd(), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
        for (int j = 0; j < nv; ++j) {
            const double *row = M + (size_t)j * nv + blk;
            const __m256d fj = _mm256_set1_pd(f[j]);
            a0 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 0, m0), fj, a0);
            a1 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 4, m1), fj, a1);
            a2 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 8, m2), fj, a2);
            a3 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 12, m3), fj, a3);
            a4 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 16, m4), fj, a4);
            a5 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 20, m5), fj, a5);
            a6 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 24, m6), fj, a6);
            a7 = _mm256_fmadd_pd(_mm256_maskload_pd(row + 28, m7), fj, a7);
        }
        double *o = acc + blk;
        _mm256_maskstore_pd(o + 0, m0, a0); _mm256_maskstore_pd(o + 4, m1, a1); _mm256_maskstore_pd(o + 8, m2, a2);
        _mm256_maskstore_pd(o + 12, m3, a3); _mm256_maskstore_pd(o + 16, m4, a4); _mm256_maskstore_pd(o + 20, m5, a5);
        _mm256_maskstore_pd(o + 24, m6, a6); _mm256_maskstore_pd(o + 28, m7, a7);
    }
}

__attribute__((target("avx512f"))) static void matvec_axpy_avx512(const double *M, const double *f, double *acc, int nv) {
    // nv <= 64: eight zmm accumulators hold the whole result; vectors past nv are loaded / stored under an empty lane mask
    __mmask8 mk[8];
    for (int k = 0; k < 8; ++k) {
        const int left = nv - 8 * k;
        mk[k] = left >= 8 ? (__mmask8)0xFF : (left > 0 ? (__mmask8)((1u << left) - 1u) : (__mmask8)0);
    }
    const __mmask8 m0 = mk[0], m1 = mk[1], m2 = mk[2], m3 = mk[3], m4 = mk[4], m5 = mk[5], m6 = mk[6], m7 = mk[7];
    __m512d a0 = _mm512_setzero_pd(), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    for (int j = 0; j < nv; ++j) {
        const double *row = M + (size_t)j * nv;
        const __m512d fj = _mm512_set1_pd(f[j]);
        a0 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m0, row + 0), fj, a0);
        a1 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m1, row + 8), fj, a1);
        a2 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m2, row + 16), fj, a2);
        a3 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m3, row + 24), fj, a3);
        a4 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m4, row + 32), fj, a4);
        a5 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m5, row + 40), fj, a5);
        a6 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m6, row + 48), fj, a6);
        a7 = _mm512_fmadd_pd(_mm512_maskz_loadu_pd(m7, row + 56), fj, a7);
    }
    _mm512_mask_storeu_pd(acc + 0, m0, a0); _mm512_mask_storeu_pd(acc + 8, m1, a1); _mm512_mask_storeu_pd(acc + 16, m2, a2);
    _mm512_mask_storeu_pd(acc + 24, m3, a3); _mm512_mask_storeu_pd(acc + 32, m4, a4); _mm512_mask_storeu_pd(acc + 40, m5, a5);
    _mm512_mask_storeu_pd(acc + 48, m6, a6); _mm512_mask_storeu_pd(acc + 56, m7, a7);
}

typedef void (*matvec_fn)(const double *, const double *, double *, int);
static matvec_fn pick_matvec() {
    const char *e = getenv("EGP_SURROGATE_SIMD");          // "plain" | "avx2" | "avx512" (default: the widest the CPU has)
    __builtin_cpu_init();
    const bool has512 = __builtin_cpu_supports("avx512f"), has2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (e && !strcmp(e, "plain")) return matvec_axpy_plain;
    if (e && !strcmp(e, "avx2")) return has2 ? matvec_axpy_avx2 : matvec_axpy_plain;
    if (has512) return matvec_axpy_avx512;
    return has2 ? matvec_axpy_avx2 : matvec_axpy_plain;
}
static inline void matvec_axpy(const double *M, const double *f, double *acc, int nv) {
    static const matvec_fn fn = pick_matvec();
    if (nv > 64) { matvec_axpy_plain(M, f, acc, nv); return; }
    fn(M, f, acc, nv);
}

#endif

void compute_bias(const Surrogate &S, int env, double *C) {
    const double *q = &S.qpos[(size_t)env * S.nq];
    const double *v = &S.qvel[(size_t)env * S.nv];
    for (int i = 0; i < 6; ++i) C[i] = 0.0;
    C[2] = S.support_k * (q[2] - S.zref[env]) + S.support_c * v[2];
    for (int i = 6; i < S.nv; ++i) C[i] = S.damping * v[i];
}

int sur_reset(void *user, int32_t env, const double *qpos, const double *qvel) {
    Surrogate &S = *(Surrogate *)user;
    if (env < 0 || env >= S.n_env) return EGP_E_INVALID;
    memcpy(&S.qpos[(size_t)env * S.nq], qpos, S.nq * sizeof(double));
    memcpy(&S.qvel[(size_t)env * S.nv], qvel, S.nv * sizeof(double));
    S.zref[env] = qpos[2];
    compute_bias(S, env, &S.bias[(size_t)env * S.nv]);   // sim.forward(): bias at the reset state
    return EGP_OK;
}

int sur_step(void *user, int32_t env, const double *ctrl) {
    Surrogate &S = *(Surrogate *)user;
    if (env < 0 || env >= S.n_env) return EGP_E_INVALID;
    double *q = &S.qpos[(size_t)env * S.nq];
    double *v = &S.qvel[(size_t)env * S.nv];
    double *C = &S.bias[(size_t)env * S.nv];
    const int nv = S.nv;
    std::chrono::steady_clock::time_point t_in;
    if (S.min_step_ns > 0) t_in = std::chrono::steady_clock::now();
    compute_bias(S, env, C);
    double f[EGP_MAX_NV], acc[EGP_MAX_NV];
    for (int i = 0; i < 6; ++i) f[i] = -C[i];
    for (int i = 6; i < nv; ++i) f[i] = ctrl[i - 6] - C[i];
    // acc = Minv0 * f as nv axpys over contiguous rows (Minv0 is symmetric): acc_i = fma(M[j][i], f[j], acc_i), j ascending
    matvec_axpy(S.Minv0.data(), f, acc, nv);
    for (int i = 0; i < nv; ++i) v[i] += S.dt * acc[i];
    for (int k = 0; k < 3; ++k) q[k] += S.dt * v[k];
    // q <- q * exp(dt * omega / 2), omega in the body frame
    const double wx = v[3], wy = v[4], wz = v[5];
    const double wn = sqrt(wx * wx + wy * wy + wz * wz);
    if (wn > 1e-12) {
        const double h = 0.5 * S.dt * wn, sh = sin(h) / wn;
        const double dq[4] = {cos(h), wx * sh, wy * sh, wz * sh};
        double o[4];
        quat_mul(q + 3, dq, o);
        const double n = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        for (int k = 0; k < 4; ++k) q[3 + k] = o[k] / n;
    }
    for (int i = 6; i < nv; ++i) q[i + 1] += S.dt * v[i];
    if (S.min_step_ns > 0)
        while (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_in).count() < S.min_step_ns) {}
    return EGP_OK;
}

int sur_drain(void *user, int32_t env, double *qpos, double *qvel, double *qM, double *bias, double *xpos) {
    Surrogate &S = *(Surrogate *)user;
    if (env < 0 || env >= S.n_env) return EGP_E_INVALID;
    if (qpos) memcpy(qpos, &S.qpos[(size_t)env * S.nq], S.nq * sizeof(double));
    if (qvel) memcpy(qvel, &S.qvel[(size_t)env * S.nv], S.nv * sizeof(double));
    if (qM) memcpy(qM, S.qM0.data(), S.nM * sizeof(double));
    if (bias) memcpy(bias, &S.bias[(size_t)env * S.nv], S.nv * sizeof(double));
    if (xpos) forward_kinematics(S, &S.qpos[(size_t)env * S.nq], xpos);
    return EGP_OK;
}

int64_t sur_epoch(void *, int32_t) { return 1; }   // M0 never changes

void sur_destroy(void *user) { delete (Surrogate *)user; }

}  // namespace

extern "C" {

int egp_physics_register(const egp_physics_vtable *vt, int32_t n_env, egp_physics **out) {
    EGP_REQUIRE(vt && out, "vtable/out is NULL");
    EGP_REQUIRE(vt->reset && vt->step && vt->drain, "vtable needs reset, step and drain");
    EGP_REQUIRE(n_env > 0, "n_env must be positive");
    egp_physics *p = new egp_physics();
    p->vt = *vt;
    p->n_env = n_env;
    *out = p;
    return EGP_OK;
}

int egp_physics_create_surrogate(const egp_surrogate_desc *d, int32_t n_env, egp_physics **out) {
    EGP_REQUIRE(d && out, "desc/out is NULL");
    EGP_REQUIRE(n_env > 0, "n_env must be positive");
    EGP_REQUIRE(d->nv > 6 && d->nv <= EGP_MAX_NV && d->nq == d->nv + 1 && d->nu == d->nv - 6, "bad dims");
    EGP_REQUIRE(d->njoint == d->nv - 6 && d->nbody >= 2 && d->nbody <= EGP_MAX_BODY, "bad joint/body count");
    EGP_REQUIRE(d->qM0 && d->Minv0 && d->body_parent && d->body_pos && d->body_ndof && d->joint_axis && d->joint_anchor, "NULL table");
    EGP_REQUIRE(d->sub_dt > 0, "sub_dt must be positive");
    Surrogate *S = new Surrogate();
    S->nq = d->nq; S->nv = d->nv; S->nu = d->nu; S->nbody = d->nbody; S->nM = d->nM; S->njoint = d->njoint;
    S->qM0.assign(d->qM0, d->qM0 + d->nM);
    S->Minv0.assign(d->Minv0, d->Minv0 + (size_t)d->nv * d->nv);
    S->body_parent.assign(d->body_parent, d->body_parent + d->nbody);
    S->body_ndof.assign(d->body_ndof, d->body_ndof + d->nbody);
    S->body_pos.assign(d->body_pos, d->body_pos + (size_t)d->nbody * 3);
    S->joint_axis.assign(d->joint_axis, d->joint_axis + (size_t)d->njoint * 3);
    S->joint_anchor.assign(d->joint_anchor, d->joint_anchor + (size_t)d->njoint * 3);
    S->dt = d->sub_dt; S->damping = d->damping; S->support_k = d->support_k; S->support_c = d->support_c;
    S->n_env = n_env;
    S->qpos.assign((size_t)n_env * d->nq, 0.0);
    S->qvel.assign((size_t)n_env * d->nv, 0.0);
    S->bias.assign((size_t)n_env * d->nv, 0.0);
    S->zref.assign(n_env, 0.0);
    for (int e = 0; e < n_env; ++e) S->qpos[(size_t)e * d->nq + 3] = 1.0;
    egp_physics_vtable vt{};
    vt.user = S; vt.reset = sur_reset; vt.step = sur_step; vt.drain = sur_drain; vt.destroy = sur_destroy;
    if (const char *e = getenv("EGP_SURROGATE_SUBSTEP_US")) {
        const double us = atof(e);
        if (us > 0) {
            S->min_step_ns = (long long)(us * 1e3);
            char buf[64];
            snprintf(buf, sizeof buf, "+%gus-per-substep", us);
            S->name += buf;
        }
    }
    vt.name = S->name.c_str();
    vt.inertia_epoch = sur_epoch;
    // EGP_SURROGATE_ALWAYS_DIRTY=1: report "inertia changed" on every drain, as a simulator with a pose-dependent qM
    // (MuJoCo) would -- same numbers (M0), but the full 7.3 kB inertia row crosses to the GPU every substep. Traffic
    // emulation for measurements (DESIGN: inertia epochs / device dynamics).
    if (const char *e = getenv("EGP_SURROGATE_ALWAYS_DIRTY")) if (atoi(e) != 0) vt.inertia_epoch = nullptr;
    egp_physics *p = new egp_physics();
    p->vt = vt; p->n_env = n_env; p->owns_user = true;
    *out = p;
    return EGP_OK;
}

int egp_physics_destroy(egp_physics *p) {
    if (!p) return EGP_OK;
    if (p->vt.destroy) p->vt.destroy(p->vt.user);
    delete p;
    return EGP_OK;
}

const char *egp_physics_name(const egp_physics *p) { return p && p->vt.name ? p->vt.name : "unnamed"; }
int32_t egp_physics_n_env(const egp_physics *p) { return p ? p->n_env : 0; }

int egp_physics_reset_host(egp_physics *p, int32_t env, const double *qpos, const double *qvel) {
    EGP_REQUIRE(p && qpos && qvel, "NULL pointer");
    EGP_REQUIRE(env >= 0 && env < p->n_env, "env out of range");
    return p->vt.reset(p->vt.user, env, qpos, qvel) == 0 ? EGP_OK : EGP_E_PHYSICS;
}
int egp_physics_step_host(egp_physics *p, int32_t env, const double *ctrl) {
    EGP_REQUIRE(p && ctrl, "NULL pointer");
    EGP_REQUIRE(env >= 0 && env < p->n_env, "env out of range");
    return p->vt.step(p->vt.user, env, ctrl) == 0 ? EGP_OK : EGP_E_PHYSICS;
}
int egp_physics_drain_host(egp_physics *p, int32_t env, double *qpos, double *qvel, double *qM, double *bias, double *xpos) {
    EGP_REQUIRE(p, "NULL pointer");
    EGP_REQUIRE(env >= 0 && env < p->n_env, "env out of range");
    return p->vt.drain(p->vt.user, env, qpos, qvel, qM, bias, xpos) == 0 ? EGP_OK : EGP_E_PHYSICS;
}

}  // extern "C"

// internal accessors for the engine
const egp_physics_vtable *egp_physics_vt(const egp_physics *p) { return &p->vt; }
