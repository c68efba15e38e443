// egp_physics_mujoco.cpp -- MuJoCo behind the host physics boundary (include/egopose_hip.h: egp_physics_vtable).
//
// What the reference does through mujoco_py, one env per forked worker:
//   load_model_from_path / MjSim                 envs/common/mujoco_env.py:18-23
//   sim.reset / set_state / sim.forward          envs/common/mujoco_env.py:84-101
//   data.ctrl[:] = torque; sim.step()            ego_pose/envs/humanoid_v1.py:173-174   (frame_skip = 15 times per env-step)
//   mjf.mj_fullM(model, M, data.qM)              ego_pose/envs/humanoid_v1.py:133-135   -> drain() hands out data.qM itself (nM legacy
//                                                sparse entries); the K1 kernels expand it through LDS, which IS mj_fullM
//   data.qfrc_bias                               ego_pose/envs/humanoid_v1.py:136
//   data.body_xpos / get_body_com('Head')        ego_pose/envs/humanoid_v1.py:106,187   -> drain()'s xpos = mjData.xpos[1:]
// here: ONE mjModel shared by n_env mjData, each stepped by whichever engine thread owns the env (MuJoCo's documented
// multi-threading model: mjModel is read-only during simulation, one mjData per thread of control).
//
// This file is NOT part of libegopose_hip.so: MuJoCo is an un-vendored dependency of the reference and is not in the build
// image. `python -m egopose_amd.build_mujoco` compiles it into egopose_amd/libegopose_mujoco.so when MUJOCO_DIR points at a
// MuJoCo tree (include/mujoco.h or include/mujoco/mujoco.h, lib/ or bin/ with libmujoco.so / libmujoco210.so); the plugin
// talks to the main library through its public C-ABI only (egp_physics_register). Parity of everything physics-related is
// pinned on a MuJoCo-equipped machine by tools/gen_mujoco_golden.py -> tests/golden/mujoco_dynamics.npz (K8 against mjData).
//
// The reference's MJCF (assets/mujoco_models/humanoid_1205_v1.xml:14) is written with coordinate="global", which MuJoCo
// dropped in 2.1.2: use the release the reference ran (mujoco200 / mujoco210), or convert the model once with that release's
// `compile` tool (local coordinates) for newer ones.
#if __has_include(<mujoco/mujoco.h>)
#include <mujoco/mujoco.h>
#else
#include <mujoco.h>
#endif

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "../../include/egopose_hip.h"

namespace {

struct MjBackend {
    mjModel *m = nullptr;
    std::vector<mjData *> d;                       // one per env
    std::unique_ptr<std::atomic<int64_t>[]> epoch; // bumped whenever an env's qM may have changed (every step / reset)
    std::string name;
    int n_env = 0;
    ~MjBackend() {
        for (mjData *x : d) if (x) mj_deleteData(x);
        if (m) mj_deleteModel(m);
    }
};

// MujocoEnv.set_state (envs/common/mujoco_env.py:95-101): overwrite qpos / qvel, forward kinematics + dynamics terms
int mj_reset_cb(void *user, int32_t env, const double *qpos, const double *qvel) {
    MjBackend *B = static_cast<MjBackend *>(user);
    if (env < 0 || env >= B->n_env) return -1;
    mjData *d = B->d[env];
    mj_resetData(B->m, d);
    memcpy(d->qpos, qpos, sizeof(double) * B->m->nq);
    memcpy(d->qvel, qvel, sizeof(double) * B->m->nv);
    mju_zero(d->ctrl, B->m->nu);
    mj_forward(B->m, d);
    B->epoch[env].fetch_add(1, std::memory_order_relaxed);
    return 0;
}

// one substep of do_simulation (ego_pose/envs/humanoid_v1.py:172-174): data.ctrl[:] = torque; sim.step()
int mj_step_cb(void *user, int32_t env, const double *ctrl) {
    MjBackend *B = static_cast<MjBackend *>(user);
    if (env < 0 || env >= B->n_env) return -1;
    mjData *d = B->d[env];
    memcpy(d->ctrl, ctrl, sizeof(double) * B->m->nu);
    mj_step(B->m, d);
    B->epoch[env].fetch_add(1, std::memory_order_relaxed);
    // a diverged simulation (mj_step resets the data and raises a warning) must not be stepped on silently
    if (d->warning[mjWARN_BADQACC].number > 0 || d->warning[mjWARN_BADQPOS].number > 0 || d->warning[mjWARN_BADQVEL].number > 0) return -2;
    return 0;
}

// copy out what compute_torque / get_ee_pos / the termination test read. Like the reference between two sim.step() calls,
// qM and qfrc_bias are those mj_step computed at the START of the step that produced this qpos (one substep stale,
// SURVEY section 3.5) -- exactly what mjData holds after mj_step.
int mj_drain_cb(void *user, int32_t env, double *qpos, double *qvel, double *qM, double *qfrc_bias, double *xpos) {
    MjBackend *B = static_cast<MjBackend *>(user);
    if (env < 0 || env >= B->n_env) return -1;
    const mjModel *m = B->m;
    const mjData *d = B->d[env];
    if (qpos) memcpy(qpos, d->qpos, sizeof(double) * m->nq);
    if (qvel) memcpy(qvel, d->qvel, sizeof(double) * m->nv);
    if (qM) memcpy(qM, d->qM, sizeof(double) * m->nM);
    if (qfrc_bias) memcpy(qfrc_bias, d->qfrc_bias, sizeof(double) * m->nv);
    if (xpos) memcpy(xpos, d->xpos + 3, sizeof(double) * 3 * (m->nbody - 1));      // body 0 is the world
    return 0;
}

void mj_destroy_cb(void *user) { delete static_cast<MjBackend *>(user); }

int64_t mj_epoch_cb(void *user, int32_t env) {
    MjBackend *B = static_cast<MjBackend *>(user);
    return (env >= 0 && env < B->n_env) ? B->epoch[env].load(std::memory_order_relaxed) : 0;
}

void say(char *err, int errlen, const char *msg) {
    if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg);
}

}  // namespace

extern "C" {

// Loads `mjcf_path` (the model HumanoidEnv.__init__ names, cfg.mujoco_model_file) and registers an n_env-wide backend with
// libegopose_hip.so. Checks the dimensions the kernels are built for against the model (nq = nv + 1, nu = nv - 6, one free
// root joint). `frame_skip` is the engine's business, not the backend's. Returns EGP_OK or a negative code with a message in `err`.
int egp_physics_create_mujoco(const char *mjcf_path, int32_t n_env, egp_physics **out, char *err, int32_t errlen) {
    if (!mjcf_path || !out || n_env <= 0) { say(err, errlen, "egp_physics_create_mujoco: bad arguments"); return EGP_E_INVALID; }
#ifdef EGP_MUJOCO_ACTIVATE                           // mujoco200 and older need a licence key (mujoco_py read MUJOCO_PY_MJKEY_PATH)
    const char *key = getenv("MUJOCO_KEY_PATH") ? getenv("MUJOCO_KEY_PATH") : getenv("MUJOCO_PY_MJKEY_PATH");
    if (!key || !mj_activate(key)) { say(err, errlen, "mj_activate failed: set MUJOCO_KEY_PATH"); return EGP_E_PHYSICS; }
#endif
    char lerr[1000] = "";
    std::unique_ptr<MjBackend> B(new MjBackend());
    B->m = mj_loadXML(mjcf_path, nullptr, lerr, (int)sizeof lerr);
    if (!B->m) { say(err, errlen, lerr[0] ? lerr : "mj_loadXML failed"); return EGP_E_PHYSICS; }
    const mjModel *m = B->m;
    if (m->nq != m->nv + 1 || m->nu != m->nv - 6 || m->nv > EGP_MAX_NV || m->njnt < 1 || m->jnt_type[0] != mjJNT_FREE) {
        say(err, errlen, "model is not a free-root humanoid with one actuator per hinge (nq = nv + 1, nu = nv - 6, nv <= 64)");
        return EGP_E_INVALID;
    }
    B->n_env = n_env;
    B->d.assign((size_t)n_env, nullptr);
    B->epoch.reset(new std::atomic<int64_t>[(size_t)n_env]);
    for (int e = 0; e < n_env; ++e) {
        B->epoch[e].store(0);
        B->d[e] = mj_makeData(m);
        if (!B->d[e]) { say(err, errlen, "mj_makeData failed (out of memory?)"); return EGP_E_PHYSICS; }
        mj_forward(m, B->d[e]);
    }
    char nm[64];
    snprintf(nm, sizeof nm, "mujoco-%d", mj_version());
    B->name = nm;
    egp_physics_vtable vt;
    memset(&vt, 0, sizeof vt);
    vt.user = B.get();
    vt.reset = mj_reset_cb;
    vt.step = mj_step_cb;
    vt.drain = mj_drain_cb;
    vt.destroy = mj_destroy_cb;
    vt.name = B->name.c_str();
    vt.inertia_epoch = mj_epoch_cb;                  // qM depends on qpos: it moves with every step
    const int rc = egp_physics_register(&vt, n_env, out);
    if (rc != EGP_OK) { say(err, errlen, egp_last_error()); return rc; }
    B.release();                                     // owned by the registered backend from here (vt.destroy)
    return EGP_OK;
}

// mjModel quantities the kernel context is built from (what egopose_amd.skeleton reads from the MJCF text otherwise):
// lets a caller cross-check the two. Arrays must hold nv (dof_parentid, dof_Madr) and nbody - 1 entries (the rest).
int egp_mujoco_model_tables(const char *mjcf_path, int32_t *nq, int32_t *nv, int32_t *nu, int32_t *nbody, int32_t *nM, double *timestep,
                            int32_t *dof_parentid, int32_t *dof_Madr, int32_t *body_jntadr_qpos, int32_t *body_ndof, char *err, int32_t errlen) {
    char lerr[1000] = "";
    mjModel *m = mj_loadXML(mjcf_path, nullptr, lerr, (int)sizeof lerr);
    if (!m) { say(err, errlen, lerr[0] ? lerr : "mj_loadXML failed"); return EGP_E_PHYSICS; }
    if (nq) *nq = m->nq;
    if (nv) *nv = m->nv;
    if (nu) *nu = m->nu;
    if (nbody) *nbody = m->nbody - 1;
    if (nM) *nM = m->nM;
    if (timestep) *timestep = m->opt.timestep;
    for (int i = 0; i < m->nv; ++i) {
        if (dof_parentid) dof_parentid[i] = m->dof_parentid[i];
        if (dof_Madr) dof_Madr[i] = m->dof_Madr[i];
    }
    for (int b = 1; b < m->nbody; ++b) {             // utils/tools.py:55-68 (get_body_qposaddr): first qpos index and joint count per body
        const int j0 = m->body_jntadr[b];
        if (body_jntadr_qpos) body_jntadr_qpos[b - 1] = j0 >= 0 ? m->jnt_qposadr[j0] : -1;
        if (body_ndof) body_ndof[b - 1] = m->body_dofnum[b];
    }
    mj_deleteModel(m);
    return EGP_OK;
}

}  // extern "C"
