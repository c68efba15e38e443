// TEST DOUBLE -- see include/mujoco.h in this directory. The entry points of MuJoCo's C API that egp_physics_mujoco.cpp calls,
// implemented on this package's surrogate integrator (egp_physics_create_surrogate behind the public C-ABI): one 1-env
// surrogate backend per mjData. The "model file" is a side-car written by tests (mujoco_api.write_fake_model): mj_loadXML
// reads <filename>.fakemj. Pins no physics.
#include <mujoco.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/egopose_hip.h"

namespace {
struct FakeModel {
    std::vector<int> jnt_type, jnt_qposadr, dof_parentid, dof_Madr, body_jntadr, body_dofnum;
    std::vector<double> qM0, Minv0, body_pos, joint_axis, joint_anchor;
    std::vector<int> body_parent, body_ndof;
    egp_surrogate_desc desc;
};
struct FakeData {
    egp_physics *phys = nullptr;
    std::vector<double> buf;
};
template <typename T> bool rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); return n == 0 || fread(v.data(), sizeof(T), n, f) == n; }
void refresh(const mjModel *m, mjData *d) {       // mjData arrays <- the surrogate's drained state (body 0 = the world, at the origin)
    FakeData *fd = (FakeData *)d->fake;
    d->xpos[0] = d->xpos[1] = d->xpos[2] = 0.0;
    egp_physics_drain_host(fd->phys, 0, d->qpos, d->qvel, d->qM, d->qfrc_bias, d->xpos + 3);
}
}  // namespace

extern "C" {

mjModel *mj_loadXML(const char *filename, const void *, char *error, int error_sz) {
    char path[4096];
    snprintf(path, sizeof path, "%s.fakemj", filename);
    FILE *f = fopen(path, "rb");
    if (!f) { if (error) snprintf(error, (size_t)error_sz, "fake mujoco: cannot open %s", path); return nullptr; }
    int h[8];
    double t[4];
    FakeModel *fm = new FakeModel();
    mjModel *m = (mjModel *)calloc(1, sizeof(mjModel));
    bool ok = fread(h, sizeof(int), 8, f) == 8 && fread(t, sizeof(double), 4, f) == 4;
    const int nq = h[0], nv = h[1], nu = h[2], nbody = h[3] /* without the world */, nM = h[4], njoint = h[5];
    ok = ok && rd(f, fm->dof_parentid, nv) && rd(f, fm->dof_Madr, nv) && rd(f, fm->body_parent, nbody) && rd(f, fm->body_ndof, nbody) &&
         rd(f, fm->jnt_qposadr, nbody) /* first qpos index per body */ && rd(f, fm->qM0, nM) && rd(f, fm->Minv0, (size_t)nv * nv) &&
         rd(f, fm->body_pos, (size_t)nbody * 3) && rd(f, fm->joint_axis, (size_t)njoint * 3) && rd(f, fm->joint_anchor, (size_t)njoint * 3);
    fclose(f);
    if (!ok) { if (error) snprintf(error, (size_t)error_sz, "fake mujoco: %s is truncated", path); delete fm; free(m); return nullptr; }
    m->nq = nq; m->nv = nv; m->nu = nu; m->nbody = nbody + 1; m->nM = nM;
    m->opt.timestep = t[0];
    // joints as MuJoCo numbers them: body b (1-based; 0 = world) owns one joint entry here (the free joint / its first hinge)
    m->njnt = nbody;
    std::vector<int> first_qpos = fm->jnt_qposadr;
    fm->jnt_type.assign(nbody, mjJNT_HINGE); fm->jnt_type[0] = mjJNT_FREE;
    fm->body_jntadr.assign(nbody + 1, -1); fm->body_dofnum.assign(nbody + 1, 0);
    for (int b = 0; b < nbody; ++b) { fm->body_jntadr[b + 1] = b; fm->body_dofnum[b + 1] = fm->body_ndof[b]; }
    m->jnt_type = fm->jnt_type.data(); m->jnt_qposadr = fm->jnt_qposadr.data();
    m->dof_parentid = fm->dof_parentid.data(); m->dof_Madr = fm->dof_Madr.data();
    m->body_jntadr = fm->body_jntadr.data(); m->body_dofnum = fm->body_dofnum.data();
    egp_surrogate_desc &d = fm->desc;
    memset(&d, 0, sizeof d);
    d.nq = nq; d.nv = nv; d.nu = nu; d.nbody = nbody; d.nM = nM; d.njoint = njoint;
    d.qM0 = fm->qM0.data(); d.Minv0 = fm->Minv0.data(); d.body_parent = fm->body_parent.data(); d.body_pos = fm->body_pos.data();
    d.body_ndof = fm->body_ndof.data(); d.joint_axis = fm->joint_axis.data(); d.joint_anchor = fm->joint_anchor.data();
    d.sub_dt = t[0]; d.damping = t[1]; d.support_k = t[2]; d.support_c = t[3];
    m->fake = fm;
    return m;
}
void mj_deleteModel(mjModel *m) { if (m) { delete (FakeModel *)m->fake; free(m); } }

mjData *mj_makeData(const mjModel *m) {
    FakeData *fd = new FakeData();
    if (egp_physics_create_surrogate(&((FakeModel *)m->fake)->desc, 1, &fd->phys) != EGP_OK) { delete fd; return nullptr; }
    mjData *d = (mjData *)calloc(1, sizeof(mjData));
    fd->buf.assign((size_t)m->nq + m->nv + m->nu + m->nM + m->nv + 3 * m->nbody, 0.0);
    double *p = fd->buf.data();
    d->qpos = p; p += m->nq; d->qvel = p; p += m->nv; d->ctrl = p; p += m->nu; d->qM = p; p += m->nM; d->qfrc_bias = p; p += m->nv; d->xpos = p;
    d->fake = fd;
    d->qpos[3] = 1.0;
    return d;
}
void mj_deleteData(mjData *d) { if (d) { FakeData *fd = (FakeData *)d->fake; egp_physics_destroy(fd->phys); delete fd; free(d); } }
void mj_resetData(const mjModel *m, mjData *d) {
    memset(d->qpos, 0, sizeof(double) * m->nq); d->qpos[3] = 1.0;
    memset(d->qvel, 0, sizeof(double) * m->nv);
    memset(d->ctrl, 0, sizeof(double) * m->nu);
    memset(d->warning, 0, sizeof d->warning);
}
void mj_forward(const mjModel *m, mjData *d) {      // "set_state + forward": the surrogate takes (qpos, qvel), everything derived follows
    egp_physics_reset_host(((FakeData *)d->fake)->phys, 0, d->qpos, d->qvel);
    refresh(m, d);
}
void mj_step(const mjModel *m, mjData *d) {
    for (int i = 0; i < m->nu; ++i)
        if (!(d->ctrl[i] == d->ctrl[i]) || d->ctrl[i] > 1e12 || d->ctrl[i] < -1e12) {      // what MuJoCo reports as a diverged step
            d->warning[mjWARN_BADQACC].number += 1;
            return;
        }
    egp_physics_step_host(((FakeData *)d->fake)->phys, 0, d->ctrl);
    refresh(m, d);
}
void mju_zero(mjtNum *res, int n) { memset(res, 0, sizeof(mjtNum) * (size_t)n); }
int mj_version(void) { return 0; }               // "mujoco-0": nobody mistakes it for a release
int mj_activate(const char *) { return 1; }

}  // extern "C"
