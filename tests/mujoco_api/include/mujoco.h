/* TEST DOUBLE -- not MuJoCo, and not derived from MuJoCo's headers.
 *
 * The MuJoCo plugin (egopose_amd/csrc/egp_physics_mujoco.cpp) is written against MuJoCo's public C API, which is not in the
 * build image. This header declares ONLY the entry points, struct fields and enum values that plugin touches, with MuJoCo's
 * published names and meanings, so that the plugin compiles and links here against tests/mujoco_api/fake_mujoco.cpp -- a
 * stand-in "simulator" that runs this package's own surrogate integrator behind those names. It exists to turn the plugin
 * from never-compiled source into compiled, exercised code (registration with the engine, reset / step / drain / inertia
 * epochs, the `xpos + 3` world-body offset, nM, the bad-state return of a step). IT PINS NO PHYSICS: nothing computed
 * through it says anything about MuJoCo's arithmetic; tests/golden/mujoco_dynamics.npz (tools/gen_mujoco_golden.py on a
 * MuJoCo-equipped machine) remains the only way to pin that. Field order and struct sizes here are arbitrary. */
#ifndef EGP_TEST_FAKE_MUJOCO_H
#define EGP_TEST_FAKE_MUJOCO_H

#ifdef __cplusplus
extern "C" {
#endif

typedef double mjtNum;

enum { mjJNT_FREE = 0, mjJNT_BALL = 1, mjJNT_SLIDE = 2, mjJNT_HINGE = 3 };
enum { mjWARN_INERTIA = 0, mjWARN_CONTACTFULL, mjWARN_CNSTRFULL, mjWARN_VGEOMFULL, mjWARN_BADQPOS, mjWARN_BADQVEL, mjWARN_BADQACC,
       mjWARN_BADCTRL, mjNWARNING };

typedef struct mjWarningStat_ { int lastinfo; int number; } mjWarningStat;
typedef struct mjOption_ { mjtNum timestep; } mjOption;

typedef struct mjModel_ {
    int nq, nv, nu, nbody, njnt, nM;
    mjOption opt;
    int *jnt_type;          /* [njnt] */
    int *jnt_qposadr;       /* [njnt] */
    int *dof_parentid;      /* [nv] */
    int *dof_Madr;          /* [nv] */
    int *body_jntadr;       /* [nbody] first joint of the body, -1: none */
    int *body_dofnum;       /* [nbody] */
    void *fake;             /* the stand-in's own data */
} mjModel;

typedef struct mjData_ {
    mjtNum *qpos, *qvel, *ctrl;     /* [nq], [nv], [nu] */
    mjtNum *qM;                     /* [nM] inertia in the sparse dof-tree order */
    mjtNum *qfrc_bias;              /* [nv] */
    mjtNum *xpos;                   /* [nbody * 3], body 0 = the world */
    mjWarningStat warning[mjNWARNING];
    void *fake;
} mjData;

mjModel *mj_loadXML(const char *filename, const void *vfs, char *error, int error_sz);
void mj_deleteModel(mjModel *m);
mjData *mj_makeData(const mjModel *m);
void mj_deleteData(mjData *d);
void mj_resetData(const mjModel *m, mjData *d);
void mj_forward(const mjModel *m, mjData *d);
void mj_step(const mjModel *m, mjData *d);
void mju_zero(mjtNum *res, int n);
int mj_version(void);
int mj_activate(const char *filename);

#ifdef __cplusplus
}
#endif
#endif
