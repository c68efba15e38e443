/*
 * egopose_hip.h -- C-ABI of the MI355X-native EgoPose PPO rollout+update hot path.
 *
 * The reference (Khrylx/EgoPose) is 100 % Python and has no FFI of its own; every entry point
 * below is the drop-in for a Python function on the hot path, cited as reference file:line.
 * The binding a maintainer adds on the reference side is a ctypes stub.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types. All array arguments are DEVICE pointers
 *     (HBM) unless the name ends in `_host`. Arrays are env-major, row-contiguous:
 *     qpos[n_env][nq], qvel[n_env][nv], ... exactly the layout MuJoCo's mjData has per env and
 *     the layout TrajBatch exposes ((N,115) rows).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). Calls are
 *     stream-ordered and asynchronous; nothing is retained past the call except inside `egp_ctx`.
 *   - every function returns an int status: EGP_OK (0) or a negative EGP_E_* code; it never
 *     throws. egp_last_error() gives a thread-local message.
 *   - two arithmetic variants per kernel: `_f64` (the reference's float64 arithmetic; parity
 *     <= 1e-10) and `_f32` (parity <= 1e-5; K1 keeps a float64 factorisation inside).
 */
#ifndef EGOPOSE_HIP_H
#define EGOPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGP_OK 0
#define EGP_E_INVALID (-1)   /* bad argument (NULL pointer, size mismatch, unsupported dims) */
#define EGP_E_HIP (-2)       /* a HIP runtime call failed; see egp_last_error() */
#define EGP_E_STATE (-3)     /* call order violated (e.g. reward before egp_upload_experts) */
#define EGP_E_PHYSICS (-4)   /* the physics backend reported a failure */

#define EGP_MAX_NV 64        /* K1 maps one dof per lane of a 64-wide wavefront */
#define EGP_MAX_BODY 32      /* K2 maps one body per lane of a half wavefront */
#define EGP_EXPERT_ROW 168   /* packed reward row per expert frame (166 used, padded) */

typedef struct egp_ctx egp_ctx;          /* model constants + expert table resident in HBM */
typedef struct egp_physics egp_physics;  /* host physics backend (MuJoCo-shaped boundary) */
typedef struct egp_engine egp_engine;    /* lockstep rollout engine: host threads + pinned staging */

/* ----------------------------------------------------------------------------------------
 * Model description: what the reference reads from mujoco_py's model + its YAML Config.
 *   skeleton tables   utils/tools.py:55-68 (body -> qpos address), mj_fullM's dof tree
 *   PD gains          ego_pose/utils/egomimic_config.py:108-116
 *   reward weights    ego_pose/core/reward_function.py:6-12, config/egomimic/<id>.yml
 * All pointers are HOST pointers, copied at egp_create. */
typedef struct egp_model_desc {
    int32_t nq, nv, nu, nbody, nM;        /* 59, 58, 52, 21, 910 for humanoid_1205_v1 */
    const int32_t *body_qpos_start;       /* [nbody] first qpos index of the body's joints (root: 0) */
    const int32_t *body_ndof;             /* [nbody] hinge count (root: 6) */
    const int32_t *dof_parentid;          /* [nv]    MuJoCo dof tree */
    const int32_t *dof_Madr;              /* [nv]    start of dof i's chain in the sparse inertia qM */
    const int32_t *ee_body;               /* [5]     LeftFoot, RightFoot, LeftHand, RightHand, Head */
    const double *jkp, *jkd, *a_ref, *a_scale, *torque_lim;   /* [nu] each */
    const double *b_diffw;                /* [nbody-1] */
    double sub_dt;                        /* model timestep (1/450 s) */
    int32_t frame_skip;                   /* 15 */
    int32_t episode_len;                  /* env_episode_len (200) */
    double w_p, w_v, w_e, w_rp, w_rv;     /* reward blend weights */
    double k_p, k_v, k_e, k_rh, k_rq, k_rl, k_ra;
    double v_ord;                         /* norm order of the body-angular-velocity term (2) */
    int32_t decay;                        /* reward *= 1 - t/episode_len */
    /* observation variants of HumanoidEnv.get_full_obs (humanoid_v1.py:73-96, defaults of egomimic_config.py:99-103 = all
     * zero here): obs = [heading angle]? ++ qpos[2:] (root quaternion de-headed unless obs_keep_root_heading)
     * ++ {qvel | qvel[:6] | nothing} with qvel[:3] rotated into the heading frame (obs_coord 'heading') or the root frame */
    int32_t obs_heading;                  /* cfg.obs_heading: prepend get_heading(root quaternion) */
    int32_t obs_keep_root_heading;        /* cfg.root_deheading == False */
    int32_t obs_coord_root;               /* cfg.obs_coord == 'root' */
    int32_t obs_vel;                      /* cfg.obs_vel: 0 'full', 1 'root', 2 none */
    /* cfg.action_type (egomimic_config.py:105, humanoid_v1.py:167-172): 0 'position' -- the action is a PD target and K1
     * solves the stable-PD system per substep; 1 'torque' -- torque = clip(a_ref + action * a_scale, +-torque_lim), the
     * same on every substep of the env-step, no solve (every K1 entry point and the engine follow this switch) */
    int32_t action_torque;
    /* cfg.obs_phase (humanoid_v1.py:92-94, egoforecast_config.py:107): the observation ends with one more column,
     * min(cur_t / episode_len, 1). Every entry point that forms observations from (qpos, qvel) then needs the rows' cur_t
     * (`phase_t`, int32 per row, device or device-visible memory); with obs_phase == 0 that argument is ignored (NULL). */
    int32_t obs_phase;
} egp_model_desc;
/* width of an observation row under those options (115 for the defaults) */
int32_t egp_obs_dim(const egp_ctx *ctx);

/* Expert table (ego_pose/data_process/gen_expert.py:28-83): all takes concatenated, HOST pointers.
 * frame f of take k lives at row take_offset[k] + f. */
typedef struct egp_expert_table {
    int32_t n_takes;
    int32_t n_frames;                     /* total rows */
    const int32_t *take_offset;           /* [n_takes + 1] */
    const double *qpos;                   /* [n_frames][59] */
    const double *qvel;                   /* [n_frames][58] */
    const double *rlinv_local;            /* [n_frames][3]  */
    const double *rangv;                  /* [n_frames][3]  */
    const double *rq_rmh;                 /* [n_frames][4]  */
    const double *ee_pos;                 /* [n_frames][15] */
    const double *bquat;                  /* [n_frames][84] */
    const double *bangvel;                /* [n_frames][63] */
    const double *head_height_lb;         /* [n_takes] */
} egp_expert_table;

const char *egp_last_error(void);
const char *egp_version(void);
/* sizeof() of the descriptor struct called `name` ("egp_gemm_desc", ...) as this library was built, or -1: lets a
 * binding check its own mirror of the layout before the first call */
int64_t egp_abi_sizeof(const char *name);

int egp_create(const egp_model_desc *desc, int device, egp_ctx **out);
int egp_destroy(egp_ctx *ctx);
/* replaces reward weights in place (cfg.reward_weights can change between iterations) */
int egp_set_reward_weights(egp_ctx *ctx, const egp_model_desc *desc);
/* K1 implementation switch: 0 = tree-ordered in-register elimination (humanoid_1205_v1 dof tree, default; a launch of
 * one substep runs it on a 4 x 16 lane grid per env), 3 = the same with one lane per matrix row (what the resident
 * engine kernel uses), 2 = dense in-register Gauss-Jordan (any tree with nv == 58), 1 = generic LDS kernel (any nv <= 64) */
int egp_set_pd_variant(egp_ctx *ctx, int variant);
/* HumanoidEnv.load_experts (ego_pose/envs/humanoid_v1.py:47-54): packs the reward rows into HBM */
int egp_upload_experts(egp_ctx *ctx, const egp_expert_table *tbl);


/* a7 -- quaternion algebra, (w, x, y, z) order, one row per thread. Replaces the per-call numpy functions of
 * utils/transformation.py:348-356 (rotation_from_quaternion), :1194-1248 (quaternion_from_euler 'sxyz'), :1267-1291
 * (quaternion_matrix), :1379-1393 (quaternion_multiply), :1410-1421 (quaternion_inverse) and utils/math.py:47-59
 * (transform_vec), :62-67 (get_heading_q), :80-81 (de_heading), :84-100 (multi_quat_diff / multi_quat_norm).
 *   op                              a         b         out
 *   EGP_QUAT_MUL                    q1 [n][4] q0 [n][4] q1*q0 [n][4]
 *   EGP_QUAT_INV                    q [n][4]  -         conj(q)/(q.q) [n][4]
 *   EGP_QUAT_FROM_EULER_SXYZ        e [n][3]  -         [n][4]
 *   EGP_QUAT_HEADING_Q              q [n][4]  -         [n][4]
 *   EGP_QUAT_DE_HEADING             q [n][4]  -         [n][4]
 *   EGP_QUAT_TRANSFORM_VEC_ROOT     v [n][3]  q [n][4]  R(q)^T v [n][3]
 *   EGP_QUAT_TRANSFORM_VEC_HEADING  v [n][3]  q [n][4]  R(heading_q(q))^T v [n][3]
 *   EGP_QUAT_ROTATION               q [n][4]  -         axis[3] | angle [n][4]   (identity branch 1 - w < 1e-8; no clamp, no wrap)
 *   EGP_QUAT_DIFF_HALF_ANGLE        q1 [n][4] q0 [n][4] acos(clip(w(q1 * q0^-1))) [n]   (= multi_quat_norm(multi_quat_diff))
 * The float32 variant evaluates the two angle ops through atan2(|xyz|, w) (deviation iii). */
enum {
    EGP_QUAT_MUL = 0, EGP_QUAT_INV = 1, EGP_QUAT_FROM_EULER_SXYZ = 2, EGP_QUAT_HEADING_Q = 3, EGP_QUAT_DE_HEADING = 4,
    EGP_QUAT_TRANSFORM_VEC_ROOT = 5, EGP_QUAT_TRANSFORM_VEC_HEADING = 6, EGP_QUAT_ROTATION = 7, EGP_QUAT_DIFF_HALF_ANGLE = 8,
    EGP_QUAT_N_OPS = 9
};
int egp_quat_op_f64(int32_t op, const double *a, const double *b, int32_t n, double *out, void *stream);
int egp_quat_op_f32(int32_t op, const float *a, const float *b, int32_t n, float *out, void *stream);

/* ---------------------------------------------------------------------------------------- K4
 * HumanoidEnv.get_body_quat (ego_pose/envs/humanoid_v1.py:113-125): qpos[n][nq] -> bquat[n][4*nbody] */
int egp_body_quat_f64(egp_ctx *ctx, const double *qpos, int32_t n, double *bquat, void *stream);
int egp_body_quat_f32(egp_ctx *ctx, const float *qpos, int32_t n, float *bquat, void *stream);

/* ---------------------------------------------------------------------------------------- K3
 * HumanoidEnv.get_full_obs (ego_pose/envs/humanoid_v1.py:73-96), obs_coord='heading',
 * root_deheading, obs_vel='full': obs[n][nq-2+nv] */
int egp_obs_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, int32_t n, double *obs, void *stream);
int egp_obs_f32(egp_ctx *ctx, const float *qpos, const float *qvel, const int32_t *phase_t, int32_t n, float *obs, void *stream);

/* ---------------------------------------------------------------------------------------- K1
 * HumanoidEnv.compute_torque + compute_desired_accel + the target/clip lines of do_simulation
 * (ego_pose/envs/humanoid_v1.py:130-156,167-172) for ONE physics substep of n envs.
 *   qM        [n][nM]  MuJoCo sparse inertia as left by the previous mj_step (mj_fullM is done on device)
 *   qfrc_bias [n][nv]
 *   action    [n][nu]  policy action (target = a_ref + action*a_scale)
 *   torque    [n][nu]  out: clipped to +-torque_lim (what is written to data.ctrl)
 *   torque_raw[n][nu]  out, optional (NULL to skip): before the clip */
int egp_pd_torque_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const double *action,
                      const double *qM, const double *qfrc_bias, int32_t n,
                      double *torque, double *torque_raw, void *stream);
int egp_pd_torque_f32(egp_ctx *ctx, const float *qpos, const float *qvel, const float *action,
                      const float *qM, const float *qfrc_bias, int32_t n,
                      float *torque, float *torque_raw, void *stream);

/* ---------------------------------------------------------------------------------------- K2
 * quat_space_reward_v3 (ego_pose/core/reward_function.py:4-60) on drained per-env state.
 *   cur_qpos, prev_qpos [n][nq];  ee_wpos [n][15] world positions of the 5 end effectors
 *   t      [n] env.cur_t after the step;  frame [n] global expert row (take_offset + start_ind + t)
 *   end    [n] info['end'] (0/1);  active [n] optional (NULL = all): inactive envs write 0
 *   reward [n], c_info [n][5] */
int egp_reward_quat_v3_f64(egp_ctx *ctx, const double *cur_qpos, const double *prev_qpos,
                           const double *ee_wpos, const int32_t *t, const int32_t *frame,
                           const int32_t *end, const int32_t *active, double end_reward, int32_t n,
                           double *reward, double *c_info, void *stream);
int egp_reward_quat_v3_f32(egp_ctx *ctx, const float *cur_qpos, const float *prev_qpos,
                           const float *ee_wpos, const int32_t *t, const int32_t *frame,
                           const int32_t *end, const int32_t *active, double end_reward, int32_t n,
                           float *reward, float *c_info, void *stream);
/* the other two entries of the reward registry (ego_pose/core/reward_function.py:63-80), float64, one c_info column:
 * EGP_REWARD_CONSTANT: reward 1.0, c_info 0;  EGP_REWARD_POSE_DIST: d = |(expert qpos - qpos)[2:]| at expert row frame[e]
 * (HumanoidEnv.get_pose_dist, humanoid_v1.py:275-280), reward 5 - 3 d (+ end_reward where end[e]), c_info d */
#define EGP_REWARD_CONSTANT 1
#define EGP_REWARD_POSE_DIST 2
int egp_reward_simple_f64(egp_ctx *ctx, int32_t kind, const double *qpos, const int32_t *frame, const int32_t *end, const int32_t *active,
                          double end_reward, int32_t n, double *reward, double *cinfo, void *stream);

/* ---------------------------------------------------------------------------------------- K7
 * Pose features of a (previous, current) frame pair in the reference's expert formats
 * (ego_pose/data_process/gen_expert.py:28-83; the learner side of reward_function.py:18-26):
 *   qvel[n][nv]        get_qvel_fd(prev, cur, dt)            (utils/math.py:20-35, world-frame root lin-vel)
 *   rlinv_local[n][3]  root lin-vel in the heading frame of the CURRENT root quat when expert_convention
 *                      != 0 (gen_expert.py:53), of the PREVIOUS one otherwise (reward_function.py:19-21)
 *   rangv[n][3], rq_rmh[n][4] (de_heading), ee_pos[n][15] (get_ee_pos 'heading'),
 *   bquat[n][4*nbody] (get_body_quat), bangvel[n][3*nbody] (get_angvel_fd) */
int egp_pose_features_f64(egp_ctx *ctx, const double *cur_qpos, const double *prev_qpos, const double *ee_wpos,
                          int32_t n, int32_t expert_convention, double *qvel, double *rlinv_local, double *rangv,
                          double *rq_rmh, double *ee_pos, double *bquat, double *bangvel, void *stream);
int egp_pose_features_f32(egp_ctx *ctx, const float *cur_qpos, const float *prev_qpos, const float *ee_wpos,
                          int32_t n, int32_t expert_convention, float *qvel, float *rlinv_local, float *rangv,
                          float *rq_rmh, float *ee_pos, float *bquat, float *bangvel, void *stream);

/* ---------------------------------------------------------------------------------------- K6
 * ZFilter / RunningStat (utils/zfilter.py:7-67), batched: Chan-merge the `active` rows of
 * x[n][dim] into the running state, then y = clip((x-mean)/(std+1e-8), +-clip) for all rows.
 *   state layout (device, float64 always): [0]=count, [1..dim]=mean, [1+dim..2*dim]=S
 *   state_in may equal state_out only when update==0.
 *   workspace: >= egp_zfilter_workspace_bytes(n, dim) bytes of device scratch */
int64_t egp_zfilter_workspace_bytes(int32_t n, int32_t dim);
int egp_zfilter_f64(const double *x, const int32_t *active, int32_t n, int32_t dim,
                    const double *state_in, double *state_out, int32_t update, double clip,
                    double *y, void *workspace, void *stream);
int egp_zfilter_f32(const float *x, const int32_t *active, int32_t n, int32_t dim,
                    const double *state_in, double *state_out, int32_t update, double clip,
                    float *y, void *workspace, void *stream);

/* K3 + K6 fused (what the rollout calls every tick): get_full_obs of the drained state pushed through the
 * running filter without an intermediate raw-observation array.
 *   phase_t [n]: the rows' cur_t when the model has obs_phase (egp_model_desc), else ignored
 *   active [n] (optional): rows that update the statistics; write_only_active != 0: only those rows are written
 *   state_in == NULL: no filter (y = raw observation);  y2 (optional): second copy of the output rows */
int egp_obs_zfilter_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, const int32_t *active, int32_t n,
                        const double *state_in, double *state_out, double clip, double *y, double *y2,
                        int32_t write_only_active, void *workspace, void *stream);
int egp_obs_zfilter_f32(egp_ctx *ctx, const float *qpos, const float *qvel, const int32_t *phase_t, const int32_t *active, int32_t n,
                        const double *state_in, double *state_out, double clip, float *y, float *y2,
                        int32_t write_only_active, void *workspace, void *stream);
/* egp_obs_zfilter_f64 (all rows written, no write mask) in two calls, for batches of at most egp_obs_zfilter_split_max_rows() rows:
 * _stats = its first launch (tile statistics of the rows with active != 0 into `workspace`), _apply = its second (merge into
 * state_in -> state_out, normalise, clip, write y / y2): the same kernels with the same arguments, bit-identical results.
 * The apply pass can instead ride in the policy step that consumes y2 (egp_policy_gaussian_filter_f32). */
int32_t egp_obs_zfilter_split_max_rows(void);
int egp_obs_zfilter_stats_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, const int32_t *active, int32_t n, void *workspace,
                              void *stream);
int egp_obs_zfilter_apply_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, int32_t n, const double *state_in, double *state_out,
                              double clip, double *y, double *y2, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------- K5
 * estimate_advantages (core/common.py:5-25) over the flat concatenated batch.
 *   rewards, masks, values [n] -> adv_raw [n] (before standardisation), returns [n]
 *   stats (device, float64[3]) receives {n, mean, M2 = sum (a-mean)^2} of adv_raw so that several
 *   ranks can Chan-merge before standardising; egp_gae_standardize applies
 *   (a - stats[1]) / sqrt(stats[2] / (stats[0] - 1))  (torch.std is unbiased) reading stats on device. */
int64_t egp_gae_workspace_bytes(int32_t n);
int egp_gae_f64(const double *rewards, const double *masks, const double *values, int32_t n,
                double gamma, double tau, double *adv_raw, double *returns, double *stats,
                void *workspace, void *stream);
int egp_gae_f32(const float *rewards, const float *masks, const float *values, int32_t n,
                double gamma, double tau, float *adv_raw, float *returns, double *stats,
                void *workspace, void *stream);
int egp_gae_standardize_f64(double *adv, int32_t n, const double *stats, void *stream);
int egp_gae_standardize_f32(float *adv, int32_t n, const double *stats, void *stream);


/* ------------------------------------------------------------------------------------ update GEMMs
 * float32 matrix products of the PPO update on the bf16 matrix cores with split operands (csrc/egp_gemm.hip): replaces
 * the rocBLAS / hipBLASLt calls behind the reference's nn.Linear layers (models/mlp.py:22-25, core/policy_gaussian.py:19-24,
 * core/critic.py:15-18) and behind the LSTM input projection / weight gradients (models/rnn.py:45-61 under autograd).
 *   C[M][N] = A[M][K] B[K][N]  (+ bias[n]) (ReLU) (* (mask[m][n] > 0))
 *   a_kcontig: A is given as [m][k] (element (m, k) at A[m*lda + k]); otherwise as [k][m] (A[k*lda + m])
 *   b_kcontig: B is given as [n][k] (B[n*ldb + k], e.g. an nn.Linear weight in a forward pass); otherwise [k][n]
 *   terms = 3: every operand is split x = hi + lo (two bf16) and a product is hi*hi + hi*lo + lo*hi in float32
 *              accumulators (~16 mantissa bits per product); terms = 1: bf16 inputs (hi*hi only)
 *   splits > 1: split-K, partial sums through `workspace` and a fixed-order reduction (deterministic); no epilogue then.
 *   bias_grad != NULL (needs b_kcontig = 0): B gets a virtual column of ones, its result -- the column sums of A given
 *              as [k][m], i.e. the bias gradient of a weight-gradient product -- goes to bias_grad[M].
 *   accumulate: split-K / bias_grad launches add to C (and bias_grad) instead of overwriting.
 * workspace: egp_gemm_workspace_floats(M, N, bias_grad != NULL, splits) floats, caller-owned, needed when splits > 1 or
 * bias_grad is set. All pointers are device memory, rows need 4-byte alignment only. */
typedef struct egp_gemm_desc {
    int32_t M, N, K;
    const float *A; int64_t lda; int32_t a_kcontig;
    const float *B; int64_t ldb; int32_t b_kcontig;
    float *C; int64_t ldc;
    const float *bias; int32_t relu;
    const float *mask; int64_t ldmask;
    int32_t terms, splits, accumulate;
    float *bias_grad;
    float *workspace;
    /* Optional gather / scatter fused into the product: the update's first MLP layer reads its input
     * [ ctx[idx[i]] | state[i] ] (models/video_state_net.py:65-69) straight from the two tensors, its weight gradient does the
     * same along k, and its data gradient writes the context rows it belongs to -- no concatenated copy, no scatter pass.
     * Only with three-piece products on the persistent kernel (terms = 6, k ranges >= 32, N % 4 == 0 unless split-K);
     * anything else is refused with EGP_E_INVALID. All index arrays are device int64; gathered sources must stay below
     * 2^30 elements (a_src_rows * lda, b_src_rows * ldb).
     *   a_rows / A2 / a_split (A k-contiguous): row m of the operand is A[a_rows[m]][k] for k < a_split and
     *       A2[m][k - a_split] for k >= a_split (a_split a multiple of 32, K - a_split >= 32, no split-K; A2 = NULL: every k from A).
     *   b_krows / B2 / b_split (B given as [k][n]): element (k, n) is B[b_krows[k]][n] for n < b_split and
     *       B2[k][n - b_split] for n >= b_split (b_split a multiple of 128; B2 = NULL: every n from B).
     *   c_rows: result row m goes to row c_rows[m] of C (rows must not repeat; rows nobody writes keep their content).
     *   a_krows (A given as [k][m]): k-row k of the operand is A[a_krows[k]] (a weight gradient over a subset of the rows). */
    const int64_t *a_rows; const float *A2; int64_t lda2; int32_t a_split; int64_t a_src_rows;
    const int64_t *b_krows; const float *B2; int64_t ldb2; int32_t b_split; int64_t b_src_rows;
    const int64_t *c_rows;
    const int64_t *a_krows;
} egp_gemm_desc;
/* ------------------------------------------------------------------------------------ update: losses and the optimizer step
 * (csrc/egp_update.hip) The element-wise tail of a PPO epoch in two launches instead of ~45 small library kernels.
 *
 * egp_ppo_loss_f32 -- the critic's MSE loss (agents/agent_pg.py:19-26: (values_pred - returns).pow(2).mean()) and the clipped
 * surrogate (agents/agent_ppo.py:58-65 over core/distributions.py:6-25 / utils/math.py:14-17):
 *     logp_i   = sum_j -(a_ij - mean_ij)^2 / (2 var_j) - 0.5 log(2 pi) - log_std_j
 *     ratio_i  = exp(logp_i - fixed_logp_i);  surr = min(ratio_i A_i, clamp(ratio_i, 1 - eps, 1 + eps) A_i)
 *     losses[0] = inv_n_val * sum_i (pred_i - returns_i)^2,  losses[1] = -inv_n_exp * sum_i surr_i
 * together with their gradients w.r.t. the nets' outputs -- d_pred = 2 inv_n_val (pred - returns), d_mean (through
 * ratio -> logp -> mean, with torch.min / torch.clamp's sub-gradients: ties share, clamp passes inside [1 - eps, 1 + eps]),
 * d_log_std (only when the policy learns its standard deviation) -- so the caller back-propagates from (pred, mean) directly.
 * inv_n_val / inv_n_exp carry the GLOBAL counts (every rank divides by them; the gradient all-reduce sums).
 * `rows` (or NULL): policy row i is sample rows[i] of actions / adv (the rows with exps == 1, agent_ppo.py:45-46); mean,
 * fixed_logp and d_mean are dense over the n_pol policy rows. write_fixed = 1: this is the pass that defines the sampling
 * policy's log-probabilities -- logp is written to fixed_logp and used (ratio == 1), what get_log_prob under no_grad
 * (agent_ppo.py:40-42) followed by epoch 0 computes. Sums in float64, fixed order (deterministic). All pointers device memory. */
typedef struct egp_ppo_loss_desc {
    int32_t n, n_pol, act_dim;
    const int64_t *rows;
    const float *pred; const float *returns;
    const float *mean; int64_t ld_mean;
    const float *actions; int64_t ld_act;
    const float *log_std;
    const float *adv;
    float *fixed_logp; int32_t write_fixed;
    double clip_eps, inv_n_val, inv_n_exp;
    float *d_pred;
    float *d_mean; int64_t ld_dmean;
    float *d_log_std;
    double *losses;
    void *workspace;
} egp_ppo_loss_desc;
int64_t egp_ppo_loss_workspace_bytes(int32_t n, int32_t n_pol, int32_t act_dim);
int egp_ppo_loss_f32(const egp_ppo_loss_desc *desc, void *stream);

/* egp_adam_step_* -- gradient-norm clip + Adam over FLAT parameter buffers: clip_policy_grad (agents/agent_ppo.py:53-56,
 * torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g||_2 + 1e-6))) followed by optimizer_value.step() /
 * optimizer_policy.step() (agent_ppo.py:24-30; torch.optim.Adam, no amsgrad) in two launches for all parameter sets.
 * A segment = one optimizer parameter group laid out contiguously in the flat buffers:
 *     g = clip_coef * grad (+ weight_decay * p);  m = m + (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g^2
 *     p -= (lr / bias1) * m / (sqrt(v) / sqrt(bias2) + eps)        bias1 = 1 - beta1^t, bias2 = 1 - beta2^t (host-computed)
 * Segments with the same clip_group > 0 share one norm (the reference clips policy_net + policy_vs_net together); 0 = no clip.
 * _f32: float32 parameters / moments / gradients. _f64: float64 MASTER parameters and moments, float32 gradients (of the
 * float32 compute copies), and `shadow` (or NULL) receives the stepped parameters rounded to float32 -- the compute copy is
 * refreshed by the same launch. _f64g: float64 throughout. */
typedef struct egp_adam_segment {
    int64_t begin, end;
    double lr, beta1, beta2, eps, weight_decay, bias1, bias2, max_norm;
    int32_t clip_group;
} egp_adam_segment;
#define EGP_ADAM_MAX_SEGMENTS 8
int64_t egp_adam_workspace_bytes(void);
int egp_adam_step_f32(int32_t n_seg, const egp_adam_segment *seg, const float *grad, float *param, float *exp_avg, float *exp_avg_sq,
                      void *workspace, double *norms_out, void *stream);
int egp_adam_step_f64(int32_t n_seg, const egp_adam_segment *seg, const float *grad, double *param, double *exp_avg, double *exp_avg_sq,
                      float *shadow, void *workspace, double *norms_out, void *stream);
int egp_adam_step_f64g(int32_t n_seg, const egp_adam_segment *seg, const double *grad, double *param, double *exp_avg, double *exp_avg_sq,
                       void *workspace, double *norms_out, void *stream);

/* The update's policy / value input in one pass (VideoStateNet.forward('train'), models/video_state_net.py:65-69):
 * out[i] = [ ctx[idx[i]][0:H] | x[i][0:S] ] for i < n, and the adjoint for the context rows, dctx[idx[i]][0:H] = dout[i][0:H]
 * (idx must not repeat: every (time, episode) row of the video net's output belongs to one sample; dctx is zero-filled by
 * the caller). Row strides in elements. */
int egp_gather_concat_f32(const float *ctx, int64_t ld_ctx, const int64_t *idx, const float *x, int64_t ldx, int32_t n, int32_t H, int32_t S,
                          float *out, int64_t ldo, void *stream);
int egp_scatter_rows_f32(const float *dout, int64_t ldd, const int64_t *idx, int32_t n, int32_t H, float *dctx, int64_t ld_ctx, void *stream);
int64_t egp_gemm_workspace_floats(int32_t M, int32_t N, int32_t ones_col, int32_t splits);
int egp_gemm_f32(const egp_gemm_desc *desc, void *stream);

/* ---------------------------------------------------------------------------------------- LSTM
 * Recurrent sweep of ONE direction of the video-context LSTM (hidden size 64 or 128, float32, zero initial state):
 * the t-loop of RNN.batch_forward (models/rnn.py:45-61) in one launch.
 *   gates_x [T][B][4H] = x_t W_ih^T + b_ih + b_hh;  w_hh [4H][H] (torch.nn.LSTMCell order: rows i, f, g, o)
 *   h_out [T][B][H];  gates_save [T][B][4H] (may alias gates_x) / cells_save [T][B][H]: both NULL for inference
 * gates_x, gates_save and d_pre use the layout egp_lstm_gate_layout() reports: EGP_LSTM_GATES_TORCH = [gate][unit] as
 * torch (column g*H + u), EGP_LSTM_GATES_UNIT_MAJOR = [unit][gate] (column 4*u + g; what the matrix-core kernels use:
 * the four gates of a unit are one 16-byte access). The caller permutes the rows of W_ih and of the bias accordingly
 * (4H x D, once per call) and un-permutes the rows of the weight gradient it forms from d_pre. */
#define EGP_LSTM_GATES_TORCH 0
#define EGP_LSTM_GATES_UNIT_MAJOR 1
int32_t egp_lstm_gate_layout(void);
int egp_lstm_fwd_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t reverse,
                     float *h_out, float *gates_save, float *cells_save, void *stream);
/* backward-through-time of the same sweep: d_pre [T][B][256] = gradient w.r.t. the pre-activation gates; the
 * caller forms dW_ih = d_pre^T X, dW_hh = d_pre^T H_prev, db = sum d_pre with library GEMMs */
int egp_lstm_bwd_f32(const float *dh_out, const float *gates_save, const float *cells_save, const float *w_hh,
                     int32_t T, int32_t B, int32_t hidden, int32_t reverse, float *d_pre, void *stream);
/* n_problems (1..4) sweeps over the same (T, B) in ONE launch -- the two directions of a bi-LSTM and/or the LSTMs of several
 * nets reading the same windows (a sweep is latency-bound: grouped problems fill the CUs a single one leaves idle).
 * Needs the unit-major layout (egp_lstm_gate_layout() == EGP_LSTM_GATES_UNIT_MAJOR).
 *   gates_x / gates_save / d_pre [T*B][n_problems*4H]: columns p*4H .. (p+1)*4H-1 belong to problem p (what one
 *   projection GEMM against the row-stacked W_ih of all problems produces);  w_hh [n_problems][4H][H];
 *   reverse_mask bit p: problem p runs backwards in time;  h_out[p] -> problem p's hidden states, row (t, b) at
 *   h_out[p] + (t*B + b)*ld_h (ld_h = H for a dense [T][B][H]; 2H when the two directions of a bi-LSTM fill the halves of
 *   one [T][B][2H] buffer);  cells_save [n_problems][T][B][H];  dh_out[p] / ld_dh likewise;  d_bias (optional)
 *   [n_problems][4H], zeroed by the caller, receives sum over (t, b) of d_pre (float atomics). */
int egp_lstm_group_fwd_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t n_problems,
                           int32_t reverse_mask, float *const *h_out, int32_t ld_h, float *gates_save, float *cells_save, void *stream);
int egp_lstm_group_bwd_f32(const float *const *dh_out, int32_t ld_dh, const float *gates_save, const float *cells_save, const float *w_hh,
                           int32_t T, int32_t B, int32_t hidden, int32_t n_problems, int32_t reverse_mask, float *d_pre, float *d_bias,
                           void *stream);

/* The same for a RAGGED batch (the update's video contexts: episodes of 1..200 steps padded to one window length, as
 * VideoStateNet.initialize('train') pads them, models/video_state_net.py:40-59). seq_order[p] (device, B entries, a
 * permutation of 0..B-1): the sequence at position p -- workgroups take consecutive positions, sort them by length;
 * seq_steps[p]: the time steps the sequence at position p needs counted from t = 0 (its outputs at t >= seq_steps[p] are
 * never read and receive no gradient). Problems that run forward in time stop there (h_out / d_pre of the skipped steps are
 * written as zeros; with leave_skipped != 0 only up to the longest sequence of the ALIGNED GROUP OF 8 POSITIONS and not at
 * all beyond: the sweeps move 3.6-3.9 TB/s, and a caller whose products visit only the rows t < that maximum of each group
 * of 8 -- row lists, egp_gemm_desc.a_rows / a_krows -- saves a tenth of their bytes);
 * problems that run backward in time go through all T steps. NULL / NULL = the plain functions.
 * seq_base (forward, optional; device, B entries): the sequences are windows of consecutive FRAMES of one table -- x[t][b] = frame
 * seq_base[b] + t, as VideoStateNet's windows cnn_feat[take][start - m : start + len + m] are (models/video_state_net.py:52-55) --
 * and gates_x is the projection of that table, [frames][P * 4H]: row seq_base[b] + t is read for (t, b). The projection then runs
 * once per unique frame instead of once per window row. gates_save must be a buffer of its own ([T * B][P * 4H]) in that case. */
int egp_lstm_group_fwd_len_f32(const float *gates_x, const float *w_hh, int32_t T, int32_t B, int32_t hidden, int32_t n_problems,
                               int32_t reverse_mask, float *const *h_out, int32_t ld_h, float *gates_save, float *cells_save,
                               const int32_t *seq_order, const int32_t *seq_steps, int32_t leave_skipped, const int32_t *seq_base,
                               void *stream);
int egp_lstm_group_bwd_len_f32(const float *const *dh_out, int32_t ld_dh, const float *gates_save, const float *cells_save, const float *w_hh,
                               int32_t T, int32_t B, int32_t hidden, int32_t n_problems, int32_t reverse_mask, float *d_pre, float *d_bias,
                               const int32_t *seq_order, const int32_t *seq_steps, int32_t leave_skipped, void *stream);

/* ----------------------------------------------------------------------------------------
 * K8: dynamics terms on the GPU. From (qpos, qvel) per env: body frame positions (mjData.xpos[1:]),
 * the joint-space inertia in MuJoCo's legacy sparse order (mjData.qM, what mj_fullM expands) and the bias force
 * (mjData.qfrc_bias) -- the mjData fields compute_torque / get_ee_pos read (ego_pose/envs/humanoid_v1.py:98-111,130-144).
 * Conventions are MuJoCo's (root angular velocity in the body frame, MJCF coordinate="global", armature on the hinge
 * diagonal). Tables are HOST arrays describing the zero pose in global coordinates; bodies parents-first, body 0 = root.
 * qM / qfrc_bias / xpos may each be NULL. MuJoCo is not available to pin against: see oracle/dynamics.py. */
typedef struct egp_dynamics_desc {
    int32_t nbody, njoint;
    const int32_t *body_parent;   /* [nbody], -1 for the root */
    const double *body_pos;       /* [nbody*3] */
    const double *body_com;       /* [nbody*3] */
    const double *body_inertia;   /* [nbody*9] about the COM, global axes */
    const double *body_mass;      /* [nbody] */
    const int32_t *body_ndof;     /* [nbody] hinges per body (entry 0 ignored: free joint) */
    const double *joint_axis;     /* [njoint*3] */
    const double *joint_anchor;   /* [njoint*3] */
    double armature;
    double gravity[3];
} egp_dynamics_desc;
int egp_set_dynamics_model(egp_ctx *ctx, const egp_dynamics_desc *desc);
int egp_dynamics_f64(egp_ctx *ctx, const double *qpos, const double *qvel, int32_t n, double *qM, int64_t ld_m,
                     double *qfrc_bias, double *xpos, void *stream);


/* ----------------------------------------------------------------------------------------
 * Rollout-time policy step in one launch (replaces, for all envs of a group at once, the chain
 * VideoStateNet.forward concat (models/video_state_net.py:37-43) -> MLP (models/mlp.py:5-25) ->
 * PolicyGaussian.forward / select_action (models/policy_gaussian.py:19-27, core/agent.py:38-44)):
 *   x[r] = [ctx_rows[r*ctx_row_stride + t_idx[r]*ctx_dim ...] (float32) | state[r] (float64 -> float32)]
 *   hidden layers with `activation` (0 tanh, 1 relu, 2 sigmoid), last layer = action_mean (no activation)
 *   action[r] = mean + exp(log_std) * noise[r]   (noise == NULL: action = mean), float32 arithmetic, stored float64
 * The layer products run on the matrix cores in exact float32 (v_mfma_f32_4x4x1_16b_f32).
 * `layers[l].wt` is the weight in the kernel's PACKED form (egp_mlp_pack_f32, 16-byte aligned): for each group g of 64 output
 * columns and each quad kq of input features one block of 64 x 4 floats, packed[((g * ceil(in/4) + kq) * 64 + c) * 4 + kk] =
 * W[64 g + c][4 kq + kk] (zero outside the matrix) -- a wave loads a block with one coalesced 16-byte load per lane.
 * mean_out (float32) may be NULL. */
typedef struct egp_mlp_layer {
    const float *wt;
    const float *bias;
    int32_t in_dim, out_dim;
} egp_mlp_layer;
/* nn.Linear weight W[out_dim][in_dim] (device, row stride ldw floats) -> packed (device, egp_mlp_pack_floats(in_dim, out_dim) floats) */
int64_t egp_mlp_pack_floats(int32_t in_dim, int32_t out_dim);
int egp_mlp_pack_f32(const float *weight, int64_t ldw, int32_t in_dim, int32_t out_dim, float *packed, void *stream);
int egp_policy_gaussian_f32(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                            const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                            int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                            double *action, float *mean_out, void *stream);
/* The same launch also moving a staging slab: `stage_bytes` (a multiple of 4) bytes from `stage_src` -- device-visible pinned
 * host memory, `t_idx` may point into it -- to `stage_dst` in HBM, spread over the launch's workgroups. The rollout's tick
 * uses it for its per-tick flag slab (egp_rollout_tick.slab_host / slab_dev) instead of a copy-engine transfer in front of
 * the policy step; kernels launched later on the stream read `stage_dst`. */
int egp_policy_gaussian_staged_f32(const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                   const double *state, int32_t state_dim, int32_t n, const egp_mlp_layer *layers,
                                   int32_t n_layers, int32_t activation, const float *log_std, const float *noise,
                                   double *action, float *mean_out, const void *stage_src, void *stage_dst, int64_t stage_bytes,
                                   void *stream);
/* ... with the observation filter's apply pass in front (agents/agent.py:50-51 followed by :42-46 of the next step: the filtered
 * next state IS the policy's next input): the state columns of the policy input are the observations of (qpos, qvel) -- the
 * group's n rows, n <= egp_obs_zfilter_split_max_rows() -- normalised with `zf_in` merged with the tile statistics that
 * egp_obs_zfilter_stats_f64 left in `zf_workspace`; they are written to y (and y2) and the merged statistics to zf_out. One launch
 * computes exactly what egp_obs_zfilter_apply_f64 followed by egp_policy_gaussian_staged_f32 on y2 computes (bit-identical). */
int egp_policy_gaussian_filter_f32(egp_ctx *ctx, const float *ctx_rows, int64_t ctx_row_stride, int32_t ctx_dim, const int64_t *t_idx,
                                   const double *qpos, const double *qvel, const int32_t *phase_t, int32_t n, const double *zf_in, double *zf_out,
                                   double clip, double *y, double *y2, const void *zf_workspace,
                                   const egp_mlp_layer *layers, int32_t n_layers, int32_t activation, const float *log_std,
                                   const float *noise, double *action, float *mean_out, const void *stage_src, void *stage_dst,
                                   int64_t stage_bytes, void *stream);

/* ----------------------------------------------------------------------------------------
 * Host physics boundary (replaces mujoco_py's MjSim inside HumanoidEnv: envs/common/mujoco_env.py:84-105,
 * ego_pose/envs/humanoid_v1.py:158-177). A backend is a vtable of plain C callbacks working on one
 * env at a time; all buffers are HOST memory owned by the engine.
 *   reset(user, env, qpos[nq], qvel[nv])                      set_state + forward
 *   step (user, env, ctrl[nu])                                data.ctrl = ctrl; mj_step
 *   drain(user, env, qpos, qvel, qM[nM], qfrc_bias[nv], xpos[nbody*3])   copy out mjData fields (qM / xpos may be NULL)
 * Threading: the engine calls reset / step / drain of DIFFERENT envs concurrently from its host threads (an env is only ever
 * touched by one thread at a time): a backend keeps per-env state apart (MuJoCo: one mjData per env over a shared mjModel).
 * A MuJoCo adapter fills this from mj_step / mjData; the built-in surrogate is egp_physics_create_surrogate. */
typedef struct egp_physics_vtable {
    void *user;
    int (*reset)(void *user, int32_t env, const double *qpos, const double *qvel);
    int (*step)(void *user, int32_t env, const double *ctrl);
    int (*drain)(void *user, int32_t env, double *qpos, double *qvel, double *qM, double *qfrc_bias,
                 double *xpos);
    void (*destroy)(void *user);
    const char *name;
    /* optional (may be NULL): a counter that changes whenever env's qM changed since it was last drained.
     * The engine re-uploads an env group's inertia rows only when some epoch in the group moved; a MuJoCo
     * adapter returns a per-step counter (M depends on qpos), the surrogate a constant (its M0 is fixed).
     * `drain` is then called with qM == NULL for unchanged envs. */
    int64_t (*inertia_epoch)(void *user, int32_t env);
} egp_physics_vtable;

int egp_physics_register(const egp_physics_vtable *vt, int32_t n_env, egp_physics **out);
/* Deterministic surrogate: semi-implicit Euler on qacc = M0^-1 (tau - C), fixed tree-sparse SPD M0
 * (host arrays: qM0[nM], Minv0[nv*nv], body tree for FK). NOT MuJoCo; physics parity is unpinned. */
typedef struct egp_surrogate_desc {
    int32_t nq, nv, nu, nbody, nM, njoint;
    const double *qM0;            /* [nM] */
    const double *Minv0;          /* [nv*nv] */
    const int32_t *body_parent;   /* [nbody] */
    const double *body_pos;       /* [nbody*3] */
    const int32_t *body_ndof;     /* [nbody] */
    const double *joint_axis;     /* [njoint*3] */
    const double *joint_anchor;   /* [njoint*3] */
    double sub_dt;
    double damping;               /* viscous joint damping in the bias force */
    double support_k, support_c;  /* vertical root support spring / damper (stands in for contacts) */
} egp_surrogate_desc;
int egp_physics_create_surrogate(const egp_surrogate_desc *desc, int32_t n_env, egp_physics **out);
int egp_physics_destroy(egp_physics *p);
const char *egp_physics_name(const egp_physics *p);
/* single-env host access (used by the CPU baseline and tests) */
int egp_physics_reset_host(egp_physics *p, int32_t env, const double *qpos_host, const double *qvel_host);
int egp_physics_step_host(egp_physics *p, int32_t env, const double *ctrl_host);
int egp_physics_drain_host(egp_physics *p, int32_t env, double *qpos_host, double *qvel_host,
                           double *qM_host, double *qfrc_bias_host, double *xpos_host);

/* ----------------------------------------------------------------------------------------
 * Lockstep rollout engine: n_env envs advance one env-step (frame_skip substeps of
 * {K1 on the GPU <-> physics on host threads}) per egp_engine_step. Replaces the 15x
 * compute_torque/sim.step loop of do_simulation (ego_pose/envs/humanoid_v1.py:158-177) for all
 * envs at once. State, inertia and torque rows live in pinned host memory the kernels address in place (zero-copy). Two forms of the
 * env-step: RESIDENT (default: one K1 launch serves all frame_skip substeps, trading go words / torque rows with the physics threads;
 * taken when every workgroup of every group fits on the chip at once) and PER-SUBSTEP (one K1 launch per substep; the fallback for other
 * dof trees / K1 variants, more slots than the chip holds resident workgroups for, or EGP_SERVER=0). */
typedef struct egp_engine_desc {
    int32_t n_env;
    int32_t n_threads;       /* host physics worker threads (reference: --num-threads samplers) */
    int32_t n_groups;        /* env groups that can be stepped independently (policy/physics overlap) */
    int32_t device_dynamics; /* 1: qM and qfrc_bias of every substep come from K8 (egp_set_dynamics_model must have been
                              * called) instead of the backend's drain -- `drain` is then always called with qM == NULL and its
                              * qfrc_bias is ignored. Timing as in the reference (ego_pose/envs/humanoid_v1.py:130-144 reads
                              * data.qM / data.qfrc_bias as the previous mj_step left them): a substep's torque is solved with
                              * M, C of the state the PREVIOUS substep started from; egp_engine_reset evaluates them at the reset
                              * state (sim.forward(), envs/common/mujoco_env.py:97-101) */
} egp_engine_desc;

int egp_engine_create(egp_ctx *ctx, egp_physics *phys, const egp_engine_desc *desc, egp_engine **out);
int egp_engine_destroy(egp_engine *e);
/* state of all envs after reset/wait, env-major float64. Device: qpos[n][nq], qvel[n][nv], ee_wpos[n][15]
 * (world positions of the 5 end effectors, data.body_xpos rows). Pinned host mirrors: head_z[n]
 * (get_body_com('Head')[2], humanoid_v1.py:191-196), qpos/qvel. prev_qpos[n][nq] (device) is qpos as it was when the
 * current/last env-step started (env.prev_qpos, humanoid_v1.py:182). Any out-pointer may be NULL. */
int egp_engine_state(egp_engine *e, double **qpos, double **qvel, double **ee_wpos, double **head_z_host,
                     double **qpos_host, double **qvel_host, double **prev_qpos);
/* HumanoidEnv.reset_model's set_state (+ sim.forward) for the listed envs (strictly increasing ids):
 * rows qpos_host[k][nq], qvel_host[k][nv]; their drained state is uploaded on `stream`. */
int egp_engine_reset(egp_engine *e, const int32_t *env_ids_host, int32_t n, const double *qpos_host,
                     const double *qvel_host, void *stream);
/* start one env-step (frame_skip substeps) for group g: `action` is a DEVICE pointer [n_env][nu];
 * `active_host` [n_env] (optional) marks the envs to step; `ready_event` (optional hipEvent_t) is waited
 * on by every worker stream before its first K1 launch (action produced on another stream; the event must also cover the
 * caller's last egp_engine_reset, i.e. be recorded on that reset's stream after it). Without an event the env-step orders
 * itself behind the last reset's upload and nothing else. */
int egp_engine_step_async(egp_engine *e, int32_t group, const double *action, const int32_t *active_host,
                          void *ready_event);
/* block until group g finished; makes `stream` wait for the final uploads so kernels enqueued on it
 * afterwards see the new device state */
int egp_engine_wait(egp_engine *e, int32_t group, void *stream);
/* accumulated since creation / egp_engine_reset_timing, summed over workers: host physics seconds,
 * seconds blocked on the GPU round trip, and (when profiling) K1 time by HIP events on the launch
 * streams (empty-bracket overhead subtracted) + number of K1 launches */
int egp_engine_timing(egp_engine *e, double *phys_s, double *gpu_wait_s, double *k1_ms_events, int64_t *k1_launches);
int egp_engine_reset_timing(egp_engine *e);
/* env-substeps (stepped envs x frame_skip) served by the event-bracketed K1 launches counted in egp_engine_timing:
 * finished slots of a rollout's tail are not stepped and do not count (bench.py's roofline is quoted on this) */
int64_t egp_engine_k1_env_substeps(egp_engine *e);
int64_t egp_engine_inertia_uploads(egp_engine *e);
double egp_engine_event_overhead_ms(egp_engine *e);   /* calibrated cost of an empty begin/end event pair, already subtracted from k1_ms_events */   /* group-level qM uploads done inside step (not resets) */
int egp_engine_set_profile(egp_engine *e, int on);   /* 0 off; 1: HIP events around every K1 launch; N>1: on every Nth env-step */
int egp_engine_layout(egp_engine *e, int32_t *pack_ld, int32_t *n_env, int32_t *n_threads, int32_t *n_groups);
int egp_engine_group_range(egp_engine *e, int32_t group, int32_t *env_begin, int32_t *env_end);
/* ----------------------------------------------------------------------------------------
 * One tick of the lockstep sampler (Agent.sample_worker's loop body, core/agent.py:31-66, for every slot of an env group
 * at once) on the native side: the per-slot bookkeeping and the calls the rollout driver otherwise makes one by one from
 * Python (flags of the coming env-step, their upload, the fused policy step, the reward job, the env-step; then the wait,
 * K3 + K6, K2 and the termination flags of HumanoidEnv.step, ego_pose/envs/humanoid_v1.py:182-199). The driver fills the
 * descriptor once per rollout; the arrays are its own (NumPy / device tensors) and stay valid for the rollout.
 *   the tick's flag slab (pinned) reaches its device copy inside the policy step (egp_policy_gaussian_staged_f32 / _filter_f32)
 *   pre : flags / context rows of tick k for slots [a, b) -> policy -> env-step (asynchronous)
 *   post: wait for the env-step, observation + filter into states[k + 1] / next_states[k], reward, cur_t / done / record rows;
 *         *n_done = slots whose episode ended, *wait_s = seconds blocked on the env-step */
typedef struct egp_rollout_tick {
    egp_ctx *ctx; egp_engine *eng; void *stream;
    int32_t n_env, nmax, obs_dim, nu, nq, nv, ctx_dim, ctx_T, episode_len, reward_job, has_fix_head_lb;
    double end_reward, zf_clip, fix_head_lb;
    /* host state of the env slots */
    int64_t *cur_t, *frame_base, *e_ind, *s_ind, *steps_done;
    uint8_t *active; int32_t *active_i32;
    const double *head_z, *head_lb;
    /* host record of the rollout, [T_max][n_env] */
    uint8_t *rec_valid, *rec_done; int64_t *rec_e_ind, *rec_s_ind;
    /* device record, [T_max (+ 1)][n_env][...] */
    double *states, *next_states, *actions, *rewards, *cinfo;
    const float *noise;                          /* [T_max][n_env][nu], NULL: mean action */
    const float *v_out; int64_t v_stride;        /* per-slot context rows of the video net */
    const egp_mlp_layer *layers; int32_t n_layers, activation; const float *log_std;
    uint8_t *slab_host, *slab_dev;               /* [n_groups][2][24 * nmax] flags + context-row slabs (pinned / device) */
    const double *qpos, *qvel, *prev_qpos, *ee;  /* the engine's device state */
    void *zf_workspace;
    int32_t *reset_scratch;                      /* [n_groups][2][3 * nmax] pinned, device-visible: ids | group mask | cur_t of egp_rollout_reset */
    int32_t defer_apply;                         /* 1 (needs a filter, groups of at most egp_obs_zfilter_split_max_rows() slots):
                                                  * `post` runs the filter's statistics pass only; its apply pass rides in the next tick's policy
                                                  * step (egp_rollout_tick_pre with apply_pending = 1 and the same zf_cur / zf_new) or, in a tick with
                                                  * in-batch resets and in a group's last tick, is run by egp_rollout_tick_apply */
} egp_rollout_tick;
int egp_rollout_tick_pre(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, void *ready_event,
                         int32_t apply_pending, const double *zf_cur, double *zf_new);
int egp_rollout_tick_apply(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const double *zf_cur, double *zf_new);
int egp_rollout_tick_post(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const double *zf_cur, double *zf_new,
                          int32_t *n_done, double *wait_s);
/* HumanoidEnv.reset_model + the first observation of the new episodes (ego_pose/envs/humanoid_v1.py:201-226, core/agent.py:35-38)
 * for the slots `ids` (n of them, strictly increasing, all inside [a, b)) whose episode ended in tick k, called right behind
 * egp_rollout_tick_post: physics reset to (qpos, qvel) rows [n][nq] / [n][nv] (host), slot bookkeeping (take, start frame, expert
 * row of frame 0, cur_t = cur_t0[j] -- NULL: 0; cfg.random_cur_t, humanoid_v1.py:218-220: the state rows then are those of frame
 * start + cur_t0), the slots' video-context rows `ctx_rows` (device, [n][ctx_T][ctx_dim] float32) into v_out, and K3 + K6
 * over the group with only those slots active: their filtered observation replaces states[k + 1] (the running filter advances
 * zf_cur -> zf_new exactly as one more egp_obs_zfilter_f64 call). Uses reset_scratch slot k & 1 of the group. */
int egp_rollout_reset(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const int32_t *ids, int32_t n,
                      const int64_t *e_ind, const int64_t *s_ind, const int64_t *frame_rows, const int64_t *cur_t0, const double *qpos, const double *qvel,
                      const float *ctx_rows, const double *zf_cur, double *zf_new);
/* the stream group g's env-step kernels are launched on (owned by the engine) */
void *egp_engine_group_stream(egp_engine *e, int32_t group);

/* Resident-K1 mode only: hand the NEXT egp_engine_step_async of `group` its reward launch. The flag arrays (device
 * memory, group-local: t / frame / end / active of the state the step will produce) must be ready by the step's
 * ready_event; the engine launches egp_reward_quat_v3_f64 on the group's stream right behind the env-step's kernel (inputs:
 * its own qpos / prev_qpos / ee_wpos rows), i.e. ordered before the next env-step and off the caller's stream. The caller
 * must synchronise with the device before reading `reward` / `cinfo`. */
int egp_engine_set_reward_job(egp_engine *e, int32_t group, const int32_t *t, const int32_t *frame, const int32_t *end,
                              const int32_t *active, double end_reward, double *reward, double *cinfo);
/* diagnostic: `blocks` workgroups of arithmetic for `us` microseconds on `stream` (does a busy GPU clock the rollout's short
 * kernels differently? tools/probes/burn_probe.py); `sink` = any device float */
int egp_debug_burn(int64_t us, int32_t blocks, float *sink, void *stream);
/* substeps one K1 launch serves: frame_skip when the engine runs the resident K1 (one launch per env-step that
 * trades go/done words with the physics threads through pinned memory; EGP_SERVER=0 turns it off), else 1 */
int egp_engine_substeps_per_launch(egp_engine *e);
/* the resident K1's shape: envs a wavefront serves in turn per substep (1: four envs per workgroup, one per wave; 2 / 4 when the
 * slots do not fit the chip that way -- more than 4 envs per CU, or fewer CUs to be had; EGP_SERVER_KE forces a count) and, in
 * `resident_capacity` (may be NULL), how many workgroups of that kernel the chip holds at once as the kernel's own residency probe
 * counted them at engine creation (the occupancy calculator's figure when EGP_SERVER_PROBE=0). The reference scales its sampler
 * by the number of worker processes (agents/agent.py:93-100); this is the engine's counterpart. 0 = per-substep form. */
int32_t egp_engine_envs_per_wave(egp_engine *e, int32_t *resident_capacity);
/* where the resident K1's go words live: 1 = fine-grained device memory the host writes through the PCIe BAR (large-BAR systems),
 * 0 = pinned host memory the waves poll over PCIe, -1 = the engine does not run the resident K1 */
int32_t egp_engine_go_words_in_vram(egp_engine *e);
/* diagnostics (engine created with EGP_SERVER_TRACE=1): the last env-step of `group` as seen by block 0 of the
 * resident K1 -- device_ticks[frame_skip*8], 100 MHz wall_clock64 stamps per substep: poll start, go seen, state
 * loaded, solved, torques stored -- and by the owner of slice 0 -- host_us[frame_skip*4], microseconds since the
 * step was posted: wait start, first torque row in, go written */
int egp_engine_server_trace(egp_engine *e, int32_t group, int64_t *device_ticks, double *host_us);
/* Host probe (egp_probe.hip): ~2 s of measurements of what the rollout's env-step depends on OUTSIDE this library's code -- the box.
 * The env-step is a latency chain through the host (per substep: go word host -> GPU, state rows GPU <- pinned memory over PCIe,
 * torque rows GPU -> pinned memory; host threads spinning on those rows): the reference has no such path (its sampler is
 * CPU-only, agents/agent.py:29-111), so this has no counterpart there; bench.py reports the numbers next to T_sample (the
 * reference prints T_sample / T_update per iteration, ego_pose/ego_mimic.py:115-126) so that a slow run can be told from a slow box.
 *   pcie_read_*   1 024 pinned state rows (176 doubles) read by one wave per row with the resident K1's access shape
 *   go_rtt_*      microseconds from the host's go-word store to the arrival of the 52-double row a resident wave writes back after
 *                 reading one state row (one substep's round trip without the solve); go_in_vram: the go word lived in fine-grained
 *                 device memory behind a large BAR, as the engine's do by default
 *   spin_*        n_threads threads spin on the clock for `millis` ms: gaps between two reads = time a spinning thread was not running */
typedef struct egp_host_probe_result {
    double pcie_read_gbps, pcie_read_us_per_pass;
    double go_rtt_us_p50, go_rtt_us_p99, go_rtt_us_max;
    double spin_gap_us_max, spin_gap_us_median_of_thread_max, spin_lost_frac;
    int32_t pcie_read_rows, go_rtt_n, go_in_vram, large_bar, spin_threads, spin_gaps_over_5us;
} egp_host_probe_result;
int egp_host_probe(int32_t device, int32_t n_threads, int32_t millis, egp_host_probe_result *out);
/* CUs of `device` this process really gets: workgroups that each take more than half a CU's LDS are launched one per reported CU and
 * count how many are on the chip at once (cached per device; EGP_SERVER_PROBE=0: the attribute's figure). The attribute
 * multiProcessorCount is the data sheet's answer -- under HSA_CU_MASK it stays 256, under ROC_GLOBAL_CU_MASK it says 240 where 225
 * fit --, and kernels that launch ONE persistent workgroup per CU (the update's products, gemm.pick_splits) would run a second round
 * for the missing ones. The reference has no counterpart (it sizes its sampler by --num-threads, agents/agent.py:93-100). */
int32_t egp_device_usable_cus(int32_t device);
int32_t egp_physics_n_env(const egp_physics *p);

#ifdef __cplusplus
}
#endif
#endif /* EGOPOSE_HIP_H */
