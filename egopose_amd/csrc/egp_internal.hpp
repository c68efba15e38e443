// Internal declarations shared by the kernel, physics and engine translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/egopose_hip.h"

namespace egp {

void set_error(const char *fmt, ...);
#define EGP_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            egp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return EGP_E_HIP;                                                                \
        }                                                                                    \
    } while (0)

#define EGP_REQUIRE(cond, msg)                       \
    do {                                             \
        if (!(cond)) {                               \
            egp::set_error("invalid argument: %s", msg); \
            return EGP_E_INVALID;                    \
        }                                            \
    } while (0)

// Reward coefficients, passed to the kernel by value (SGPR-resident).
struct RewardW {
    double w_p, w_v, w_e, w_rp, w_rv, w_sum;
    double k_p, k_v, k_e, k_rh, k_rq, k_rl, k_ra;
    double v_ord;
    int decay;
    int episode_len;
};

// Device-side view of the model: small read-only tables the kernels stage in LDS.
struct DevModel {
    int nq, nv, nu, nbody, nM;
    const int *body_qpos_start;   // [nbody]
    const int *body_ndof;         // [nbody]
    const int *m_row;             // [nM] row of sparse-inertia entry
    const int *m_col;             // [nM] col
    const short *m_map;           // [nv*nv] dense (i,j) -> index into qM, or -1
    const double *jkp, *jkd, *a_ref, *a_scale, *torque_lim;   // [nu]
    const double *b_diffw;        // [nbody-1]
    double sub_dt;                // model timestep
    double dt;                    // env step = frame_skip * sub_dt
    int obs_heading, obs_keep, obs_root, obs_vel;   // observation variants (egp_model_desc), all 0 = the shipped configs
    int obs_phase, episode_len;   // cfg.obs_phase: a last column min(cur_t / episode_len, 1)
    int obs_dim;                  // width of an observation row
    int action_torque;            // cfg.action_type == 'torque': torque = clip(a_ref + action * a_scale), K1 solves nothing
};

}  // namespace egp

struct egp_ctx {
    int device = 0;
    egp::DevModel dm{};
    egp::RewardW rw{};
    int frame_skip = 15;
    std::vector<int> ee_body;
    std::vector<void *> allocs;        // every device allocation owned by the ctx
    // expert table in HBM
    int n_takes = 0, n_frames = 0;
    double *expert_rows_f64 = nullptr; // [n_frames][EGP_EXPERT_ROW]
    float *expert_rows_f32 = nullptr;
    double *expert_qpos_f64 = nullptr; // [n_frames][nq]: the pose_dist reward compares whole poses
    int pd_variant = 0;                // 0 = tree-ordered in-register elimination, 2 = dense in-register, 1 = LDS
    bool tree58 = false;               // runtime dof tree == compiled-in humanoid tree
    bool pd_grid = false;              // variant 0, one substep per launch: the lane-grid kernel (variant 3: the row kernel)
    const unsigned short *pd_grid_off = nullptr;   // its [register][lane] gather table (device)
    void *dyn_tables = nullptr;        // device copy of the dynamics tree (egp_set_dynamics_model), owned through allocs
};

// launches used by the engine (same TU as the kernels)
int egp_launch_pd_torque_strided(egp_ctx *ctx, const double *qpos, long ld_qpos, const double *qvel, long ld_qvel,
                                 const double *bias, long ld_bias, const double *qM, long ld_qM, const double *action,
                                 int32_t n, double *torque, hipStream_t stream, unsigned *done_counter,
                                 unsigned long long *host_flag, unsigned long long seq);
int egp_launch_pd_server(egp_ctx *ctx, const double *qpos, long ld_qpos, const double *qvel, long ld_qvel, const double *bias,
                         long ld_bias, const double *qM, long ld_qM, const double *qM_host, const double *action, int32_t n,
                         double *torque, hipStream_t stream, const int *block_slice, const unsigned long long *go,
                         unsigned long long base, int n_sub, int *err, double timeout_s, long long *trace, const double *ee_host,
                         double *out_qpos, double *out_prev_qpos, double *out_qvel, double *out_ee, const int *active, bool device_dynamics,
                         int envs_per_wave = 1, const int *block_env0 = nullptr, int n_blocks = 0);
size_t egp_pd_server_dyn_lds_bytes();
int egp_pd_server_resident_blocks(int device, bool device_dynamics, int envs_per_wave = 1);
int egp_launch_dynamics_strided(egp_ctx *ctx, const double *qpos, long ld_q, const double *qvel, long ld_v, int32_t n, double *qM,
                                long ld_m, double *bias, long ld_b, double *xpos, hipStream_t stream, const int *list = nullptr,
                                int list_stride = 0, double *qM_alt = nullptr, double *bias_alt = nullptr);
// bit pattern the engine pre-fills pinned torque rows with in resident-K1 mode (a quiet NaN no clipped torque can equal)
constexpr unsigned long long EGP_TORQUE_SENTINEL = 0x7FF8DEADBEEF0001ull;
const egp_physics_vtable *egp_physics_vt(const egp_physics *p);
extern "C" int32_t egp_physics_n_env(const egp_physics *p);
