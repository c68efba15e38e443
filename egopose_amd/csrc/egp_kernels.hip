// HIP kernels K1-K6 of the EgoPose rollout+update hot path for gfx950 (MI355X), plus the
// context object and the C-ABI entry points declared in include/egopose_hip.h.
//
// Data layout: every per-env array is env-major and row-contiguous (qpos[n][59], qvel[n][58],
// qM[n][910], ...), which is what the host physics writes and what TrajBatch exposes. Kernels
// map work so that consecutive lanes touch consecutive addresses of one env row (or of adjacent
// rows), and the shared skeleton tables (body->qpos map, dof tree, PD gains) are staged in LDS.
//
//   K1 pd_torque     one 64-lane wavefront per env, one dof per lane; MuJoCo sparse inertia is
//                    expanded (mj_fullM) through LDS, the 58x58 system lives in VGPRs (lane i owns
//                    row i) and is solved by in-register Gauss-Jordan with v_readlane broadcasts.
//   K2 reward        one half-wavefront per env: lane = body (pose / body ang-vel terms), lane 0 =
//                    root terms, 5 lanes = end effectors; width-32 butterfly reduction.
//   K3 obs, K4 body_quat   one thread per output element / body.
//   K5 gae           chunked affine-recurrence scan (3 launches) + Welford statistics.
//   K6 zfilter       per-tile (count, mean, M2) partials + Chan merge + normalise.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>

#include "egp_internal.hpp"
#include "egp_dynamics_dev.hpp"
#include "egp_quat.hpp"
#include "egp_filter_dev.hpp"

namespace egp {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ============================================================================================ K4
// get_body_quat (ego_pose/envs/humanoid_v1.py:113-125)
template <typename T>
__global__ __launch_bounds__(256) void k_body_quat(DevModel m, const T *__restrict__ qpos, int n,
                                                   T *__restrict__ bquat) {
    __shared__ int s_start[EGP_MAX_BODY], s_ndof[EGP_MAX_BODY];
    if (threadIdx.x < m.nbody) {
        s_start[threadIdx.x] = m.body_qpos_start[threadIdx.x];
        s_ndof[threadIdx.x] = m.body_ndof[threadIdx.x];
    }
    __syncthreads();
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)n * m.nbody) return;
    const int env = gid / m.nbody, b = gid % m.nbody;
    const T *q = qpos + (long)env * m.nq;
    Q4<T> o;
    if (b == 0) {
        o.w = q[3]; o.x = q[4]; o.y = q[5]; o.z = q[6];
    } else {
        const int s = s_start[b], nd = s_ndof[b];
        const T e0 = nd > 0 ? q[s] : T(0), e1 = nd > 1 ? q[s + 1] : T(0), e2 = nd > 2 ? q[s + 2] : T(0);
        o = q_from_euler_sxyz<T>(e0, e1, e2);
    }
    T *dst = bquat + gid * 4;
    dst[0] = o.w; dst[1] = o.x; dst[2] = o.y; dst[3] = o.z;
}

// ============================================================================================ K3
// (ObsOpt, obs_element: egp_filter_dev.hpp)

template <typename T>
__global__ __launch_bounds__(256) void k_obs(DevModel m, const T *__restrict__ qpos, const T *__restrict__ qvel,
                                             const int *__restrict__ phase_t, int n, T *__restrict__ obs) {
    const int od = m.obs_dim;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)n * od) return;
    const int env = gid / od, c = gid % od;
    obs[gid] = obs_element<T>(qpos + (long)env * m.nq, qvel + (long)env * m.nv, obs_opt_of(m), c, m.obs_phase ? phase_t[env] : 0);
}

// Large batches (the HBM-resident form): with one thread per element the wave that holds a row's root quaternion columns pays the
// whole de-heading (two float64 divisions, a square root) and the one with the root velocity columns the rotation -- ~600 cycles
// of divergent float64 work per wave for 1 KiB of traffic: 65 536 rows took 47 us at 32 % of the HBM roofline, vector-ALU bound (now 23 us, 65 %; 262 144 rows, beyond the 256 MB cache: 52 %).
// Here a workgroup takes 64 rows. First the columns that need arithmetic (heading, de-headed quaternion, root velocity, phase:
// at most 9): one wave per column, one lane per row, obs_element itself (same bits) into LDS -- every lane of the wave works.
// Then the rows stream through: 128 threads per row, 8-byte loads and stores of consecutive columns, the special ones from LDS.
constexpr int OBS_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void k_obs_rows(DevModel m, const T *__restrict__ qpos, const T *__restrict__ qvel,
                                                  const int *__restrict__ phase_t, int n, T *__restrict__ obs) {
    __shared__ T s_spec[9][OBS_ROWS];
    const ObsOpt o = obs_opt_of(m);
    const int od = m.obs_dim, h = o.heading ? 1 : 0;
    const int n_vel = o.vel == 0 ? o.nv : (o.vel == 1 ? 6 : 0);
    // special column s -> its index in the row (-1: not present)
    auto spec_col = [&](int s) {
        if (s == 0) return o.heading ? 0 : -1;
        if (s <= 4) return o.keep ? -1 : h + s;
        if (s <= 7) return n_vel >= 3 ? h + o.np + (s - 5) : -1;
        return o.phase ? h + o.np + n_vel : -1;
    };
    const long r0 = (long)blockIdx.x * OBS_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < 9; s += 4) {                       // (wave-uniform: one column per wave and round)
        const int c = spec_col(s);
        if (c < 0) continue;
        const long r = min(r0 + lane, (long)n - 1);
        s_spec[s][lane] = obs_element<T>(qpos + r * m.nq, qvel + r * m.nv, o, c, o.phase ? phase_t[r] : 0);
    }
    __syncthreads();
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    if (c >= od) return;
    int sp = -1;
#pragma unroll
    for (int s = 0; s < 9; ++s) sp = spec_col(s) == c ? s : sp;
    const int cc = c - h;
    const bool from_q = cc < o.np;
    const int src = from_q ? cc + 2 : cc - o.np;             // (unused for special columns)
#pragma unroll 4
    for (int e = half; e < OBS_ROWS; e += 2) {
        const long r = r0 + e;
        if (r >= n) break;
        T v;
        if (sp >= 0) v = s_spec[sp][e];
        else v = from_q ? qpos[r * m.nq + src] : qvel[r * m.nv + src];
        obs[r * od + c] = v;
    }
}

// ============================================================================================ K1
// compute_torque / compute_desired_accel (ego_pose/envs/humanoid_v1.py:130-156) + clip (:172)
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// row strides (in elements) of the five K1 inputs: dense C-ABI arrays use (nq, nv, nu, nM, nv); the
// rollout engine passes its packed per-env staging row instead
struct PdLd { long qpos, qvel, action, qM, bias; };
// optional completion signal of a K1 launch (counter in HBM, flag in pinned host memory)
struct PdDone { unsigned *counter; unsigned long long *host_flag; unsigned long long seq; };

template <typename TIO>
__device__ __forceinline__ void pd_rhs(const DevModel &m, const PdLd &ld, const TIO *qpos, const TIO *qvel, const TIO *action,
                                       const TIO *C, long env, int row, double &kp, double &kd, double &eq,
                                       double &qv, double &b) {
    kp = 0.0; kd = 0.0; eq = 0.0;
    if (row >= 6) {
        const int a = row - 6;
        kp = m.jkp[a];
        kd = m.jkd[a];
        const double target = m.a_ref[a] + (double)action[env * ld.action + a] * m.a_scale[a];
        eq = (double)qpos[env * ld.qpos + 7 + a] - target;
    }
    qv = (double)qvel[env * ld.qvel + row];
    b = -(double)C[env * ld.bias + row] - kp * eq - kd * qv;
}

template <typename TIO>
__device__ __forceinline__ void pd_store(const DevModel &m, long env, int row, bool owner, double kp, double kd,
                                         double eq, double qv, double qacc, TIO *torque, TIO *torque_raw) {
    if (owner && row >= 6) {
        const int a = row - 6;
        const double ev = qv + qacc * m.sub_dt;
        const double tau = -kp * eq - kd * ev;
        const double lim = m.torque_lim[a];
        const double tc = fmin(fmax(tau, -lim), lim);
        torque[env * m.nu + a] = (TIO)tc;
        if (torque_raw) torque_raw[env * m.nu + a] = (TIO)tau;
    }
}

// Fast path for the humanoid (nv == 58): 4 envs per 256-thread block, one wavefront each.
constexpr int PD_NV = 58;
constexpr int PD_NM_MAX = 960;   // >= nM (910)

template <typename TIO>
__global__ __launch_bounds__(256) void k_pd_torque_reg58(DevModel m, PdLd ld, const TIO *__restrict__ qpos,
                                                         const TIO *__restrict__ qvel, const TIO *__restrict__ action,
                                                         const TIO *__restrict__ qM, const TIO *__restrict__ C, int n,
                                                         TIO *__restrict__ torque, TIO *__restrict__ torque_raw) {
    __shared__ short s_map[PD_NV * PD_NV];        // dense (i,j) -> qM index: the dof tree, staged once per block
    __shared__ double s_qM[4][PD_NM_MAX];         // this wave's sparse inertia
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < PD_NV * PD_NV; i += 256) s_map[i] = m.m_map[i];
    const long env = (long)blockIdx.x * 4 + wave;
    const bool valid = env < n;
    if (valid) {
        const TIO *src = qM + env * ld.qM;
        for (int i = lane; i < m.nM; i += 64) s_qM[wave][i] = (double)src[i];
    }
    __syncthreads();
    if (!valid) return;
    const int row = lane < PD_NV ? lane : PD_NV - 1;   // spare lanes shadow the last row
    double kp, kd, eq, qv, b;
    pd_rhs<TIO>(m, ld, qpos, qvel, action, C, env, row, kp, kd, eq, qv, b);
    const double kd_dt = kd * m.sub_dt;
    // mj_fullM: lane `row` gathers its dense row from the sparse chain layout; + Kd*dt on the diagonal
    double a[PD_NV];
#pragma unroll
    for (int j = 0; j < PD_NV; ++j) {
        const int id = s_map[row * PD_NV + j];
        double v = id >= 0 ? s_qM[wave][id] : 0.0;
        a[j] = v + (j == row ? kd_dt : 0.0);
    }
    // Gauss-Jordan without pivoting (matrix is SPD): after step k column k is zero off the diagonal.
    double dinv = 0.0;
#pragma unroll
    for (int k = 0; k < PD_NV; ++k) {
        const double pk = readlane_f64(a[k], k);
        const double inv = 1.0 / pk;
        const bool me = row == k;
        const double f = me ? 0.0 : a[k] * inv;
        dinv = me ? inv : dinv;
#pragma unroll
        for (int j = k + 1; j < PD_NV; ++j) {
            const double r = readlane_f64(a[j], k);
            a[j] = fma(-f, r, a[j]);
        }
        const double bk = readlane_f64(b, k);
        b = fma(-f, bk, b);
    }
    const double qacc = b * dinv;
    pd_store<TIO>(m, env, row, lane < PD_NV, kp, kd, eq, qv, qacc, torque, torque_raw);
}

// Tree-ordered fast path (default for the humanoid): same lane-owns-row layout, but the dofs are eliminated
// leaves -> root (descending index). When dof k is the pivot, every descendant column of row k has already
// been zeroed, so row k is non-zero only on k's ANCESTOR columns: a pivot step updates |anc(k)| columns
// instead of all remaining ones (852 + 58 row updates instead of 1711 + 58; MuJoCo's L^T D L sparsity,
// no fill-in). The ancestor chains are compile-time tables (egp_tree58.inc, generated from the skeleton asset),
// so every register index stays static; egp_create checks the runtime dof tree against them.
#include <utility>
#include "egp_tree58.inc"

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);          // v_rcp_f64 + two Newton steps: <= 1 ulp for normal x
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

template <int K, int... T>
__device__ __forceinline__ void tree_pivot_update(double (&a)[PD_NV], double f, std::integer_sequence<int, T...>) {
    if constexpr (sizeof...(T) > 0) {
        // all broadcasts first (v_readlane -> SGPR pairs), then the FMAs: no SGPR-hazard nops in between
        const double r[sizeof...(T)] = {readlane_f64(a[Tree58::ANC[K][T]], K)...};
        __builtin_amdgcn_sched_barrier(0);
        ((a[Tree58::ANC[K][T]] = fma(-f, r[T], a[Tree58::ANC[K][T]])), ...);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K>
__device__ __forceinline__ void tree_eliminate(double (&a)[PD_NV], double &b, double &dinv, int row) {
    const double pk = readlane_f64(a[K], K);
    const double inv = fast_rcp(pk);
    const bool me = row == K;
    const double f = me ? 0.0 : a[K] * inv;
    dinv = me ? inv : dinv;
    tree_pivot_update<K>(a, f, std::make_integer_sequence<int, Tree58::NANC[K]>{});
    const double bk = readlane_f64(b, K);
    b = fma(-f, bk, b);
    if constexpr (K > 0) tree_eliminate<K - 1>(a, b, dinv, row);
}

// The same elimination split in two: the matrix work once (the multiplier of pivot K replaces column K of the lane's
// row, like L stored in place), the right-hand side per solve. The operations on b are exactly those of tree_eliminate,
// in the same order, so factor + solve is bit-identical to the one-pass form.
template <int K>
__device__ __forceinline__ void tree_factor(double (&a)[PD_NV], double &dinv, int row) {
    const double pk = readlane_f64(a[K], K);
    const double inv = fast_rcp(pk);
    const bool me = row == K;
    const double f = me ? 0.0 : a[K] * inv;
    dinv = me ? inv : dinv;
    tree_pivot_update<K>(a, f, std::make_integer_sequence<int, Tree58::NANC[K]>{});
    a[K] = f;                       // column K is dead from here on (no later pivot has K among its ancestors)
    if constexpr (K > 0) tree_factor<K - 1>(a, dinv, row);
}

template <int K>
__device__ __forceinline__ void tree_solve(const double (&a)[PD_NV], double &b) {
    const double bk = readlane_f64(b, K);
    b = fma(-a[K], bk, b);
    if constexpr (K > 0) tree_solve<K - 1>(a, b);
}

#include "egp_pd_grid.hpp"

template <typename TIO>
__global__ __launch_bounds__(256) void k_pd_torque_tree58(DevModel m, PdLd ld, const TIO *__restrict__ qpos,
                                                          const TIO *__restrict__ qvel, const TIO *__restrict__ action,
                                                          const TIO *__restrict__ qM, const TIO *__restrict__ C, int n,
                                                          TIO *__restrict__ torque, TIO *__restrict__ torque_raw, PdDone done) {
    __shared__ short s_map[PD_NV * PD_NV];
    __shared__ double s_qM[4][PD_NM_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    const bool valid = env < n;
    const int row = lane < PD_NV ? lane : PD_NV - 1;
    // the state row may live in pinned host memory (zero-copy engine): issue those loads first so the PCIe round
    // trip overlaps the LDS fills below
    const int act = row >= 6 ? row - 6 : 0;
    TIO r_q = TIO(0), r_v = TIO(0), r_c = TIO(0), r_a = TIO(0);
    if (valid) {
        r_q = qpos[env * ld.qpos + 7 + act];
        r_v = qvel[env * ld.qvel + row];
        r_c = C[env * ld.bias + row];
        r_a = action[env * ld.action + act];
    }
    // per-dof constants, also up front (each dependent global load costs a full memory round trip at 2 waves/CU)
    const double c_kp = m.jkp[act], c_kd = m.jkd[act], c_ref = m.a_ref[act], c_scale = m.a_scale[act], c_lim = m.torque_lim[act];
    // LDS fills with every load of a thread in flight at once (the rolled loops waited for each load in turn)
    {
        constexpr int MAP_IT = (PD_NV * PD_NV + 255) / 256;
        short t_map[MAP_IT];
#pragma unroll
        for (int k = 0; k < MAP_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            t_map[k] = i < PD_NV * PD_NV ? m.m_map[i] : (short)0;
        }
        constexpr int QM_IT = PD_NM_MAX / 64;
        TIO t_qM[QM_IT];
        const TIO *src = qM + (valid ? env : 0) * ld.qM;
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) {
            const int i = lane + 64 * k;
            t_qM[k] = (valid && i < m.nM) ? src[i] : TIO(0);
        }
#pragma unroll
        for (int k = 0; k < MAP_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < PD_NV * PD_NV) s_map[i] = t_map[k];
        }
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) s_qM[wave][lane + 64 * k] = (double)t_qM[k];
    }
    __syncthreads();
    if (valid) {
        // same arithmetic as pd_rhs / pd_store
        double kp = 0.0, kd = 0.0, eq = 0.0;
        if (row >= 6) {
            kp = c_kp;
            kd = c_kd;
            const double target = c_ref + (double)r_a * c_scale;
            eq = (double)r_q - target;
        }
        const double qv = (double)r_v;
        double b = -(double)r_c - kp * eq - kd * qv;
        const double kd_dt = kd * m.sub_dt;
        double a[PD_NV];
#pragma unroll
        for (int j = 0; j < PD_NV; ++j) {
            const int id = s_map[row * PD_NV + j];
            double v = id >= 0 ? s_qM[wave][id] : 0.0;
            a[j] = v + (j == row ? kd_dt : 0.0);
        }
        double dinv = 0.0;
        tree_eliminate<PD_NV - 1>(a, b, dinv, row);
        const double qacc = b * dinv;
        if (lane < PD_NV && row >= 6) {
            const double ev = qv + qacc * m.sub_dt;
            const double tau = -kp * eq - kd * ev;
            const double tc = fmin(fmax(tau, -c_lim), c_lim);
            torque[env * m.nu + act] = (TIO)tc;
            if (torque_raw) torque_raw[env * m.nu + act] = (TIO)tau;
        }
    }
    if (done.counter) {
        // completion signal for a host that polls pinned memory instead of synchronising the stream: every wave
        // releases its torques system-wide, the block counts in; the last one publishes the launch's sequence number
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

// cfg.action_type == 'torque' (humanoid_v1.py:167-172): torque = clip(a_ref + action * a_scale, +-torque_lim); nothing of
// the state enters. Same launch geometry and completion signal as the PD kernels it stands in for (4 envs per block).
template <typename TIO>
__global__ __launch_bounds__(256) void k_torque_direct(DevModel m, PdLd ld, const TIO *__restrict__ action, int n,
                                                       TIO *__restrict__ torque, TIO *__restrict__ torque_raw, PdDone done) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    if (env < n && lane < m.nu) {
        const double tau = m.a_ref[lane] + (double)action[env * ld.action + lane] * m.a_scale[lane];
        const double lim = m.torque_lim[lane];
        torque[env * m.nu + lane] = (TIO)fmin(fmax(tau, -lim), lim);
        if (torque_raw) torque_raw[env * m.nu + lane] = (TIO)tau;
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

// Resident form of the tree kernel for the rollout engine ("K1 server"): ONE launch covers all frame_skip
// substeps of an env-step. The launch is cut into slices of whole blocks, each owned by one host physics thread.
// For every substep a block waits until its slice's `go` word (pinned host memory, written by the owner after it
// advanced the slice's envs) reaches the substep, reads the fresh state rows over PCIe, solves and writes the
// clipped torques straight into the pinned torque rows. There is no completion signal: the owner pre-fills the
// rows with a NaN sentinel before it raises `go` and steps an env as soon as its row holds no sentinel any more
// (clipped torques are never NaN), so the device side needs no fence, no counter and no flag write.
// What stays on the chip across substeps: the dof map, the gains, the action and the sparse inertia rows in LDS
// (re-read from the host rows only when the owner flags that some qM of the slice changed).
// No kernel launch, no stream sync and no group-wide barrier is left on the substep path.
struct PdServe {
    const int *block_slice;               // [gridDim.x]
    const unsigned long long *go;         // [n_slices * 8]  host; value = (substep sequence << 1) | refresh_qM
    unsigned long long base;              // sequence of substep 0 of this env-step
    int n_sub;
    const double *qM_host;                // device alias of the pinned inertia rows (same row stride as qM)
    double *qM_dev;                       // HBM inertia rows, refreshed together with LDS
    int *err;                             // host: set when a wait timed out
    long long timeout_ticks;              // wall_clock64 ticks (100 MHz)
    long long *trace;                     // optional [n_sub * 8] wall_clock64 stamps of block 0 (diagnostics)
    // epilogue (go word reaches base + n_sub: the slice's last physics step is drained): final state -> HBM
    const double *ee_host;                // [n][15] pinned end-effector rows (device alias)
    double *out_qpos, *out_prev_qpos, *out_qvel, *out_ee;   // HBM [n][nq] / [n][nq] / [n][nv] / [n][15]
    int nq, nv;
    int poll_sleep;                       // s_sleep(1) (64 clocks) repeats between two polls of a go word
    const int *active;                    // optional [n] (pinned host): envs with 0 are not stepped this env-step -- their waves
                                          // move no state, torque or epilogue rows (in a rollout's tail that is most of the PCIe traffic)
    const void *dyn;                      // DYN kernels: the egp_dyn::DynTables of the context (device)
    double *bias_dev;                     // DYN kernels: [n][nv] HBM bias rows -- with qM_dev what the env's last mj_step "left behind"
    unsigned *probe;                      // residency probe (egp_pd_server_resident_blocks): count the workgroups on the chip at once and leave
    const int *block_env0;                // multi-env kernels: [gridDim.x + 1] first env of every workgroup (the envs dealt out evenly)
    int row_contig;                       // qpos | qvel | bias are ONE row of nq + 2 nv doubles (the engine's state rows): read it as a
                                          // contiguous stream (see the substep loop)
};

// Poll a word of pinned host memory through the SCALAR memory path (s_load ... glc = always fetch from beyond the
// scalar cache). A vector load that is out on PCIe for ~2 us sits in the CU's in-order vector-memory return queue,
// and with a polling block on every CU that stalled the loads of every other kernel on the chip (a 17 us policy
// kernel took 160 us next to the polling engine); scalar loads do not go through that queue.
__device__ __forceinline__ unsigned long long scalar_poll_u64(const unsigned long long *p) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ __forceinline__ double sys_load_f64(const double *p) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __longlong_as_double((long long)u);
}

// Residency probe: every workgroup that gets a place on the chip counts itself in, holds that place until all gridDim.x are there
// (probe[2] is raised) or `timeout_ticks` have passed, and counts itself out; probe[1] ends up as the largest head count any of them
// saw = the workgroups of THIS kernel (its registers, its LDS) the chip really holds at once -- under a CU mask, next to a co-tenant,
// on a part with CUs fused off -- where the occupancy calculator only knows the data sheet. The resident env-step needs that number.
__device__ __forceinline__ void server_residency_probe(const PdServe &sv) {
    if (threadIdx.x == 0) {
        unsigned present = atomicAdd(sv.probe, 1u) + 1u, best = present;
        const long long t0 = wall_clock64();
        for (;;) {
            if (present >= gridDim.x) { atomicExch(sv.probe + 2, 1u); break; }
            if (__hip_atomic_load(sv.probe + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > sv.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(8);
            present = __hip_atomic_load(sv.probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            best = present > best ? present : best;
        }
        atomicMax(sv.probe + 1, best);
        atomicSub(sv.probe, 1u);
    }
}

// DYN (device_dynamics engines): the wave computes the inertia and the bias force itself from the (qpos, qvel) rows it reads
// (K8's wave function: FK + CRBA + RNE, egp_dynamics_dev.hpp) instead of taking qM / qfrc_bias from the host -- a backend whose
// inertia changes with every substep (MuJoCo's does) then sends 117 doubles per env-substep, not 1 085.
// With the reference's timing: compute_torque (ego_pose/envs/humanoid_v1.py:130-144) reads data.qM / data.qfrc_bias as the
// PREVIOUS mj_step left them -- evaluated at the state that step started from -- and only a reset's sim.forward()
// (envs/common/mujoco_env.py:97-101) makes them fresh. So substep k solves with the factors of M(q_{k-1}) and with C(q_{k-1}, v_{k-1})
// it already holds, stores the torque, and only THEN runs K8 + the factorisation on the row it has just read, for substep k + 1
// -- behind the host's physics step instead of in front of the torque. Across launches the env's (qM, bias) rows persist in
// HBM (qM_dev / bias_dev: written after the last substep, and by the engine's reset = sim.forward()). Needs the dynamic LDS.
template <bool DYN>
__global__ __launch_bounds__(256) void k_pd_server_tree58(DevModel m, PdLd ld, const double *qpos, const double *qvel,
                                                          const double *__restrict__ action, const double *qM, const double *C,
                                                          int n, double *torque, PdServe sv) {
    extern __shared__ double s_dynmem[];
    __shared__ short s_map[PD_NV * PD_NV];
    __shared__ double s_qM[4][PD_NM_MAX];
    __shared__ unsigned long long s_go[2];
    __shared__ int s_abort;
    if (sv.probe) { server_residency_probe(sv); return; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + wave;
    const bool valid = env < n;
    const int row = lane < PD_NV ? lane : PD_NV - 1;
    const int act = row >= 6 ? row - 6 : 0;
    const int slice = sv.block_slice[blockIdx.x];
    const bool live = valid && (!sv.active || sv.active[env] != 0);
    // a workgroup none of whose envs is stepped (finished slots: most of the grid in a rollout's tail) has nothing to solve, nothing
    // to hand back in the epilogue and nobody waiting for it: leave, instead of polling its slice's go word over PCIe for 15 substeps
    if (__syncthreads_or(live ? 1 : 0) == 0) return;
    const double r_a = valid ? action[env * ld.action + act] : 0.0;
    const double c_kp = row >= 6 ? m.jkp[act] : 0.0, c_kd = row >= 6 ? m.jkd[act] : 0.0;
    const double c_ref = m.a_ref[act], c_scale = m.a_scale[act], c_lim = m.torque_lim[act];
    constexpr int QM_IT = PD_NM_MAX / 64;
    {
        constexpr int MAP_IT = (PD_NV * PD_NV + 255) / 256;
        short t_map[MAP_IT];
#pragma unroll
        for (int k = 0; k < MAP_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            t_map[k] = i < PD_NV * PD_NV ? m.m_map[i] : (short)0;
        }
        double t_qM[QM_IT];
        const double *src = qM + (valid ? env : 0) * ld.qM;
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) {
            const int i = lane + 64 * k;
            t_qM[k] = (valid && i < m.nM) ? src[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < MAP_IT; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < PD_NV * PD_NV) s_map[i] = t_map[k];
        }
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) s_qM[wave][lane + 64 * k] = t_qM[k];
    }
    constexpr int TB_DOUBLES = DYN ? (int)((sizeof(egp_dyn::DynTables) + 7) / 8) : 0;
    egp_dyn::DynTables *tb = reinterpret_cast<egp_dyn::DynTables *>(s_dynmem);
    double *s_scr = s_dynmem + TB_DOUBLES + wave * egp_dyn::DY_ENV_DOUBLES;                   // K8 scratch of this wave
    double *s_q = s_dynmem + TB_DOUBLES + 4 * egp_dyn::DY_ENV_DOUBLES + wave * 192;            // qpos[64] | qvel[64] | bias[64]
    if constexpr (DYN) {
        const int words = sizeof(egp_dyn::DynTables) / 4;
        const int *src = reinterpret_cast<const int *>(sv.dyn);
        int *dst = reinterpret_cast<int *>(tb);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
    }
    if (threadIdx.x == 0) s_abort = 0;
    const double target = c_ref + r_a * c_scale;
    const bool tracer = sv.trace && blockIdx.x == 0 && threadIdx.x == 0;
    // (M + Kd dt) only changes when the owner flags a new inertia row: factor it then (and at substep 0), and run only
    // the right-hand side through the stored multipliers otherwise -- 9 us of elimination become 0.6 us per substep.
    double a[PD_NV];
    double dinv = 0.0;
    const double kd_dt = c_kd * m.sub_dt;
    double c_held = 0.0;                    // DYN: the bias entry of this lane's dof that the previous substep left behind
    double nx_q = 0.0, nx_v = 0.0;          // DYN: the next substep's state row, when it could be requested early
    bool have_next = false;
    auto factor_from_lds = [&]() {
#pragma unroll
        for (int j = 0; j < PD_NV; ++j) {
            const int id = s_map[row * PD_NV + j];
            double v = id >= 0 ? s_qM[wave][id] : 0.0;
            a[j] = v + (j == row ? kd_dt : 0.0);
        }
        tree_factor<PD_NV - 1>(a, dinv, row);
    };
    if constexpr (DYN) {
        // what the last env-step's final mj_step (or the reset's forward) left in HBM: factor it before the first go word
        __syncthreads();
        if (live && !m.action_torque) {
            c_held = C[env * ld.bias + row];
            factor_from_lds();
        }
    }
    for (int sub = 0; sub < sv.n_sub; ++sub) {
        if (threadIdx.x == 0) {
            const unsigned long long want = sv.base + (unsigned long long)sub;
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = scalar_poll_u64(sv.go + slice * 8);
                if ((v >> 1) >= want) break;
                for (int z = 0; z < sv.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > sv.timeout_ticks) { s_abort = 1; break; }
            }
            s_go[sub & 1] = v;            // double-buffered: one barrier per substep is enough
            if (tracer) { sv.trace[sub * 8 + 0] = t0; sv.trace[sub * 8 + 1] = wall_clock64(); }
        }
        __syncthreads();
        if (s_abort) {
            if (threadIdx.x == 0) __hip_atomic_store(sv.err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const bool refresh = DYN || (s_go[sub & 1] & 1ull) != 0ull;
        if (live && m.action_torque) {          // action_type 'torque' (humanoid_v1.py:170-172): the clipped control itself
            if (lane < PD_NV && row >= 6)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(torque + env * m.nu + act),
                                   (unsigned long long)__double_as_longlong(fmin(fmax(target, -c_lim), c_lim)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (live) {
            double r_q, r_v, r_c;
            if constexpr (DYN) {
                if (have_next) {                // requested under the previous substep's factorisation (below)
                    if (lane < sv.nq) s_q[lane] = nx_q;
                    if (lane < sv.nv) s_q[64 + lane] = nx_v;
                } else {
                    if (lane < sv.nq) s_q[lane] = sys_load_f64(qpos + env * ld.qpos + lane);
                    if (lane < sv.nv) s_q[64 + lane] = sys_load_f64(qvel + env * ld.qvel + lane);
                }
                egp_dyn::wave_sync();
                r_q = s_q[7 + act]; r_v = s_q[64 + row]; r_c = c_held;
            } else if (sv.row_contig) {
                // The state row over PCIe, lane l taking doubles l, 64 + l, 128 + l: whole 64-byte lines, each once (22 for the
                // humanoid's 175 doubles). As three lane = dof segments (qpos + 7.., qvel, bias) the same row costs ~25 line reads,
                // the segments start mid-line -- tools/probes/pcie_read_probe.hip: 48 against 57 GB/s, and the full-activity
                // env-step is bound by exactly this traffic. The lanes then fetch their dof's three values with ds_bpermute.
                const double *rowp = qpos + env * ld.qpos;
                const int tot = sv.nq + 2 * sv.nv;
                const double c0 = lane < tot ? sys_load_f64(rowp + lane) : 0.0;
                const double c1 = 64 + lane < tot ? sys_load_f64(rowp + 64 + lane) : 0.0;
                const double c2 = 128 + lane < tot ? sys_load_f64(rowp + 128 + lane) : 0.0;
                auto pick = [&](int i) {
                    const double a0 = __shfl(c0, i & 63), a1 = __shfl(c1, i & 63), a2 = __shfl(c2, i & 63);
                    return i < 64 ? a0 : (i < 128 ? a1 : a2);
                };
                r_q = pick(7 + act);
                r_v = pick(sv.nq + row);
                r_c = pick(sv.nq + sv.nv + row);
            } else {
                r_q = sys_load_f64(qpos + env * ld.qpos + 7 + act);
                r_v = sys_load_f64(qvel + env * ld.qvel + row);
                r_c = sys_load_f64(C + env * ld.bias + row);
            }
            if (!DYN && refresh) {              // this wave's inertia row changed on the host: LDS and HBM copies
                const double *src = sv.qM_host + env * ld.qM;
                double *dst = sv.qM_dev + env * ld.qM;
                double t_qM[QM_IT];
#pragma unroll
                for (int k = 0; k < QM_IT; ++k) {
                    const int i = lane + 64 * k;
                    t_qM[k] = i < m.nM ? sys_load_f64(src + i) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < QM_IT; ++k) {
                    const int i = lane + 64 * k;
                    s_qM[wave][i] = t_qM[k];
                    if (i < m.nM) dst[i] = t_qM[k];
                }
            }
            if (!DYN && (sub == 0 || refresh)) factor_from_lds();      // (wave-uniform: the go word is per slice, the wave per env)
            const double kp = c_kp, kd = c_kd;
            const double eq = row >= 6 ? r_q - target : 0.0;
            const double qv = r_v;
            double b = -r_c - kp * eq - kd * qv;
            if (tracer) sv.trace[sub * 8 + 2] = b != 12345.678 ? wall_clock64() : 0;
            tree_solve<PD_NV - 1>(a, b);
            const double qacc = b * dinv;
            if (tracer) sv.trace[sub * 8 + 3] = qacc != 12345.678 ? wall_clock64() : 0;
            if (lane < PD_NV && row >= 6) {
                const double ev = qv + qacc * m.sub_dt;
                const double tau = -kp * eq - kd * ev;
                // system-scope store: straight to the pinned row, nothing lingers in a device cache
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(torque + env * m.nu + act),
                                   (unsigned long long)__double_as_longlong(fmin(fmax(tau, -c_lim), c_lim)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (tracer) sv.trace[sub * 8 + 4] = wall_clock64();
            if constexpr (DYN) {
                // the torque is on its way and the host steps: now what that mj_step leaves behind for the NEXT compute_torque --
                // M and C at the state just read -- and its factors
                egp_dyn::dynamics_wave(*tb, s_scr, s_q, s_q + 64, lane, true, &s_qM[wave][0], s_q + 128, nullptr);
                egp_dyn::wave_sync();
                c_held = s_q[128 + row];
                // K8 took its 11 us: if the owner has raised the next go word meanwhile (with free physics it has; the rows were
                // written before the word, fenced), the next state row is requested NOW and crosses PCIe under the factorisation
                // instead of in front of the next solve (2.6 us of every 24.6 us substep in the rollout's tail)
                have_next = false;
                if (sub + 1 < sv.n_sub && (scalar_poll_u64(sv.go + slice * 8) >> 1) >= sv.base + (unsigned long long)sub + 1ull) {
                    nx_q = lane < sv.nq ? sys_load_f64(qpos + env * ld.qpos + lane) : 0.0;
                    nx_v = lane < sv.nv ? sys_load_f64(qvel + env * ld.qvel + lane) : 0.0;
                    have_next = true;
                }
                factor_from_lds();
                if (sub == sv.n_sub - 1) {       // ... and across launches
                    double *dst = sv.qM_dev + env * ld.qM;
#pragma unroll
                    for (int k = 0; k < QM_IT; ++k) {
                        const int i = lane + 64 * k;
                        if (i < m.nM) dst[i] = s_qM[wave][i];
                    }
                    if (lane < PD_NV) sv.bias_dev[env * ld.bias + row] = c_held;
                }
                if (tracer) sv.trace[sub * 8 + 5] = wall_clock64();
            }
        }
    }
    // epilogue: prev_qpos <- qpos; qpos | qvel | ee_wpos <- the rows the owner drained after the last substep.
    // Runs slice by slice as the physics threads finish, so the env-step ends one PCIe round trip after the last
    // physics call instead of a barrier + a copy launch later.
    {
        const int slot = sv.n_sub & 1;
        if (threadIdx.x == 0) {
            const unsigned long long want = sv.base + (unsigned long long)sv.n_sub;
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = scalar_poll_u64(sv.go + slice * 8);
                if ((v >> 1) >= want) break;
                for (int z = 0; z < sv.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > sv.timeout_ticks) { s_abort = 1; break; }
            }
            s_go[slot] = v;
        }
        __syncthreads();
        if (s_abort) {
            if (threadIdx.x == 0) __hip_atomic_store(sv.err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (live) {
            // the row base of qpos: `qpos` points at the state rows' qpos column (offset 0 of the row)
            if (lane < sv.nq) {
                const long d = env * sv.nq + lane;
                const double q = sys_load_f64(qpos + env * ld.qpos + lane);
                sv.out_prev_qpos[d] = sv.out_qpos[d];
                sv.out_qpos[d] = q;
            }
            if (lane < sv.nv) sv.out_qvel[env * sv.nv + lane] = sys_load_f64(qvel + env * ld.qvel + lane);
            if (lane < 15) sv.out_ee[env * 15 + lane] = sys_load_f64(sv.ee_host + env * 15 + lane);
        }
    }
}

// KE envs per wavefront ("multi" form of the resident K1): the grid must fit the chip at once -- one 350-register workgroup per
// CU -- so beyond 4 envs per CU (1 024 slots on 256 CUs), or when fewer CUs are to be had, a wave serves its KE envs in turn in
// every substep. The wave's registers hold ONE env's factors while it eliminates; what the solves need is then kept in LDS in
// place of the env's inertia row, in the same sparse layout: entry (K, r), r an ancestor of K, holds the multiplier of pivot K in
// row r -- L[K][r] of M + Kd dt = L^T D L (the tree elimination's upper multipliers ARE Featherstone's L) -- and the diagonal
// entry 1 / D_K. A solve is two sweeps over those 852 + 58 numbers: leaves -> root through L^T, scale, root -> leaves through L
// (the one-env kernel keeps both triangles' multipliers in registers and needs one sweep: the two forms agree to rounding).
// All state rows of the wave's envs are requested before the first solve, so their PCIe round trips overlap.
template <int KE>
__global__ __launch_bounds__(256) void k_pd_server_tree58_multi(DevModel m, PdLd ld, const double *qpos, const double *qvel,
                                                                const double *__restrict__ action, const double *qM, const double *C,
                                                                int n, double *torque, PdServe sv) {
    extern __shared__ double s_fac[];                  // [4 waves][KE][PD_NM_MAX]
    __shared__ short s_map[PD_NV * PD_NV];
    __shared__ unsigned long long s_go[2];
    __shared__ int s_abort;
    if (sv.probe) { server_residency_probe(sv); return; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the workgroup's envs [be0, be1) (at most 4 KE of them, dealt out evenly by the engine: with 4.5 envs per workgroup only every
    // second workgroup has a wave that serves two); wave w takes be0 + w, be0 + w + 4, ...
    const long be0 = sv.block_env0[blockIdx.x], be1 = sv.block_env0[blockIdx.x + 1];
    const long env0 = be0 + wave;
    const int row = lane < PD_NV ? lane : PD_NV - 1;
    const int act = row >= 6 ? row - 6 : 0;
    const int slice = sv.block_slice[blockIdx.x];
    bool live[KE];
    bool any = false;
#pragma unroll
    for (int e = 0; e < KE; ++e) {
        live[e] = env0 + 4 * e < be1 && env0 + 4 * e < n && (!sv.active || sv.active[env0 + 4 * e] != 0);
        any = any || live[e];
    }
    if (__syncthreads_or(any ? 1 : 0) == 0) return;
    const double c_kp = row >= 6 ? m.jkp[act] : 0.0, c_kd = row >= 6 ? m.jkd[act] : 0.0;
    const double c_ref = m.a_ref[act], c_scale = m.a_scale[act], c_lim = m.torque_lim[act];
    const double kd_dt = c_kd * m.sub_dt;
    double target[KE];
#pragma unroll
    for (int e = 0; e < KE; ++e) target[e] = c_ref + (live[e] ? action[(env0 + 4 * e) * ld.action + act] : 0.0) * c_scale;
    constexpr int QM_IT = PD_NM_MAX / 64;
    for (int i = threadIdx.x; i < PD_NV * PD_NV; i += 256) s_map[i] = m.m_map[i];
#pragma unroll
    for (int e = 0; e < KE; ++e) {
        double *F = s_fac + (size_t)(wave * KE + e) * PD_NM_MAX;
        const double *src = qM + (env0 + 4 * e) * ld.qM;
#pragma unroll
        for (int k = 0; k < QM_IT; ++k) {
            const int i = lane + 64 * k;
            F[i] = (live[e] && i < m.nM) ? src[i] : 0.0;
        }
    }
    if (threadIdx.x == 0) s_abort = 0;
    const int tot = sv.nq + 2 * sv.nv;
    const bool tracer = sv.trace && blockIdx.x == 0 && threadIdx.x == 0;
    // sparse index of entry (row, K) / (K, row) -- the same for every env -- or, where the tree has no such entry, a slot of the
    // row's padding that stays zero (nM = 910 of PD_NM_MAX = 960 doubles)
    constexpr int ZERO_SLOT = PD_NM_MAX - 1;
    short idv[PD_NV];
#pragma unroll
    for (int K = 0; K < PD_NV; ++K) {
        const short id = m.m_map[row * PD_NV + K];
        idv[K] = id >= 0 ? id : (short)ZERO_SLOT;
    }
    const int id_diag = m.m_map[row * PD_NV + row];
    for (int sub = 0; sub < sv.n_sub; ++sub) {
        if (threadIdx.x == 0) {
            const unsigned long long want = sv.base + (unsigned long long)sub;
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = scalar_poll_u64(sv.go + slice * 8);
                if ((v >> 1) >= want) break;
                for (int z = 0; z < sv.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > sv.timeout_ticks) { s_abort = 1; break; }
            }
            s_go[sub & 1] = v;
            if (tracer) { sv.trace[sub * 8 + 0] = t0; sv.trace[sub * 8 + 1] = wall_clock64(); }
        }
        __syncthreads();
        if (s_abort) {
            if (threadIdx.x == 0) __hip_atomic_store(sv.err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const bool refresh = (s_go[sub & 1] & 1ull) != 0ull;
        if (m.action_torque) {                  // action_type 'torque' (humanoid_v1.py:170-172): the clipped control itself
#pragma unroll
            for (int e = 0; e < KE; ++e)
                if (live[e] && lane < PD_NV && row >= 6)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(torque + (env0 + 4 * e) * m.nu + act),
                                       (unsigned long long)__double_as_longlong(fmin(fmax(target[e], -c_lim), c_lim)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            continue;
        }
        // every live env's state row first (whole 64-byte lines when the row is contiguous: see k_pd_server_tree58)
        double c0[KE], c1[KE], c2[KE];
#pragma unroll
        for (int e = 0; e < KE; ++e) {
            c0[e] = c1[e] = c2[e] = 0.0;
            if (!live[e]) continue;
            const long env = env0 + 4 * e;
            if (sv.row_contig) {
                const double *rowp = qpos + env * ld.qpos;
                if (lane < tot) c0[e] = sys_load_f64(rowp + lane);
                if (64 + lane < tot) c1[e] = sys_load_f64(rowp + 64 + lane);
                if (128 + lane < tot) c2[e] = sys_load_f64(rowp + 128 + lane);
            } else {
                c0[e] = sys_load_f64(qpos + env * ld.qpos + 7 + act);
                c1[e] = sys_load_f64(qvel + env * ld.qvel + row);
                c2[e] = sys_load_f64(C + env * ld.bias + row);
            }
        }
#pragma unroll
        for (int e = 0; e < KE; ++e) {
            if (!live[e]) continue;             // (wave-uniform)
            const long env = env0 + 4 * e;
            double *F = s_fac + (size_t)(wave * KE + e) * PD_NM_MAX;
            double r_q = c0[e], r_v = c1[e], r_c = c2[e];
            if (sv.row_contig) {
                auto pick = [&](int i) {
                    const double a0 = __shfl(c0[e], i & 63), a1 = __shfl(c1[e], i & 63), a2 = __shfl(c2[e], i & 63);
                    return i < 64 ? a0 : (i < 128 ? a1 : a2);
                };
                r_q = pick(7 + act);
                r_v = pick(sv.nq + row);
                r_c = pick(sv.nq + sv.nv + row);
            }
            if (refresh) {                      // this env's inertia row changed on the host: LDS and HBM copies
                const double *src = sv.qM_host + env * ld.qM;
                double *dst = sv.qM_dev + env * ld.qM;
                double t_qM[QM_IT];
#pragma unroll
                for (int k = 0; k < QM_IT; ++k) {
                    const int i = lane + 64 * k;
                    t_qM[k] = i < m.nM ? sys_load_f64(src + i) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < QM_IT; ++k) {
                    const int i = lane + 64 * k;
                    F[i] = t_qM[k];
                    if (i < m.nM) dst[i] = t_qM[k];
                }
                egp_dyn::wave_sync();
            }
            if (sub == 0 || refresh) {
                double a[PD_NV];
                double dinv = 0.0;
#pragma unroll
                for (int j = 0; j < PD_NV; ++j) {
                    const int id = s_map[row * PD_NV + j];
                    const double v = id >= 0 ? F[id] : 0.0;
                    a[j] = v + (j == row ? kd_dt : 0.0);
                }
                tree_factor<PD_NV - 1>(a, dinv, row);
                egp_dyn::wave_sync();           // every lane has its row: the inertia entries may go
#pragma unroll
                for (int K = 0; K < PD_NV; ++K) {
                    const int id = s_map[row * PD_NV + K];
                    if (lane < PD_NV && K >= row && id >= 0) F[id] = K == row ? dinv : a[K];
                }
                egp_dyn::wave_sync();
            }
            const double eq = row >= 6 ? r_q - target[e] : 0.0;
            double b = -r_c - c_kp * eq - c_kd * r_v;
            if (tracer && e == 0) sv.trace[sub * 8 + 2] = b != 12345.678 ? wall_clock64() : 0;     // (stamps of block 0: its first env, then its last)
            // the lane's column of L for the sweep towards the root (entries (K, row), K a descendant), then its row for the sweep back
            // (entries (row, K), K an ancestor): 58 reads in flight at once each time, the entries that do not exist (and the other
            // sweep's) read a zero slot of the row's padding -- no select, no LDS round trip on the dependent chain readlane -> fma
            // (with the reads and a K > row select inside the sweeps a solve took 13, then 4 us; now as the one-env kernel's: ~2)
            double cf[PD_NV];
#pragma unroll
            for (int K = 0; K < PD_NV; ++K) cf[K] = F[K > row ? idv[K] : ZERO_SLOT];
            const double d_own = F[id_diag];
#pragma unroll
            for (int K = PD_NV - 1; K >= 0; --K) b = fma(-cf[K], readlane_f64(b, K), b);      // L^T y = b: leaves -> root
            b *= d_own;
#pragma unroll
            for (int K = 0; K < PD_NV; ++K) cf[K] = F[K < row ? idv[K] : ZERO_SLOT];
#pragma unroll
            for (int J = 0; J < PD_NV; ++J) b = fma(-cf[J], readlane_f64(b, J), b);           // L x = z: root -> leaves
            if (tracer && e == 0) sv.trace[sub * 8 + 3] = b != 12345.678 ? wall_clock64() : 0;
            if (lane < PD_NV && row >= 6) {
                const double ev = r_v + b * m.sub_dt;
                const double tau = -c_kp * eq - c_kd * ev;
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(torque + env * m.nu + act),
                                   (unsigned long long)__double_as_longlong(fmin(fmax(tau, -c_lim), c_lim)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (tracer) sv.trace[sub * 8 + (e == 0 ? 4 : 5)] = wall_clock64();
        }
    }
    // epilogue (see k_pd_server_tree58): the final state of the wave's envs to HBM once the slice's last step is drained
    {
        const int slot = sv.n_sub & 1;
        if (threadIdx.x == 0) {
            const unsigned long long want = sv.base + (unsigned long long)sv.n_sub;
            const long long t0 = wall_clock64();
            unsigned long long v;
            for (;;) {
                v = scalar_poll_u64(sv.go + slice * 8);
                if ((v >> 1) >= want) break;
                for (int z = 0; z < sv.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > sv.timeout_ticks) { s_abort = 1; break; }
            }
            s_go[slot] = v;
        }
        __syncthreads();
        if (s_abort) {
            if (threadIdx.x == 0) __hip_atomic_store(sv.err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
#pragma unroll
        for (int e = 0; e < KE; ++e) {
            if (!live[e]) continue;
            const long env = env0 + 4 * e;
            if (lane < sv.nq) {
                const long d = env * sv.nq + lane;
                const double q = sys_load_f64(qpos + env * ld.qpos + lane);
                sv.out_prev_qpos[d] = sv.out_qpos[d];
                sv.out_qpos[d] = q;
            }
            if (lane < sv.nv) sv.out_qvel[env * sv.nv + lane] = sys_load_f64(qvel + env * ld.qvel + lane);
            if (lane < 15) sv.out_ee[env * 15 + lane] = sys_load_f64(sv.ee_host + env * 15 + lane);
        }
    }
}

// Generic path (any nv <= 64): one wavefront per env, system in LDS.
template <typename TIO>
__global__ __launch_bounds__(64) void k_pd_torque_lds(DevModel m, PdLd ld, const TIO *__restrict__ qpos,
                                                      const TIO *__restrict__ qvel, const TIO *__restrict__ action,
                                                      const TIO *__restrict__ qM, const TIO *__restrict__ C, int n,
                                                      TIO *__restrict__ torque, TIO *__restrict__ torque_raw) {
    extern __shared__ double s_dyn[];
    const int nv = m.nv, lda = nv + 1;
    double *A = s_dyn;            // [nv][ld]
    double *rhs = s_dyn + nv * lda; // [nv]
    const int lane = threadIdx.x;
    const long env = blockIdx.x;
    const int row = lane < nv ? lane : nv - 1;
    for (int i = lane; i < nv * lda; i += 64) A[i] = 0.0;
    __syncthreads();
    for (int i = lane; i < m.nM; i += 64) {
        const int r = m.m_row[i], c = m.m_col[i];
        const double v = (double)qM[env * ld.qM + i];
        A[r * lda + c] = v;
        A[c * lda + r] = v;
    }
    __syncthreads();
    double kp, kd, eq, qv, b;
    pd_rhs<TIO>(m, ld, qpos, qvel, action, C, env, row, kp, kd, eq, qv, b);
    if (lane < nv) {
        A[row * lda + row] += kd * m.sub_dt;
        rhs[row] = b;
    }
    __syncthreads();
    for (int k = 0; k < nv; ++k) {
        const double pk = A[k * lda + k];
        if (lane < nv && row != k) {
            const double f = A[row * lda + k] / pk;
            for (int j = k + 1; j < nv; ++j) A[row * lda + j] -= f * A[k * lda + j];
            rhs[row] -= f * rhs[k];
        }
        __syncthreads();
    }
    const double qacc = rhs[row] / A[row * lda + row];
    pd_store<TIO>(m, env, row, lane < nv, kp, kd, eq, qv, qacc, torque, torque_raw);
}

// ============================================================================================ K2
// quat_space_reward_v3 (ego_pose/core/reward_function.py:4-60)
// packed expert row: [0] qpos_z | [1:4] rlinv_local | [4:7] rangv | [7:11] rq_rmh | [11:26] ee_pos |
//                    [26:26+4(nb-1)] bquat[4:] | [106:106+3(nb-1)] bangvel[3:]
constexpr int ER_Z = 0, ER_RLINV = 1, ER_RANGV = 4, ER_RQ = 7, ER_EE = 11, ER_BQ = 26, ER_BAV = 106;

template <typename T>
__device__ __forceinline__ T half_wave_sum(T v) {
    v += __shfl_xor(v, 16, 32);
    v += __shfl_xor(v, 8, 32);
    v += __shfl_xor(v, 4, 32);
    v += __shfl_xor(v, 2, 32);
    v += __shfl_xor(v, 1, 32);
    return v;
}

// ---- per-lane feature helpers shared by the reward kernel (K2) and the feature kernel (K7)
// body lane: orientation of one non-root body now and one step ago, and its finite-difference angular
// velocity rotation_from_quaternion(q_t q_{t-1}^-1)/dt  (get_angvel_fd, utils/math.py:38-44)
template <typename T>
__device__ __forceinline__ void body_feature(const T *cq, const T *pq, int s, int nd, T dt, Q4<T> *qc, V3<T> *omega) {
    const T c0 = nd > 0 ? cq[s] : T(0), c1 = nd > 1 ? cq[s + 1] : T(0), c2 = nd > 2 ? cq[s + 2] : T(0);
    const T p0 = nd > 0 ? pq[s] : T(0), p1 = nd > 1 ? pq[s + 1] : T(0), p2 = nd > 2 ? pq[s + 2] : T(0);
    *qc = q_from_euler_sxyz<T>(c0, c1, c2);
    const Q4<T> qp = q_from_euler_sxyz<T>(p0, p1, p2);   // == env.prev_bquat (humanoid_v1.py:184,188)
    const Q4<T> dv = qmul(*qc, qinv(qp));
    V3<T> ax; T ang;
    rot_axis_angle<T>(dv, &ax, &ang);
    if (sizeof(T) == 8) {          // (float64: one division instead of three, see qinv)
        const T rate = ang / dt;
        omega->x = ax.x * rate; omega->y = ax.y * rate; omega->z = ax.z * rate;
    } else {
        omega->x = ax.x * ang / dt; omega->y = ax.y * ang / dt; omega->z = ax.z * ang / dt;
    }
}

// root lane: get_qvel_fd(prev, cur, dt) root part (utils/math.py:20-35): world-frame linear velocity `v`,
// root-frame angular velocity `rv` (frame of the PREVIOUS root quaternion, angle wrapped to (-pi, pi])
template <typename T>
__device__ __forceinline__ void root_velocity(const T *cq, const T *pq, T dt, V3<T> *v, V3<T> *rv) {
    const Q4<T> rc{cq[3], cq[4], cq[5], cq[6]};
    const Q4<T> rp{pq[3], pq[4], pq[5], pq[6]};
    v->x = (cq[0] - pq[0]) / dt; v->y = (cq[1] - pq[1]) / dt; v->z = (cq[2] - pq[2]) / dt;
    const Q4<T> qrel = qmul(rc, qinv(rp));
    V3<T> ax; T ang;
    rot_axis_angle<T>(qrel, &ax, &ang);
    const T pi = T(3.14159265358979323846);
    if (ang > pi) ang -= T(2) * pi;
    else if (ang < -pi) ang += T(2) * pi;
    const V3<T> rvw{ax.x * ang / dt, ax.y * ang / dt, ax.z * ang / dt};
    *rv = rotate_T(rp, rvw);
}

// transform_vec(v, q, cfg.obs_coord) (utils/math.py:47-59): R(q)^T v in the 'root' frame, R(heading_q(q))^T v in the 'heading' frame
template <typename T>
__device__ __forceinline__ V3<T> coord_vec(const Q4<T> &q, const V3<T> &v, bool root_frame) {
    return rotate_T(root_frame ? q : heading_q(q), v);
}

// end-effector lane: world position -> root-relative, in the frame cfg.obs_coord names (get_ee_pos(transform),
// humanoid_v1.py:98-111; the reward passes cfg.obs_coord, reward_function.py:23; gen_expert.py:18-22 always 'heading')
template <typename T>
__device__ __forceinline__ V3<T> ee_local(const T *cq, const T *wp, bool root_frame) {
    const Q4<T> rc{cq[3], cq[4], cq[5], cq[6]};
    const V3<T> rel{wp[0] - cq[0], wp[1] - cq[1], wp[2] - cq[2]};
    return coord_vec(rc, rel, root_frame);
}

// Work split (round 3). A wavefront used to carry two envs, lane = body, with lane 0 doing the root terms and five lanes the end
// effectors: three divergent branches that run one after the other, the ~1 000-instruction root branch with ONE lane per env
// active (the kernel is float64-VALU bound: VALUBusy 87 %, 18.6 % of HBM peak at 65 536 envs). Now a workgroup owns a tile of
// envs and works in two phases: (1) lane = (env, body): pose and angular-velocity terms of every body into LDS; (2) lane = env
// for the root terms, lane = (env, end effector) on the other waves, then lane = env adds up the bodies' terms and combines the
// five parts. With multi-pass tiles (PASSES = 5: 60 envs for the humanoid, large batches) the root branch runs once per tile
// with nearly every lane busy instead of once per two envs; rollout-sized batches keep one-pass tiles (12 envs: the latency of
// a tick's reward is unchanged).
// Phase 1 packs as many envs into a wavefront as its 64 lanes hold bodies (three for the humanoid's 20 non-root bodies, 60 of
// 64 lanes busy instead of 40): a pass of the workgroup covers 4 * (64 / (nbody - 1)) envs, the per-body terms go to LDS and
// phase 2's env lane adds them up in body order.
__host__ __device__ inline int reward_envs_per_wave(int nbody) { return nbody > 1 ? (64 / (nbody - 1) > 0 ? 64 / (nbody - 1) : 1) : 1; }
__host__ __device__ inline int reward_tile_envs(int nbody, int passes) {
    const int t = 4 * reward_envs_per_wave(nbody) * passes, cap = passes == 1 ? 16 : 64;     // (cap = the tile arrays' capacity)
    return t < cap ? t : cap;
}
template <typename T, int PASSES>
__device__ __forceinline__ void reward_body(const DevModel &m, const RewardW &w, const T *__restrict__ expert_rows,
                                            const T *__restrict__ cur_qpos, const T *__restrict__ prev_qpos,
                                            const T *__restrict__ ee_wpos, const int *__restrict__ tcur,
                                            const int *__restrict__ frame, const int *__restrict__ endf,
                                            const int *__restrict__ active, T end_reward, int n,
                                            T *__restrict__ reward, T *__restrict__ cinfo, int block_id) {
    constexpr int ENVS = PASSES == 1 ? 16 : 64;                    // capacity of the tile arrays; the tile itself: reward_tile_envs
    __shared__ int s_start[EGP_MAX_BODY], s_ndof[EGP_MAX_BODY];
    __shared__ double s_bw[EGP_MAX_BODY];
    __shared__ T s_bp[ENVS][EGP_MAX_BODY], s_bv[ENVS][EGP_MAX_BODY];      // per-body pose / angular-velocity terms
    __shared__ T s_rp[ENVS], s_rv[ENVS], s_ee[ENVS][5];
    if (threadIdx.x < m.nbody) {
        s_start[threadIdx.x] = m.body_qpos_start[threadIdx.x];
        s_ndof[threadIdx.x] = m.body_ndof[threadIdx.x];
        s_bw[threadIdx.x] = threadIdx.x > 0 ? m.b_diffw[threadIdx.x - 1] : 0.0;
    }
    __syncthreads();
    const int per = m.nbody - 1, epw = reward_envs_per_wave(m.nbody), tile = reward_tile_envs(m.nbody, PASSES);
    const long env0 = (long)block_id * tile;
    const T dt = (T)m.dt;
    // ---- phase 1: lane = (env of the wavefront, body)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int sub = lane / per, bl = lane - sub * per;
#pragma unroll 1
        for (int ps = 0; ps < PASSES; ++ps) {
            const int el = (ps * 4 + wv) * epw + sub;
            const long env = env0 + el;
            if (sub >= epw || el >= tile) continue;
            T pose_sq = T(0), vel_acc = T(0);
            if (env < n && (!active || active[env])) {
                const int l = 1 + bl;
                const T *cq = cur_qpos + env * m.nq;
                const T *pq = prev_qpos + env * m.nq;
                const T *er = expert_rows + (long)frame[env] * EGP_EXPERT_ROW;
                Q4<T> qc; V3<T> om;
                body_feature<T>(cq, pq, s_start[l], s_ndof[l], dt, &qc, &om);
                const T *eb = er + ER_BQ + 4 * (l - 1);
                const Q4<T> qe{eb[0], eb[1], eb[2], eb[3]};
                const T pd = half_angle<T>(qmul(qc, qinv(qe))) * (T)s_bw[l];
                pose_sq = pd * pd;
                const T *ev = er + ER_BAV + 3 * (l - 1);
                const T dx = om.x - ev[0], dy = om.y - ev[1], dz = om.z - ev[2];
                if (w.v_ord == 2.0) {
                    vel_acc = dx * dx + dy * dy + dz * dz;
                } else {
                    const T p = (T)w.v_ord;
                    vel_acc = t_pow<T>(fabs(dx), p) + t_pow<T>(fabs(dy), p) + t_pow<T>(fabs(dz), p);
                }
            }
            s_bp[el][bl] = pose_sq;
            s_bv[el][bl] = vel_acc;
        }
    }
    __syncthreads();
    // ---- phase 2: lane = env (root terms, threads [0, ENVS)), lane = (env, end effector) (threads [64, 256))
    if (threadIdx.x < tile) {
        const long env = env0 + threadIdx.x;
        T rp_r = T(0), rv_r = T(0);
        if (env < n && (!active || active[env])) {
            const T *cq = cur_qpos + env * m.nq;
            const T *pq = prev_qpos + env * m.nq;
            const T *er = expert_rows + (long)frame[env] * EGP_EXPERT_ROW;
            V3<T> v, rv;
            root_velocity<T>(cq, pq, dt, &v, &rv);
            const Q4<T> rp{pq[3], pq[4], pq[5], pq[6]};
            // learner: get_qvel_fd(prev, cur, dt, cfg.obs_coord) (reward_function.py:19): frame of the PREVIOUS root quat
            const V3<T> vl = coord_vec(rp, v, m.obs_root != 0);
            const T dl = (vl.x - er[ER_RLINV]) * (vl.x - er[ER_RLINV]) + (vl.y - er[ER_RLINV + 1]) * (vl.y - er[ER_RLINV + 1]) +
                         (vl.z - er[ER_RLINV + 2]) * (vl.z - er[ER_RLINV + 2]);
            const T da = (rv.x - er[ER_RANGV]) * (rv.x - er[ER_RANGV]) + (rv.y - er[ER_RANGV + 1]) * (rv.y - er[ER_RANGV + 1]) +
                         (rv.z - er[ER_RANGV + 2]) * (rv.z - er[ER_RANGV + 2]);
            rv_r = t_exp<T>(-(T)w.k_rl * dl - (T)w.k_ra * da);
            const Q4<T> rc{cq[3], cq[4], cq[5], cq[6]};
            const Q4<T> rq = de_heading(rc);
            const Q4<T> erq{er[ER_RQ], er[ER_RQ + 1], er[ER_RQ + 2], er[ER_RQ + 3]};
            const T dq = half_angle<T>(qmul(rq, qinv(erq)));
            const T dh = cq[2] - er[ER_Z];
            rp_r = t_exp<T>(-(T)w.k_rh * dh * dh - (T)w.k_rq * dq * dq);
        }
        s_rp[threadIdx.x] = rp_r;
        s_rv[threadIdx.x] = rv_r;
    } else if (threadIdx.x >= 64) {
        for (int i = threadIdx.x - 64; i < 5 * tile; i += 192) {
            const int el = i / 5, k = i - 5 * el;
            const long env = env0 + el;
            T ee_sq = T(0);
            if (env < n && (!active || active[env])) {
                const T *cq = cur_qpos + env * m.nq;
                const T *er = expert_rows + (long)frame[env] * EGP_EXPERT_ROW;
                const V3<T> o = ee_local<T>(cq, ee_wpos + env * 15 + 3 * k, m.obs_root != 0);
                const T *ee = er + ER_EE + 3 * k;
                ee_sq = (o.x - ee[0]) * (o.x - ee[0]) + (o.y - ee[1]) * (o.y - ee[1]) + (o.z - ee[2]) * (o.z - ee[2]);
            }
            s_ee[el][k] = ee_sq;
        }
    }
    __syncthreads();
    if (threadIdx.x < tile) {
        const long env = env0 + threadIdx.x;
        if (env >= n) return;
        T *ci = cinfo + env * 5;
        if (active && !active[env]) {
            reward[env] = T(0);
            for (int k = 0; k < 5; ++k) ci[k] = T(0);
            return;
        }
        const T ee_sq = (((s_ee[threadIdx.x][0] + s_ee[threadIdx.x][1]) + s_ee[threadIdx.x][2]) + s_ee[threadIdx.x][3]) + s_ee[threadIdx.x][4];
        T pose_sum = T(0), vel_sq = T(0);
        for (int bl = 0; bl < per; ++bl) { pose_sum += s_bp[threadIdx.x][bl]; vel_sq += s_bv[threadIdx.x][bl]; }
        const T pose_r = t_exp<T>(-(T)w.k_p * pose_sum);
        if (w.v_ord != 2.0) vel_sq = t_pow<T>(vel_sq, T(2) / (T)w.v_ord);
        const T vel_r = t_exp<T>(-(T)w.k_v * vel_sq);
        const T ee_r = t_exp<T>(-(T)w.k_e * ee_sq);
        const T rp_r = s_rp[threadIdx.x], rv_r = s_rv[threadIdx.x];
        T r = ((T)w.w_p * pose_r + (T)w.w_v * vel_r + (T)w.w_e * ee_r + (T)w.w_rp * rp_r + (T)w.w_rv * rv_r) / (T)w.w_sum;
        if (w.decay) r *= T(1) - (T)tcur[env] / (T)w.episode_len;
        if (endf[env]) r += end_reward;
        reward[env] = r;
        ci[0] = pose_r; ci[1] = vel_r; ci[2] = ee_r; ci[3] = rp_r; ci[4] = rv_r;
    }
}

// (four workgroups per CU: the multi-pass float64 form needed 132 VGPRs -- one granule past 128 -- and ran three waves per SIMD;
//  VALUBusy 63 % at 1 M envs said the fourth was missing)
template <typename T, int PASSES>
__global__ __launch_bounds__(256, 4) void k_reward_quat_v3(DevModel m, RewardW w, const T *__restrict__ expert_rows,
                                                        const T *__restrict__ cur_qpos, const T *__restrict__ prev_qpos,
                                                        const T *__restrict__ ee_wpos, const int *__restrict__ tcur,
                                                        const int *__restrict__ frame, const int *__restrict__ endf,
                                                        const int *__restrict__ active, T end_reward, int n,
                                                        T *__restrict__ reward, T *__restrict__ cinfo) {
    reward_body<T, PASSES>(m, w, expert_rows, cur_qpos, prev_qpos, ee_wpos, tcur, frame, endf, active, end_reward, n, reward, cinfo, blockIdx.x);
}

// ============================================================================================ K7
// Expert / learner pose features of one frame pair, reference formats (gen_expert.py:28-83 keys):
//   qvel[nv] = get_qvel_fd(prev, cur, dt) (world-frame root lin-vel), rlinv_local[3], rangv[3], rq_rmh[4],
//   ee_pos[15], bquat[4*nbody], bangvel[3*nbody].
// expert_convention != 0: rlinv_local uses the heading of the CURRENT root quat (gen_expert.py:53) and the heading
// frame whatever cfg.obs_coord says (gen_expert.py:18-22 builds its own config with obs_coord 'heading');
// otherwise the previous root quat and cfg.obs_coord's frame (get_qvel_fd(..., cfg.obs_coord) and
// get_ee_pos(cfg.obs_coord) as in the reward, reward_function.py:19-23).
template <typename T>
__global__ __launch_bounds__(256) void k_pose_features(DevModel m, const T *__restrict__ cur_qpos,
                                                       const T *__restrict__ prev_qpos, const T *__restrict__ ee_wpos,
                                                       int n, int expert_convention, T *__restrict__ qvel,
                                                       T *__restrict__ rlinv_local, T *__restrict__ rangv,
                                                       T *__restrict__ rq_rmh, T *__restrict__ ee_pos,
                                                       T *__restrict__ bquat, T *__restrict__ bangvel) {
    __shared__ int s_start[EGP_MAX_BODY], s_ndof[EGP_MAX_BODY];
    if (threadIdx.x < m.nbody) {
        s_start[threadIdx.x] = m.body_qpos_start[threadIdx.x];
        s_ndof[threadIdx.x] = m.body_ndof[threadIdx.x];
    }
    __syncthreads();
    const long env = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    if (env >= n) return;
    const T *cq = cur_qpos + env * m.nq;
    const T *pq = prev_qpos + env * m.nq;
    const T dt = (T)m.dt;
    if (l >= 1 && l < m.nbody) {
        Q4<T> qc; V3<T> om;
        body_feature<T>(cq, pq, s_start[l], s_ndof[l], dt, &qc, &om);
        T *bq = bquat + env * 4 * m.nbody + 4 * l;
        bq[0] = qc.w; bq[1] = qc.x; bq[2] = qc.y; bq[3] = qc.z;
        T *bv = bangvel + env * 3 * m.nbody + 3 * l;
        bv[0] = om.x; bv[1] = om.y; bv[2] = om.z;
        // hinge velocities of this body
        for (int k = 0; k < s_ndof[l]; ++k) qvel[env * m.nv + s_start[l] - 1 + k] = (cq[s_start[l] + k] - pq[s_start[l] + k]) / dt;
    } else if (l == 0) {
        V3<T> v, rv;
        root_velocity<T>(cq, pq, dt, &v, &rv);
        const Q4<T> rc{cq[3], cq[4], cq[5], cq[6]};
        const Q4<T> rp{pq[3], pq[4], pq[5], pq[6]};
        T *qv = qvel + env * m.nv;
        qv[0] = v.x; qv[1] = v.y; qv[2] = v.z; qv[3] = rv.x; qv[4] = rv.y; qv[5] = rv.z;
        const V3<T> vl = coord_vec(expert_convention ? rc : rp, v, !expert_convention && m.obs_root != 0);
        T *o = rlinv_local + env * 3; o[0] = vl.x; o[1] = vl.y; o[2] = vl.z;
        o = rangv + env * 3; o[0] = rv.x; o[1] = rv.y; o[2] = rv.z;
        const Q4<T> rq = de_heading(rc);
        o = rq_rmh + env * 4; o[0] = rq.w; o[1] = rq.x; o[2] = rq.y; o[3] = rq.z;
        // root entries of bquat / bangvel
        o = bquat + env * 4 * m.nbody; o[0] = rc.w; o[1] = rc.x; o[2] = rc.y; o[3] = rc.z;
        const Q4<T> dv = qmul(rc, qinv(rp));
        V3<T> ax; T ang;
        rot_axis_angle<T>(dv, &ax, &ang);
        o = bangvel + env * 3 * m.nbody; o[0] = ax.x * ang / dt; o[1] = ax.y * ang / dt; o[2] = ax.z * ang / dt;
    } else if (l < m.nbody + 5) {
        const int k = l - m.nbody;
        const V3<T> e = ee_local<T>(cq, ee_wpos + env * 15 + 3 * k, !expert_convention && m.obs_root != 0);
        T *o = ee_pos + env * 15 + 3 * k; o[0] = e.x; o[1] = e.y; o[2] = e.z;
    }
}

// ============================================================================================ K6
// RunningStat / ZFilter (utils/zfilter.py:7-67), batched.
// partial layout per tile p: ws[p*(1+2*dim)] = count, then mean[dim], then M2[dim]  (float64)
// (ZfSrc, zf_merge_column: egp_filter_dev.hpp)

// tile statistics: a workgroup is G = blockDim / 128 row groups x 128 columns. A thread takes its group's share of the
// tile's rows in chunks of 8 (8 independent loads in flight, then an exact two-pass mean / M2 of the chunk, Chan-merged
// into the thread's running statistics); the groups' results meet in LDS and group 0 merges them in row order. With
// 1 024 threads and 64-row tiles every thread has ONE round of loads and a 512-env tick leaves 8 partials -- few
// enough for k_zf_apply to merge them itself (one launch and its memory round trips less on the tick's critical path).
// ROWS: the source is the drained state (src.x == nullptr) and takes ZfSrc's rows form; else src.at() per element
// (two instantiations, two kernels: one function with both paths needs more than the 128 registers of a 1 024-thread workgroup)
template <typename T, bool ROWS>
__device__ __forceinline__ void zf_partial_body(const ZfSrc<T> &src, const int *__restrict__ active, int n, int dim,
                                                int rows_per_tile, double *__restrict__ ws, int p) {
    __shared__ double s_mean[8][128], s_m2[8][128], s_cnt[8];
    const int G = blockDim.x >> 7, g = threadIdx.x >> 7, lc = threadIdx.x & 127;
    const int r0 = p * rows_per_tile, r1 = min(n, r0 + rows_per_tile);
    // row group g takes the tile's 8-row chunks g, g + G, g + 2 G, ...: at any time the workgroup reads 8 G consecutive rows (with a
    // contiguous share per group -- 256 rows each in a 2 048-row tile -- a workgroup kept 16 streams a quarter of a megabyte apart
    // going; 64-row tiles, the rollout's, have one chunk per group either way: same partials bit for bit)
    const int g0 = r0 + 8 * g, g1 = r1, g_step = 8 * G;
    double *out = ws + (long)p * (1 + 2 * dim);
    for (int cb = 0; cb < dim; cb += 128) {
        const int c = cb + lc;
        double cnt = 0.0, mean = 0.0, m2 = 0.0;
        // (rows of the drained state: every load of the chunk's 8 rows in flight at once and the rows' quaternion work shared
        //  out over the wave's lanes -- ZfSrc::load_rows / finish_rows, wave-uniform here: a wave is 64 columns of ONE row group;
        //  with src.at() per row the chunk was 8 dependent rounds of flag -> branch -> loads -> arithmetic)
        if (ROWS || c < dim)
            for (int rb = g0; rb < g1; rb += g_step) {
                double v[8];
                unsigned on = 0;          // bit i: row rb + i counts
                int k = 0;
                if constexpr (ROWS) {
                    // (two half-chunks of 4 rows: the 8-row form needs more than the 128 registers a thread of this workgroup has)
                    const int cl = min(c, dim - 1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        typename ZfSrc<T>::template Rows<4> P = src.template load_rows<4>(rb + 4 * h, n, cl);
                        int flag[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) flag[i] = 1;
                        if (active) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) flag[i] = active[min(rb + 4 * h + i, n - 1)];
                        }
                        src.template finish_rows<4>(P, rb + 4 * h, n, cl);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool o_ = rb + 4 * h + i < g1 && flag[i] && c < dim;
                            on |= (unsigned)o_ << (4 * h + i);
                            v[4 * h + i] = o_ ? (double)P.own[i] : 0.0;
                            k += o_;
                        }
                    }
                } else {
                    // (requesting the next chunk before this one's arithmetic was tried in round 6: the 1 024-thread workgroup's 128
                    //  registers do not hold it -- 305 -> 400 us at 1 M rows, 16 -> 34 us at 1 024)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int r = rb + i;
                        const bool o_ = r < g1 && (!active || active[r]);
                        on |= (unsigned)o_ << i;
                        v[i] = o_ ? (double)src.at(r, c) : 0.0;
                        k += o_;
                    }
                }
                if (k == 0) continue;
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[i];
                const double mb = sum / k;
                double sb = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) sb += (on >> i) & 1 ? (v[i] - mb) * (v[i] - mb) : 0.0;
                if (cnt == 0.0) {
                    cnt = k; mean = mb; m2 = sb;
                } else {
                    const double tot = cnt + k, d = mb - mean;
                    m2 += sb + d * d * (cnt * k / tot);
                    mean += d * (k / tot);
                    cnt = tot;
                }
            }
        s_mean[g][lc] = mean;
        s_m2[g][lc] = m2;
        if (lc == 0 && cb == 0) s_cnt[g] = cnt;          // (the count is the same for every column)
        __syncthreads();
        if (g == 0 && c < dim) {
            double C = s_cnt[0], M = s_mean[0][lc], S = s_m2[0][lc];
            for (int j = 1; j < G; ++j) {
                const double nb = s_cnt[j];
                if (nb > 0.0) {
                    if (C == 0.0) {
                        C = nb; M = s_mean[j][lc]; S = s_m2[j][lc];
                    } else {
                        const double d = s_mean[j][lc] - M, tot = C + nb;
                        S = S + s_m2[j][lc] + d * d * (C * nb / tot);
                        M = M + d * (nb / tot);
                        C = tot;
                    }
                }
            }
            out[1 + c] = M;
            out[1 + dim + c] = S;
            if (c == 0) out[0] = C;
        }
        __syncthreads();
    }
}

template <typename T, bool ROWS>
__global__ __launch_bounds__(1024) void k_zf_partial(ZfSrc<T> src, const int *__restrict__ active, int n, int dim,
                                                    int rows_per_tile, double *__restrict__ ws) {
    zf_partial_body<T, ROWS>(src, active, n, dim, rows_per_tile, ws, blockIdx.x);
}
template <typename T>
static void launch_zf_partial(const ZfSrc<T> &src, const int *active, int n, int dim, int rpt, int nt, double *ws, hipStream_t stream) {
    if (src.x == nullptr)
        k_zf_partial<T, true><<<dim3(nt), dim3(1024), 0, stream>>>(src, active, n, dim, rpt, ws);
    else
        k_zf_partial<T, false><<<dim3(nt), dim3(1024), 0, stream>>>(src, active, n, dim, rpt, ws);
}

// one block: the merged state of a batch with many tiles (few tiles: k_zf_apply merges them itself). 8 tile groups x 128
// columns: group g merges its contiguous eighth of the tiles in order, then group 0 merges the eight results in order
// through LDS -- eight times fewer dependent rounds of loads than one pass over all tiles (90 -> 15 us for 512 tiles).
// Large batches (round 3): ONE block merging 512 partials took 31 of K6's 70 us at 65 536 rows. Now gridDim.x blocks each
// merge a contiguous share of the tiles into a partial record of their own (st_in == nullptr: block b writes record b of
// st_out, layout as ws) and the apply kernel merges those <= 16 records into the running state itself, as it does for small batches.
__global__ __launch_bounds__(1024) void k_zf_merge(int dim, int n_tiles_all, const double *__restrict__ ws_all,
                                                   const double *__restrict__ st_in, double *__restrict__ st_out_all) {
    __shared__ double s_n[8], s_mean[8][128], s_S[8][128];
    const int g = threadIdx.x >> 7, lc = threadIdx.x & 127;
    const int share = (n_tiles_all + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * share, n_tiles = max(0, min(n_tiles_all, t0 + share) - t0);
    const double *ws = ws_all + (long)t0 * (1 + 2 * dim);
    double *st_out = st_out_all + (st_in ? 0 : (long)blockIdx.x * (1 + 2 * dim));
    const int per = (n_tiles + 7) / 8, q0 = g * per, q1 = min(n_tiles, q0 + per);
    for (int cb = 0; cb < dim; cb += 128) {
        const int c = cb + lc;
        double cnt = 0.0, mean = 0.0, S = 0.0;
        if (c < dim && q0 < q1) {
            for (int qa = q0; qa < q1; qa += 8) {
                double nb[8], mb[8], Sb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = qa + i;
                    const double *pp = ws + (long)(q < q1 ? q : q0) * (1 + 2 * dim);
                    nb[i] = q < q1 ? pp[0] : 0.0;
                    mb[i] = pp[1 + c];
                    Sb[i] = pp[1 + dim + c];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (nb[i] > 0.0) {
                        if (cnt == 0.0) {
                            cnt = nb[i]; mean = mb[i]; S = Sb[i];
                        } else {
                            const double d = mb[i] - mean, tot = cnt + nb[i];
                            S = S + Sb[i] + d * d * (cnt * nb[i] / tot);
                            mean = mean + d * (nb[i] / tot);
                            cnt = tot;
                        }
                    }
                }
            }
        }
        s_mean[g][lc] = mean; s_S[g][lc] = S;
        if (lc == 0 && cb == 0) s_n[g] = cnt;            // (the count is the same for every column)
        __syncthreads();
        if (g == 0 && c < dim) {
            double C = st_in ? st_in[0] : 0.0, M = st_in ? st_in[1 + c] : 0.0, SS = st_in ? st_in[1 + dim + c] : 0.0;
            for (int j = 0; j < 8; ++j) {
                const double nb = s_n[j];
                if (nb > 0.0) {
                    if (C == 0.0) {
                        C = nb; M = s_mean[j][lc]; SS = s_S[j][lc];
                    } else {
                        const double d = s_mean[j][lc] - M, tot = C + nb;
                        SS = SS + s_S[j][lc] + d * d * (C * nb / tot);
                        M = M + d * (nb / tot);
                        C = tot;
                    }
                }
            }
            st_out[1 + c] = M;
            st_out[1 + dim + c] = SS;
            if (c == 0) st_out[0] = C;
        }
        __syncthreads();
    }
}

// y = clip((x - mean) / (std + 1e-8)) with the statistics in `st` (identity: raw copy)
template <typename T>
// `ws` != nullptr: `st` is the state BEFORE this batch and every block merges the batch's n_tiles partials into it for
// itself (same operations in the same order: identical in every block); block 0 writes the new state to st_out
__global__ __launch_bounds__(128) void k_zf_apply(ZfSrc<T> src, int n, int dim, int rows_per_block, const double *__restrict__ st,
                                                  double clip, T *__restrict__ y, T *__restrict__ y2,
                                                  const int *__restrict__ write_mask, int identity,
                                                  const double *__restrict__ ws, int n_tiles, double *__restrict__ st_out) {
    extern __shared__ double s_ms[];   // mean[dim], inv[dim]
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        if (identity) { s_ms[c] = 0.0; s_ms[dim + c] = 1.0; continue; }
        double cnt, mean, S;
        if (ws) {
            zf_merge_column(dim, n_tiles, ws, st, c, cnt, mean, S);
            if (blockIdx.x == 0) {
                st_out[1 + c] = mean;
                st_out[1 + dim + c] = S;
                if (c == 0) st_out[0] = cnt;
            }
        } else {
            cnt = st[0]; mean = st[1 + c]; S = st[1 + dim + c];
        }
        const double var = cnt > 1.0 ? S / (cnt - 1.0) : mean * mean;
        s_ms[c] = mean;
        s_ms[dim + c] = 1.0 / (sqrt(var) + 1e-8);
    }
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    if (src.x == nullptr && dim <= (int)blockDim.x) {
        // rows of the drained state, one column per thread: ZfSrc's rows form (every load of 4 rows in flight at once, the rows'
        // quaternion work shared out over the wave's lanes) instead of one src.at() -- branch, loads, arithmetic -- per element
        const int c = min((int)threadIdx.x, dim - 1);
        const bool mine = (int)threadIdx.x < dim;
        const double mean = s_ms[c], inv = s_ms[dim + c];
        for (int rb = r0; rb < r1; rb += 4) {
            typename ZfSrc<T>::template Rows<4> P = src.template load_rows<4>(rb, n, c);
            int keep[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) keep[i] = 1;
            if (write_mask) {
#pragma unroll
                for (int i = 0; i < 4; ++i) keep[i] = write_mask[min(rb + i, n - 1)];
            }
            src.template finish_rows<4>(P, rb, n, c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rb + i;
                if (r < r1 && mine && keep[i]) {
                    double v = ((double)P.own[i] - mean) * inv;
                    if (clip > 0.0 && !identity) v = fmin(fmax(v, -clip), clip);
                    const long e = (long)r * dim + c;
                    y[e] = (T)v;
                    if (y2) y2[e] = (T)v;
                }
            }
        }
        return;
    }
    if (src.x != nullptr && dim <= (int)blockDim.x && !write_mask) {
        // a plain matrix, one column per thread, eight rows' loads in flight: no per-element division by `dim` (the flat loop below
        // spent 64-bit e / dim and e % dim on every element: 603 us over 1 M rows of 115 where this form takes ~400)
        const int c = threadIdx.x;
        if (c >= dim) return;
        const double mean = s_ms[c], inv = s_ms[dim + c];
        const bool clamp = clip > 0.0 && !identity;
        for (int rb = r0; rb < r1; rb += 8) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = rb + i < r1 ? (double)src.x[(long)(rb + i) * dim + c] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (rb + i >= r1) break;
                double w = (v[i] - mean) * inv;
                if (clamp) w = fmin(fmax(w, -clip), clip);
                const long e = (long)(rb + i) * dim + c;
                y[e] = (T)w;
                if (y2) y2[e] = (T)w;
            }
        }
        return;
    }
    const long e0 = (long)r0 * dim, e1 = (long)r1 * dim;
    for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const int c = e % dim;
        const long r = e / dim;
        if (write_mask && !write_mask[r]) continue;
        double v = ((double)src.at(r, c) - s_ms[c]) * s_ms[dim + c];
        if (clip > 0.0 && !identity) v = fmin(fmax(v, -clip), clip);
        y[e] = (T)v;
        if (y2) y2[e] = (T)v;
    }
}

// ============================================================================================ K5
// estimate_advantages (core/common.py:5-25):  a_i = delta_i + (gamma*tau*m_i) a_{i+1},
// delta_i = r_i + gamma*m_i*v_{i+1} - v_i   (v_N = a_N = 0; one reverse sweep over the flat batch).
// Affine recurrence -> chunk summaries (P, A) -> block scan over chunks -> per-chunk replay.
// (round 3: 8 samples per thread instead of 32 -- a workgroup then covers 2 048 samples and keeps 33 kB of LDS, so a 1.6 M-sample
//  batch is 800 workgroups, four per CU, instead of 200 on 256 CUs, and the update's 135 k samples 66 instead of 17)
constexpr int GAE_CHUNK = 4;

template <typename T>
__device__ __forceinline__ void gae_coeffs(const T *r, const T *mk, const T *v, int i, int n, double gamma, double gt,
                                           double &delta, double &c) {
    const double mi = (double)mk[i];
    const double vn = i + 1 < n ? (double)v[i + 1] : 0.0;
    delta = (double)r[i] + gamma * vn * mi - (double)v[i];
    c = gt * mi;
}

// Staging. A thread owns a chunk of GAE_CHUNK CONSECUTIVE samples (the recurrence is sequential), so its own loads and stores
// would touch 64 different cache lines per wave instruction. The block's samples therefore pass through LDS: coalesced
// loads form (delta_i, c_i) straight away -- element e of the block at s[e + e / GAE_CHUNK], one pad word per chunk so that the
// lanes of a wave, GAE_CHUNK + 1 words apart, spread over the banks -- and the replay's results leave the same way (round 3: K5
// at 1.6 M samples 142 -> see docs/DESIGN_TRAIL.md section 4).
constexpr int GAE_BLOCK_ELEMS = 256 * GAE_CHUNK;
constexpr int GAE_LDS_DOUBLES = 2 * (GAE_BLOCK_ELEMS + GAE_BLOCK_ELEMS / GAE_CHUNK);
__device__ __forceinline__ int gae_pad(int e) { return e + e / GAE_CHUNK; }      // one pad word per chunk: lanes GAE_CHUNK + 1 words apart

template <typename T>
__device__ __forceinline__ void gae_stage(const T *__restrict__ r, const T *__restrict__ mk, const T *__restrict__ v, int n, double gamma,
                                          double gt, double *s_d, double *s_c, int tile) {
    const long base = (long)tile * GAE_BLOCK_ELEMS;
    double d[GAE_CHUNK], c[GAE_CHUNK];
#pragma unroll
    for (int j = 0; j < GAE_CHUNK; ++j) {                 // (every load of the thread in flight before the first LDS store)
        const long i = base + j * 256 + threadIdx.x;
        d[j] = 0.0; c[j] = 1.0;                           // (past the end: the identity map)
        if (i < n) gae_coeffs<T>(r, mk, v, (int)i, n, gamma, gt, d[j], c[j]);
    }
#pragma unroll
    for (int j = 0; j < GAE_CHUNK; ++j) {
        const int e = j * 256 + threadIdx.x;
        s_d[gae_pad(e)] = d[j];
        s_c[gae_pad(e)] = c[j];
    }
    __syncthreads();
}

// ONE pass over the batch (round 6; rounds 2-5: summary -> scan -> replay, which read r, mask, v twice and sent a (P, A) pair per
// 8 samples through memory: 68 B per sample moved for 40 algorithmic). A workgroup = a tile of 2 048 consecutive samples:
//   stage (delta, c) through LDS | compose the tile's 256 chunk maps (suffix scan in LDS) | PUBLISH the tile's map (P, A) |
//   look RIGHT over the tiles after it, composing their published maps in order until the product of the P's is below 2^-200 (an
//   episode end makes it exactly 0; 2 048 samples without one leave (gamma tau)^2048) or the batch ends: the carry entering the tile |
//   replay the chunks with their carries, write adv / ret through LDS, Welford partial of the tile.
// Nothing waits for another tile's RESULT, only for its map, which every tile publishes before it looks anywhere: the look-back is
// short (one tile for every gamma tau < 0.93) and its association is fixed -- bit-reproducible, unlike the classic form that takes
// whichever of (map, inclusive prefix) a predecessor has ready. No fences (a device-scope release writes back the XCD's L2 here,
// docs/DESIGN_TRAIL.md): the two words of a map are their own flags -- the slots are armed with a NaN pattern no map contains
// (one hipMemsetD32Async in front of the launch) and travel as relaxed agent-scope atomics. Tiles go right to left in workgroup-index
// order (see the kernel), so a tile only ever waits for workgroups that were dispatched before it.
constexpr unsigned GAE_ARM32 = 0x7FF8DEADu;                               // both halves of an armed 64-bit slot
constexpr unsigned long long GAE_ARM64 = ((unsigned long long)GAE_ARM32 << 32) | GAE_ARM32;

template <typename T>
__global__ __launch_bounds__(256) void k_gae_onepass(const T *__restrict__ r, const T *__restrict__ mk, const T *__restrict__ v, int n,
                                                     double gamma, double gt, unsigned long long *maps, T *__restrict__ adv,
                                                     T *__restrict__ ret, double *__restrict__ part) {
    extern __shared__ double s_gae[];
    double *s_d = s_gae, *s_c = s_gae + GAE_LDS_DOUBLES / 2;
    __shared__ double s_n[256], s_mean[256], s_m2[256];
    __shared__ double s_bc;
    const int t = threadIdx.x, nb = gridDim.x;
    // Tiles right to left in workgroup-index order: a tile only waits for maps of workgroups with a LOWER index, and the dispatcher
    // hands workgroups out in index order (round-robin over the XCDs, in order within each), so whoever it waits for is on the chip
    // or done. (A ticket counter makes that independent of the dispatcher -- and cost 30 ns per workgroup, serialised on one
    // address across eight L2s: 24 of the 36 us at 800 tiles, 390 us at 12 800.) The wait below is bounded all the same.
    const int tile = nb - 1 - (int)blockIdx.x;
    const int i0 = (tile * 256 + t) * GAE_CHUNK;
    gae_stage<T>(r, mk, v, n, gamma, gt, s_d, s_c, tile);
    double P = 1.0, A = 0.0;                              // this thread's chunk as a map of the carry entering it (past the end: identity)
    if (i0 < n) {
        const int i1 = min(n, i0 + GAE_CHUNK);
        for (int i = i1 - 1; i >= i0; --i) {
            const int e = gae_pad(t * GAE_CHUNK + (i - i0));
            const double d = s_d[e], c = s_c[e];
            A = d + c * A;
            P = c * P;
        }
    }
    // suffix scan: (s_n, s_mean)[t] = f_t o f_{t+1} o ... o f_255 (the sweep runs right to left; the arrays are rewritten further down):
    // inside a wave by shuffles, then the waves to its right folded in (two barriers instead of sixteen)
    {
        const int lane = t & 63, w = t >> 6;
        double sP_ = P, sA_ = A;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double pP = __shfl_down(sP_, off), pA = __shfl_down(sA_, off);
            if (lane + off < 64) {
                sA_ = sA_ + sP_ * pA;
                sP_ = sP_ * pP;
            }
        }
        if (lane == 0) { s_m2[w] = sP_; s_m2[4 + w] = sA_; }           // the wave's whole map
        __syncthreads();
        double rP = 1.0, rA = 0.0;                                     // waves w + 1 .. 3 composed left to right
        for (int k = w + 1; k < 4; ++k) {
            rA = rA + rP * s_m2[4 + k];
            rP = rP * s_m2[k];
        }
        s_mean[t] = sA_ + sP_ * rA;
        s_n[t] = sP_ * rP;
        __syncthreads();
    }
    if (t == 0) {
        __hip_atomic_store(maps + 2 * (size_t)tile, (unsigned long long)__double_as_longlong(s_n[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(maps + 2 * (size_t)tile + 1, (unsigned long long)__double_as_longlong(s_mean[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < 64) {                                         // wave 0 looks right, 64 tiles per round, composing strictly in tile order
        double Pacc = 1.0, Aacc = 0.0;
        int next = tile + 1;
        bool done = next >= nb;
        const long long t_wait0 = wall_clock64();
        while (!done) {
            const int j = next + t;
            unsigned long long pb = GAE_ARM64, ab = GAE_ARM64;
            bool have = j >= nb;
            if (!have) {
                pb = __hip_atomic_load(maps + 2 * (size_t)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ab = __hip_atomic_load(maps + 2 * (size_t)j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                have = pb != GAE_ARM64 && ab != GAE_ARM64;
            }
            const unsigned long long ok = __ballot(have);
            const int avail = ok == ~0ull ? 64 : __ffsll((long long)~ok) - 1;      // the leading run of published tiles
            for (int l = 0; l < avail && !done; ++l) {
                if (next + l >= nb) { done = true; break; }
                const double Pj = __longlong_as_double((long long)__shfl(pb, l)), Aj = __longlong_as_double((long long)__shfl(ab, l));
                Aacc = Aacc + Pacc * Aj;
                Pacc = Pacc * Pj;
                if (fabs(Pacc) < 0x1p-200) done = true;
            }
            if (!done) {
                next += avail;
                done = next >= nb;
                if (avail == 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t_wait0 > 200000000ll) {      // 2 s without a neighbour's map: give up loudly (NaN), never hang
                        Aacc = __longlong_as_double((long long)GAE_ARM64);
                        done = true;
                    }
                }
            }
        }
        if (t == 0) s_bc = Aacc;                          // a at the first sample of the next tile (0 behind the last tile)
    }
    __syncthreads();
    const double bc = s_bc;
    const double carry_in = t < 255 ? s_mean[t + 1] + s_n[t + 1] * bc : bc;
    __syncthreads();
    double cnt = 0.0, mean = 0.0, m2 = 0.0;
    if (i0 < n) {
        const int len = min(n, i0 + GAE_CHUNK) - i0;
        double a = carry_in, x[GAE_CHUNK], sum = 0.0;
#pragma unroll
        for (int k = GAE_CHUNK - 1; k >= 0; --k) {
            x[k] = 0.0;
            if (k < len) {
                const int e = gae_pad(t * GAE_CHUNK + k);
                a = s_d[e] + s_c[e] * a;
                s_d[e] = a;                               // leaves through the coalesced pass below
                x[k] = (double)(T)a;                      // statistics of the stored (rounded) advantage
                sum += x[k];
            }
        }
        // the chunk's (n, mean, M2) by two passes over its registers (a Welford step per sample cost a float64 division each)
        cnt = (double)len;
        mean = len == GAE_CHUNK ? sum * (1.0 / GAE_CHUNK) : sum / cnt;
#pragma unroll
        for (int k = 0; k < GAE_CHUNK; ++k)
            if (k < len) { const double dl = x[k] - mean; m2 = fma(dl, dl, m2); }
    }
    __syncthreads();
    {
        const long base = (long)tile * GAE_BLOCK_ELEMS;
#pragma unroll 4
        for (int j = 0; j < GAE_CHUNK; ++j) {
            const int e = j * 256 + t;
            const long i = base + e;
            if (i < n) {
                const double a = s_d[gae_pad(e)];
                adv[i] = (T)a;
                ret[i] = (T)((double)v[i] + a);
            }
        }
    }
    s_n[t] = cnt; s_mean[t] = mean; s_m2[t] = m2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) {
            const double na = s_n[t], nbb = s_n[t + off];
            if (nbb > 0.0) {
                const double tot = na + nbb, d = s_mean[t + off] - s_mean[t];
                s_m2[t] += s_m2[t + off] + d * d * (na * nbb / tot);
                s_mean[t] += d * (nbb / tot);
                s_n[t] = tot;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        part[tile * 3 + 0] = s_n[0];
        part[tile * 3 + 1] = s_mean[0];
        part[tile * 3 + 2] = s_m2[0];
    }
}

constexpr int GAE_STATS_THREADS = 1024;
// stats = {n, mean, M2}: Chan merge of the tile partials in a fixed order (deterministic): every thread its strided share, then a
// tree over the threads (one thread walking 200 partials took 45 us of dependent divisions)
__global__ __launch_bounds__(GAE_STATS_THREADS) void k_gae_stats(int n_parts, const double *__restrict__ part, double *__restrict__ stats) {
    constexpr int NT = GAE_STATS_THREADS;
    __shared__ double s_n[NT], s_mean[NT], s_m2[NT];
    const int t = threadIdx.x;
    double cnt = 0.0, mean = 0.0, m2 = 0.0;
    // thread t merges the partials t, t + NT, t + 2 NT, ...: a wave's loads cover consecutive partials (contiguous runs per thread
    // made every load instruction touch 64 cache lines, and the one CU this kernel runs on took 75 us over 25 600 partials), eight
    // of a thread's partials in flight per round
    for (int p0 = t; p0 < n_parts; p0 += 8 * NT) {
        double pn[8], pm[8], ps[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = p0 + k * NT;
            pn[k] = p < n_parts ? part[p * 3] : 0.0;
            pm[k] = p < n_parts ? part[p * 3 + 1] : 0.0;
            ps[k] = p < n_parts ? part[p * 3 + 2] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double nb = pn[k];
            if (nb > 0.0) {
                const double tot = cnt + nb, d = pm[k] - mean, w = nb / tot;      // (one division per merge)
                m2 += ps[k] + d * d * (cnt * w);
                mean += d * w;
                cnt = tot;
            }
        }
    }
    s_n[t] = cnt; s_mean[t] = mean; s_m2[t] = m2;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        if ((t & (2 * off - 1)) == 0) {
            const double na = s_n[t], nb = s_n[t + off];
            if (nb > 0.0) {
                const double tot = na + nb, d = s_mean[t + off] - s_mean[t];
                s_m2[t] += s_m2[t + off] + d * d * (na * nb / tot);
                s_mean[t] += d * (nb / tot);
                s_n[t] = tot;
            }
        }
        __syncthreads();
    }
    if (t == 0) { stats[0] = s_n[0]; stats[1] = s_mean[0]; stats[2] = s_m2[0]; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_standardize(T *__restrict__ a, int n, const double *__restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = stats[1];
    const double sd = sqrt(stats[2] / (stats[0] - 1.0));   // unbiased, as torch.std
    a[i] = (T)(((double)a[i] - mean) / sd);
}

}  // namespace egp

// ================================================================================================
//                                           host side / C-ABI
// ================================================================================================
using namespace egp;

template <typename T>
static int dev_copy(egp_ctx *ctx, const T *host, size_t count, const T **out) {
    void *d = nullptr;
    EGP_HIP_CHECK(hipMalloc(&d, count * sizeof(T)));
    ctx->allocs.push_back(d);
    EGP_HIP_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)d;
    return EGP_OK;
}

static int fill_reward(egp_ctx *ctx, const egp_model_desc *d) {
    RewardW &w = ctx->rw;
    w.w_p = d->w_p; w.w_v = d->w_v; w.w_e = d->w_e; w.w_rp = d->w_rp; w.w_rv = d->w_rv;
    w.w_sum = d->w_p + d->w_v + d->w_e + d->w_rp + d->w_rv;
    w.k_p = d->k_p; w.k_v = d->k_v; w.k_e = d->k_e; w.k_rh = d->k_rh; w.k_rq = d->k_rq; w.k_rl = d->k_rl; w.k_ra = d->k_ra;
    w.v_ord = d->v_ord; w.decay = d->decay; w.episode_len = d->episode_len;
    EGP_REQUIRE(d->v_ord >= 1.0, "v_ord must be >= 1");
    EGP_REQUIRE(d->episode_len > 0, "episode_len must be positive");
    return EGP_OK;
}

extern "C" {

const char *egp_last_error(void) { return g_err; }
const char *egp_version(void) { return "egopose_hip 0.2.0 (gfx950)"; }
int64_t egp_abi_sizeof(const char *name) {
    if (!name) return -1;
#define EGP_ABI_SIZE(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T);
    EGP_ABI_SIZE(egp_model_desc) EGP_ABI_SIZE(egp_expert_table) EGP_ABI_SIZE(egp_gemm_desc)
    EGP_ABI_SIZE(egp_dynamics_desc) EGP_ABI_SIZE(egp_mlp_layer) EGP_ABI_SIZE(egp_physics_vtable)
    EGP_ABI_SIZE(egp_surrogate_desc) EGP_ABI_SIZE(egp_engine_desc) EGP_ABI_SIZE(egp_rollout_tick)
    EGP_ABI_SIZE(egp_ppo_loss_desc) EGP_ABI_SIZE(egp_adam_segment) EGP_ABI_SIZE(egp_host_probe_result)
#undef EGP_ABI_SIZE
    return -1;
}
int32_t egp_obs_dim(const egp_ctx *ctx) { return ctx ? ctx->dm.obs_dim : -1; }

int egp_create(const egp_model_desc *d, int device, egp_ctx **out) {
    EGP_REQUIRE(d && out, "desc/out is NULL");
    EGP_REQUIRE(d->nv > 6 && d->nv <= EGP_MAX_NV, "nv must be in (6, 64]");
    EGP_REQUIRE(d->nq == d->nv + 1 && d->nu == d->nv - 6, "expects one free root joint + hinges (nq=nv+1, nu=nv-6)");
    EGP_REQUIRE(d->nbody >= 2 && d->nbody + 5 <= EGP_MAX_BODY, "nbody must be in [2, 27]");
    EGP_REQUIRE(d->nM > 0 && d->nM <= PD_NM_MAX, "nM out of range");
    EGP_REQUIRE(d->body_qpos_start && d->body_ndof && d->dof_parentid && d->dof_Madr && d->ee_body, "skeleton table is NULL");
    EGP_REQUIRE(d->jkp && d->jkd && d->a_ref && d->a_scale && d->torque_lim && d->b_diffw, "gain table is NULL");
    EGP_REQUIRE(d->sub_dt > 0 && d->frame_skip > 0, "timestep/frame_skip must be positive");
    EGP_REQUIRE(26 + 4 * (d->nbody - 1) <= 106 && 106 + 3 * (d->nbody - 1) <= EGP_EXPERT_ROW, "expert row overflow");
    EGP_HIP_CHECK(hipSetDevice(device));
    egp_ctx *ctx = new egp_ctx();
    ctx->device = device;
    ctx->frame_skip = d->frame_skip;
    int rc = fill_reward(ctx, d);
    if (rc != EGP_OK) { delete ctx; return rc; }
    DevModel &m = ctx->dm;
    m.nq = d->nq; m.nv = d->nv; m.nu = d->nu; m.nbody = d->nbody; m.nM = d->nM;
    m.sub_dt = d->sub_dt;
    m.dt = d->sub_dt * d->frame_skip;
    m.obs_heading = d->obs_heading != 0; m.obs_keep = d->obs_keep_root_heading != 0; m.obs_root = d->obs_coord_root != 0;
    m.obs_vel = d->obs_vel;
    m.action_torque = d->action_torque != 0;
    if (m.obs_vel < 0 || m.obs_vel > 2) { delete ctx; set_error("obs_vel must be 0 (full), 1 (root) or 2 (none)"); return EGP_E_INVALID; }
    m.obs_phase = d->obs_phase != 0;
    m.episode_len = d->episode_len;
    if (m.obs_phase && d->episode_len <= 0) { delete ctx; set_error("obs_phase needs episode_len > 0"); return EGP_E_INVALID; }
    m.obs_dim = obs_width(d->nq, d->nv, m.obs_heading, m.obs_vel, m.obs_phase);
    // sparse-inertia index tables from the dof tree (what mj_fullM walks)
    std::vector<int> rows(d->nM), cols(d->nM);
    std::vector<short> mmap((size_t)d->nv * d->nv, (short)-1);
    int total = 0;
    for (int i = 0; i < d->nv; ++i) {
        int adr = d->dof_Madr[i], j = i;
        while (j >= 0) {
            if (adr < 0 || adr >= d->nM) { delete ctx; set_error("dof_Madr/dof_parentid inconsistent with nM"); return EGP_E_INVALID; }
            rows[adr] = i; cols[adr] = j;
            mmap[(size_t)i * d->nv + j] = (short)adr;
            mmap[(size_t)j * d->nv + i] = (short)adr;
            ++adr; ++total;
            j = d->dof_parentid[j];
        }
    }
    if (total != d->nM) { delete ctx; set_error("dof tree has %d inertia entries, nM says %d", total, d->nM); return EGP_E_INVALID; }
    for (int k = 0; k < 5; ++k) ctx->ee_body.push_back(d->ee_body[k]);
#define EGP_TRY(x) do { rc = (x); if (rc != EGP_OK) { egp_destroy(ctx); return rc; } } while (0)
    EGP_TRY(dev_copy<int>(ctx, d->body_qpos_start, d->nbody, &m.body_qpos_start));
    EGP_TRY(dev_copy<int>(ctx, d->body_ndof, d->nbody, &m.body_ndof));
    EGP_TRY(dev_copy<int>(ctx, rows.data(), d->nM, &m.m_row));
    EGP_TRY(dev_copy<int>(ctx, cols.data(), d->nM, &m.m_col));
    EGP_TRY(dev_copy<short>(ctx, mmap.data(), mmap.size(), &m.m_map));
    EGP_TRY(dev_copy<double>(ctx, d->jkp, d->nu, &m.jkp));
    EGP_TRY(dev_copy<double>(ctx, d->jkd, d->nu, &m.jkd));
    EGP_TRY(dev_copy<double>(ctx, d->a_ref, d->nu, &m.a_ref));
    EGP_TRY(dev_copy<double>(ctx, d->a_scale, d->nu, &m.a_scale));
    EGP_TRY(dev_copy<double>(ctx, d->torque_lim, d->nu, &m.torque_lim));
    EGP_TRY(dev_copy<double>(ctx, d->b_diffw, d->nbody - 1, &m.b_diffw));
#undef EGP_TRY
    // fast paths: 0 = tree-ordered elimination (needs the compiled-in humanoid dof tree; one substep per launch: the lane-grid
    // kernel, 3 = its lane-per-row predecessor), 2 = dense in-register Gauss-Jordan (any tree with nv == 58), 1 = generic LDS kernel
    bool tree_ok = d->nv == Tree58::NV;
    for (int i = 0; tree_ok && i < d->nv; ++i) tree_ok = d->dof_parentid[i] == Tree58::PARENT[i];
    ctx->tree58 = tree_ok;
    if (tree_ok) {          // gather table of the lane-grid K1 (k_pd_torque_grid58)
        const std::vector<unsigned short> off = grid58_offsets(mmap);
        const unsigned short *dev = nullptr;
        rc = dev_copy<unsigned short>(ctx, off.data(), off.size(), &dev);
        if (rc != EGP_OK) { egp_destroy(ctx); return rc; }
        ctx->pd_grid_off = dev;
        ctx->pd_grid = true;
    }
    ctx->pd_variant = tree_ok ? 0 : (d->nv == PD_NV ? 2 : 1);
    *out = ctx;
    return EGP_OK;
}

int egp_destroy(egp_ctx *ctx) {
    if (!ctx) return EGP_OK;
    for (void *p : ctx->allocs) (void)hipFree(p);
    delete ctx;
    return EGP_OK;
}

int egp_set_reward_weights(egp_ctx *ctx, const egp_model_desc *d) {
    EGP_REQUIRE(ctx && d, "ctx/desc is NULL");
    return fill_reward(ctx, d);
}

int egp_set_pd_variant(egp_ctx *ctx, int variant) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(variant == 1 || (variant == 2 && ctx->dm.nv == PD_NV) || ((variant == 0 || variant == 3) && ctx->tree58),
                "variants 0 and 3 need the humanoid_1205_v1 dof tree, variant 2 needs nv == 58");
    ctx->pd_variant = variant == 3 ? 0 : variant;
    ctx->pd_grid = variant == 0;
    return EGP_OK;
}

int egp_upload_experts(egp_ctx *ctx, const egp_expert_table *t) {
    EGP_REQUIRE(ctx && t, "ctx/table is NULL");
    EGP_REQUIRE(t->n_takes > 0 && t->n_frames > 0 && t->take_offset, "empty expert table");
    EGP_REQUIRE(t->qpos && t->rlinv_local && t->rangv && t->rq_rmh && t->ee_pos && t->bquat && t->bangvel, "expert column is NULL");
    EGP_REQUIRE(t->take_offset[t->n_takes] == t->n_frames, "take_offset[n_takes] != n_frames");
    const int nb = ctx->dm.nbody, nq = ctx->dm.nq;
    std::vector<double> rows((size_t)t->n_frames * EGP_EXPERT_ROW, 0.0);
    for (int f = 0; f < t->n_frames; ++f) {
        double *r = rows.data() + (size_t)f * EGP_EXPERT_ROW;
        r[ER_Z] = t->qpos[(size_t)f * nq + 2];
        for (int k = 0; k < 3; ++k) r[ER_RLINV + k] = t->rlinv_local[(size_t)f * 3 + k];
        for (int k = 0; k < 3; ++k) r[ER_RANGV + k] = t->rangv[(size_t)f * 3 + k];
        for (int k = 0; k < 4; ++k) r[ER_RQ + k] = t->rq_rmh[(size_t)f * 4 + k];
        for (int k = 0; k < 15; ++k) r[ER_EE + k] = t->ee_pos[(size_t)f * 15 + k];
        for (int k = 0; k < 4 * (nb - 1); ++k) r[ER_BQ + k] = t->bquat[(size_t)f * 4 * nb + 4 + k];
        for (int k = 0; k < 3 * (nb - 1); ++k) r[ER_BAV + k] = t->bangvel[(size_t)f * 3 * nb + 3 + k];
    }
    std::vector<float> rows32(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) rows32[i] = (float)rows[i];
    EGP_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->expert_rows_f64) { (void)hipFree(ctx->expert_rows_f64); (void)hipFree(ctx->expert_rows_f32); (void)hipFree(ctx->expert_qpos_f64); }
    EGP_HIP_CHECK(hipMalloc((void **)&ctx->expert_qpos_f64, (size_t)t->n_frames * nq * sizeof(double)));
    EGP_HIP_CHECK(hipMemcpy(ctx->expert_qpos_f64, t->qpos, (size_t)t->n_frames * nq * sizeof(double), hipMemcpyHostToDevice));
    EGP_HIP_CHECK(hipMalloc((void **)&ctx->expert_rows_f64, rows.size() * sizeof(double)));
    EGP_HIP_CHECK(hipMalloc((void **)&ctx->expert_rows_f32, rows.size() * sizeof(float)));
    EGP_HIP_CHECK(hipMemcpy(ctx->expert_rows_f64, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    EGP_HIP_CHECK(hipMemcpy(ctx->expert_rows_f32, rows32.data(), rows32.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->n_takes = t->n_takes;
    ctx->n_frames = t->n_frames;
    return EGP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- launchers
static inline int after_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return EGP_E_HIP;
    }
    return EGP_OK;
}

template <typename T>
static int launch_body_quat(egp_ctx *ctx, const T *qpos, int n, T *bquat, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(qpos && bquat, "NULL pointer");
    const long total = (long)n * ctx->dm.nbody;
    k_body_quat<T><<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(ctx->dm, qpos, n, bquat);
    return after_launch("k_body_quat");
}

// ---------------------------------------------------------------------------------------------------------------
// a7: the quaternion algebra of egp_quat.hpp as a batched entry point of its own (one thread per row). The rollout
// kernels inline these functions; this launch exists so that callers (and the parity tests against the reference's
// utils/transformation.py / utils/math.py vectors) can reach each of them directly.
template <typename T>
__global__ __launch_bounds__(256) void k_quat_op(int op, const T *__restrict__ a, const T *__restrict__ b, int n, T *__restrict__ o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto ldq = [](const T *p) { Q4<T> q; q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3]; return q; };
    auto stq = [](T *p, const Q4<T> &q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; };
    switch (op) {
    case EGP_QUAT_MUL: stq(o + 4 * i, qmul(ldq(a + 4 * i), ldq(b + 4 * i))); break;
    case EGP_QUAT_INV: stq(o + 4 * i, qinv(ldq(a + 4 * i))); break;
    case EGP_QUAT_FROM_EULER_SXYZ: stq(o + 4 * i, q_from_euler_sxyz<T>(a[3 * i], a[3 * i + 1], a[3 * i + 2])); break;
    case EGP_QUAT_HEADING_Q: stq(o + 4 * i, heading_q(ldq(a + 4 * i))); break;
    case EGP_QUAT_DE_HEADING: stq(o + 4 * i, de_heading(ldq(a + 4 * i))); break;
    case EGP_QUAT_TRANSFORM_VEC_ROOT:
    case EGP_QUAT_TRANSFORM_VEC_HEADING: {
        Q4<T> q = ldq(b + 4 * i);
        if (op == EGP_QUAT_TRANSFORM_VEC_HEADING) q = heading_q(q);
        V3<T> v; v.x = a[3 * i]; v.y = a[3 * i + 1]; v.z = a[3 * i + 2];
        const V3<T> r = rotate_T(q, v);
        o[3 * i] = r.x; o[3 * i + 1] = r.y; o[3 * i + 2] = r.z;
        break;
    }
    case EGP_QUAT_ROTATION: {
        V3<T> ax; T an;
        rot_axis_angle(ldq(a + 4 * i), &ax, &an);
        o[4 * i] = ax.x; o[4 * i + 1] = ax.y; o[4 * i + 2] = ax.z; o[4 * i + 3] = an;
        break;
    }
    case EGP_QUAT_DIFF_HALF_ANGLE: o[i] = half_angle(qmul(ldq(a + 4 * i), qinv(ldq(b + 4 * i)))); break;
    default: break;
    }
}

template <typename T>
static int launch_quat_op(int op, const T *a, const T *b, int n, T *out, void *stream) {
    EGP_REQUIRE(op >= 0 && op < EGP_QUAT_N_OPS, "unknown quaternion op");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    const bool two = op == EGP_QUAT_MUL || op == EGP_QUAT_TRANSFORM_VEC_ROOT || op == EGP_QUAT_TRANSFORM_VEC_HEADING ||
                     op == EGP_QUAT_DIFF_HALF_ANGLE;
    EGP_REQUIRE(a && out && (b || !two), "NULL pointer");
    k_quat_op<T><<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(op, a, b, n, out);
    return after_launch("k_quat_op");
}

// The two small entries of the reward registry (ego_pose/core/reward_function.py:63-80), one thread per env:
//   EGP_REWARD_CONSTANT   reward 1.0 (the reference computes 1 + end_reward at an episode's end but returns 1.0), c_info [0]
//   EGP_REWARD_POSE_DIST  d = |expert qpos[2:] - qpos[2:]| of the step's expert frame (HumanoidEnv.get_pose_dist,
//                         humanoid_v1.py:275-280), reward 5 - 3 d (+ end_reward on the last step), c_info [d]
__global__ __launch_bounds__(256) void k_reward_simple(int kind, int nq, const double *__restrict__ qpos, const double *__restrict__ expert_qpos,
                                                       const int *__restrict__ frame, const int *__restrict__ endf, const int *__restrict__ active,
                                                       double end_reward, int n, double *__restrict__ reward, double *__restrict__ cinfo) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || (active && !active[e])) return;
    if (kind == EGP_REWARD_CONSTANT) {
        reward[e] = 1.0;
        cinfo[e] = 0.0;
        return;
    }
    const double *q = qpos + (long)e * nq, *x = expert_qpos + (long)frame[e] * nq;
    double s = 0.0;
    for (int k = 2; k < nq; ++k) {
        const double d = x[k] - q[k];
        s += d * d;
    }
    const double dist = sqrt(s);
    reward[e] = 5.0 - 3.0 * dist + (endf[e] ? end_reward : 0.0);
    cinfo[e] = dist;
}

template <typename T>
static int launch_obs(egp_ctx *ctx, const T *qpos, const T *qvel, const int *phase_t, int n, T *obs, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(qpos && qvel && obs, "NULL pointer");
    EGP_REQUIRE(!ctx->dm.obs_phase || phase_t, "the model has obs_phase: phase_t (the rows' cur_t) is required");
    const long total = (long)n * ctx->dm.obs_dim;
    if (n >= 32768 && ctx->dm.obs_dim <= 128) {        // whole rounds of the chip: the row-streaming form (k_obs_rows); below, launch-bound either way
        k_obs_rows<T><<<dim3((n + OBS_ROWS - 1) / OBS_ROWS), dim3(256), 0, (hipStream_t)stream>>>(ctx->dm, qpos, qvel, phase_t, n, obs);
        return after_launch("k_obs_rows");
    }
    k_obs<T><<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(ctx->dm, qpos, qvel, phase_t, n, obs);
    return after_launch("k_obs");
}

template <typename T>
static int launch_pd(egp_ctx *ctx, const PdLd &ld, const T *qpos, const T *qvel, const T *action, const T *qM, const T *C, int n,
                     T *torque, T *torque_raw, hipStream_t stream, PdDone done = PdDone{nullptr, nullptr, 0}) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    if (ctx->dm.action_torque) {        // action_type 'torque': no state, no inertia
        EGP_REQUIRE(action && torque, "NULL pointer");
        k_torque_direct<T><<<dim3((n + 3) / 4), dim3(256), 0, stream>>>(ctx->dm, ld, action, n, torque, torque_raw, done);
        return after_launch("k_torque_direct");
    }
    EGP_REQUIRE(qpos && qvel && action && qM && C && torque, "NULL pointer");
    if (ctx->pd_variant == 0 && ctx->pd_grid) {
        k_pd_torque_grid58<T><<<dim3((n + 3) / 4), dim3(256), 0, stream>>>(ctx->dm, ld, ctx->pd_grid_off, qpos, qvel, action, qM, C, n, torque, torque_raw, done);
        return after_launch("k_pd_torque_grid58");
    }
    if (ctx->pd_variant == 0) {
        k_pd_torque_tree58<T><<<dim3((n + 3) / 4), dim3(256), 0, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, C, n, torque, torque_raw, done);
        return after_launch("k_pd_torque_tree58");
    }
    if (ctx->pd_variant == 2) {
        k_pd_torque_reg58<T><<<dim3((n + 3) / 4), dim3(256), 0, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, C, n, torque, torque_raw);
        return after_launch("k_pd_torque_reg58");
    }
    const int nv = ctx->dm.nv;
    const size_t lds = ((size_t)nv * (nv + 1) + nv) * sizeof(double);
    k_pd_torque_lds<T><<<dim3(n), dim3(64), lds, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, C, n, torque, torque_raw);
    return after_launch("k_pd_torque_lds");
}

static inline PdLd dense_ld(const egp_ctx *c) { return PdLd{c->dm.nq, c->dm.nv, c->dm.nu, c->dm.nM, c->dm.nv}; }

// engine entry: inputs live in the engine's staging layouts (row strides in doubles)
int egp_launch_pd_torque_strided(egp_ctx *ctx, const double *qpos, long ld_qpos, const double *qvel, long ld_qvel,
                                 const double *bias, long ld_bias, const double *qM, long ld_qM, const double *action,
                                 int32_t n, double *torque, hipStream_t stream, unsigned *done_counter,
                                 unsigned long long *host_flag, unsigned long long seq) {
    PdLd ld{ld_qpos, ld_qvel, ctx->dm.nu, ld_qM, ld_bias};
    PdDone done{ctx->pd_variant == 0 ? done_counter : nullptr, host_flag, seq};
    return launch_pd<double>(ctx, ld, qpos, qvel, action, qM, bias, n, torque, nullptr, stream, done);
}

size_t egp_pd_server_dyn_lds_bytes() {
    return ((sizeof(egp_dyn::DynTables) + 7) / 8 + 4 * (size_t)egp_dyn::DY_ENV_DOUBLES + 4 * 192) * sizeof(double);
}

namespace {
// the resident K1's variants: (device dynamics, envs per wavefront) -> kernel, dynamic LDS
struct ServerKernel { const void *fn; size_t lds; };
ServerKernel server_kernel(bool device_dynamics, int ke) {
    const size_t multi = (size_t)4 * ke * PD_NM_MAX * sizeof(double);
    if (device_dynamics) return {ke == 1 ? reinterpret_cast<const void *>(&k_pd_server_tree58<true>) : nullptr, egp_pd_server_dyn_lds_bytes()};
    switch (ke) {
        case 1: return {reinterpret_cast<const void *>(&k_pd_server_tree58<false>), 0};
        case 2: return {reinterpret_cast<const void *>(&k_pd_server_tree58_multi<2>), multi};
        case 4: return {reinterpret_cast<const void *>(&k_pd_server_tree58_multi<4>), multi};
        default: return {nullptr, 0};
    }
}
int server_kernel_prepare(const ServerKernel &k) {
    if (!k.fn) { set_error("no resident K1 for this (device dynamics, envs per wave) pair"); return EGP_E_INVALID; }
    if (k.lds > 0) {
        hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds);
        if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipFuncSetAttribute(resident K1, %zu B of LDS): %s", k.lds, hipGetErrorString(e)); return EGP_E_HIP; }
    }
    return EGP_OK;
}
}  // namespace

// How many workgroups of the resident K1 (variant: device dynamics, `envs_per_wave`) the chip holds at once: the engine runs the
// resident form only when every workgroup of every group is resident at the same time -- they wait on the host, and a workgroup
// that is not resident cannot answer its slice's go word. The occupancy calculator's figure x the CUs is the data sheet's answer;
// the kernel itself, launched in probe mode (server_residency_probe), gives the one that holds on THIS device as this process
// sees it (EGP_SERVER_PROBE=0: calculator only). 0 when the kernel cannot run at all. Cached per (device, variant).
int egp_pd_server_resident_blocks(int device, bool device_dynamics, int envs_per_wave) {
    static std::mutex mu;
    static std::map<long, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    const long key = ((long)device << 8) | ((long)envs_per_wave << 1) | (device_dynamics ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int result = 0;
    do {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); break; }
        const ServerKernel k = server_kernel(device_dynamics, envs_per_wave);
        if (!k.fn || server_kernel_prepare(k) != EGP_OK) break;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, 256, k.lds) != hipSuccess) { (void)hipGetLastError(); break; }
        result = per_cu * prop.multiProcessorCount;
        const char *pe = getenv("EGP_SERVER_PROBE");
        if (result <= 0 || (pe && atoi(pe) == 0)) break;
        unsigned *d_probe = nullptr;
        if (hipMalloc((void **)&d_probe, 4 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); break; }
        unsigned h[4] = {0, 0, 0, 0};
        bool ok = hipMemset(d_probe, 0, sizeof(h)) == hipSuccess;
        if (ok) {
            PdServe sv{};
            sv.probe = d_probe;
            sv.timeout_ticks = 100000;           // 1 ms: a workgroup that is not on the chip by then is not resident
            DevModel dm{};
            PdLd ld{};
            void *args[] = {&dm, &ld, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sv};
            const double *np = nullptr; double *npw = nullptr; int nn = 0;
            args[2] = &np; args[3] = &np; args[4] = &np; args[5] = &np; args[6] = &np; args[7] = &nn; args[8] = &npw;
            ok = hipLaunchKernel(k.fn, dim3(result), dim3(256), args, k.lds, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
                 hipMemcpy(h, d_probe, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
        }
        (void)hipFree(d_probe);
        if (!ok) { (void)hipGetLastError(); break; }
        if ((int)h[1] > 0 && (int)h[1] < result) result = (int)h[1];
    } while (false);
    cache[key] = result;
    return result;
}

// engine entry for the resident K1 (see k_pd_server_tree58); all flag arrays are device-visible addresses. `envs_per_wave` = 1: the
// one-env kernel (4 envs per workgroup); 2 / 4: k_pd_server_tree58_multi over `n_blocks` workgroups, workgroup b serving the envs
// [block_env0[b], block_env0[b + 1]) (at most 4 * envs_per_wave; block_slice has one entry per workgroup)
int egp_launch_pd_server(egp_ctx *ctx, const double *qpos, long ld_qpos, const double *qvel, long ld_qvel, const double *bias,
                         long ld_bias, const double *qM, long ld_qM, const double *qM_host, const double *action, int32_t n,
                         double *torque, hipStream_t stream, const int *block_slice, const unsigned long long *go,
                         unsigned long long base, int n_sub, int *err, double timeout_s, long long *trace, const double *ee_host,
                         double *out_qpos, double *out_prev_qpos, double *out_qvel, double *out_ee, const int *active, bool device_dynamics,
                         int envs_per_wave, const int *block_env0, int n_blocks) {
    EGP_REQUIRE(ctx && ctx->tree58 && ctx->pd_variant == 0, "the K1 server needs the humanoid tree kernel");
    EGP_REQUIRE(!device_dynamics || ctx->dyn_tables, "device dynamics needs egp_set_dynamics_model on the context");
    EGP_REQUIRE(ee_host && out_qpos && out_prev_qpos && out_qvel && out_ee, "NULL epilogue pointer");
    EGP_REQUIRE(ctx->dm.nq <= 64 && ctx->dm.nv <= 64, "the epilogue moves one state row per wavefront");
    EGP_REQUIRE(qpos && qvel && bias && qM && qM_host && action && torque && block_slice && go && err, "NULL pointer");
    EGP_REQUIRE(n > 0 && n_sub > 0, "n and n_sub must be positive");
    const ServerKernel k = server_kernel(device_dynamics, envs_per_wave);
    EGP_REQUIRE(k.fn, "no resident K1 for this (device dynamics, envs per wave) pair");
    EGP_REQUIRE(envs_per_wave == 1 || ctx->dm.nM < PD_NM_MAX, "the multi-env K1 keeps a zero slot behind the inertia row");
    {
        static std::atomic<int> prepared[2][5];          // (the LDS attribute once per variant; egp_pd_server_resident_blocks set it already)
        std::atomic<int> &pf = prepared[device_dynamics ? 1 : 0][envs_per_wave];
        if (pf.load(std::memory_order_acquire) == 0) {
            const int rc = server_kernel_prepare(k);
            if (rc != EGP_OK) return rc;
            pf.store(1, std::memory_order_release);
        }
    }
    PdLd ld{ld_qpos, ld_qvel, ctx->dm.nu, ld_qM, ld_bias};
    const int poll_sleep = 2;          // s_sleep(1) repeats between two polls of a go word
    const int nq = ctx->dm.nq, nv = ctx->dm.nv;
    // (the engine's state rows are qpos | qvel | bias back to back: read as one contiguous stream; separate arrays: three segments)
    const int row_contig = !device_dynamics && qvel == qpos + nq && bias == qvel + nv && ld_qpos == ld_qvel &&
                           ld_qvel == ld_bias && nq + 2 * nv <= 192 && ld_qpos >= nq + 2 * nv;
    PdServe sv{block_slice, go, base, n_sub, qM_host, const_cast<double *>(qM), err, (long long)(timeout_s * 100.0e6), trace,
               ee_host, out_qpos, out_prev_qpos, out_qvel, out_ee, nq, nv, poll_sleep, active, ctx->dyn_tables,
               device_dynamics ? const_cast<double *>(bias) : nullptr, nullptr, block_env0, row_contig};
    EGP_REQUIRE(envs_per_wave == 1 || (block_env0 && n_blocks > 0), "the multi-env K1 needs the workgroups' env ranges");
    const dim3 grid(envs_per_wave == 1 ? (n + 3) / 4 : n_blocks);
    if (device_dynamics) {
        k_pd_server_tree58<true><<<grid, dim3(256), k.lds, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, bias, n, torque, sv);
    } else if (envs_per_wave == 1) {
        k_pd_server_tree58<false><<<grid, dim3(256), 0, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, bias, n, torque, sv);
    } else if (envs_per_wave == 2) {
        k_pd_server_tree58_multi<2><<<grid, dim3(256), k.lds, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, bias, n, torque, sv);
    } else {
        k_pd_server_tree58_multi<4><<<grid, dim3(256), k.lds, stream>>>(ctx->dm, ld, qpos, qvel, action, qM, bias, n, torque, sv);
    }
    return after_launch("k_pd_server_tree58");
}

template <typename T>
static int launch_reward(egp_ctx *ctx, const T *expert_rows, const T *cur_qpos, const T *prev_qpos, const T *ee_wpos,
                         const int *t, const int *frame, const int *endf, const int *active, double end_reward, int n,
                         T *reward, T *cinfo, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (!expert_rows) { set_error("egp_upload_experts must be called before the reward kernel"); return EGP_E_STATE; }
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(cur_qpos && prev_qpos && ee_wpos && t && frame && endf && reward && cinfo, "NULL pointer");
    // multi-pass tiles (60 envs for the humanoid) once the launch fills the chip several times over, one-pass tiles for
    // rollout-sized batches
    if (n >= 16384) {
        const int tile = reward_tile_envs(ctx->dm.nbody, 5);
        k_reward_quat_v3<T, 5><<<dim3((n + tile - 1) / tile), dim3(256), 0, (hipStream_t)stream>>>(
            ctx->dm, ctx->rw, expert_rows, cur_qpos, prev_qpos, ee_wpos, t, frame, endf, active, (T)end_reward, n, reward, cinfo);
    } else {
        const int tile = reward_tile_envs(ctx->dm.nbody, 1);
        k_reward_quat_v3<T, 1><<<dim3((n + tile - 1) / tile), dim3(256), 0, (hipStream_t)stream>>>(
            ctx->dm, ctx->rw, expert_rows, cur_qpos, prev_qpos, ee_wpos, t, frame, endf, active, (T)end_reward, n, reward, cinfo);
    }
    return after_launch("k_reward_quat_v3");
}

template <typename T>
static int launch_features(egp_ctx *ctx, const T *cur, const T *prev, const T *ee_w, int n, int expert_conv, T *qvel, T *rlinv,
                           T *rangv, T *rq, T *ee, T *bq, T *bav, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(cur && prev && ee_w && qvel && rlinv && rangv && rq && ee && bq && bav, "NULL pointer");
    const long threads = (long)n * 32;
    k_pose_features<T><<<dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(ctx->dm, cur, prev, ee_w, n, expert_conv,
                                                                                         qvel, rlinv, rangv, rq, ee, bq, bav);
    return after_launch("k_pose_features");
}

template <typename T>
static int launch_zfilter_src(const ZfSrc<T> &src, const int *active, int n, int dim, const double *st_in, double *st_out, int update,
                              double clip, T *y, T *y2, const int *write_mask, void *ws, void *stream) {
    const int identity = st_in == nullptr;
    EGP_REQUIRE(n >= 0 && dim > 0 && dim <= 4096, "bad n/dim");
    EGP_REQUIRE(n == 0 || y, "NULL output");
    EGP_REQUIRE(identity || !update || (st_out && ws && st_out != st_in), "update needs workspace and a distinct state_out");
    if (identity) update = 0;
    if (n == 0) {
        if (update) EGP_HIP_CHECK(hipMemcpyAsync(st_out, st_in, (1 + 2 * (size_t)dim) * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return EGP_OK;
    }
    int rpt, nt;
    zf_tiling(n, &rpt, &nt);
    // <= 16 tiles: the apply kernel merges them itself (2 launches). Up to 256 tiles: one block merges them into the new state
    // (a few rounds of loads). Beyond (>= 32 k rows): 16 blocks reduce the tiles to 16 records, which the apply kernel's
    // (16-row) blocks merge themselves -- the single block took 31 of K6's 70 us at 65 536 rows.
    const bool direct = update && nt <= ZF_FUSED_TILES;
    const bool two_level = update && nt > 256;
    const double *records = nullptr;                            // what the apply kernel merges into the running state, if anything
    int n_records = 0;
    if (update) {
        // (1 024 threads = one round of loads per thread: in the rollout 16.9 us per call against 20.1 with 512 and 33.6 with 256)
        launch_zf_partial<T>(src, active, n, dim, rpt, nt, (double *)ws, (hipStream_t)stream);
        if (direct) {
            records = (const double *)ws; n_records = nt;
        } else if (two_level) {
            double *lvl1 = (double *)ws + (size_t)nt * (1 + 2 * dim);            // behind the tiles in the workspace
            k_zf_merge<<<dim3(ZF_FUSED_TILES), dim3(1024), 0, (hipStream_t)stream>>>(dim, nt, (const double *)ws, nullptr, lvl1);
            records = lvl1; n_records = ZF_FUSED_TILES;
        } else {
            k_zf_merge<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(dim, nt, (const double *)ws, st_in, st_out);
        }
        int rc = after_launch("k_zf_partial/merge");
        if (rc != EGP_OK) return rc;
    }
    const int rows_per_block = direct ? 8 : (n <= 8192 ? 2 : 16);     // small batches: enough blocks to cover the latency
    k_zf_apply<T><<<dim3((n + rows_per_block - 1) / rows_per_block), dim3(128), 2 * dim * sizeof(double), (hipStream_t)stream>>>(
        src, n, dim, rows_per_block, update && !records ? st_out : st_in, clip, y, y2, write_mask, identity, records, n_records, st_out);
    return after_launch("k_zf_apply");
}

template <typename T>
static int launch_zfilter(const T *x, const int *active, int n, int dim, const double *st_in, double *st_out, int update,
                          double clip, T *y, void *ws, void *stream) {
    EGP_REQUIRE(st_in && (n == 0 || (x && y)), "NULL pointer");
    ZfSrc<T> src{x, nullptr, nullptr, 0, 0, dim, ObsOpt{0, 0, 0, 0, 0, 0}};
    return launch_zfilter_src<T>(src, active, n, dim, st_in, st_out, update, clip, y, nullptr, nullptr, ws, stream);
}

template <typename T>
static int launch_obs_zfilter(egp_ctx *ctx, const T *qpos, const T *qvel, const int *phase_t, const int *active, int n, const double *st_in, double *st_out,
                              double clip, T *y, T *y2, int write_only_active, void *ws, void *stream) {
    EGP_REQUIRE(ctx, "ctx is NULL");
    EGP_REQUIRE(n == 0 || (qpos && qvel), "NULL pointer");
    EGP_REQUIRE(!write_only_active || active, "write_only_active needs the active mask");
    EGP_REQUIRE(n == 0 || !ctx->dm.obs_phase || phase_t, "the model has obs_phase: phase_t (the rows' cur_t) is required");
    const int dim = ctx->dm.obs_dim;
    ZfSrc<T> src{nullptr, qpos, qvel, ctx->dm.nq, ctx->dm.nv, dim, obs_opt_of(ctx->dm), phase_t};
    return launch_zfilter_src<T>(src, active, n, dim, st_in, st_out, 1, clip, y, y2, write_only_active ? active : nullptr, ws, stream);
}

template <typename T>
static int launch_gae(const T *r, const T *mk, const T *v, int n, double gamma, double tau, T *adv, T *ret, double *stats,
                      void *ws, void *stream) {
    EGP_REQUIRE(r && mk && v && adv && ret && stats && ws, "NULL pointer");
    EGP_REQUIRE(n > 0, "n must be positive");
    const int n_chunks = (n + GAE_CHUNK - 1) / GAE_CHUNK;
    const int n_blocks = (n_chunks + 255) / 256;
    // workspace: [maps: 2 x n_blocks 64-bit slots][tile partials: 3 x n_blocks doubles]
    unsigned long long *maps = (unsigned long long *)ws;
    double *part = (double *)(maps + 2 * (size_t)n_blocks);
    hipStream_t s = (hipStream_t)stream;
    constexpr size_t lds = (size_t)GAE_LDS_DOUBLES * sizeof(double);          // the tile's (delta, c) pairs, padded
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gae_onepass<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) { set_error("hipFuncSetAttribute(k_gae_onepass, %zu B of LDS): %s", lds, hipGetErrorString(attr)); return EGP_E_HIP; }
    EGP_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)maps, (int)GAE_ARM32, (size_t)n_blocks * 4, s));     // arm the map slots
    k_gae_onepass<T><<<dim3(n_blocks), dim3(256), lds, s>>>(r, mk, v, n, gamma, gamma * tau, maps, adv, ret, part);
    k_gae_stats<<<dim3(1), dim3(GAE_STATS_THREADS), 0, s>>>(n_blocks, part, stats);
    return after_launch("k_gae_*");
}

template <typename T>
static int launch_standardize(T *a, int n, const double *stats, void *stream) {
    EGP_REQUIRE(a && stats, "NULL pointer");
    EGP_REQUIRE(n > 0, "n must be positive");
    k_standardize<T><<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(a, n, stats);
    return after_launch("k_standardize");
}

extern "C" {

int egp_body_quat_f64(egp_ctx *c, const double *q, int32_t n, double *o, void *s) { return launch_body_quat<double>(c, q, n, o, s); }
int egp_body_quat_f32(egp_ctx *c, const float *q, int32_t n, float *o, void *s) { return launch_body_quat<float>(c, q, n, o, s); }
int egp_reward_simple_f64(egp_ctx *c, int32_t kind, const double *qpos, const int32_t *frame, const int32_t *endf, const int32_t *active,
                          double end_reward, int32_t n, double *reward, double *cinfo, void *s) {
    EGP_REQUIRE(c, "ctx is NULL");
    EGP_REQUIRE(kind == EGP_REWARD_CONSTANT || kind == EGP_REWARD_POSE_DIST, "unknown reward kind");
    EGP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return EGP_OK;
    EGP_REQUIRE(reward && cinfo && (kind == EGP_REWARD_CONSTANT || (qpos && frame && endf)), "NULL pointer");
    if (kind == EGP_REWARD_POSE_DIST && !c->expert_qpos_f64) { set_error("egp_upload_experts must be called before the pose_dist reward"); return EGP_E_STATE; }
    k_reward_simple<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s>>>(kind, c->dm.nq, qpos, c->expert_qpos_f64, frame, endf, active,
                                                                              end_reward, n, reward, cinfo);
    return after_launch("k_reward_simple");
}
int egp_quat_op_f64(int32_t op, const double *a, const double *b, int32_t n, double *o, void *s) { return launch_quat_op<double>(op, a, b, n, o, s); }
int egp_quat_op_f32(int32_t op, const float *a, const float *b, int32_t n, float *o, void *s) { return launch_quat_op<float>(op, a, b, n, o, s); }
int egp_obs_f64(egp_ctx *c, const double *q, const double *v, const int32_t *phase_t, int32_t n, double *o, void *s) { return launch_obs<double>(c, q, v, phase_t, n, o, s); }
int egp_obs_f32(egp_ctx *c, const float *q, const float *v, const int32_t *phase_t, int32_t n, float *o, void *s) { return launch_obs<float>(c, q, v, phase_t, n, o, s); }

int egp_pd_torque_f64(egp_ctx *c, const double *qpos, const double *qvel, const double *action, const double *qM,
                      const double *bias, int32_t n, double *torque, double *torque_raw, void *s) {
    EGP_REQUIRE(c, "ctx is NULL");
    return launch_pd<double>(c, dense_ld(c), qpos, qvel, action, qM, bias, n, torque, torque_raw, (hipStream_t)s);
}
int egp_pd_torque_f32(egp_ctx *c, const float *qpos, const float *qvel, const float *action, const float *qM,
                      const float *bias, int32_t n, float *torque, float *torque_raw, void *s) {
    EGP_REQUIRE(c, "ctx is NULL");
    return launch_pd<float>(c, dense_ld(c), qpos, qvel, action, qM, bias, n, torque, torque_raw, (hipStream_t)s);
}

int egp_reward_quat_v3_f64(egp_ctx *c, const double *cq, const double *pq, const double *ee, const int32_t *t,
                           const int32_t *frame, const int32_t *endf, const int32_t *active, double end_reward, int32_t n,
                           double *reward, double *cinfo, void *s) {
    EGP_REQUIRE(c, "ctx is NULL");
    return launch_reward<double>(c, c->expert_rows_f64, cq, pq, ee, t, frame, endf, active, end_reward, n, reward, cinfo, s);
}
int egp_reward_quat_v3_f32(egp_ctx *c, const float *cq, const float *pq, const float *ee, const int32_t *t,
                           const int32_t *frame, const int32_t *endf, const int32_t *active, double end_reward, int32_t n,
                           float *reward, float *cinfo, void *s) {
    EGP_REQUIRE(c, "ctx is NULL");
    return launch_reward<float>(c, c->expert_rows_f32, cq, pq, ee, t, frame, endf, active, end_reward, n, reward, cinfo, s);
}

int egp_pose_features_f64(egp_ctx *c, const double *cur, const double *prev, const double *ee_w, int32_t n, int32_t expert_conv,
                          double *qvel, double *rlinv, double *rangv, double *rq, double *ee, double *bq, double *bav, void *s) {
    return launch_features<double>(c, cur, prev, ee_w, n, expert_conv, qvel, rlinv, rangv, rq, ee, bq, bav, s);
}
int egp_pose_features_f32(egp_ctx *c, const float *cur, const float *prev, const float *ee_w, int32_t n, int32_t expert_conv,
                          float *qvel, float *rlinv, float *rangv, float *rq, float *ee, float *bq, float *bav, void *s) {
    return launch_features<float>(c, cur, prev, ee_w, n, expert_conv, qvel, rlinv, rangv, rq, ee, bq, bav, s);
}

int64_t egp_zfilter_workspace_bytes(int32_t n, int32_t dim) {
    int rpt, nt;
    zf_tiling(n > 0 ? n : 1, &rpt, &nt);
    return ((int64_t)nt + ZF_FUSED_TILES) * (1 + 2 * (int64_t)dim) * sizeof(double);      // the tiles + the level-1 records
}
int egp_zfilter_f64(const double *x, const int32_t *active, int32_t n, int32_t dim, const double *si, double *so,
                    int32_t update, double clip, double *y, void *ws, void *s) {
    return launch_zfilter<double>(x, active, n, dim, si, so, update, clip, y, ws, s);
}
int egp_zfilter_f32(const float *x, const int32_t *active, int32_t n, int32_t dim, const double *si, double *so,
                    int32_t update, double clip, float *y, void *ws, void *s) {
    return launch_zfilter<float>(x, active, n, dim, si, so, update, clip, y, ws, s);
}

/* K3+K6 fused: observations of the drained state (get_full_obs) pushed through the running filter in one call.
 *   active [n] (optional): rows that update the statistics; write_only_active: only those rows are written.
 *   state_in == NULL: no filter (raw observations). y2 (optional) receives a second copy of the output. */
int egp_obs_zfilter_f64(egp_ctx *c, const double *qpos, const double *qvel, const int32_t *phase_t, const int32_t *active, int32_t n, const double *si,
                        double *so, double clip, double *y, double *y2, int32_t write_only_active, void *ws, void *s) {
    return launch_obs_zfilter<double>(c, qpos, qvel, phase_t, active, n, si, so, clip, y, y2, write_only_active, ws, s);
}
int egp_obs_zfilter_f32(egp_ctx *c, const float *qpos, const float *qvel, const int32_t *phase_t, const int32_t *active, int32_t n, const double *si,
                        double *so, double clip, float *y, float *y2, int32_t write_only_active, void *ws, void *s) {
    return launch_obs_zfilter<float>(c, qpos, qvel, phase_t, active, n, si, so, clip, y, y2, write_only_active, ws, s);
}
// egp_obs_zfilter_f64 in two calls, for batches of at most egp_obs_zfilter_split_max_rows() rows (the apply pass merges the tile
// statistics itself there): _stats = the first launch, _apply = the second, same kernels with the same arguments. Whoever runs
// the apply pass may instead be the policy step of the next tick (egp_policy_gaussian_filter_f32).
int32_t egp_obs_zfilter_split_max_rows(void) { return 64 * ZF_FUSED_TILES; }
int egp_obs_zfilter_stats_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, const int32_t *active, int32_t n, void *ws, void *stream) {
    EGP_REQUIRE(ctx && ws, "NULL pointer");
    EGP_REQUIRE(n > 0 && n <= 64 * ZF_FUSED_TILES && qpos && qvel, "1 .. egp_obs_zfilter_split_max_rows() rows");
    EGP_REQUIRE(!ctx->dm.obs_phase || phase_t, "the model has obs_phase: phase_t (the rows' cur_t) is required");
    const int dim = ctx->dm.obs_dim;
    ZfSrc<double> src{nullptr, qpos, qvel, ctx->dm.nq, ctx->dm.nv, dim, obs_opt_of(ctx->dm), phase_t};
    int rpt, nt;
    zf_tiling(n, &rpt, &nt);
    launch_zf_partial<double>(src, active, n, dim, rpt, nt, (double *)ws, (hipStream_t)stream);
    return after_launch("k_zf_partial");
}
int egp_obs_zfilter_apply_f64(egp_ctx *ctx, const double *qpos, const double *qvel, const int32_t *phase_t, int32_t n, const double *st_in, double *st_out,
                              double clip, double *y, double *y2, void *ws, void *stream) {
    EGP_REQUIRE(ctx && ws && st_in && st_out && st_in != st_out && y, "NULL pointer / state_out must differ from state_in");
    EGP_REQUIRE(n > 0 && n <= 64 * ZF_FUSED_TILES && qpos && qvel, "1 .. egp_obs_zfilter_split_max_rows() rows");
    EGP_REQUIRE(!ctx->dm.obs_phase || phase_t, "the model has obs_phase: phase_t (the rows' cur_t) is required");
    const int dim = ctx->dm.obs_dim;
    ZfSrc<double> src{nullptr, qpos, qvel, ctx->dm.nq, ctx->dm.nv, dim, obs_opt_of(ctx->dm), phase_t};
    int rpt, nt;
    zf_tiling(n, &rpt, &nt);
    k_zf_apply<double><<<dim3((n + 7) / 8), dim3(128), 2 * dim * sizeof(double), (hipStream_t)stream>>>(
        src, n, dim, 8, st_in, clip, y, y2, nullptr, 0, (const double *)ws, nt, st_out);
    return after_launch("k_zf_apply");
}


int64_t egp_gae_workspace_bytes(int32_t n) {
    const int64_t n_chunks = ((int64_t)n + GAE_CHUNK - 1) / GAE_CHUNK;
    const int64_t n_blocks = (n_chunks + 255) / 256;
    return (2 * n_chunks + 6 * n_blocks) * (int64_t)sizeof(double);
}
int egp_gae_f64(const double *r, const double *m, const double *v, int32_t n, double gamma, double tau, double *adv,
                double *ret, double *stats, void *ws, void *s) {
    return launch_gae<double>(r, m, v, n, gamma, tau, adv, ret, stats, ws, s);
}
int egp_gae_f32(const float *r, const float *m, const float *v, int32_t n, double gamma, double tau, float *adv,
                float *ret, double *stats, void *ws, void *s) {
    return launch_gae<float>(r, m, v, n, gamma, tau, adv, ret, stats, ws, s);
}
int egp_gae_standardize_f64(double *a, int32_t n, const double *stats, void *s) { return launch_standardize<double>(a, n, stats, s); }
int egp_gae_standardize_f32(float *a, int32_t n, const double *stats, void *s) { return launch_standardize<float>(a, n, stats, s); }

}  // extern "C"
