// Device pieces of K3 (observation) and K6 (running filter) shared by the rollout kernels (egp_kernels.hip) and the fused
// policy step (egp_policy.hip): one element of get_full_obs, the row source of the filter kernels, the ordered Chan merge of
// the tile statistics, and the host-side tiling rule.
#pragma once
#include <hip/hip_runtime.h>

#include "egp_internal.hpp"
#include "egp_quat.hpp"

namespace egp {

// ============================================================================================ K3
// get_full_obs (ego_pose/envs/humanoid_v1.py:73-96): obs = [qpos[2:] (root quat de-headed), qvel
// (root linear velocity in the heading frame)]
// Observation variants (cfg.obs_heading / root_deheading / obs_coord / obs_vel), all zero for every shipped config.
struct ObsOpt { int heading, keep, root, vel, np, nv, phase, episode_len; };
__host__ __device__ inline ObsOpt obs_opt_of(const DevModel &m) {
    return ObsOpt{m.obs_heading, m.obs_keep, m.obs_root, m.obs_vel, m.nq - 2, m.nv, m.obs_phase, m.episode_len};
}
__host__ __device__ inline int obs_width(int nq, int nv, int heading, int vel, int phase = 0) {
    return (heading ? 1 : 0) + (nq - 2) + (vel == 0 ? nv : (vel == 1 ? 6 : 0)) + (phase ? 1 : 0);
}

// one element of get_full_obs (humanoid_v1.py:73-96) for env row (q, v): column c of
// [heading]? ++ qpos[2:] (root quat de-headed unless `keep`) ++ {qvel | qvel[:6] | -} (root linear velocity in the
// heading frame, or in the root frame with `root`) ++ [phase]?
// cfg.obs_phase (:92-94; ego_forecast only): one more column at the end, min(cur_t / env_episode_len, 1) -- `t` = the env's cur_t
template <typename T>
__device__ __forceinline__ T obs_element(const T *q, const T *v, const ObsOpt &o, int c, int t = 0) {
    if (o.phase && c == (o.heading ? 1 : 0) + o.np + (o.vel == 0 ? o.nv : (o.vel == 1 ? 6 : 0))) {
        const T ph = T(t) / T(o.episode_len);
        return ph < T(1) ? ph : T(1);
    }
    if (o.heading) {
        if (c == 0) {        // get_heading (utils/math.py:70-77): angle of the yaw-only quaternion, z made non-negative
            T w = q[3], z = q[6];
            if (z < T(0)) { w = -w; z = -z; }
            if (sizeof(T) == 8) return T(2) * t_acos<T>(w / t_sqrt<T>(w * w + z * z));
            return T(2) * (T)atan2f((float)z, (float)w);          // float32: acos loses its digits near w = 1
        }
        c -= 1;
    }
    const int np = o.np;
    T out;
    if (c >= 1 && c <= 4 && !o.keep) {
        Q4<T> r{q[3], q[4], q[5], q[6]};
        Q4<T> d = de_heading(r);
        out = c == 1 ? d.w : (c == 2 ? d.x : (c == 3 ? d.y : d.z));
    } else if (c < np) {
        out = q[c + 2];
    } else if (c < np + 3) {
        Q4<T> r{q[3], q[4], q[5], q[6]};
        V3<T> lv{v[0], v[1], v[2]};
        V3<T> w = rotate_T(o.root ? r : heading_q(r), lv);
        const int k = c - np;
        out = k == 0 ? w.x : (k == 1 ? w.y : w.z);
    } else {
        out = v[c - np];
    }
    return out;
}

// ============================================================================================ K6
// RunningStat / ZFilter (utils/zfilter.py:7-67), batched.
// partial layout per tile p: ws[p*(1+2*dim)] = count, then mean[dim], then M2[dim]  (float64)
// Source of the rows being filtered: a dense array x[n][dim], or (x == nullptr) the observation computed on the
// fly from the drained state (K3 fused into K6: no intermediate raw-observation array)
template <typename T>
struct ZfSrc {
    const T *x; const T *qpos; const T *qvel; int nq, nv, dim;
    ObsOpt opt;
    const int *t;               // opt.phase: the rows' cur_t (egp_* `phase_t`); unused otherwise
    __device__ __forceinline__ T at(long r, int c) const {
        return x ? x[r * dim + c] : obs_element<T>(qpos + r * nq, qvel + r * nv, opt, c, opt.phase ? t[r] : 0);
    }
};

// Chan-merge of the tile partials of column c into the running state, fixed order (deterministic)
__device__ __forceinline__ void zf_merge_column(int dim, int n_tiles, const double *__restrict__ ws, const double *__restrict__ st_in,
                                                int c, double &cnt, double &mean, double &S) {
    cnt = st_in[0]; mean = st_in[1 + c]; S = st_in[1 + dim + c];
    for (int q0 = 0; q0 < n_tiles; q0 += 8) {
        double nb[8], mb[8], Sb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // 24 independent loads in flight, then the ordered merge
            const int q = q0 + i;
            const double *pp = ws + (long)(q < n_tiles ? q : 0) * (1 + 2 * dim);
            nb[i] = q < n_tiles ? pp[0] : 0.0;
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

// 64-row tiles (8 rows per thread of a 1 024-thread workgroup) while that gives <= 512 partials, larger tiles beyond.
// Up to ZF_FUSED_TILES partials the apply kernel merges them itself (two launches per update), beyond that k_zf_merge does.
constexpr int ZF_FUSED_TILES = 16;
static inline void zf_tiling(int n, int *rows_per_tile, int *n_tiles) {
    int rpt = 64;
    while ((n + rpt - 1) / rpt > 512) rpt *= 2;
    *rows_per_tile = rpt;
    *n_tiles = n > 0 ? (n + rpt - 1) / rpt : 1;
}

}  // namespace egp
