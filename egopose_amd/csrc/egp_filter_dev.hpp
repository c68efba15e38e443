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
// (in two steps -- the loads, then the arithmetic on what they returned -- so that a caller can have the loads of several rows
//  in flight before the first use: obs_element = obs_from(obs_load))
template <typename T> struct ObsIn { T qw, qx, qy, qz, v0, v1, v2, own; };

template <typename T>
__device__ __forceinline__ ObsIn<T> obs_load(const T *q, const T *v, const ObsOpt &o, int c) {
    ObsIn<T> in{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    if (o.phase && c == (o.heading ? 1 : 0) + o.np + (o.vel == 0 ? o.nv : (o.vel == 1 ? 6 : 0))) return in;
    if (o.heading) {
        if (c == 0) { in.qw = q[3]; in.qz = q[6]; return in; }
        c -= 1;
    }
    const int np = o.np;
    if (c >= 1 && c <= 4 && !o.keep) {
        in.qw = q[3]; in.qx = q[4]; in.qy = q[5]; in.qz = q[6];
    } else if (c < np) {
        in.own = q[c + 2];
    } else if (c < np + 3) {
        in.qw = q[3]; in.qx = q[4]; in.qy = q[5]; in.qz = q[6];
        in.v0 = v[0]; in.v1 = v[1]; in.v2 = v[2];
    } else {
        in.own = v[c - np];
    }
    return in;
}

// The quaternion arithmetic of get_full_obs, written out here under `fp contract(off)` (egp_quat.hpp's de_heading / rotate_T
// are compiled with the default contraction): which products the compiler fuses into FMAs depends on the code around the
// inlined function, and the kernels that compute the same observation -- K3, the filter's two passes, the policy step's
// prologue -- must agree bit for bit (the de-headed z component is a difference of two nearly equal products: one fused
// product moved it by 1e-8 after normalisation). Without contraction every kernel rounds each product and sum the same way.
template <typename T>
__device__ __forceinline__ Q4<T> obs_heading_q(const Q4<T> &q) {       // get_heading_q (utils/math.py:62-67)
#pragma clang fp contract(off)
    const T n = t_sqrt<T>(q.w * q.w + q.z * q.z);
    return Q4<T>{q.w / n, T(0), T(0), q.z / n};
}
template <typename T>
__device__ __forceinline__ Q4<T> obs_de_heading(const Q4<T> &q) {       // de_heading: inverse(heading_q(q)) * q (utils/math.py:80-81)
#pragma clang fp contract(off)
    const Q4<T> h = obs_heading_q<T>(q);
    const T nn = h.w * h.w + h.z * h.z;
    T iw, iz;                                  // quaternion_inverse: conjugate / (h.h)
    if (sizeof(T) == 8) {
        const T inv = T(1) / nn;
        iw = h.w * inv; iz = -h.z * inv;
    } else {
        iw = h.w / nn; iz = -h.z / nn;
    }
    Q4<T> r;                                   // (iw, 0, 0, iz) * q
    r.w = iw * q.w - iz * q.z;
    r.x = iw * q.x - iz * q.y;
    r.y = iw * q.y + iz * q.x;
    r.z = iw * q.z + iz * q.w;
    return r;
}
template <typename T>
__device__ __forceinline__ V3<T> obs_rotate_T(const Q4<T> &q, const V3<T> &v) {      // rotate_T of egp_quat.hpp: R(q)^T v
#pragma clang fp contract(off)
    const T n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    const T eps4 = sizeof(T) == 8 ? T(8.881784197001252e-16) : T(4.76837158203125e-07);
    if (n < eps4) return v;
    const T s = T(2) / n;
    const T xx = s * q.x * q.x, yy = s * q.y * q.y, zz = s * q.z * q.z;
    const T xy = s * q.x * q.y, xz = s * q.x * q.z, yz = s * q.y * q.z;
    const T wx = s * q.w * q.x, wy = s * q.w * q.y, wz = s * q.w * q.z;
    V3<T> o;
    o.x = (T(1) - yy - zz) * v.x + (xy + wz) * v.y + (xz - wy) * v.z;
    o.y = (xy - wz) * v.x + (T(1) - xx - zz) * v.y + (yz + wx) * v.z;
    o.z = (xz + wy) * v.x + (yz - wx) * v.y + (T(1) - xx - yy) * v.z;
    return o;
}

template <typename T>
__device__ __forceinline__ T obs_from(const ObsIn<T> &in, const ObsOpt &o, int c, int t = 0) {
#pragma clang fp contract(off)
    if (o.phase && c == (o.heading ? 1 : 0) + o.np + (o.vel == 0 ? o.nv : (o.vel == 1 ? 6 : 0))) {
        const T ph = T(t) / T(o.episode_len);
        return ph < T(1) ? ph : T(1);
    }
    if (o.heading) {
        if (c == 0) {        // get_heading (utils/math.py:70-77): angle of the yaw-only quaternion, z made non-negative
            T w = in.qw, z = in.qz;
            if (z < T(0)) { w = -w; z = -z; }
            if (sizeof(T) == 8) return T(2) * t_acos<T>(w / t_sqrt<T>(w * w + z * z));
            return T(2) * (T)atan2f((float)z, (float)w);          // float32: acos loses its digits near w = 1
        }
        c -= 1;
    }
    const int np = o.np;
    T out;
    if (c >= 1 && c <= 4 && !o.keep) {
        Q4<T> r{in.qw, in.qx, in.qy, in.qz};
        Q4<T> d = obs_de_heading<T>(r);
        out = c == 1 ? d.w : (c == 2 ? d.x : (c == 3 ? d.y : d.z));
    } else if (c < np) {
        out = in.own;
    } else if (c < np + 3) {
        Q4<T> r{in.qw, in.qx, in.qy, in.qz};
        V3<T> lv{in.v0, in.v1, in.v2};
        V3<T> w = obs_rotate_T<T>(o.root ? r : obs_heading_q<T>(r), lv);
        const int k = c - np;
        out = k == 0 ? w.x : (k == 1 ? w.y : w.z);
    } else {
        out = in.own;
    }
    return out;
}

template <typename T>
__device__ __forceinline__ T obs_element(const T *q, const T *v, const ObsOpt &o, int c, int t = 0) {
    return obs_from<T>(obs_load<T>(q, v, o, c), o, c, t);
}

// ============================================================================================ K6
// RunningStat / ZFilter (utils/zfilter.py:7-67), batched.
// partial layout per tile p: ws[p*(1+2*dim)] = count, then mean[dim], then M2[dim]  (float64)
// Source of the rows being filtered: a dense array x[n][dim], or (x == nullptr) the observation computed on the
// fly from the drained state (K3 fused into K6: no intermediate raw-observation array)
__device__ __forceinline__ double zf_readlane(double x, int l) {          // l: a constant or wave-uniform lane number
    union { double d; int i[2]; } u;
    u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}
__device__ __forceinline__ float zf_readlane(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

template <typename T>
struct ZfSrc {
    const T *x; const T *qpos; const T *qvel; int nq, nv, dim;
    ObsOpt opt;
    const int *t;               // opt.phase: the rows' cur_t (egp_* `phase_t`); unused otherwise
    __device__ __forceinline__ T at(long r, int c) const {
        return x ? x[r * dim + c] : obs_element<T>(qpos + r * nq, qvel + r * nv, opt, c, opt.phase ? t[r] : 0);
    }
    // at() for R consecutive rows of one column of the drained state (x == nullptr), in two steps.
    // load_rows() issues every load without a branch, a select or any arithmetic on what comes back (each of those, in front
    // of a block boundary, makes the compiler wait for the loads right there): the column's own element of each row, and the
    // root quaternion + root velocity of ONE row per lane -- the quaternion work of a row (de-heading the root quaternion for
    // columns 1..4, turning the root velocity into the heading frame for columns np..np+2) is the same for every lane that
    // needs it, and a wave pays for a divergent case once per ROW it holds whatever the number of lanes in it: ~170 float64
    // operations per row, 2.2 us for four rows on a lone wave (phase stamps). So the wave shares it out: lane r de-heads row
    // r, lane R + r rotates row r's velocity, and finish_rows() hands the components to the lanes that own those columns
    // with v_readlane. Both under wave-uniform control flow. Same helpers as obs_from on the same values: the same bits.
    template <int R> struct Rows { T qw, qx, qy, qz, v0, v1, v2; T own[R]; int t_raw[R]; };
    template <int R>
    __device__ __forceinline__ Rows<R> load_rows(long r0, long n, int c) const {
        // (relaxed wavefront-scope atomic loads = plain global_load instructions the compiler may not move: as ordinary loads
        //  it sinks them down to finish_rows, behind whatever the caller wanted them to overlap with)
        auto ld = [](const T *a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
        const int lane = threadIdx.x & 63;
        const long mr = min(r0 + (lane < R ? lane : (lane < 2 * R ? lane - R : 0)), n - 1);
        const T *q = qpos + mr * nq, *v = qvel + mr * nv;
        Rows<R> p;
        p.qw = ld(q + 3); p.qx = ld(q + 4); p.qy = ld(q + 5); p.qz = ld(q + 6);
        p.v0 = ld(v); p.v1 = ld(v + 1); p.v2 = ld(v + 2);
        const int cc = c - (opt.heading ? 1 : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = min(r0 + r, n - 1);
            const T *qr = qpos + row * nq, *vr = qvel + row * nv;
            p.own[r] = ld(cc < opt.np ? qr + min(max(cc + 2, 0), nq - 1) : vr + min(max(cc - opt.np, 0), nv - 1));
        }
        if (opt.phase) {           // (t_raw stays unset without a phase column: nothing reads it then, and a zero would be a copy at this block's end)
#pragma unroll
            for (int r = 0; r < R; ++r) p.t_raw[r] = __hip_atomic_load(t + min(r0 + r, n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        return p;
    }
    template <int R>
    __device__ __forceinline__ void finish_rows(Rows<R> &p, long r0, long n, int c) const {          // results in p.own
#pragma clang fp contract(off)
        static_assert(2 * R <= 64, "one task lane per row and kind");
        const ObsOpt &o = opt;
        const int np = o.np;
        const int cc = c - (o.heading ? 1 : 0);
        const bool is_phase = o.phase && c == (o.heading ? 1 : 0) + np + (o.vel == 0 ? o.nv : (o.vel == 1 ? 6 : 0));
        const bool is_head = o.heading && c == 0;
        const bool spec_q = !is_phase && !is_head && !o.keep && cc >= 1 && cc <= 4;
        const bool spec_v = !is_phase && !is_head && !spec_q && cc >= np && cc < np + 3;
        T t0 = T(0), t1 = T(0), t2 = T(0), t3 = T(0);            // this lane's share: a de-headed quaternion or a rotated velocity
        if (__any(spec_q || spec_v)) {
            const int lane = threadIdx.x & 63;
            if (lane < R) {
                const Q4<T> d = obs_de_heading<T>(Q4<T>{p.qw, p.qx, p.qy, p.qz});
                t0 = d.w; t1 = d.x; t2 = d.y; t3 = d.z;
            } else if (lane < 2 * R) {
                const Q4<T> q{p.qw, p.qx, p.qy, p.qz};
                const V3<T> w = obs_rotate_T<T>(o.root ? q : obs_heading_q<T>(q), V3<T>{p.v0, p.v1, p.v2});
                t0 = w.x; t1 = w.y; t2 = w.z;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T dw = zf_readlane(t0, r), dx = zf_readlane(t1, r), dy = zf_readlane(t2, r), dz = zf_readlane(t3, r);
            const T wx = zf_readlane(t0, R + r), wy = zf_readlane(t1, R + r), wz = zf_readlane(t2, R + r);
            if (spec_q) p.own[r] = cc == 1 ? dw : (cc == 2 ? dx : (cc == 3 ? dy : dz));
            if (spec_v) p.own[r] = cc == np ? wx : (cc == np + 1 ? wy : wz);
            __builtin_amdgcn_sched_barrier(0);          // (row by row: hoisted, the 14 scalar registers a row reads become 14 R and spill)
        }
        if (is_head) {             // (cfg.obs_heading, no shipped config: the heading angle of each row, from memory again)
#pragma unroll
            for (int r = 0; r < R; ++r) p.own[r] = at(min(r0 + r, n - 1), c);
        }
        if (is_phase) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int tr = p.t_raw[r];
                asm volatile("" : "+v"(tr));          // (keeps the int -> float conversion here: hoisted to the load, it waits for the load there)
                const T ph = T(tr) / T(o.episode_len);
                p.own[r] = ph < T(1) ? ph : T(1);
            }
        }
    }
};

// Chan-merge of the tile partials of column c into the running state, fixed order (deterministic)
__device__ __forceinline__ void zf_merge_column(int dim, int n_tiles, const double *__restrict__ ws, const double *__restrict__ st_in,
                                                int c, double &cnt, double &mean, double &S) {
#pragma clang fp contract(off)          // (every kernel that merges must round alike: see obs_from)
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

// zf_merge_column for a whole wave at once (every lane its own column c; use it under wave-uniform control flow), in two steps
// so that the caller can put other loads between a chunk's loads and its merge: begin() + load(0) ... merge(), then
// load(q0) + merge(q0) for q0 = 8, 16, ... The merge's coefficients cnt * nb / tot and nb / tot are the same for every column:
// lane i computes those of tile q0 + i and every lane reads them with v_readlane (one pair of float64 divisions per eight
// tiles instead of eight). Same operations on the same values in the same order as zf_merge_column: the same bits
// (tests/test_hip_parity.py holds the two against each other).
struct ZfWaveMerge {
    double cnt, mean, S;
    double nb[8], mb[8], Sb[8];
    __device__ __forceinline__ void begin(int dim, const double *__restrict__ st_in, int c) {
        cnt = st_in[0]; mean = st_in[1 + c]; S = st_in[1 + dim + c];
    }
    // (raw loads only -- tiles beyond n_tiles read tile 0 and are zeroed in merge(): no select on a value in flight, see ZfSrc::load_state)
    __device__ __forceinline__ void load(int dim, int n_tiles, const double *__restrict__ ws, int c, int q0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = q0 + i;
            const double *pp = ws + (long)(q < n_tiles ? q : 0) * (1 + 2 * dim);
            nb[i] = pp[0];
            mb[i] = pp[1 + c];
            Sb[i] = pp[1 + dim + c];
        }
    }
    __device__ __forceinline__ void merge(int n_tiles, int q0) {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (q0 + i >= n_tiles) nb[i] = 0.0;
        const int li = threadIdx.x & 7;
        double before = cnt, mine = 0.0;        // the running count in front of this lane's tile (zero counts add nothing, as in the ordered merge)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == li) mine = nb[j];
            if (j < li && nb[j] > 0.0) before += nb[j];
        }
        const double tot = before + mine;
        const double ca = before * mine / tot, cb = mine / tot;       // (tot == 0 only for tiles the merge skips)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double bi = zf_readlane(before, i), ai = zf_readlane(ca, i), ri = zf_readlane(cb, i);
            if (nb[i] > 0.0) {
                if (bi == 0.0) {
                    mean = mb[i]; S = Sb[i];
                } else {
                    const double d = mb[i] - mean;
                    S = S + Sb[i] + d * d * ai;
                    mean = mean + d * ri;
                }
                cnt = bi + nb[i];
            }
        }
    }
};

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
