// Lockstep rollout engine: replaces the per-env Python loop of HumanoidEnv.do_simulation
// (ego_pose/envs/humanoid_v1.py:158-177: 15 x {compute_torque; clip; data.ctrl = torque; sim.step()})
// for all envs of a GPU at once.
//
//   * physics stays on the host (egp_physics vtable), run by a pool of worker threads; each worker owns
//     a contiguous slice of envs, split into two half-slices that ping-pong so that one half's
//     K1 launch + PCIe round trip overlaps the other half's physics;
//   * per env and substep the drained MuJoCo fields travel as ONE packed row
//     [qpos | qvel | qfrc_bias | qM | pad] (1088 doubles) from pinned host memory with hipMemcpyAsync
//     on the half-slice's own stream; K1 reads the packed row in place and the clipped torque comes
//     back with one D2H copy;
//   * env groups (n_groups) can be stepped independently so the caller's GPU work for one group
//     (reward / observation kernels, policy inference) overlaps the other group's physics.
#include <hip/hip_runtime.h>

#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "egp_internal.hpp"

namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Half {
    int e0 = 0, e1 = 0;              // env range [e0, e1)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;       // recorded after the last H2D of an env-step
    std::vector<hipEvent_t> k_beg, k_end;   // per substep, only when profiling K1
};

struct Worker {
    int group = 0;
    Half half[2];
    std::thread th;
    // accumulated timing
    double phys_s = 0.0, wait_s = 0.0;
    double k1_ms = 0.0;
    long k1_launches = 0;
    int status = EGP_OK;
    char err[256] = "";
};

struct Group {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long job = 0;                    // incremented per step request
    int pending = 0;                 // workers still running the current job
    bool quit = false;
    const double *action = nullptr;
    hipEvent_t ready = nullptr;
    std::vector<int> active;         // per env of the whole engine (copied from the caller)
    bool has_active = false;
    std::vector<int> workers;
};

}  // namespace

struct egp_engine {
    egp_ctx *ctx = nullptr;
    egp_physics *phys = nullptr;
    const egp_physics_vtable *vt = nullptr;
    int n_env = 0, n_threads = 0, n_groups = 0;
    int nq = 0, nv = 0, nu = 0, nM = 0, nbody = 0, frame_skip = 0;
    int pack_ld = 0, off_qpos = 0, off_qvel = 0, off_bias = 0, off_qM = 0;
    bool profile_k1 = false;
    // device
    double *d_pack = nullptr, *d_qpos = nullptr, *d_qvel = nullptr, *d_torque = nullptr, *d_ee = nullptr;
    // pinned host
    double *h_pack = nullptr, *h_qpos = nullptr, *h_qvel = nullptr, *h_torque = nullptr, *h_ee = nullptr,
           *h_headz = nullptr, *h_xpos = nullptr;
    std::vector<Worker> workers;
    std::vector<Group> groups;
    std::vector<int> env_group;
};

namespace {

int drain_env(egp_engine *E, int env, bool with_xpos) {
    double *row = E->h_pack + (size_t)env * E->pack_ld;
    double *xp = with_xpos ? E->h_xpos + (size_t)env * E->nbody * 3 : nullptr;
    int rc = E->vt->drain(E->vt->user, env, row + E->off_qpos, row + E->off_qvel, row + E->off_qM, row + E->off_bias, xp);
    if (rc != 0) return EGP_E_PHYSICS;
    if (with_xpos) {
        memcpy(E->h_qpos + (size_t)env * E->nq, row + E->off_qpos, E->nq * sizeof(double));
        memcpy(E->h_qvel + (size_t)env * E->nv, row + E->off_qvel, E->nv * sizeof(double));
        for (int k = 0; k < 5; ++k) {
            const int b = E->ctx->ee_body[k];
            for (int c = 0; c < 3; ++c) E->h_ee[(size_t)env * 15 + 3 * k + c] = xp[b * 3 + c];
        }
        E->h_headz[env] = xp[E->ctx->ee_body[4] * 3 + 2];   // 'Head' is the 5th end effector
    }
    return EGP_OK;
}

#define W_HIP(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            snprintf(W.err, sizeof(W.err), "%s failed: %s", #expr, hipGetErrorString(_e));                 \
            W.status = EGP_E_HIP;                                                                          \
            return;                                                                                        \
        }                                                                                                  \
    } while (0)

void enqueue_k1(egp_engine *E, Worker &W, Half &H, const double *action, int substep) {
    const int m = H.e1 - H.e0;
    if (m <= 0) return;
    if (E->profile_k1) W_HIP(hipEventRecord(H.k_beg[substep], H.stream));
    int rc = egp_launch_pd_torque_packed(E->ctx, E->d_pack + (size_t)H.e0 * E->pack_ld, E->pack_ld, E->off_qpos, E->off_qvel,
                                         E->off_bias, E->off_qM, action + (size_t)H.e0 * E->nu, m,
                                         E->d_torque + (size_t)H.e0 * E->nu, H.stream);
    if (rc != EGP_OK) {
        snprintf(W.err, sizeof(W.err), "K1 launch failed: %s", egp_last_error());
        W.status = rc;
        return;
    }
    if (E->profile_k1) W_HIP(hipEventRecord(H.k_end[substep], H.stream));
    W_HIP(hipMemcpyAsync(E->h_torque + (size_t)H.e0 * E->nu, E->d_torque + (size_t)H.e0 * E->nu, (size_t)m * E->nu * sizeof(double),
                         hipMemcpyDeviceToHost, H.stream));
}

void run_step(egp_engine *E, Worker &W, Group &G) {
    const double *action = G.action;
    const int FS = E->frame_skip;
    for (int h = 0; h < 2; ++h) {
        Half &H = W.half[h];
        if (H.e1 <= H.e0) continue;
        if (G.ready) W_HIP(hipStreamWaitEvent(H.stream, G.ready, 0));
        enqueue_k1(E, W, H, action, 0);
        if (W.status != EGP_OK) return;
    }
    for (int s = 0; s < FS; ++s) {
        const bool last = s == FS - 1;
        for (int h = 0; h < 2; ++h) {
            Half &H = W.half[h];
            const int m = H.e1 - H.e0;
            if (m <= 0) continue;
            auto t0 = clk::now();
            W_HIP(hipStreamSynchronize(H.stream));          // torque of substep s is on the host
            auto t1 = clk::now();
            for (int e = H.e0; e < H.e1; ++e) {
                if (G.has_active && !G.active[e]) continue;
                if (E->vt->step(E->vt->user, e, E->h_torque + (size_t)e * E->nu) != 0 || drain_env(E, e, last) != EGP_OK) {
                    snprintf(W.err, sizeof(W.err), "physics backend failed on env %d", e);
                    W.status = EGP_E_PHYSICS;
                    return;
                }
            }
            auto t2 = clk::now();
            W.wait_s += secs(t0, t1);
            W.phys_s += secs(t1, t2);
            W_HIP(hipMemcpyAsync(E->d_pack + (size_t)H.e0 * E->pack_ld, E->h_pack + (size_t)H.e0 * E->pack_ld,
                                 (size_t)m * E->pack_ld * sizeof(double), hipMemcpyHostToDevice, H.stream));
            if (!last) {
                enqueue_k1(E, W, H, action, s + 1);
                if (W.status != EGP_OK) return;
            } else {
                W_HIP(hipMemcpyAsync(E->d_qpos + (size_t)H.e0 * E->nq, E->h_qpos + (size_t)H.e0 * E->nq, (size_t)m * E->nq * sizeof(double),
                                     hipMemcpyHostToDevice, H.stream));
                W_HIP(hipMemcpyAsync(E->d_qvel + (size_t)H.e0 * E->nv, E->h_qvel + (size_t)H.e0 * E->nv, (size_t)m * E->nv * sizeof(double),
                                     hipMemcpyHostToDevice, H.stream));
                W_HIP(hipMemcpyAsync(E->d_ee + (size_t)H.e0 * 15, E->h_ee + (size_t)H.e0 * 15, (size_t)m * 15 * sizeof(double),
                                     hipMemcpyHostToDevice, H.stream));
                W_HIP(hipEventRecord(H.done, H.stream));
            }
        }
    }
    if (E->profile_k1) {
        for (int h = 0; h < 2; ++h) {
            Half &H = W.half[h];
            if (H.e1 <= H.e0) continue;
            W_HIP(hipStreamSynchronize(H.stream));
            for (int s = 0; s < FS; ++s) {
                float ms = 0.f;
                W_HIP(hipEventElapsedTime(&ms, H.k_beg[s], H.k_end[s]));
                W.k1_ms += ms;
                W.k1_launches += 1;
            }
        }
    }
}

void worker_main(egp_engine *E, int wi) {
    Worker &W = E->workers[wi];
    Group &G = E->groups[W.group];
    (void)hipSetDevice(E->ctx->device);
    long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(G.mu);
            G.cv_go.wait(lk, [&] { return G.quit || G.job != seen; });
            if (G.quit) return;
            seen = G.job;
        }
        if (W.status == EGP_OK) run_step(E, W, G);
        {
            std::lock_guard<std::mutex> lk(G.mu);
            if (--G.pending == 0) G.cv_done.notify_all();
        }
    }
}

}  // namespace

extern "C" {

int egp_engine_create(egp_ctx *ctx, egp_physics *phys, const egp_engine_desc *d, egp_engine **out) {
    EGP_REQUIRE(ctx && phys && d && out, "NULL pointer");
    EGP_REQUIRE(d->n_env > 0 && d->n_threads > 0 && d->n_groups > 0, "n_env/n_threads/n_groups must be positive");
    EGP_REQUIRE(d->n_groups <= d->n_threads && d->n_threads <= d->n_env, "need n_groups <= n_threads <= n_env");
    EGP_REQUIRE(egp_physics_n_env(phys) >= d->n_env, "physics backend has fewer envs than the engine");
    EGP_HIP_CHECK(hipSetDevice(ctx->device));
    egp_engine *E = new egp_engine();
    E->ctx = ctx; E->phys = phys; E->vt = egp_physics_vt(phys);
    E->n_env = d->n_env; E->n_threads = d->n_threads; E->n_groups = d->n_groups;
    E->nq = ctx->dm.nq; E->nv = ctx->dm.nv; E->nu = ctx->dm.nu; E->nM = ctx->dm.nM; E->nbody = ctx->dm.nbody;
    E->frame_skip = ctx->frame_skip;
    E->off_qpos = 0; E->off_qvel = E->nq; E->off_bias = E->nq + E->nv; E->off_qM = E->nq + 2 * E->nv;
    E->pack_ld = ((E->off_qM + E->nM + 15) / 16) * 16;
    const char *prof = getenv("EGP_PROFILE_K1");
    E->profile_k1 = prof && atoi(prof) != 0;
    const size_t N = (size_t)E->n_env;
#define E_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { egp::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); egp_engine_destroy(E); return EGP_E_HIP; } } while (0)
    E_TRY(hipMalloc((void **)&E->d_pack, N * E->pack_ld * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_qpos, N * E->nq * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_qvel, N * E->nv * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_torque, N * E->nu * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_ee, N * 15 * sizeof(double)));
    E_TRY(hipMemset(E->d_pack, 0, N * E->pack_ld * sizeof(double)));
    E_TRY(hipHostMalloc((void **)&E->h_pack, N * E->pack_ld * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_qpos, N * E->nq * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_qvel, N * E->nv * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_torque, N * E->nu * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_ee, N * 15 * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_headz, N * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_xpos, N * E->nbody * 3 * sizeof(double), hipHostMallocDefault));
    memset(E->h_pack, 0, N * E->pack_ld * sizeof(double));
    memset(E->h_headz, 0, N * sizeof(double));
    // partition: groups get contiguous env ranges; each group's range is split over its workers
    E->workers.resize(E->n_threads);
    E->groups = std::vector<Group>(E->n_groups);
    E->env_group.assign(E->n_env, 0);
    int wi = 0;
    for (int g = 0; g < E->n_groups; ++g) {
        const int ge0 = (int)((long)E->n_env * g / E->n_groups), ge1 = (int)((long)E->n_env * (g + 1) / E->n_groups);
        const int w0 = (int)((long)E->n_threads * g / E->n_groups), w1 = (int)((long)E->n_threads * (g + 1) / E->n_groups);
        const int nw = w1 - w0;
        for (int e = ge0; e < ge1; ++e) E->env_group[e] = g;
        for (int k = 0; k < nw; ++k, ++wi) {
            Worker &W = E->workers[wi];
            W.group = g;
            const int a = ge0 + (int)((long)(ge1 - ge0) * k / nw), b = ge0 + (int)((long)(ge1 - ge0) * (k + 1) / nw);
            const int mid = a + (b - a + 1) / 2;
            W.half[0].e0 = a; W.half[0].e1 = mid;
            W.half[1].e0 = mid; W.half[1].e1 = b;
            for (int h = 0; h < 2; ++h) {
                E_TRY(hipStreamCreateWithFlags(&W.half[h].stream, hipStreamNonBlocking));
                E_TRY(hipEventCreateWithFlags(&W.half[h].done, hipEventDisableTiming));
                if (E->profile_k1) {
                    W.half[h].k_beg.resize(E->frame_skip);
                    W.half[h].k_end.resize(E->frame_skip);
                    for (int s = 0; s < E->frame_skip; ++s) {
                        E_TRY(hipEventCreate(&W.half[h].k_beg[s]));
                        E_TRY(hipEventCreate(&W.half[h].k_end[s]));
                    }
                }
            }
            E->groups[g].workers.push_back(wi);
        }
        E->groups[g].active.assign(E->n_env, 1);
    }
#undef E_TRY
    for (int i = 0; i < E->n_threads; ++i) E->workers[i].th = std::thread(worker_main, E, i);
    *out = E;
    return EGP_OK;
}

int egp_engine_destroy(egp_engine *E) {
    if (!E) return EGP_OK;
    for (auto &G : E->groups) {
        std::lock_guard<std::mutex> lk(G.mu);
        G.quit = true;
        G.cv_go.notify_all();
    }
    for (auto &W : E->workers) {
        if (W.th.joinable()) W.th.join();
        for (int h = 0; h < 2; ++h) {
            if (W.half[h].stream) (void)hipStreamDestroy(W.half[h].stream);
            if (W.half[h].done) (void)hipEventDestroy(W.half[h].done);
            for (auto ev : W.half[h].k_beg) (void)hipEventDestroy(ev);
            for (auto ev : W.half[h].k_end) (void)hipEventDestroy(ev);
        }
    }
    void *dev[] = {E->d_pack, E->d_qpos, E->d_qvel, E->d_torque, E->d_ee};
    for (void *p : dev) if (p) (void)hipFree(p);
    void *host[] = {E->h_pack, E->h_qpos, E->h_qvel, E->h_torque, E->h_ee, E->h_headz, E->h_xpos};
    for (void *p : host) if (p) (void)hipHostFree(p);
    delete E;
    return EGP_OK;
}

int egp_engine_state(egp_engine *E, double **qpos, double **qvel, double **ee_wpos, double **head_z_host,
                     double **qpos_host, double **qvel_host) {
    EGP_REQUIRE(E, "engine is NULL");
    if (qpos) *qpos = E->d_qpos;
    if (qvel) *qvel = E->d_qvel;
    if (ee_wpos) *ee_wpos = E->d_ee;
    if (head_z_host) *head_z_host = E->h_headz;
    if (qpos_host) *qpos_host = E->h_qpos;
    if (qvel_host) *qvel_host = E->h_qvel;
    return EGP_OK;
}

int egp_engine_reset(egp_engine *E, const int32_t *ids, int32_t n, const double *qpos, const double *qvel, void *stream) {
    EGP_REQUIRE(E && (n == 0 || (ids && qpos && qvel)), "NULL pointer");
    EGP_HIP_CHECK(hipSetDevice(E->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < n; ++k) {
        const int e = ids[k];
        EGP_REQUIRE(e >= 0 && e < E->n_env, "env id out of range");
        EGP_REQUIRE(k == 0 || ids[k] > ids[k - 1], "env ids must be strictly increasing");
        if (E->vt->reset(E->vt->user, e, qpos + (size_t)k * E->nq, qvel + (size_t)k * E->nv) != 0 || drain_env(E, e, true) != EGP_OK) {
            egp::set_error("physics backend failed to reset env %d", e);
            return EGP_E_PHYSICS;
        }
    }
    // upload maximal runs of consecutive env ids
    int k = 0;
    while (k < n) {
        int j = k;
        while (j + 1 < n && ids[j + 1] == ids[j] + 1) ++j;
        const size_t e0 = ids[k], m = (size_t)(j - k + 1);
        EGP_HIP_CHECK(hipMemcpyAsync(E->d_pack + e0 * E->pack_ld, E->h_pack + e0 * E->pack_ld, m * E->pack_ld * sizeof(double), hipMemcpyHostToDevice, s));
        EGP_HIP_CHECK(hipMemcpyAsync(E->d_qpos + e0 * E->nq, E->h_qpos + e0 * E->nq, m * E->nq * sizeof(double), hipMemcpyHostToDevice, s));
        EGP_HIP_CHECK(hipMemcpyAsync(E->d_qvel + e0 * E->nv, E->h_qvel + e0 * E->nv, m * E->nv * sizeof(double), hipMemcpyHostToDevice, s));
        EGP_HIP_CHECK(hipMemcpyAsync(E->d_ee + e0 * 15, E->h_ee + e0 * 15, m * 15 * sizeof(double), hipMemcpyHostToDevice, s));
        k = j + 1;
    }
    return EGP_OK;
}

int egp_engine_step_async(egp_engine *E, int32_t group, const double *action, const int32_t *active_host, void *ready_event) {
    EGP_REQUIRE(E && action, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Group &G = E->groups[group];
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.pending != 0) { egp::set_error("group %d is still stepping", group); return EGP_E_STATE; }
    G.action = action;
    G.ready = (hipEvent_t)ready_event;
    G.has_active = active_host != nullptr;
    if (active_host) memcpy(G.active.data(), active_host, E->n_env * sizeof(int));
    G.pending = (int)G.workers.size();
    G.job += 1;
    G.cv_go.notify_all();
    return EGP_OK;
}

int egp_engine_wait(egp_engine *E, int32_t group, void *stream) {
    EGP_REQUIRE(E, "engine is NULL");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Group &G = E->groups[group];
    {
        std::unique_lock<std::mutex> lk(G.mu);
        G.cv_done.wait(lk, [&] { return G.pending == 0; });
    }
    for (int wi : G.workers) {
        Worker &W = E->workers[wi];
        if (W.status != EGP_OK) {
            egp::set_error("rollout worker %d: %s", wi, W.err);
            return W.status;
        }
        for (int h = 0; h < 2; ++h)
            if (W.half[h].e1 > W.half[h].e0) EGP_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, W.half[h].done, 0));
    }
    return EGP_OK;
}

int egp_engine_timing(egp_engine *E, double *phys_s, double *gpu_wait_s, double *k1_ms, int64_t *k1_launches) {
    EGP_REQUIRE(E, "engine is NULL");
    double p = 0, w = 0, k = 0;
    long l = 0;
    for (auto &W : E->workers) { p += W.phys_s; w += W.wait_s; k += W.k1_ms; l += W.k1_launches; }
    if (phys_s) *phys_s = p;
    if (gpu_wait_s) *gpu_wait_s = w;
    if (k1_ms) *k1_ms = k;
    if (k1_launches) *k1_launches = l;
    return EGP_OK;
}

int egp_engine_set_profile(egp_engine *E, int on) {
    EGP_REQUIRE(E, "engine is NULL");
    EGP_REQUIRE(!on || !E->workers.empty(), "no workers");
    if (on && E->workers[0].half[0].k_beg.empty()) {
        EGP_HIP_CHECK(hipSetDevice(E->ctx->device));
        for (auto &W : E->workers)
            for (int h = 0; h < 2; ++h) {
                W.half[h].k_beg.resize(E->frame_skip);
                W.half[h].k_end.resize(E->frame_skip);
                for (int s = 0; s < E->frame_skip; ++s) {
                    EGP_HIP_CHECK(hipEventCreate(&W.half[h].k_beg[s]));
                    EGP_HIP_CHECK(hipEventCreate(&W.half[h].k_end[s]));
                }
            }
    }
    E->profile_k1 = on != 0;
    return EGP_OK;
}

int egp_engine_reset_timing(egp_engine *E) {
    EGP_REQUIRE(E, "engine is NULL");
    for (auto &W : E->workers) { W.phys_s = 0; W.wait_s = 0; W.k1_ms = 0; W.k1_launches = 0; }
    return EGP_OK;
}

int egp_engine_layout(egp_engine *E, int32_t *pack_ld, int32_t *n_env, int32_t *n_threads, int32_t *n_groups) {
    EGP_REQUIRE(E, "engine is NULL");
    if (pack_ld) *pack_ld = E->pack_ld;
    if (n_env) *n_env = E->n_env;
    if (n_threads) *n_threads = E->n_threads;
    if (n_groups) *n_groups = E->n_groups;
    return EGP_OK;
}

int egp_engine_group_range(egp_engine *E, int32_t group, int32_t *e0, int32_t *e1) {
    EGP_REQUIRE(E && e0 && e1, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    *e0 = (int)((long)E->n_env * group / E->n_groups);
    *e1 = (int)((long)E->n_env * (group + 1) / E->n_groups);
    return EGP_OK;
}

}  // extern "C"
