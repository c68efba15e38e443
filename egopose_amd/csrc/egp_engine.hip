// Lockstep rollout engine: replaces the per-env Python loop of HumanoidEnv.do_simulation
// (ego_pose/envs/humanoid_v1.py:158-177: 15 x {compute_torque; clip; data.ctrl = torque; sim.step()})
// for all envs of a GPU at once.
//
//   * envs are partitioned into `n_groups` contiguous groups with their own host threads and HIP stream; state rows
//     [qpos | qvel | qfrc_bias | pad] (176 doubles/env), inertia rows qM (912 doubles/env) and torque rows live in pinned host
//     memory that the kernels read / write in place over PCIe (zero-copy);
//   * TWO forms of the env-step (docs/DESIGN_TRAIL.md section 2 has the forms that were measured and dropped):
//        resident   ONE launch of k_pd_server_tree58 serves all 15 substeps: its workgroups wait for the go word of their
//                   slice, pull the slice's state rows, solve, write torques; every host thread owns a few slices and steps an
//                   env the moment its torque row has arrived (run_step_server). The default -- whenever every workgroup of
//                   every group fits on the chip at the same time (they wait on the host: a workgroup that is not resident
//                   while its slice's owner waits for it would stall the env-step until another group's kernel ends).
//        per-substep  one K1 launch per substep over the whole group, the leader polls the kernel's completion flag, the
//                   group's threads meet at spin barriers (run_step). The fallback: other dof trees / K1 variants, more
//                   slots than the chip holds resident workgroups for, EGP_SERVER=0.
//   * different groups run independently: while one group's rows are on the PCIe link / in K1, another
//     group's threads do physics, and the caller's GPU work for a finished group (reward / observation
//     kernels, policy inference) overlaps the other groups' stepping.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "egp_internal.hpp"

namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

// A go word of the resident kernel. In pinned host memory a release store is all it takes (the waves read it over PCIe). In
// fine-grained DEVICE memory (EGP_BAR_GO) the host's mapping is write-combining: without a store fence the word can sit in a
// write-combining buffer until something else evicts it -- usually microseconds, occasionally the rest of a time slice, which
// is what the "heavier tail" of the pushed form was (single rollouts of 136 / 165 / 228 ms): the fence pushes it out now.
// (And a write-combining store is not ordered behind the ordinary stores that filled the state rows: a fence in front keeps the
//  rows globally visible before the word that announces them.)
inline void store_go(unsigned long long *word, unsigned long long value, bool in_vram) {
#if defined(__x86_64__)
    if (in_vram) _mm_sfence();
#endif
    __atomic_store_n(word, value, __ATOMIC_RELEASE);
#if defined(__x86_64__)
    if (in_vram) _mm_sfence();
#endif
}

// sense-reversing spin barrier for the threads of one group
struct SpinBarrier {
    std::atomic<int> count{0};
    std::atomic<int> phase{0};
    int n = 1;
    void wait() {
        const int ph = phase.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            phase.store(ph + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (phase.load(std::memory_order_acquire) == ph) {
                if (++spins < 4096) cpu_relax();
                else { std::this_thread::yield(); }
            }
        }
    }
};

// Resident-K1 mode: the group's envs as a few slices per host thread (whole 4-env blocks each), each with the `go`
// word its owner raises after a substep of physics; torques come back through sentinel-filled pinned rows
struct Server {
    int n_slices = 0, n_blocks = 0, per_thread = 0;
    std::vector<int> e0, e1;                  // env range of each slice
    int *d_block_slice = nullptr;
    int *d_block_env0 = nullptr;              // multi-env K1: [n_blocks + 1] first env of every workgroup, relative to the group's first
    unsigned long long *h_go = nullptr, *hd_go = nullptr;   // [n_slices * 8]
    bool go_in_vram = false;                  // the go words live in fine-grained device memory (host writes through the BAR)
    int *h_err = nullptr, *hd_err = nullptr;
    unsigned long long seq = 1, base = 0;     // next free sequence number / first substep of the env-step in flight
    std::unique_ptr<std::atomic<int>[]> dirty;   // some qM row of the slice changed since the kernel last read it
    long long *d_trace = nullptr;             // EGP_SERVER_TRACE=1: device stamps of block 0 + host stamps of slice 0
    std::vector<double> host_trace;           // [frame_skip * 4] us since the job started: wait start, torques seen, go written
    // Which thread walks which slices in the env-step in flight: thread t has order[first[t] .. first[t + 1]). With every
    // env active that is its own K consecutive slices; with an active mask the slices are dealt out per env-step so that
    // every thread steps about the same number of envs (towards the end of a rollout the running episodes sit in a few
    // slots, and the env-step takes as long as the thread with the most of them)
    // Dealing pays when a substep of physics costs more than moving an env's state to another core's cache: at 20 us
    // per env-substep (an mj_step-like cost) the rollout gains 7 %; with the 0.3 us surrogate it LOSES 10 % (measured both
    // ways). So the engine measures what a substep costs and deals only above `balance_min_ns`.
    std::vector<int> order, first;
    std::atomic<long long> substep_ns{0};     // running estimate of one env-substep on a host thread (step + drain)
    long long balance_min_ns = 2000;
    bool can_balance = true;
};

// deal the slices of a group out to its threads for one env-step (called by the submitter, before the workers wake)
inline void assign_slices(Server &S, int n_threads, const int *active) {
    const int ns = S.n_slices, K = S.per_thread;
    S.order.resize(ns);
    S.first.resize(n_threads + 1);
    const bool deal = active && S.can_balance && S.substep_ns.load(std::memory_order_relaxed) >= S.balance_min_ns;
    if (!deal) {
        for (int sl = 0; sl < ns; ++sl) S.order[sl] = sl;
        for (int t = 0; t <= n_threads; ++t) S.first[t] = std::min(K * t, ns);      // (fewer workgroups than threads: the last threads idle)
        return;
    }
    // longest-processing-time-first: slices by falling number of active envs, each to the least loaded thread (ties:
    // fewer slices, then the lower thread). Empty slices still need their go words written: they go to whoever has fewest.
    int cnt[512], idx[512], load[64], held[64], owner[512];
    const int n = std::min(ns, 512), nt = std::min(n_threads, 64);
    for (int sl = 0; sl < n; ++sl) {
        int c = 0;
        for (int e = S.e0[sl]; e < S.e1[sl]; ++e) c += active[e] != 0;
        cnt[sl] = c; idx[sl] = sl;
    }
    std::stable_sort(idx, idx + n, [&](int a, int b) { return cnt[a] > cnt[b]; });
    for (int t = 0; t < nt; ++t) load[t] = held[t] = 0;
    for (int i = 0; i < n; ++i) {
        int best = 0;
        for (int t = 1; t < nt; ++t)
            if (load[t] < load[best] || (load[t] == load[best] && held[t] < held[best])) best = t;
        owner[idx[i]] = best; load[best] += cnt[idx[i]]; held[best] += 1;
    }
    int pos = 0;
    for (int t = 0; t < nt; ++t) {
        S.first[t] = pos;
        for (int sl = 0; sl < n; ++sl)           // rising slice order within a thread: neighbouring rows stay together
            if (owner[sl] == t) S.order[pos++] = sl;
    }
    S.first[nt] = pos;
}

struct Group {
    int e0 = 0, e1 = 0;                       // env range [e0, e1)
    Server srv;
    int n_threads = 1;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;                // recorded after the last upload of an env-step
    hipEvent_t reward_done = nullptr;         // recorded behind the reward job's kernel (it reads d_qpos / d_prev_qpos / d_ee rows
    bool reward_in_flight = false;            //  that a reset on the caller's stream overwrites): egp_engine_reset waits for it
    std::vector<hipEvent_t> k_beg, k_end;     // per substep, when profiling K1
    SpinBarrier bar;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long job = 0;
    int pending = 0;                          // threads still inside the current job
    std::atomic<long> job_pub{0};             // == job, published for threads that spin between env-steps
    std::atomic<int> pending_pub{0};          // == pending, published for a caller that spins in egp_engine_wait
    bool quit = false;
    const double *action = nullptr;
    hipEvent_t ready = nullptr;
    // reward job of the env-step in flight (egp_engine_set_reward_job): launched on the group's stream right behind K1
    struct RewardJob {
        bool armed = false;
        const int32_t *t = nullptr, *frame = nullptr, *end = nullptr, *active = nullptr;
        double end_reward = 0.0;
        double *reward = nullptr, *cinfo = nullptr;
    } rjob;
    int *active = nullptr;                    // [n_env] pinned (the resident K1 reads its group's part in place)
    int *hd_active = nullptr;
    bool has_active = false;
    std::atomic<int> status{EGP_OK};
    std::atomic<int> qM_dirty{0};             // some env of the group drained a new inertia this substep
    unsigned *d_done = nullptr;               // K1 completion counter (HBM)
    unsigned long long *h_flag = nullptr;     // K1 completion flag (pinned host), polled by the leader
    unsigned long long *hd_flag = nullptr;
    unsigned long long seq = 0;
    int dyn_gen = 0;                          // device dynamics, per-substep form: which generation of (qM, bias) rows K1 reads next
    bool polled = false;                      // the launch in flight publishes to h_flag
    bool prof_now = false;                    // this env-step brackets its K1 launches with events
    bool server_job = false;                  // form of the env-step in flight (fixed when it is posted)
    char err[256] = "";
    // timing (leader only)
    double phys_s = 0.0, wait_s = 0.0, k1_ms = 0.0, ev_overhead_ms = 0.0;
    long k1_launches = 0, qM_uploads = 0;
    long k1_env_substeps = 0;                 // env-substeps actually stepped inside the event-bracketed launches
    std::vector<std::thread> threads;
};

}  // namespace

struct egp_engine {
    egp_ctx *ctx = nullptr;
    egp_physics *phys = nullptr;
    const egp_physics_vtable *vt = nullptr;
    int n_env = 0, n_threads = 0, n_groups = 0;
    int nq = 0, nv = 0, nu = 0, nM = 0, nbody = 0, frame_skip = 0;
    int ld_s = 0, ld_m = 0, off_qpos = 0, off_qvel = 0, off_bias = 0;   // state / inertia row strides (doubles)
    std::atomic<bool> profile_k1{false};
    int profile_every = 1;                    // bracket K1 with events on every Nth env-step of a group
    double *hd_state = nullptr, *hd_torque = nullptr, *hd_qM = nullptr, *hd_ee = nullptr;   // device-side aliases of h_state / h_torque / h_qM
    // Resident-K1 mode: the slices' go words live in FINE-GRAINED DEVICE memory that the host threads write through the PCIe
    // BAR (posted stores); the resident waves then poll HBM instead of host memory -- one PCIe read round trip less per
    // substep. Default since the end of round 3 (EGP_BAR_GO=0: pinned go words; a failed fine-grained allocation falls back to
    // them too). History: round 2 measured -1 .. -3 ms on two boxes and nothing on a third; round 3's in-lease A/B found a better
    // median (94.5 against 99.9 ms) but single rollouts of 136 / 165 / 228 ms -- the host's mapping of that memory is
    // write-combining and the go word was a plain store: it could sit in a write-combining buffer until something evicted it. With
    // a store fence either side of it (store_go) the stalls are gone: four alternating pairs of 15 rollouts, medians 94.2 against
    // 97.3 ms, worst rollout 101 against 107.5; under bench.py (tools/probes/ab_bench_env.sh) 906 / 921 / 918 k against 907 / 903 / 905 k.
    // The state rows stay in pinned host memory in any case: mirroring them the same
    // way was built and measured (tools/probes/bar_pingpong.hip: 4.7 instead of 8.1 us per round trip for ONE wave's four
    // rows), but a host thread's write-combined stores move ~0.4 us per env and substep one after the other where the
    // waves' PCIe reads run in parallel: T_sample 102 -> 118 ms with every row mirrored, and erratic (100 .. 270 ms)
    // when only env-steps with few running envs used the mirror.
    bool bar_go = false;
    bool server_ok = false;                   // resident-K1 mode allowed (EGP_SERVER, every workgroup of every group fits the chip at once)
    int server_ke = 1;                        // envs a resident wave serves in turn per substep (1: k_pd_server_tree58; 2 / 4: ..._multi)
    int server_cap = 0;                       // workgroups of that kernel the chip holds at once (probed, egp_pd_server_resident_blocks)
    bool server_ke_forced = false;            // EGP_SERVER_KE (tests): workgroups packed full instead of the envs dealt out over the capacity
    // device-dynamics mode: qM / qfrc_bias come from K8 on the (qpos, qvel) rows the backend drains -- with the reference's timing
    // (ego_pose/envs/humanoid_v1.py:130-144 reads data.qM / data.qfrc_bias as the previous mj_step left them): the torque of a
    // substep is solved with M, C of the state the PREVIOUS substep started from; only a reset (sim.forward(),
    // envs/common/mujoco_env.py:97-101) evaluates them at the current state. d_qM / d_bias hold what "the last mj_step left behind"
    // per env; the per-substep form alternates between them and a second generation (K8 writes the next while K1 reads the current).
    bool device_dynamics = false;
    double *d_bias = nullptr;                 // [n_env][nv]
    double *d_qM2 = nullptr, *d_bias2 = nullptr;   // second generation (per-substep form only)
    int reward_delay_us = 0;                  // EGP_REWARD_JOB_DELAY_US (tests): a spin kernel ahead of the reward job's kernel
    // The engine's threads (and a caller in egp_engine_wait, 4 x as long) poll this long for the next env-step / its end before
    // sleeping on a condition variable: a futex wake-up per env-step and thread is ~2 % of the rollout (round 3's in-lease A/B,
    // 99.3 against 101.2 ms of T_sample; 150 vs 500 us: equal). Between rollouts the threads sleep after 150 us.
    static constexpr int spin_us = 150;
    double *d_state = nullptr, *d_qM = nullptr, *d_prev_qpos = nullptr, *d_qpos = nullptr, *d_qvel = nullptr, *d_torque = nullptr, *d_ee = nullptr;
    double *h_state = nullptr, *h_qM = nullptr, *h_qpos = nullptr, *h_qvel = nullptr, *h_torque = nullptr, *h_ee = nullptr,
           *h_headz = nullptr, *h_xpos = nullptr;
    std::vector<Group> groups;
    std::vector<int64_t> epoch;               // last drained inertia epoch per env (-1 = never)
    std::vector<int> env_group, env_slice;
    int *h_reset_list = nullptr, *hd_reset_list = nullptr;   // pinned (env, qM_changed, dynamics generation) triples of the reset in flight
    hipEvent_t reset_done = nullptr;                          // recorded behind the scatter kernel of the last reset
    bool reset_pending = false;
};

namespace {

// egp_engine_reset on a zero-copy engine: ONE launch moves the freshly drained pinned rows of the listed envs to
// their HBM mirrors (block per env; the list holds (env, qM_changed) pairs in pinned memory) instead of five
// copy-engine calls per run of consecutive env ids.
// test aid: hold the stream for `us` microseconds (wall_clock64 ticks at 100 MHz on gfx950)
__global__ void k_engine_spin(long long us) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < us * 100) __builtin_amdgcn_s_sleep(32);
}

// diagnostic (egp_debug_burn): keep `gridDim.x` workgroups arithmetically busy for `us` microseconds
__global__ __launch_bounds__(256) void k_engine_burn(long long us, float *sink) {
    const long long t0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    while (wall_clock64() - t0 < us * 100) {
#pragma unroll
        for (int i = 0; i < 256; ++i) a = fmaf(a, b, 1e-7f);
    }
    if (a == 123.456f) *sink = a;
}

__global__ __launch_bounds__(256) void k_engine_reset_scatter(const int *__restrict__ list, const double *__restrict__ h_state, int ld_s,
                                                              int off_qpos, int off_qvel, const double *__restrict__ h_ee,
                                                              const double *__restrict__ h_qM, int ld_m, int nM, int nq, int nv,
                                                              double *d_state, double *d_qpos, double *d_qvel, double *d_ee,
                                                              double *d_qM) {
    const int e = list[3 * blockIdx.x], new_qM = list[3 * blockIdx.x + 1];
    const double *row = h_state + (long)e * ld_s;
    for (int c = threadIdx.x; c < ld_s; c += blockDim.x) {
        const double v = row[c];
        d_state[(long)e * ld_s + c] = v;
        if (c >= off_qpos && c < off_qpos + nq) d_qpos[(long)e * nq + (c - off_qpos)] = v;
        if (c >= off_qvel && c < off_qvel + nv) d_qvel[(long)e * nv + (c - off_qvel)] = v;
    }
    if (threadIdx.x < 15) d_ee[(long)e * 15 + threadIdx.x] = h_ee[(long)e * 15 + threadIdx.x];
    if (new_qM)
        for (int c = threadIdx.x; c < nM; c += blockDim.x) d_qM[(long)e * ld_m + c] = h_qM[(long)e * ld_m + c];
}

// mark_dirty: flag the env's group / slice for an inertia upload (the substep loop); egp_engine_reset
// uploads the rows it drained itself and must leave the flags of its neighbours alone
int drain_env(egp_engine *E, int env, bool with_xpos, bool mark_dirty = true) {
    double *row = E->h_state + (size_t)env * E->ld_s;
    double *xp = with_xpos ? E->h_xpos + (size_t)env * E->nbody * 3 : nullptr;
    double *qM = E->device_dynamics ? nullptr : E->h_qM + (size_t)env * E->ld_m;
    if (qM && E->vt->inertia_epoch) {
        const int64_t ep = E->vt->inertia_epoch(E->vt->user, env);
        if (ep == E->epoch[env]) qM = nullptr;            // unchanged since the last drain: nothing to move
        else E->epoch[env] = ep;
    }
    int rc = E->vt->drain(E->vt->user, env, row + E->off_qpos, row + E->off_qvel, qM, row + E->off_bias, xp);
    if (rc != 0) return EGP_E_PHYSICS;
    if (qM && mark_dirty) {
        Group &G = E->groups[E->env_group[env]];
        G.qM_dirty.store(1, std::memory_order_relaxed);
        if (G.srv.n_slices) G.srv.dirty[E->env_slice[env]].store(1, std::memory_order_relaxed);
    }
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

void fail(Group &G, int code, const char *what, const char *detail) {
    int expect = EGP_OK;
    if (G.status.compare_exchange_strong(expect, code)) snprintf(G.err, sizeof(G.err), "%s: %s", what, detail);
}

#define G_HIP(expr)                                                          \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) fail(G, EGP_E_HIP, #expr, hipGetErrorString(_e)); \
    } while (0)

// leader only (per-substep form): K1 over the whole group, state rows read and torques written in pinned host memory
void enqueue_k1(egp_engine *E, Group &G, int substep) {
    const int m = G.e1 - G.e0;
    const bool prof = G.prof_now;
    if (prof) G_HIP(hipEventRecord(G.k_beg[substep], G.stream));
    const double *st = E->hd_state + (size_t)G.e0 * E->ld_s;
    double *tq = E->hd_torque + (size_t)G.e0 * E->nu;
    const double *bias = st + E->off_bias;
    long ld_bias = E->ld_s;
    const double *qM = E->d_qM + (size_t)G.e0 * E->ld_m;
    if (E->device_dynamics) {
        // K8 on the state just drained writes the NEXT generation (what this substep's mj_step will have left behind); K1 solves with
        // the current one -- the previous substep's, or the reset's. (K8 has to read the pinned rows before the host may step, so it
        // stays in front of K1 in the stream; the resident form runs it behind the torque store.)
        double *qM_gen[2] = {E->d_qM + (size_t)G.e0 * E->ld_m, E->d_qM2 + (size_t)G.e0 * E->ld_m};
        double *bias_gen[2] = {E->d_bias + (size_t)G.e0 * E->nv, E->d_bias2 + (size_t)G.e0 * E->nv};
        const int cur = G.dyn_gen, nxt = cur ^ 1;
        int rd = egp_launch_dynamics_strided(E->ctx, st + E->off_qpos, E->ld_s, st + E->off_qvel, E->ld_s, m, qM_gen[nxt], E->ld_m,
                                             bias_gen[nxt], E->nv, nullptr, G.stream);
        if (rd != EGP_OK) fail(G, rd, "K8 launch", egp_last_error());
        qM = qM_gen[cur];
        bias = bias_gen[cur];
        ld_bias = E->nv;
        G.dyn_gen = nxt;
    }
    int rc = egp_launch_pd_torque_strided(E->ctx, st + E->off_qpos, E->ld_s, st + E->off_qvel, E->ld_s, bias, ld_bias,
                                          qM, E->ld_m, G.action + (size_t)G.e0 * E->nu, m, tq, G.stream,
                                          G.polled ? G.d_done : nullptr, G.hd_flag, G.polled ? ++G.seq : 0);
    if (rc != EGP_OK) fail(G, rc, "K1 launch", egp_last_error());
    if (prof) G_HIP(hipEventRecord(G.k_end[substep], G.stream));
}

// envs of the group this env-step advances (finished slots of a rollout's tail are skipped)
inline long stepped_envs(const Group &G) {
    if (!G.has_active) return G.e1 - G.e0;
    long n = 0;
    for (int e = G.e0; e < G.e1; ++e) n += G.active[e] != 0;
    return n;
}

void run_step(egp_engine *E, Group &G, int tid) {
    const bool leader = tid == 0;
    const int FS = E->frame_skip;
    const int m = G.e1 - G.e0;
    const int my0 = G.e0 + (int)((long)m * tid / G.n_threads), my1 = G.e0 + (int)((long)m * (tid + 1) / G.n_threads);
    if (leader) {
        G.prof_now = E->profile_k1.load(std::memory_order_relaxed) && !G.k_beg.empty() && (G.job % E->profile_every == 0);
        if (G.prof_now) G.k1_env_substeps += stepped_envs(G) * (long)E->frame_skip;
        G.polled = E->ctx->pd_variant == 0;          // (the tree kernels publish a completion flag; the others: hipStreamSynchronize)
        if (G.ready) G_HIP(hipStreamWaitEvent(G.stream, G.ready, 0));
        // env.prev_qpos = data.qpos.copy() (humanoid_v1.py:182): kept on the device for the reward kernel
        G_HIP(hipMemcpyAsync(E->d_prev_qpos + (size_t)G.e0 * E->nq, E->d_qpos + (size_t)G.e0 * E->nq, (size_t)m * E->nq * sizeof(double),
                             hipMemcpyDeviceToDevice, G.stream));
        enqueue_k1(E, G, 0);
    }
    for (int s = 0; s < FS; ++s) {
        const bool last = s == FS - 1;
        if (leader) {
            auto t0 = clk::now();
            bool synced = false;
            if (G.polled) {                                // spin on the flag the last K1 block writes over PCIe
                const unsigned long long want = G.seq;
                const auto deadline = t0 + std::chrono::seconds(5);
                long spins = 0;
                while (__atomic_load_n(G.h_flag, __ATOMIC_ACQUIRE) != want) {
                    cpu_relax();
                    if ((++spins & 0xFFFF) == 0 && clk::now() > deadline) break;   // fall back to the runtime
                }
                synced = __atomic_load_n(G.h_flag, __ATOMIC_ACQUIRE) == want;
            }
            if (!synced) G_HIP(hipStreamSynchronize(G.stream));        // torques of substep s are on the host
            G.wait_s += secs(t0, clk::now());
        }
        G.bar.wait();
        auto t1 = clk::now();
        if (G.status.load(std::memory_order_relaxed) == EGP_OK) {
            for (int e = my0; e < my1; ++e) {
                if (G.has_active && !G.active[e]) continue;
                if (E->vt->step(E->vt->user, e, E->h_torque + (size_t)e * E->nu) != 0 || drain_env(E, e, last) != EGP_OK) {
                    char msg[64];
                    snprintf(msg, sizeof(msg), "env %d", e);
                    fail(G, EGP_E_PHYSICS, "physics backend failed", msg);
                    break;
                }
            }
        }
        G.bar.wait();
        if (leader) {
            G.phys_s += secs(t1, clk::now());
            if (G.status.load() == EGP_OK) {
                if (G.qM_dirty.exchange(0)) {
                    G_HIP(hipMemcpyAsync(E->d_qM + (size_t)G.e0 * E->ld_m, E->h_qM + (size_t)G.e0 * E->ld_m,
                                         (size_t)m * E->ld_m * sizeof(double), hipMemcpyHostToDevice, G.stream));
                    G.qM_uploads += 1;
                }
                if (!last) {
                    enqueue_k1(E, G, s + 1);
                } else {
                    G_HIP(hipMemcpyAsync(E->d_qpos + (size_t)G.e0 * E->nq, E->h_qpos + (size_t)G.e0 * E->nq, (size_t)m * E->nq * sizeof(double),
                                         hipMemcpyHostToDevice, G.stream));
                    G_HIP(hipMemcpyAsync(E->d_qvel + (size_t)G.e0 * E->nv, E->h_qvel + (size_t)G.e0 * E->nv, (size_t)m * E->nv * sizeof(double),
                                         hipMemcpyHostToDevice, G.stream));
                    G_HIP(hipMemcpyAsync(E->d_ee + (size_t)G.e0 * 15, E->h_ee + (size_t)G.e0 * 15, (size_t)m * 15 * sizeof(double),
                                         hipMemcpyHostToDevice, G.stream));
                    G_HIP(hipEventRecord(G.done, G.stream));
                }
            }
        }
    }
    if (leader && G.prof_now && G.status.load() == EGP_OK) {
        G_HIP(hipStreamSynchronize(G.stream));
        for (int s = 0; s < FS; ++s) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, G.k_beg[s], G.k_end[s]) == hipSuccess) {
                G.k1_ms += ms > G.ev_overhead_ms ? ms - G.ev_overhead_ms : 0.0;
                G.k1_launches += 1;
            }
        }
    }
}

inline bool server_mode(const egp_engine *E, const Group &G) {
    return E->server_ok && G.srv.n_slices > 0 && E->ctx->pd_variant == 0 && E->ctx->tree58;
}

// Resident-K1 env-step: one launch of k_pd_server_tree58 serves all substeps. Every host thread owns a few slices
// and walks them round-robin -- while the GPU turns one slice's new state into torques (PCIe read, solve, PCIe
// write), the thread advances the next -- so no thread ever waits for another one inside the env-step, and an env
// is stepped the moment its own torque row has fully arrived.
// A torque row is nu doubles; the GPU writes it in pieces of at least one 32-byte sector (4 doubles), each of which
// lands atomically. Rows start sector-aligned when nu % 4 == 0 (52 for the humanoid), so ONE sentinel word per sector
// is enough both to arm a row and to see that all of it has arrived; otherwise every word is used.
inline int sentinel_stride(int nu) { return nu % 4 == 0 ? 4 : 1; }

inline void prefetch_row(const double *row, int nu) {
#if defined(__x86_64__)
    const char *p = reinterpret_cast<const char *>(row);
    for (int b = 0; b < nu * (int)sizeof(double); b += 64) _mm_prefetch(p + b, _MM_HINT_T0);
#else
    (void)row; (void)nu;
#endif
}

inline bool row_arrived(const double *row, int nu, int stride) {
    const unsigned long long *u = reinterpret_cast<const unsigned long long *>(row);
    bool ok = true;
    for (int i = 0; i < nu; i += stride) ok &= __atomic_load_n(u + i, __ATOMIC_RELAXED) != EGP_TORQUE_SENTINEL;
    return ok;
}

// arm the rows of the slice's envs that will be stepped (inactive envs are never read: their rows stay as they are)
inline void arm_slice(egp_engine *E, const Group &G, int e0, int e1, int nu, int stride) {
    for (int e = e0; e < e1; ++e) {
        if (G.has_active && !G.active[e]) continue;
        unsigned long long *u = reinterpret_cast<unsigned long long *>(E->h_torque + (size_t)e * nu);
        for (int i = 0; i < nu; i += stride) u[i] = EGP_TORQUE_SENTINEL;
    }
}

void run_step_server(egp_engine *E, Group &G, int tid) {
    Server &S = G.srv;
    const int FS = E->frame_skip, nu = E->nu;
    const int sstride = sentinel_stride(nu);
    const unsigned long long base = S.base;
    const unsigned long long drain_all = 0x7FFFFFFFFFFFFFFFull << 1;
    const auto t_job = clk::now();
    // own slices: sentinel rows first, then the go word of substep 0 (the kernel writes torques only after it saw go)
    const int h0 = S.first[tid], h1 = S.first[tid + 1];
    for (int h = h0; h < h1; ++h) {
        const int sl = S.order[h];
        arm_slice(E, G, S.e0[sl], S.e1[sl], nu, sstride);
        store_go(S.h_go + sl * 8, (base << 1) | (unsigned long long)S.dirty[sl].exchange(0), S.go_in_vram);
    }
    if (tid == 0) {
        G.prof_now = E->profile_k1.load(std::memory_order_relaxed) && !G.k_beg.empty() && (G.job % E->profile_every == 0);
        if (G.prof_now) G.k1_env_substeps += stepped_envs(G) * (long)E->frame_skip;
        const int m = G.e1 - G.e0;
        if (G.ready) G_HIP(hipStreamWaitEvent(G.stream, G.ready, 0));
        if (G.prof_now) G_HIP(hipEventRecord(G.k_beg[0], G.stream));
        const double *st = E->hd_state + (size_t)G.e0 * E->ld_s;
        const double *bias = E->device_dynamics ? E->d_bias + (size_t)G.e0 * E->nv : st + E->off_bias;
        int rc = egp_launch_pd_server(E->ctx, st + E->off_qpos, E->ld_s, st + E->off_qvel, E->ld_s, bias, E->device_dynamics ? E->nv : E->ld_s,
                                      E->d_qM + (size_t)G.e0 * E->ld_m, E->ld_m, E->hd_qM + (size_t)G.e0 * E->ld_m,
                                      G.action + (size_t)G.e0 * nu, m, E->hd_torque + (size_t)G.e0 * nu, G.stream, S.d_block_slice,
                                      S.hd_go, base, FS, S.hd_err, 2.0, S.d_trace, E->hd_ee + (size_t)G.e0 * 15,
                                      E->d_qpos + (size_t)G.e0 * E->nq, E->d_prev_qpos + (size_t)G.e0 * E->nq,
                                      E->d_qvel + (size_t)G.e0 * E->nv, E->d_ee + (size_t)G.e0 * 15,
                                      G.has_active ? G.hd_active + G.e0 : nullptr, E->device_dynamics, E->server_ke, S.d_block_env0, S.n_blocks);
        if (rc != EGP_OK) fail(G, rc, "K1 server launch", egp_last_error());
        if (G.prof_now) G_HIP(hipEventRecord(G.k_end[0], G.stream));
        G_HIP(hipEventRecord(G.done, G.stream));      // the kernel's epilogue moves the final state to HBM
        if (G.rjob.armed) {
            // K2 of this env-step, stream-ordered behind the epilogue and ahead of the next env-step's kernel: off the
            // caller's critical path (filter -> policy -> next step), and its inputs cannot be overwritten under it
            const int m2 = G.e1 - G.e0;
            if (E->reward_delay_us > 0) k_engine_spin<<<dim3(1), dim3(1), 0, G.stream>>>((long long)E->reward_delay_us);
            int rr = egp_reward_quat_v3_f64(E->ctx, E->d_qpos + (size_t)G.e0 * E->nq, E->d_prev_qpos + (size_t)G.e0 * E->nq,
                                            E->d_ee + (size_t)G.e0 * 15, G.rjob.t, G.rjob.frame, G.rjob.end, G.rjob.active,
                                            G.rjob.end_reward, m2, G.rjob.reward, G.rjob.cinfo, G.stream);
            if (rr != EGP_OK) fail(G, rr, "reward launch", egp_last_error());
            G_HIP(hipEventRecord(G.reward_done, G.stream));
            G.reward_in_flight = true;           // (published with the job's completion: pending -> 0)
            G.rjob.armed = false;
        }
        if (!S.host_trace.empty()) S.host_trace[0 * 4 + 3] = secs(t_job, clk::now()) * 1e6;     // launch issued
    }
    const bool timekeeper = tid == (G.n_threads > 1 ? 1 : 0);
    double t_wait = 0.0, t_phys = 0.0;
    long n_stepped = 0;
    for (int s = 0; s < FS; ++s) {
        const bool last = s == FS - 1;
        for (int h = h0; h < h1; ++h) {
            const int sl = S.order[h];
            const bool tr = sl == 0 && !S.host_trace.empty();
            auto t0 = clk::now();
            if (tr) S.host_trace[s * 4 + 0] = secs(t_job, t0) * 1e6;
            double waited = 0.0;
            bool first = true;
            for (int e = S.e0[sl]; e < S.e1[sl] && G.status.load(std::memory_order_relaxed) == EGP_OK; ++e) {
                if (G.has_active && !G.active[e]) continue;
                const double *row = E->h_torque + (size_t)e * nu;
                if (!row_arrived(row, nu, sstride)) {
                    auto w0 = clk::now();
                    const auto deadline = w0 + std::chrono::seconds(5);
                    long spins = 0;
                    while (!row_arrived(row, nu, sstride)) {
                        cpu_relax();
                        if ((++spins & 0xFFF) == 0) {
                            if (G.status.load(std::memory_order_relaxed) != EGP_OK) break;
                            if (__atomic_load_n(S.h_err, __ATOMIC_ACQUIRE) != 0) { fail(G, EGP_E_HIP, "K1 server", "kernel-side wait timed out"); break; }
                            if (clk::now() > deadline) { fail(G, EGP_E_HIP, "K1 server", "no torques within 5 s"); break; }
                        }
                    }
                    waited += secs(w0, clk::now());
                    if (G.status.load(std::memory_order_relaxed) != EGP_OK) break;
                }
                if (tr && first) { S.host_trace[s * 4 + 1] = secs(t_job, clk::now()) * 1e6; first = false; }
                // the GPU's writes left the next env's torque row out of every cache: request its lines now, so that the miss
                // (a trip to memory per line) runs under this env's physics step instead of in front of the next one
                if (e + 1 < S.e1[sl]) prefetch_row(row + nu, nu);
                if (E->vt->step(E->vt->user, e, row) != 0 || drain_env(E, e, last) != EGP_OK) {
                    char msg[64];
                    snprintf(msg, sizeof(msg), "env %d", e);
                    fail(G, EGP_E_PHYSICS, "physics backend failed", msg);
                    break;
                }
                ++n_stepped;
            }
            if (G.status.load(std::memory_order_relaxed) != EGP_OK) {
                store_go(S.h_go + sl * 8, drain_all, S.go_in_vram);       // let the kernel run out
            } else if (!last) {
                arm_slice(E, G, S.e0[sl], S.e1[sl], nu, sstride);
                store_go(S.h_go + sl * 8, ((base + (unsigned long long)s + 1ull) << 1) | (unsigned long long)S.dirty[sl].exchange(0), S.go_in_vram);
            } else {
                // final state of the slice is drained: let the kernel's epilogue move it to HBM (a pending qM refresh
                // stays flagged for the next env-step)
                store_go(S.h_go + sl * 8, (base + (unsigned long long)FS) << 1, S.go_in_vram);
            }
            if (tr) S.host_trace[s * 4 + 2] = secs(t_job, clk::now()) * 1e6;
            if (timekeeper) {
                t_wait += waited;
                t_phys += secs(t0, clk::now()) - waited;
            }
        }
    }
    if (timekeeper) {
        G.wait_s += t_wait; G.phys_s += t_phys;
        if (n_stepped >= 8) {                    // (a handful of envs says little: the slice bookkeeping dominates)
            const long long now_ns = (long long)(t_phys * 1e9 / (double)n_stepped), old = S.substep_ns.load(std::memory_order_relaxed);
            S.substep_ns.store(old == 0 ? now_ns : (3 * old + now_ns) / 4, std::memory_order_relaxed);
        }
    }
    if (tid == 0 && !S.host_trace.empty()) S.host_trace[1 * 4 + 3] = secs(t_job, clk::now()) * 1e6;      // own substeps done
    if (!S.host_trace.empty() && 4 + tid < FS) S.host_trace[(4 + tid) * 4 + 3] = secs(t_job, clk::now()) * 1e6;   // per thread
    G.bar.wait();
    if (tid != 0) return;
    if (!S.host_trace.empty()) {
        S.host_trace[2 * 4 + 3] = secs(t_job, clk::now()) * 1e6;                                       // every thread done
        (void)hipEventSynchronize(G.done);
        S.host_trace[3 * 4 + 3] = secs(t_job, clk::now()) * 1e6;                                       // kernel (epilogue) done
    }
    if (G.status.load() != EGP_OK) {
        for (int sl = 0; sl < S.n_slices; ++sl) store_go(S.h_go + sl * 8, drain_all, S.go_in_vram);
        (void)hipStreamSynchronize(G.stream);
        return;
    }
    G.qM_dirty.store(0, std::memory_order_relaxed);
    if (G.prof_now) {
        G_HIP(hipStreamSynchronize(G.stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, G.k_beg[0], G.k_end[0]) == hipSuccess) {
            G.k1_ms += ms > G.ev_overhead_ms ? ms - G.ev_overhead_ms : 0.0;
            G.k1_launches += 1;
        }
    }
}

void thread_main(egp_engine *E, int gi, int tid) {
    Group &G = E->groups[gi];
    (void)hipSetDevice(E->ctx->device);
    long seen = 0;
    for (;;) {
        // the next env-step of a group usually arrives within a few hundred microseconds (the caller's reward /
        // observation / policy launches): spin that long before going to sleep on the condition variable
        bool got = false;
        {
            const auto until = clk::now() + std::chrono::microseconds(egp_engine::spin_us);
            int n = 0;
            while (!(got = G.job_pub.load(std::memory_order_acquire) != seen)) {
                cpu_relax();
                if ((++n & 63) == 0 && clk::now() > until) break;
            }
        }
        if (got) {
            seen = G.job_pub.load(std::memory_order_acquire);
        } else {
            std::unique_lock<std::mutex> lk(G.mu);
            G.cv_go.wait(lk, [&] { return G.quit || G.job != seen; });
            if (G.quit) return;
            seen = G.job;
        }
        if (G.server_job) run_step_server(E, G, tid);
        else run_step(E, G, tid);
        {
            std::lock_guard<std::mutex> lk(G.mu);
            if (--G.pending == 0) {
                G.pending_pub.store(0, std::memory_order_release);
                G.cv_done.notify_all();
            }
        }
    }
}

int make_profile_events(egp_engine *E) {
    for (auto &G : E->groups) {
        if (!G.k_beg.empty()) continue;
        const int n_ev = E->frame_skip;
        G.k_beg.resize(n_ev);
        G.k_end.resize(n_ev);
        for (int s = 0; s < n_ev; ++s) {
            EGP_HIP_CHECK(hipEventCreate(&G.k_beg[s]));
            EGP_HIP_CHECK(hipEventCreate(&G.k_end[s]));
        }
        // calibrate the cost of an empty begin/end event pair on this stream (subtracted from each K1 bracket)
        double acc = 0.0;
        const int reps = 32;
        for (int r = 0; r < reps; ++r) {
            EGP_HIP_CHECK(hipEventRecord(G.k_beg[0], G.stream));
            EGP_HIP_CHECK(hipEventRecord(G.k_end[0], G.stream));
            EGP_HIP_CHECK(hipStreamSynchronize(G.stream));
            float ms = 0.f;
            EGP_HIP_CHECK(hipEventElapsedTime(&ms, G.k_beg[0], G.k_end[0]));
            acc += ms;
        }
        G.ev_overhead_ms = acc / reps;
    }
    return EGP_OK;
}

}  // namespace

extern "C" {

int egp_engine_create(egp_ctx *ctx, egp_physics *phys, const egp_engine_desc *d, egp_engine **out) {
    EGP_REQUIRE(ctx && phys && d && out, "NULL pointer");
    EGP_REQUIRE(d->n_env > 0 && d->n_threads > 0 && d->n_groups > 0, "n_env/n_threads/n_groups must be positive");
    EGP_REQUIRE(d->n_groups <= d->n_threads && d->n_groups <= d->n_env, "need n_groups <= n_threads and n_groups <= n_env");
    EGP_REQUIRE(egp_physics_n_env(phys) >= d->n_env, "physics backend has fewer envs than the engine");
    EGP_HIP_CHECK(hipSetDevice(ctx->device));
    egp_engine *E = new egp_engine();
    E->ctx = ctx; E->phys = phys; E->vt = egp_physics_vt(phys);
    E->n_env = d->n_env; E->n_threads = d->n_threads; E->n_groups = d->n_groups;
    E->nq = ctx->dm.nq; E->nv = ctx->dm.nv; E->nu = ctx->dm.nu; E->nM = ctx->dm.nM; E->nbody = ctx->dm.nbody;
    E->frame_skip = ctx->frame_skip;
    E->device_dynamics = d->device_dynamics != 0;
    if (E->device_dynamics && !ctx->dyn_tables) {
        egp::set_error("device_dynamics needs egp_set_dynamics_model on the context first");
        delete E;
        return EGP_E_STATE;
    }
    E->off_qpos = 0; E->off_qvel = E->nq; E->off_bias = E->nq + E->nv;
    E->ld_s = ((E->nq + 2 * E->nv + 15) / 16) * 16;
    E->ld_m = ((E->nM + 15) / 16) * 16;
    E->epoch.assign(d->n_env, -1);
    E->env_group.assign(d->n_env, 0);
    E->env_slice.assign(d->n_env, 0);
    const size_t N = (size_t)E->n_env;
#define E_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { egp::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); egp_engine_destroy(E); return EGP_E_HIP; } } while (0)
    E_TRY(hipMalloc((void **)&E->d_state, N * E->ld_s * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_qM, N * E->ld_m * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_qpos, N * E->nq * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_prev_qpos, N * E->nq * sizeof(double)));
    E_TRY(hipMemset(E->d_prev_qpos, 0, N * E->nq * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_qvel, N * E->nv * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_torque, N * E->nu * sizeof(double)));
    E_TRY(hipMalloc((void **)&E->d_ee, N * 15 * sizeof(double)));
    if (E->device_dynamics) {
        E_TRY(hipMalloc((void **)&E->d_bias, N * E->nv * sizeof(double)));
        E_TRY(hipMalloc((void **)&E->d_bias2, N * E->nv * sizeof(double)));
        E_TRY(hipMalloc((void **)&E->d_qM2, N * E->ld_m * sizeof(double)));
        E_TRY(hipMemset(E->d_bias, 0, N * E->nv * sizeof(double)));
        E_TRY(hipMemset(E->d_bias2, 0, N * E->nv * sizeof(double)));
        E_TRY(hipMemset(E->d_qM2, 0, N * E->ld_m * sizeof(double)));
    }
    E_TRY(hipMemset(E->d_state, 0, N * E->ld_s * sizeof(double)));
    E_TRY(hipMemset(E->d_qM, 0, N * E->ld_m * sizeof(double)));
    E_TRY(hipHostMalloc((void **)&E->h_state, N * E->ld_s * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_qM, N * E->ld_m * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_qpos, N * E->nq * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_qvel, N * E->nv * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_torque, N * E->nu * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_ee, N * 15 * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_headz, N * sizeof(double), hipHostMallocDefault));
    E_TRY(hipHostMalloc((void **)&E->h_xpos, N * E->nbody * 3 * sizeof(double), hipHostMallocDefault));
    {
        // zero-copy is the engine's only form: the kernels address the pinned rows directly
        void *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr;
        E_TRY(hipHostGetDevicePointer(&p1, E->h_state, 0));
        E_TRY(hipHostGetDevicePointer(&p2, E->h_torque, 0));
        E_TRY(hipHostGetDevicePointer(&p3, E->h_qM, 0));
        E_TRY(hipHostGetDevicePointer(&p4, E->h_ee, 0));
        E->hd_state = (double *)p1; E->hd_torque = (double *)p2; E->hd_qM = (double *)p3; E->hd_ee = (double *)p4;
        E_TRY(hipHostMalloc((void **)&E->h_reset_list, N * 3 * sizeof(int), hipHostMallocDefault));
        E_TRY(hipHostGetDevicePointer(&p5, E->h_reset_list, 0));
        E->hd_reset_list = (int *)p5;
        E_TRY(hipEventCreateWithFlags(&E->reset_done, hipEventDisableTiming));
        if (const char *rd = getenv("EGP_REWARD_JOB_DELAY_US")) E->reward_delay_us = atoi(rd);
    }
    memset(E->h_state, 0, N * E->ld_s * sizeof(double));
    {
        const char *bg = getenv("EGP_BAR_GO");
        E->bar_go = !(bg && atoi(bg) == 0);    // default on since the stores are fenced (see the field)
        if (E->bar_go) {       // the host stores straight into device memory: only where the whole of it is mapped through the PCIe BAR
            int large_bar = 0;
            if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar) {
                (void)hipGetLastError();
                E->bar_go = false;
            }
        }
    }
    memset(E->h_qM, 0, N * E->ld_m * sizeof(double));
    memset(E->h_headz, 0, N * sizeof(double));
    E->groups = std::vector<Group>(E->n_groups);
    int total_server_blocks = 0;
    bool all_groups_sliced = true;
    {
        // The resident kernel's workgroups wait on the host, so ALL of them (every group's) must be on the chip at the same time: a
        // workgroup that waits for a CU while its slice's owner waits for its torques stalls the env-step until another group's kernel
        // ends (2 x 256 one-env workgroups on 256 CUs went as far as the kernel-side timeout in round 4). One 350-register workgroup
        // fits a CU: 4 envs per CU with one env per wave (1 024 slots on 256 CUs). Beyond that -- more slots, or fewer CUs to be had
        // (a CU mask, a co-tenant, a smaller part) -- a wave serves 2 or 4 envs in turn (k_pd_server_tree58_multi): the smallest
        // count whose grid the chip holds, by the kernel's own residency probe. Device dynamics (110 kB of LDS per workgroup) has the
        // one-env form only. EGP_SERVER_KE forces a count (tests), EGP_SERVER=0 the per-substep form.
        const char *sv = getenv("EGP_SERVER");
        const char *fk = getenv("EGP_SERVER_KE");
        const bool usable = ctx->tree58 && ctx->pd_variant == 0 && !(sv && atoi(sv) == 0);
        const int choices[3] = {1, 2, 4};
        E->server_ke = 0;
        for (int c = 0; c < 3 && usable && E->server_ke == 0; ++c) {
            const int ke = choices[c];
            if (fk && atoi(fk) != ke) continue;
            E->server_ke_forced = fk != nullptr;
            if (E->device_dynamics && ke != 1) continue;
            long blocks = 0;
            for (int g = 0; g < E->n_groups; ++g) {
                const long m = (long)E->n_env * (g + 1) / E->n_groups - (long)E->n_env * g / E->n_groups;
                blocks += (m + 4 * ke - 1) / (4 * ke);
            }
            const int cap = egp_pd_server_resident_blocks(ctx->device, E->device_dynamics, ke);
            if (blocks <= cap) { E->server_ke = ke; E->server_cap = cap; }
        }
    }
    for (int g = 0; g < E->n_groups; ++g) {
        Group &G = E->groups[g];
        G.e0 = (int)((long)E->n_env * g / E->n_groups);
        G.e1 = (int)((long)E->n_env * (g + 1) / E->n_groups);
        const int w0 = (int)((long)E->n_threads * g / E->n_groups), w1 = (int)((long)E->n_threads * (g + 1) / E->n_groups);
        G.n_threads = std::max(1, std::min(w1 - w0, G.e1 - G.e0));
        G.bar.n = G.n_threads;
        E_TRY(hipHostMalloc((void **)&G.active, (size_t)E->n_env * sizeof(int), hipHostMallocDefault));
        for (int e = 0; e < E->n_env; ++e) G.active[e] = 1;
        {
            void *p = nullptr;
            E_TRY(hipHostGetDevicePointer(&p, G.active, 0));
            G.hd_active = (int *)p;
        }
        for (int e = G.e0; e < G.e1; ++e) E->env_group[e] = g;
        E_TRY(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
        E_TRY(hipEventCreateWithFlags(&G.done, hipEventDisableTiming));
        E_TRY(hipEventCreateWithFlags(&G.reward_done, hipEventDisableTiming));
        E_TRY(hipMalloc((void **)&G.d_done, sizeof(unsigned)));
        E_TRY(hipMemset(G.d_done, 0, sizeof(unsigned)));
        E_TRY(hipHostMalloc((void **)&G.h_flag, sizeof(unsigned long long), hipHostMallocDefault));
        *G.h_flag = 0;
        {
            void *p = nullptr;
            E_TRY(hipHostGetDevicePointer(&p, G.h_flag, 0));
            G.hd_flag = (unsigned long long *)p;
        }
        // resident K1: 8 slices per thread (fewer for small groups), whole workgroups (4 waves x server_ke envs) each
        if (E->server_ke > 0) {
            Server &S = G.srv;
            const int m = G.e1 - G.e0;
            const int bw = 4 * E->server_ke;
            // one env per wave: workgroup b has the envs 4 b .. 4 b + 3. Several: the group's share of the chip's capacity is used in
            // full and the envs are dealt out evenly over it -- with 240 of 256 CUs to be had that is 4 or 5 envs per workgroup (one
            // wave in five serves two envs), not 8 in half as many
            int nb = (m + bw - 1) / bw;
            if (E->server_ke > 1 && !E->server_ke_forced) {
                const int share = (int)((long)E->server_cap * m / E->n_env);
                nb = std::max(nb, std::min((m + 3) / 4, share));
            }
            std::vector<int> blk_e0(nb + 1);
            for (int b = 0; b <= nb; ++b) blk_e0[b] = E->server_ke == 1 ? std::min(m, 4 * b) : (int)((long)m * b / nb);
            int per_thread = 8;
            while (per_thread > 1 && nb < per_thread * G.n_threads) per_thread /= 2;
            const int ns = std::min(per_thread * G.n_threads, nb);       // a slice is at least one workgroup
            if (nb >= 1) {
                S.n_slices = ns;
                S.n_blocks = nb;
                S.per_thread = per_thread;
                S.can_balance = ns <= 512 && G.n_threads <= 64;
                S.e0.resize(ns);
                S.e1.resize(ns);
                S.dirty.reset(new std::atomic<int>[ns]);
                std::vector<int> block_slice(nb);
                for (int sl = 0; sl < ns; ++sl) {
                    const int b0 = (int)((long)nb * sl / ns), b1 = (int)((long)nb * (sl + 1) / ns);
                    S.e0[sl] = G.e0 + blk_e0[b0];
                    S.e1[sl] = G.e0 + blk_e0[b1];
                    S.dirty[sl].store(0);
                    for (int b = b0; b < b1; ++b) block_slice[b] = sl;
                    for (int e = S.e0[sl]; e < S.e1[sl]; ++e) E->env_slice[e] = sl;
                }
                E_TRY(hipMalloc((void **)&S.d_block_slice, nb * sizeof(int)));
                E_TRY(hipMemcpy(S.d_block_slice, block_slice.data(), nb * sizeof(int), hipMemcpyHostToDevice));
                E_TRY(hipMalloc((void **)&S.d_block_env0, (nb + 1) * sizeof(int)));
                E_TRY(hipMemcpy(S.d_block_env0, blk_e0.data(), (nb + 1) * sizeof(int), hipMemcpyHostToDevice));
                void *p = nullptr;
                if (E->bar_go && hipExtMallocWithFlags(&p, (size_t)ns * 8 * sizeof(unsigned long long), hipDeviceMallocFinegrained) == hipSuccess && p) {
                    // go words next to the state mirror: written by the host through the BAR, polled by the waves in HBM
                    S.h_go = S.hd_go = (unsigned long long *)p;
                    S.go_in_vram = true;
                    E_TRY(hipMemset(p, 0, (size_t)ns * 8 * sizeof(unsigned long long)));
                    E_TRY(hipDeviceSynchronize());
                } else {
                    (void)hipGetLastError();
                    E_TRY(hipHostMalloc((void **)&S.h_go, (size_t)ns * 8 * sizeof(unsigned long long), hipHostMallocDefault));
                    memset(S.h_go, 0, (size_t)ns * 8 * sizeof(unsigned long long));
                    E_TRY(hipHostGetDevicePointer(&p, S.h_go, 0));   S.hd_go = (unsigned long long *)p;
                }
                E_TRY(hipHostMalloc((void **)&S.h_err, 64, hipHostMallocDefault));
                *S.h_err = 0;
                E_TRY(hipHostGetDevicePointer(&p, S.h_err, 0));  S.hd_err = (int *)p;
                if (const char *tr = getenv("EGP_SERVER_TRACE")) {
                    if (atoi(tr) != 0) {
                        E_TRY(hipMalloc((void **)&S.d_trace, (size_t)E->frame_skip * 8 * sizeof(long long)));
                        E_TRY(hipMemset(S.d_trace, 0, (size_t)E->frame_skip * 8 * sizeof(long long)));
                        S.host_trace.assign((size_t)E->frame_skip * 4, 0.0);
                    }
                }
                total_server_blocks += nb;
            } else {
                all_groups_sliced = false;
            }
        }
    }
    E->server_ok = E->server_ke > 0 && all_groups_sliced && total_server_blocks <= E->server_cap;
    if (E->server_ke == 0) E->server_ke = 1;
#undef E_TRY
    for (int g = 0; g < E->n_groups; ++g)
        for (int t = 0; t < E->groups[g].n_threads; ++t) E->groups[g].threads.emplace_back(thread_main, E, g, t);
    *out = E;
    return EGP_OK;
}

int egp_engine_destroy(egp_engine *E) {
    if (!E) return EGP_OK;
    for (auto &G : E->groups) {
        {
            std::lock_guard<std::mutex> lk(G.mu);
            G.quit = true;
            G.cv_go.notify_all();
        }
        for (auto &t : G.threads)
            if (t.joinable()) t.join();
        {
            Server &S = G.srv;
            void *dv[] = {S.d_block_slice, S.d_block_env0, S.d_trace};
            for (void *p : dv) if (p) (void)hipFree(p);
            if (S.h_go) { if (S.go_in_vram) (void)hipFree(S.h_go); else (void)hipHostFree(S.h_go); }
            if (S.h_err) (void)hipHostFree(S.h_err);
        }
        if (G.active) (void)hipHostFree(G.active);
        if (G.d_done) (void)hipFree(G.d_done);
        if (G.h_flag) (void)hipHostFree(G.h_flag);
        if (G.stream) (void)hipStreamDestroy(G.stream);
        if (G.done) (void)hipEventDestroy(G.done);
        if (G.reward_done) (void)hipEventDestroy(G.reward_done);
        for (auto ev : G.k_beg) (void)hipEventDestroy(ev);
        for (auto ev : G.k_end) (void)hipEventDestroy(ev);
    }
    void *dev[] = {E->d_state, E->d_qM, E->d_prev_qpos, E->d_qpos, E->d_qvel, E->d_torque, E->d_ee, E->d_bias, E->d_bias2, E->d_qM2};
    for (void *p : dev) if (p) (void)hipFree(p);
    if (E->reset_done) (void)hipEventDestroy(E->reset_done);
    void *host[] = {E->h_state, E->h_qM, E->h_qpos, E->h_qvel, E->h_torque, E->h_ee, E->h_headz, E->h_xpos, E->h_reset_list};
    for (void *p : host) if (p) (void)hipHostFree(p);
    delete E;
    return EGP_OK;
}

int egp_engine_state(egp_engine *E, double **qpos, double **qvel, double **ee_wpos, double **head_z_host,
                     double **qpos_host, double **qvel_host, double **prev_qpos) {
    EGP_REQUIRE(E, "engine is NULL");
    if (prev_qpos) *prev_qpos = E->d_prev_qpos;
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
    std::vector<char> new_qM(n, 0);
    // the last env-step's uploads read the pinned rows when the copy engine gets to them, which can be after
    // egp_engine_wait returned: let them finish before a reset overwrites those rows
    {
        int last_group = -1;
        for (int k = 0; k < n; ++k) {
            if (ids[k] < 0 || ids[k] >= E->n_env) continue;
            const int g = E->env_group[ids[k]];
            if (g == last_group) continue;
            last_group = g;
            EGP_HIP_CHECK(hipEventSynchronize(E->groups[g].done));
            // the reward kernel of that env-step (terminal step of the slots being reset) runs on the group's stream and
            // still reads their device rows: order this reset's scatter / uploads behind it
            if (E->groups[g].reward_in_flight) EGP_HIP_CHECK(hipStreamWaitEvent(s, E->groups[g].reward_done, 0));
        }
    }
    if (E->reset_pending) {              // the previous reset's kernel reads the pinned rows and the env list
        EGP_HIP_CHECK(hipEventSynchronize(E->reset_done));
        E->reset_pending = false;
    }
    for (int k = 0; k < n; ++k) {
        const int e = ids[k];
        EGP_REQUIRE(e >= 0 && e < E->n_env, "env id out of range");
        EGP_REQUIRE(k == 0 || ids[k] > ids[k - 1], "env ids must be strictly increasing");
        Group &G = E->groups[E->env_group[e]];
        {
            std::lock_guard<std::mutex> lk(G.mu);
            if (G.pending != 0) { egp::set_error("cannot reset env %d while its group is stepping", e); return EGP_E_STATE; }
        }
        const int64_t before = E->epoch[e];
        if (E->vt->reset(E->vt->user, e, qpos + (size_t)k * E->nq, qvel + (size_t)k * E->nv) != 0 || drain_env(E, e, true, false) != EGP_OK) {
            egp::set_error("physics backend failed to reset env %d", e);
            return EGP_E_PHYSICS;
        }
        new_qM[k] = !E->device_dynamics && (!E->vt->inertia_epoch || E->epoch[e] != before);
    }
    if (n > 0) {
        for (int k = 0; k < n; ++k) {
            E->h_reset_list[3 * k] = ids[k];
            E->h_reset_list[3 * k + 1] = new_qM[k];
            E->h_reset_list[3 * k + 2] = E->groups[E->env_group[ids[k]]].dyn_gen;
        }
        k_engine_reset_scatter<<<dim3(n), dim3(256), 0, s>>>(E->hd_reset_list, E->hd_state, E->ld_s, E->off_qpos, E->off_qvel, E->hd_ee,
                                                              E->hd_qM, E->ld_m, E->nM, E->nq, E->nv, E->d_state, E->d_qpos, E->d_qvel,
                                                              E->d_ee, E->d_qM);
        EGP_HIP_CHECK(hipGetLastError());
        if (E->device_dynamics) {
            // sim.forward() of the reset (envs/common/mujoco_env.py:97-101): M, C at the reset state -- the only time compute_torque sees
            // them fresh -- into the generation the env's group reads next
            int rd = egp_launch_dynamics_strided(E->ctx, E->d_qpos, E->nq, E->d_qvel, E->nv, n, E->d_qM, E->ld_m, E->d_bias, E->nv, nullptr, s,
                                                 E->hd_reset_list, 3, E->d_qM2, E->d_bias2);
            if (rd != EGP_OK) return rd;
        }
        EGP_HIP_CHECK(hipEventRecord(E->reset_done, s));
        E->reset_pending = true;
    }
    return EGP_OK;
}

int egp_engine_step_async(egp_engine *E, int32_t group, const double *action, const int32_t *active_host, void *ready_event) {
    EGP_REQUIRE(E && action, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Group &G = E->groups[group];
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.pending != 0) { egp::set_error("group %d is still stepping", group); return EGP_E_STATE; }
    if (G.status.load() != EGP_OK) { egp::set_error("group %d failed earlier: %s", group, G.err); return G.status.load(); }
    G.action = action;
    G.ready = (hipEvent_t)ready_event;
    // a caller that hands over no event has ordered nothing behind its last egp_engine_reset: the env-step's kernel must not
    // read the device rows (state, inertia) while that reset's scatter kernel is still writing them
    if (!G.ready && E->reset_pending) G.ready = E->reset_done;
    G.has_active = active_host != nullptr;
    if (active_host) memcpy(G.active, active_host, E->n_env * sizeof(int));
    G.server_job = server_mode(E, G);
    if (G.server_job) {
        assign_slices(G.srv, G.n_threads, G.has_active ? G.active : nullptr);
        G.srv.base = G.srv.seq;
        G.srv.seq += (unsigned long long)E->frame_skip + 1ull;
    }
    G.pending = G.n_threads;
    G.pending_pub.store(G.n_threads, std::memory_order_relaxed);
    G.job += 1;
    G.job_pub.store(G.job, std::memory_order_release);
    G.cv_go.notify_all();
    return EGP_OK;
}

int egp_engine_set_reward_job(egp_engine *E, int32_t group, const int32_t *t, const int32_t *frame, const int32_t *end,
                              const int32_t *active, double end_reward, double *reward, double *cinfo) {
    EGP_REQUIRE(E && t && frame && end && reward && cinfo, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Group &G = E->groups[group];
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.pending != 0) { egp::set_error("group %d is still stepping", group); return EGP_E_STATE; }
    if (!server_mode(E, G)) { egp::set_error("the reward job rides behind the resident K1 only"); return EGP_E_STATE; }
    G.rjob.t = t; G.rjob.frame = frame; G.rjob.end = end; G.rjob.active = active;
    G.rjob.end_reward = end_reward; G.rjob.reward = reward; G.rjob.cinfo = cinfo;
    G.rjob.armed = true;
    return EGP_OK;
}

int egp_engine_wait(egp_engine *E, int32_t group, void *stream) {
    EGP_REQUIRE(E, "engine is NULL");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Group &G = E->groups[group];
    bool done = false;
    {                                   // an env-step takes a few hundred microseconds: poll before sleeping
        const auto until = clk::now() + std::chrono::microseconds(4 * egp_engine::spin_us);
        int n = 0;
        while (!(done = G.pending_pub.load(std::memory_order_acquire) == 0)) {
            cpu_relax();
            if ((++n & 63) == 0 && clk::now() > until) break;
        }
    }
    if (!done) {
        std::unique_lock<std::mutex> lk(G.mu);
        G.cv_done.wait(lk, [&] { return G.pending == 0; });
    }
    if (G.status.load() != EGP_OK) {
        egp::set_error("rollout group %d: %s", group, G.err);
        return G.status.load();
    }
    // (a caller that works on the group's own stream is already ordered behind the env-step's kernel)
    if ((hipStream_t)stream != G.stream) EGP_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, G.done, 0));
    return EGP_OK;
}

void *egp_engine_group_stream(egp_engine *E, int32_t group) {
    if (!E || group < 0 || group >= E->n_groups) return nullptr;
    return E->groups[group].stream;
}

int egp_rollout_tick_pre(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, void *ready_event,
                         int32_t apply_pending, const double *zf_cur, double *zf_new) {
    EGP_REQUIRE(d && d->ctx && d->eng && ready_event, "NULL pointer");
    EGP_REQUIRE(!apply_pending || (d->defer_apply && k > 0 && zf_cur && zf_new), "a pending apply pass needs defer_apply, k > 0 and the filter states");
    EGP_REQUIRE(group >= 0 && group < d->eng->n_groups, "group out of range");
    EGP_REQUIRE(0 <= a && a < b && b <= d->n_env && b - a <= d->nmax && k >= 0, "slot range / tick out of range");
    const int n = b - a, nmax = d->nmax, N = d->n_env;
    hipStream_t ts = (hipStream_t)d->stream;
    const size_t soff = (size_t)(group * 2 + (k & 1)) * 24 * nmax;
    int32_t *fl = reinterpret_cast<int32_t *>(d->slab_host + soff);
    int64_t *ti = reinterpret_cast<int64_t *>(d->slab_host + soff + 16 * (size_t)nmax);
    // flags of the state this env-step will produce (they do not depend on its outcome) + context rows of this tick
    for (int i = 0; i < n; ++i) {
        const int e = a + i;
        const int act = d->active[e] ? 1 : 0;
        const int64_t t_next = d->cur_t[e] + act;
        fl[i] = (int32_t)t_next;
        fl[nmax + i] = (int32_t)(d->frame_base[e] + t_next);
        fl[2 * nmax + i] = (t_next >= d->episode_len) && act;
        fl[3 * nmax + i] = act;
        ti[i] = d->cur_t[e] < d->ctx_T - 1 ? d->cur_t[e] : d->ctx_T - 1;
    }
    uint8_t *fbase = d->slab_dev + soff;
    int rc = EGP_OK;
    const size_t row = (size_t)k * N + a;
    // The policy kernel moves the flag slab to its device copy itself and reads its context-row indices straight from the pinned
    // one (one dependent operation less per tick than a copy-engine transfer in front of it).
    if (apply_pending) {
        // the apply pass of the previous env-step's filter (its statistics pass ran in `post`) rides in this policy step:
        // next_states[k - 1] and states[k] are written on the way into the MLP
        rc = egp_policy_gaussian_filter_f32(d->ctx, d->v_out + (size_t)a * d->v_stride, d->v_stride, d->ctx_dim,
                                            reinterpret_cast<const int64_t *>(d->slab_host + soff + 16 * (size_t)nmax),
                                            d->qpos + (size_t)a * d->nq, d->qvel + (size_t)a * d->nv,
                                            // (obs_phase: cur_t of the state = the step counter staged for the PREVIOUS env-step, slab slot (k - 1) & 1)
                                            reinterpret_cast<const int32_t *>(d->slab_dev + (size_t)(group * 2 + ((k - 1) & 1)) * 24 * nmax),
                                            n, zf_cur, zf_new, d->zf_clip,
                                            d->next_states + (row - N) * d->obs_dim, d->states + row * d->obs_dim, d->zf_workspace,
                                            d->layers, d->n_layers, d->activation, d->log_std,
                                            d->noise ? d->noise + row * d->nu : nullptr, d->actions + row * d->nu, nullptr,
                                            d->slab_host + soff, fbase, 24 * (int64_t)nmax, ts);
    } else {
        rc = egp_policy_gaussian_staged_f32(d->v_out + (size_t)a * d->v_stride, d->v_stride, d->ctx_dim,
                                            reinterpret_cast<const int64_t *>(d->slab_host + soff + 16 * (size_t)nmax),
                                            d->states + row * d->obs_dim, d->obs_dim, n, d->layers, d->n_layers, d->activation, d->log_std,
                                            d->noise ? d->noise + row * d->nu : nullptr, d->actions + row * d->nu, nullptr,
                                            d->slab_host + soff, fbase, 24 * (int64_t)nmax, ts);
    }
    if (rc != EGP_OK) return rc;
    EGP_HIP_CHECK(hipEventRecord((hipEvent_t)ready_event, ts));
    if (d->reward_job) {      // K2 rides behind this env-step's kernel on the engine's stream
        const int32_t *f32 = reinterpret_cast<const int32_t *>(fbase);
        rc = egp_engine_set_reward_job(d->eng, group, f32, f32 + nmax, f32 + 2 * nmax, f32 + 3 * nmax, d->end_reward, d->rewards + row,
                                       d->cinfo + row * 5);
        if (rc != EGP_OK) return rc;
    }
    for (int e = 0; e < N; ++e) d->active_i32[e] = d->active[e] ? 1 : 0;
    return egp_engine_step_async(d->eng, group, d->actions + (size_t)k * N * d->nu, d->active_i32, ready_event);
}

int egp_rollout_tick_post(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const double *zf_cur, double *zf_new,
                          int32_t *n_done, double *wait_s) {
    EGP_REQUIRE(d && d->ctx && d->eng && n_done, "NULL pointer");
    EGP_REQUIRE(0 <= a && a < b && b <= d->n_env && b - a <= d->nmax && k >= 0, "slot range / tick out of range");
    EGP_REQUIRE(group >= 0 && group < d->eng->n_groups, "group out of range");
    const auto t0 = clk::now();
    hipStream_t ts = (hipStream_t)d->stream;
    int rc = egp_engine_wait(d->eng, group, ts);
    if (wait_s) *wait_s = secs(t0, clk::now());
    if (rc != EGP_OK) return rc;
    const int n = b - a, nmax = d->nmax, N = d->n_env;
    const size_t soff = (size_t)(group * 2 + (k & 1)) * 24 * nmax;
    const int32_t *f32 = reinterpret_cast<const int32_t *>(d->slab_dev + soff);          // the flags `pre` staged for this env-step
    const size_t row = (size_t)k * N + a;
    // the filter -> policy chain of the next tick starts here: K3 + K6 (-> next_states[k] and states[k + 1]) and K2 before the bookkeeping
    if (d->defer_apply)     // statistics pass only: the apply pass is egp_rollout_tick_apply or the next tick's policy step
        rc = egp_obs_zfilter_stats_f64(d->ctx, d->qpos + (size_t)a * d->nq, d->qvel + (size_t)a * d->nv, f32, f32 + 3 * nmax, n, d->zf_workspace, ts);
    else
        rc = egp_obs_zfilter_f64(d->ctx, d->qpos + (size_t)a * d->nq, d->qvel + (size_t)a * d->nv, f32, f32 + 3 * nmax, n, zf_cur, zf_new, d->zf_clip,
                                 d->next_states + row * d->obs_dim, d->states + (row + N) * d->obs_dim, 0, d->zf_workspace, ts);
    if (rc != EGP_OK) return rc;
    if (!d->reward_job) {
        rc = egp_reward_quat_v3_f64(d->ctx, d->qpos + (size_t)a * d->nq, d->prev_qpos + (size_t)a * d->nq, d->ee + (size_t)a * 15, f32, f32 + nmax,
                                    f32 + 2 * nmax, f32 + 3 * nmax, d->end_reward, n, d->rewards + row, d->cinfo + row * 5, ts);
        if (rc != EGP_OK) return rc;
    }
    int nd = 0;
    for (int i = 0; i < n; ++i) {
        const int e = a + i;
        const int act = d->active[e] ? 1 : 0;
        d->cur_t[e] += act;
        const bool fail = d->has_fix_head_lb ? d->head_z[e] < d->fix_head_lb : d->head_z[e] < d->head_lb[d->e_ind[e]] - 0.1;
        const bool end = d->cur_t[e] >= d->episode_len;
        const bool done = (fail || end) && act;
        const size_t r = (size_t)k * N + e;
        d->rec_valid[r] = (uint8_t)act;
        d->rec_done[r] = (uint8_t)done;
        d->rec_e_ind[r] = d->e_ind[e];
        d->rec_s_ind[r] = d->s_ind[e];
        d->steps_done[e] += act;
        nd += done;
    }
    *n_done = nd;
    return EGP_OK;
}

namespace {
// video-context rows of freshly reset slots: block (j, y) copies its share of ctx_rows[j] into v_out[ids[j]]; the id list is
// read in place from pinned memory
__global__ __launch_bounds__(256) void k_ctx_rows_scatter(const int *__restrict__ ids, const float *__restrict__ src, long row_elems,
                                                          float *__restrict__ v_out, long v_stride, int vec4) {
    const int j = blockIdx.x;
    const float *s = src + (long)j * row_elems;
    float *o = v_out + (long)ids[j] * v_stride;
    const long step = (long)gridDim.y * blockDim.x;
    if (vec4) {
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        float4 *o4 = reinterpret_cast<float4 *>(o);
        for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < row_elems / 4; i += step) o4[i] = s4[i];
    } else {
        for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < row_elems; i += step) o[i] = s[i];
    }
}
}  // namespace

int egp_rollout_reset(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const int32_t *ids, int32_t n,
                      const int64_t *e_ind, const int64_t *s_ind, const int64_t *frame_rows, const int64_t *cur_t0, const double *qpos, const double *qvel,
                      const float *ctx_rows, const double *zf_cur, double *zf_new) {
    EGP_REQUIRE(d && d->ctx && d->eng && d->reset_scratch && ids && e_ind && s_ind && frame_rows && qpos && qvel && ctx_rows, "NULL pointer");
    EGP_REQUIRE(0 <= a && a < b && b <= d->n_env && b - a <= d->nmax && k >= 0 && n > 0 && n <= b - a, "slot range / tick out of range");
    for (int j = 0; j < n; ++j) EGP_REQUIRE(ids[j] >= a && ids[j] < b, "reset slot outside its group");
    EGP_REQUIRE(group >= 0 && group < d->eng->n_groups, "group out of range");
    hipStream_t s = (hipStream_t)d->stream;
    int rc = egp_engine_reset(d->eng, ids, n, qpos, qvel, s);               // also checks that the ids increase strictly
    if (rc != EGP_OK) return rc;
    const int nmax = d->nmax, ng = b - a;
    // slot k & 1: its previous readers (tick k - 2 of this group) finished before that tick's env-step started
    int32_t *list = d->reset_scratch + (size_t)(group * 2 + (k & 1)) * 3 * nmax, *mask = list + nmax, *tcur = mask + nmax;
    memset(mask, 0, sizeof(int32_t) * ng);
    for (int j = 0; j < n; ++j) {
        const int e = ids[j];
        list[j] = e;
        mask[e - a] = 1;
        d->e_ind[e] = e_ind[j];
        d->s_ind[e] = s_ind[j];
        d->frame_base[e] = frame_rows[j];
        d->cur_t[e] = cur_t0 ? cur_t0[j] : 0;          // cfg.random_cur_t (humanoid_v1.py:218-220): the episode starts at step cur_t0 of its window
    }
    for (int i = 0; i < ng; ++i) tcur[i] = (int32_t)d->cur_t[a + i];      // obs_phase: cur_t of the group's rows (only the masked ones are read)
    const long row_elems = (long)d->ctx_T * d->ctx_dim;
    const int vec4 = (row_elems % 4 == 0 && d->v_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(ctx_rows) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(d->v_out) & 15) == 0) ? 1 : 0;
    const int per_row = (int)std::min<long>(64, std::max<long>(1, row_elems / (vec4 ? 4096 : 1024)));
    k_ctx_rows_scatter<<<dim3(n, per_row), dim3(256), 0, s>>>(list, ctx_rows, row_elems, const_cast<float *>(d->v_out), d->v_stride, vec4);
    EGP_HIP_CHECK(hipGetLastError());
    // fresh episodes: their first observation goes through the filter and replaces the policy input of tick k + 1
    return egp_obs_zfilter_f64(d->ctx, d->qpos + (size_t)a * d->nq, d->qvel + (size_t)a * d->nv, tcur, mask, ng, zf_cur, zf_new, d->zf_clip,
                               d->states + ((size_t)(k + 1) * d->n_env + a) * d->obs_dim, nullptr, 1, d->zf_workspace, s);
}

int egp_debug_burn(int64_t us, int32_t blocks, float *sink, void *stream) {
    EGP_REQUIRE(us > 0 && blocks > 0 && sink, "bad arguments");
    k_engine_burn<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>((long long)us, sink);
    EGP_HIP_CHECK(hipGetLastError());
    return EGP_OK;
}

// defer_apply: the apply pass of tick k's filter on its own (a tick with in-batch resets -- their masked pass needs the merged
// statistics -- or a group's last tick); otherwise it rides in the next egp_rollout_tick_pre
int egp_rollout_tick_apply(const egp_rollout_tick *d, int32_t group, int32_t a, int32_t b, int32_t k, const double *zf_cur, double *zf_new) {
    EGP_REQUIRE(d && d->ctx && d->eng && d->defer_apply && zf_cur && zf_new, "NULL pointer / defer_apply is off");
    EGP_REQUIRE(group >= 0 && group < d->eng->n_groups, "group out of range");
    EGP_REQUIRE(0 <= a && a < b && b <= d->n_env && b - a <= d->nmax && k >= 0, "slot range / tick out of range");
    const size_t row = (size_t)k * d->n_env + a;
    return egp_obs_zfilter_apply_f64(d->ctx, d->qpos + (size_t)a * d->nq, d->qvel + (size_t)a * d->nv,
                                     reinterpret_cast<const int32_t *>(d->slab_dev + (size_t)(group * 2 + (k & 1)) * 24 * d->nmax),      // tick k's step counter
                                     b - a, zf_cur, zf_new, d->zf_clip,
                                     d->next_states + row * d->obs_dim, d->states + (row + d->n_env) * d->obs_dim, d->zf_workspace,
                                     (hipStream_t)d->stream);
}

double egp_engine_event_overhead_ms(egp_engine *E) {
    double a = 0.0;
    if (E && !E->groups.empty()) { for (auto &G : E->groups) a += G.ev_overhead_ms; a /= E->groups.size(); }
    return a;
}

int64_t egp_engine_inertia_uploads(egp_engine *E) {
    long n = 0;
    if (E) for (auto &G : E->groups) n += G.qM_uploads;
    return n;
}

int egp_engine_timing(egp_engine *E, double *phys_s, double *gpu_wait_s, double *k1_ms, int64_t *k1_launches) {
    EGP_REQUIRE(E, "engine is NULL");
    double p = 0, w = 0, k = 0;
    long l = 0;
    for (auto &G : E->groups) { p += G.phys_s; w += G.wait_s; k += G.k1_ms; l += G.k1_launches; }
    if (phys_s) *phys_s = p;
    if (gpu_wait_s) *gpu_wait_s = w;
    if (k1_ms) *k1_ms = k;
    if (k1_launches) *k1_launches = l;
    return EGP_OK;
}

int64_t egp_engine_k1_env_substeps(egp_engine *E) {
    long n = 0;
    if (E) for (auto &G : E->groups) n += G.k1_env_substeps;
    return n;
}

int egp_engine_set_profile(egp_engine *E, int on) {
    EGP_REQUIRE(E, "engine is NULL");
    E->profile_every = on > 1 ? on : 1;            // on = N > 1: sample every Nth env-step of each group
    for (auto &G : E->groups) EGP_REQUIRE(G.pending == 0, "cannot switch profiling while a group is stepping");
    if (on) {
        EGP_HIP_CHECK(hipSetDevice(E->ctx->device));
        int rc = make_profile_events(E);
        if (rc != EGP_OK) return rc;
    }
    E->profile_k1 = on != 0;
    return EGP_OK;
}

int egp_engine_reset_timing(egp_engine *E) {
    EGP_REQUIRE(E, "engine is NULL");
    for (auto &G : E->groups) { G.phys_s = 0; G.wait_s = 0; G.k1_ms = 0; G.k1_launches = 0; G.k1_env_substeps = 0; }
    return EGP_OK;
}

int egp_engine_layout(egp_engine *E, int32_t *pack_ld, int32_t *n_env, int32_t *n_threads, int32_t *n_groups) {
    EGP_REQUIRE(E, "engine is NULL");
    if (pack_ld) *pack_ld = E->ld_s + E->ld_m;
    if (n_env) *n_env = E->n_env;
    if (n_threads) *n_threads = E->n_threads;
    if (n_groups) *n_groups = E->n_groups;
    return EGP_OK;
}

int egp_engine_server_trace(egp_engine *E, int32_t group, int64_t *device_ticks, double *host_us) {
    EGP_REQUIRE(E && device_ticks && host_us, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    Server &S = E->groups[group].srv;
    if (!S.d_trace) { egp::set_error("engine was created without EGP_SERVER_TRACE=1"); return EGP_E_STATE; }
    EGP_HIP_CHECK(hipMemcpy(device_ticks, S.d_trace, (size_t)E->frame_skip * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    memcpy(host_us, S.host_trace.data(), S.host_trace.size() * sizeof(double));
    return EGP_OK;
}

int egp_engine_substeps_per_launch(egp_engine *E) {
    if (!E || E->groups.empty()) return 0;
    return server_mode(E, E->groups[0]) ? E->frame_skip : 1;
}

int32_t egp_engine_envs_per_wave(egp_engine *E, int32_t *resident_capacity) {
    if (resident_capacity) *resident_capacity = E ? E->server_cap : 0;
    if (!E || E->groups.empty() || !server_mode(E, E->groups[0])) return 0;
    return E->server_ke;
}

int32_t egp_engine_go_words_in_vram(egp_engine *E) {
    if (!E || E->groups.empty() || !server_mode(E, E->groups[0])) return -1;
    return E->groups[0].srv.go_in_vram ? 1 : 0;
}

int egp_engine_group_range(egp_engine *E, int32_t group, int32_t *e0, int32_t *e1) {
    EGP_REQUIRE(E && e0 && e1, "NULL pointer");
    EGP_REQUIRE(group >= 0 && group < E->n_groups, "group out of range");
    *e0 = E->groups[group].e0;
    *e1 = E->groups[group].e1;
    return EGP_OK;
}

}  // extern "C"
