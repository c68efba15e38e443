// Host probe: two seconds of measurements that tell a slow BOX from slow CODE (include/egopose_hip.h: egp_host_probe).
//
// The rollout's env-step is a latency chain through the host (egp_engine.hip: per substep a go word host -> GPU, the state rows
// GPU <- pinned host memory over PCIe, the torque rows GPU -> pinned host memory, 15 times per env-step, with a dozen host threads
// spinning on those rows), so its rate moves with the box: PCIe read bandwidth, the round-trip latency of one go word, and whether the
// spinning threads keep their cores. bench.py runs this probe in front of the timed region and puts the numbers into the JSON line's
// `config`; two runs of the same commit with different env-steps/s then show which of the three moved -- or that none did.
//
// What it measures, each with the engine's own access shapes:
//   pcie_read   1 024 state rows of 176 doubles in pinned host memory, one wave per row, system-scope loads of whole 64-byte lines
//               (the resident K1's contiguous row read), HIP events around 100 passes
//   go_rtt      one resident wave: lane 0 polls a go word through the scalar path (s_load ... glc), the wave reads ONE state row over
//               PCIe and writes a 52-double "torque" row with system-scope stores into pinned host memory; the host thread that raised
//               the go word spins on the row's sentinel words: microseconds from the go store to the row's arrival, n round trips
//   spin        n_threads threads spin on the clock for `millis` ms each (as the physics threads spin on torque rows): the gaps between
//               two consecutive clock reads are time the thread was not running (preemption, cgroup throttling, SMIs)
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <mutex>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "egp_internal.hpp"

namespace {

using clk = std::chrono::steady_clock;
inline double us_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
inline void relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

constexpr int PR_LD = 176, PR_NU = 52;
constexpr unsigned long long PR_SENTINEL = 0x7FF8DEADBEEF0001ull;

__device__ __forceinline__ double pr_sys_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM));
}

// one wave per row, three system-scope 8-byte loads per lane over the contiguous row (k_pd_server_tree58's row_contig read)
__global__ __launch_bounds__(256) void k_probe_rows(const double *rows, int n_rows, int passes, double *sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    double acc = 0.0;
    if (wave < n_rows) {
        for (int it = 0; it < passes; ++it) {
            const double *p = rows + (long)wave * PR_LD;
            const double c0 = pr_sys_load(p + lane), c1 = pr_sys_load(p + 64 + lane), c2 = lane < PR_LD - 128 ? pr_sys_load(p + 128 + lane) : 0.0;
            acc += c0 + c1 + c2;
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

// the resident wave of the go-word round trip
__global__ __launch_bounds__(64) void k_probe_go(const unsigned long long *go, const double *row, double *torque, int n, long long timeout_ticks,
                                                 int *err) {
    const int lane = threadIdx.x;
    __shared__ int s_abort;
    if (lane == 0) s_abort = 0;
    for (int i = 1; i <= n; ++i) {
        if (lane == 0) {
            const long long t0 = wall_clock64();
            for (;;) {
                unsigned long long v;
                asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(go) : "memory");
                if (v >= (unsigned long long)i) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > timeout_ticks) { s_abort = 1; break; }
            }
        }
        __syncthreads();
        if (s_abort) {
            if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const double c0 = pr_sys_load(row + lane), c1 = pr_sys_load(row + 64 + lane), c2 = lane < PR_LD - 128 ? pr_sys_load(row + 128 + lane) : 0.0;
        const double v = c0 + c1 + c2;               // (never NaN: the host fills the row with small numbers)
        if (lane < PR_NU)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(torque + lane), (unsigned long long)__double_as_longlong(v + (double)i),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
}

// Usable-CU probe (egp_device_usable_cus): one workgroup per CU by construction (each takes more than half of a CU's LDS), every
// workgroup counts itself in, holds its place until all gridDim.x are there or the timeout passes, and counts itself out; probe[1] =
// the largest head count seen = the CUs this process really gets. hipDeviceAttributeMultiprocessorCount is the data sheet's answer:
// under HSA_CU_MASK it stays 256, under ROC_GLOBAL_CU_MASK it reports the mask's bits (240) where 225 workgroups fit.
__global__ __launch_bounds__(64) void k_cu_probe(unsigned *probe, long long timeout_ticks) {
    extern __shared__ char s_hold[];
    if (timeout_ticks < 0) s_hold[threadIdx.x] = 0;          // (keeps the allocation: never taken)
    if (threadIdx.x != 0) return;
    unsigned present = atomicAdd(probe, 1u) + 1u, best = present;
    const long long t0 = wall_clock64();
    for (;;) {
        if (present >= gridDim.x) { atomicExch(probe + 2, 1u); break; }
        if (__hip_atomic_load(probe + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
        present = __hip_atomic_load(probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        best = present > best ? present : best;
    }
    atomicMax(probe + 1, best);
    atomicSub(probe, 1u);
}

inline double percentile(std::vector<double> &v, double q) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    const size_t i = (size_t)std::min<double>((double)v.size() - 1.0, q * (double)(v.size() - 1) + 0.5);
    return v[i];
}

#define P_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { egp::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); rc = EGP_E_HIP; goto done; } } while (0)

}  // namespace

extern "C" int32_t egp_device_usable_cus(int32_t device) {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    int caller = -1, n_cu = 0;
    if (hipGetDevice(&caller) != hipSuccess) { (void)hipGetLastError(); caller = -1; }
    int result = 0;
    do {
        if (hipSetDevice(device) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); break; }
        result = n_cu;
        const char *pe = getenv("EGP_SERVER_PROBE");
        if (n_cu <= 0 || (pe && atoi(pe) == 0)) break;
        const size_t lds = 96 * 1024;             // more than half of a CU's 160 kB: at most one of these workgroups per CU
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cu_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); break; }
        unsigned *d_probe = nullptr, h[4] = {0, 0, 0, 0};
        if (hipMalloc((void **)&d_probe, sizeof(h)) != hipSuccess) { (void)hipGetLastError(); break; }
        bool ok = hipMemset(d_probe, 0, sizeof(h)) == hipSuccess;
        if (ok) {
            k_cu_probe<<<dim3(n_cu), dim3(64), lds, nullptr>>>(d_probe, 100000);        // 1 ms: whoever is not on the chip by then is not resident
            ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, d_probe, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
        }
        (void)hipFree(d_probe);
        if (!ok) { (void)hipGetLastError(); break; }
        if ((int)h[1] > 0 && (int)h[1] < result) result = (int)h[1];
    } while (false);
    if (caller >= 0 && caller != device) (void)hipSetDevice(caller);
    cache[device] = result;
    return result;
}

extern "C" int egp_host_probe(int32_t device, int32_t n_threads, int32_t millis, egp_host_probe_result *out) {
    EGP_REQUIRE(out && n_threads >= 0 && millis > 0 && millis <= 2000, "bad probe arguments");
    memset(out, 0, sizeof(*out));
    int rc = EGP_OK;
    double *h_rows = nullptr, *hd_rows = nullptr, *h_tq = nullptr, *hd_tq = nullptr, *d_sink = nullptr;
    unsigned long long *h_go = nullptr, *hd_go = nullptr;
    int *h_err = nullptr, *hd_err = nullptr;
    bool go_in_vram = false;
    hipStream_t st = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    const int n_rows = 1024;
    int caller_device = -1;                    // the calling thread's current device is the caller's business: put it back on the way out
    if (hipGetDevice(&caller_device) != hipSuccess) { (void)hipGetLastError(); caller_device = -1; }
    EGP_HIP_CHECK(hipSetDevice(device));
    {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) (void)hipGetLastError();
        out->large_bar = large_bar;
    }
    P_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    P_TRY(hipEventCreate(&ea));
    P_TRY(hipEventCreate(&eb));
    P_TRY(hipHostMalloc((void **)&h_rows, (size_t)n_rows * PR_LD * sizeof(double), hipHostMallocDefault));
    P_TRY(hipHostMalloc((void **)&h_tq, 64 * sizeof(double), hipHostMallocDefault));
    P_TRY(hipHostMalloc((void **)&h_err, 64, hipHostMallocDefault));
    P_TRY(hipMalloc((void **)&d_sink, 64));
    for (long i = 0; i < (long)n_rows * PR_LD; ++i) h_rows[i] = (double)(i % 977) * 1e-3;
    *h_err = 0;
    { void *p = nullptr; P_TRY(hipHostGetDevicePointer(&p, h_rows, 0)); hd_rows = (double *)p; }
    { void *p = nullptr; P_TRY(hipHostGetDevicePointer(&p, h_tq, 0)); hd_tq = (double *)p; }
    { void *p = nullptr; P_TRY(hipHostGetDevicePointer(&p, h_err, 0)); hd_err = (int *)p; }
    // ---- pinned-row PCIe read
    {
        const int passes = 100, blocks = n_rows / 4;
        k_probe_rows<<<dim3(blocks), dim3(256), 0, st>>>(hd_rows, n_rows, 5, d_sink);
        P_TRY(hipStreamSynchronize(st));
        P_TRY(hipEventRecord(ea, st));
        k_probe_rows<<<dim3(blocks), dim3(256), 0, st>>>(hd_rows, n_rows, passes, d_sink);
        P_TRY(hipEventRecord(eb, st));
        P_TRY(hipEventSynchronize(eb));
        float ms = 0.f;
        P_TRY(hipEventElapsedTime(&ms, ea, eb));
        out->pcie_read_us_per_pass = (double)ms * 1e3 / passes;
        out->pcie_read_gbps = (double)n_rows * PR_LD * 8.0 * passes / ((double)ms * 1e-3) / 1e9;
        out->pcie_read_rows = n_rows;
    }
    // ---- go-word round trip (the go word where the engine puts it: fine-grained device memory behind a large BAR, else pinned)
    {
        void *p = nullptr;
        if (out->large_bar && hipExtMallocWithFlags(&p, 64, hipDeviceMallocFinegrained) == hipSuccess && p) {
            h_go = hd_go = (unsigned long long *)p;
            go_in_vram = true;
            P_TRY(hipMemset(p, 0, 64));
            P_TRY(hipDeviceSynchronize());
        } else {
            (void)hipGetLastError();
            P_TRY(hipHostMalloc((void **)&h_go, 64, hipHostMallocDefault));
            *h_go = 0;
            P_TRY(hipHostGetDevicePointer(&p, h_go, 0));
            hd_go = (unsigned long long *)p;
        }
        out->go_in_vram = go_in_vram;
        const int n = 1500;
        std::vector<double> rtt;
        rtt.reserve(n);
        unsigned long long *tq_u = reinterpret_cast<unsigned long long *>(h_tq);
        for (int k = 0; k < PR_NU; k += 4) tq_u[k] = PR_SENTINEL;
        k_probe_go<<<dim3(1), dim3(64), 0, st>>>(hd_go, hd_rows, hd_tq, n, 200000000ll /* 2 s */, hd_err);
        P_TRY(hipGetLastError());
        bool failed = false;
        for (int i = 1; i <= n && !failed; ++i) {
            for (int k = 0; k < PR_NU; k += 4) __atomic_store_n(tq_u + k, PR_SENTINEL, __ATOMIC_RELAXED);
            const auto t0 = clk::now();
#if defined(__x86_64__)
            if (go_in_vram) _mm_sfence();
#endif
            __atomic_store_n(h_go, (unsigned long long)i, __ATOMIC_RELEASE);
#if defined(__x86_64__)
            if (go_in_vram) _mm_sfence();
#endif
            long spins = 0;
            for (;;) {
                bool ok = true;
                for (int k = 0; k < PR_NU; k += 4) ok &= __atomic_load_n(tq_u + k, __ATOMIC_RELAXED) != PR_SENTINEL;
                if (ok) break;
                relax();
                if ((++spins & 0xFFF) == 0 && (us_between(t0, clk::now()) > 1e6 || __atomic_load_n(h_err, __ATOMIC_ACQUIRE) != 0)) { failed = true; break; }
            }
            if (!failed && i > 20) rtt.push_back(us_between(t0, clk::now()));     // (the first round trips include the launch)
        }
        if (failed) {                       // let the wave run out, report nothing
            __atomic_store_n(h_go, ~0ull >> 1, __ATOMIC_RELEASE);
#if defined(__x86_64__)
            _mm_sfence();
#endif
        }
        P_TRY(hipStreamSynchronize(st));
        out->go_rtt_n = (int32_t)rtt.size();
        if (!rtt.empty()) {
            double mx = 0.0;
            for (double v : rtt) mx = std::max(mx, v);
            out->go_rtt_us_max = mx;
            out->go_rtt_us_p50 = percentile(rtt, 0.50);
            out->go_rtt_us_p99 = percentile(rtt, 0.99);
        }
    }
    // ---- spinning threads: are they left alone?
    if (n_threads > 0) {
        std::vector<double> worst(n_threads, 0.0), lost(n_threads, 0.0), total(n_threads, 0.0);
        std::vector<long> big(n_threads, 0);
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                while (!go.load(std::memory_order_acquire)) relax();
                const auto t0 = clk::now();
                auto prev = t0;
                const double budget = (double)millis * 1e3;
                for (;;) {
                    relax();
                    const auto now = clk::now();
                    const double gap = us_between(prev, now);
                    if (gap > worst[t]) worst[t] = gap;
                    if (gap > 5.0) { lost[t] += gap; big[t] += 1; }
                    prev = now;
                    if (us_between(t0, now) > budget) { total[t] = us_between(t0, now); break; }
                }
            });
        go.store(1, std::memory_order_release);
        for (auto &x : th) x.join();
        double l = 0.0, tt = 0.0, mx = 0.0;
        long nb = 0;
        for (int t = 0; t < n_threads; ++t) { l += lost[t]; tt += total[t]; mx = std::max(mx, worst[t]); nb += big[t]; }
        out->spin_threads = n_threads;
        out->spin_gap_us_max = mx;
        out->spin_gap_us_median_of_thread_max = percentile(worst, 0.5);
        out->spin_lost_frac = tt > 0 ? l / tt : 0.0;
        out->spin_gaps_over_5us = (int32_t)std::min<long>(nb, 0x7FFFFFFF);
    }
done:
    if (h_go) { if (go_in_vram) (void)hipFree(h_go); else (void)hipHostFree(h_go); }
    if (h_rows) (void)hipHostFree(h_rows);
    if (h_tq) (void)hipHostFree(h_tq);
    if (h_err) (void)hipHostFree(h_err);
    if (d_sink) (void)hipFree(d_sink);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    if (st) (void)hipStreamDestroy(st);
    if (caller_device >= 0 && caller_device != device) (void)hipSetDevice(caller_device);
    return rc;
}
