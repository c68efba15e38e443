// Can host threads write straight into device memory (large BAR), and how fast does a resident kernel see it?
// hipcc --offload-arch=gfx950 -O2 tools/probes/bar_probe.hip -o tools/probes/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
#include <vector>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_sum(const unsigned long long *p, int n, unsigned long long *out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += __builtin_nontemporal_load(p + i);
    atomicAdd(out, s);
}
// ping-pong: wait for dev_flag == i (host-written, device memory), answer in host_flag (pinned) ; n rounds
__global__ void k_pingpong(volatile unsigned long long *dev_flag, volatile unsigned long long *host_flag, int n) {
    for (int i = 1; i <= n; ++i) {
        while (__hip_atomic_load((unsigned long long *)dev_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)i) {}
        __hip_atomic_store((unsigned long long *)host_flag, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main() {
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    const size_t bytes = 64 << 20;
    for (int mode = 0; mode < 3; ++mode) {
        void *p = nullptr;
        hipError_t e;
        const char *name = mode == 0 ? "hipExtMallocWithFlags(Finegrained)" : mode == 1 ? "hipMalloc" : "hipExtMallocWithFlags(Uncached)";
        if (mode == 0) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        else if (mode == 1) e = hipMalloc(&p, bytes);
        else e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("%s: alloc failed: %s\n", name, hipGetErrorString(e)); continue; }
        if (sigsetjmp(jb, 1)) { printf("%s: host write FAULTED\n", name); continue; }
        volatile unsigned long long *q = (volatile unsigned long long *)p;
        q[0] = 1; q[1] = 2; q[511] = 3;
        _mm_sfence();
        printf("%s: host write ok\n", name);
        unsigned long long *out; CK(hipHostMalloc(&out, 8)); *out = 0;
        k_sum<<<1, 64>>>((const unsigned long long *)p, 512, out); CK(hipDeviceSynchronize());
        printf("  kernel sees sum %llu (expect 6 + garbage-free? first run may include stale) \n", *out);
        // bandwidth of host stores into it
        std::vector<char> src(bytes, 7);
        auto t0 = std::chrono::steady_clock::now();
        memcpy(p, src.data(), bytes); _mm_sfence();
        auto t1 = std::chrono::steady_clock::now();
        printf("  host memcpy -> device: %.1f GB/s (1 thread)\n", bytes / std::chrono::duration<double>(t1 - t0).count() / 1e9);
        // round trip: host writes flag in device memory, kernel answers in pinned host memory
        unsigned long long *hflag; CK(hipHostMalloc(&hflag, 64)); *hflag = 0; q[0] = 0; _mm_sfence();
        const int n = 2000;
        k_pingpong<<<1, 1>>>((volatile unsigned long long *)p, hflag, n);
        auto a0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= n; ++i) {
            q[0] = i; _mm_sfence();
            while (*(volatile unsigned long long *)hflag != (unsigned long long)i) _mm_pause();
        }
        auto a1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("  ping-pong (flag in device memory, answer in pinned host memory): %.2f us per round trip\n",
               std::chrono::duration<double>(a1 - a0).count() / n * 1e6);
        // reference: flag in pinned host memory (what the engine does today)
        unsigned long long *hgo; CK(hipHostMalloc(&hgo, 64)); *hgo = 0; *hflag = 0;
        unsigned long long *dgo; CK(hipHostGetDevicePointer((void **)&dgo, hgo, 0));
        k_pingpong<<<1, 1>>>(dgo, hflag, n);
        a0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= n; ++i) {
            *(volatile unsigned long long *)hgo = i; _mm_sfence();
            while (*(volatile unsigned long long *)hflag != (unsigned long long)i) _mm_pause();
        }
        a1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("  ping-pong (flag in pinned host memory): %.2f us per round trip\n", std::chrono::duration<double>(a1 - a0).count() / n * 1e6);
        hipFree(p);
    }
    return 0;
}
