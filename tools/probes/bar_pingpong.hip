// Host <-> GPU round trip per substep, two ways of handing a state row + go word to a resident wave:
//   A: row and go word in pinned HOST memory, the wave polls and reads over PCIe (what the engine does today)
//   B: row and go word in fine-grained DEVICE memory, the host writes them through the BAR (posted writes), the wave polls HBM
// In both the answer (a "torque" row + sentinel) goes to pinned host memory.   hipcc --offload-arch=gfx950 -O3 bar_pingpong.hip -o bar_pingpong.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <atomic>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ROW = 128;    // doubles per state row (the engine's is 117 + padding)
constexpr int OUT = 52;
__global__ void k_server(const double *state, const unsigned long long *go, double *out, int iters, int rows) {
    const int lane = threadIdx.x;
    for (int i = 1; i <= iters; ++i) {
        if (lane == 0) {
            while (__hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)i) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_s_barrier();
        __threadfence_system();
        double acc = 0.0;
        for (int r = 0; r < rows; ++r) {
            const double *row = state + (long)r * ROW;
            acc += __builtin_nontemporal_load(row + lane) + __builtin_nontemporal_load(row + 64 + lane);
        }
        for (int r = 0; r < rows; ++r)
            if (lane < OUT) out[(long)r * 64 + lane] = acc + i;          // every word != sentinel
        __threadfence_system();
    }
}
int main(int argc, char **argv) {
    const int iters = 20000, rows = argc > 1 ? atoi(argv[1]) : 4;
    double *h_state, *d_state_fg, *h_out; unsigned long long *h_go, *d_go_fg;
    CK(hipHostMalloc((void **)&h_state, rows * ROW * 8, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_go, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_out, rows * 64 * 8, hipHostMallocDefault));
    hipError_t e = hipExtMallocWithFlags((void **)&d_state_fg, rows * ROW * 8, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    CK(hipExtMallocWithFlags((void **)&d_go_fg, 64, hipDeviceMallocFinegrained));
    CK(hipMemset(d_go_fg, 0, 64)); CK(hipMemset(d_state_fg, 0, rows * ROW * 8)); CK(hipDeviceSynchronize());
    hipPointerAttribute_t at; CK(hipPointerGetAttributes(&at, d_go_fg));
    printf("device pointer %p, host view %p, type %d\n", at.devicePointer, at.hostPointer, (int)at.type);
    for (int mode = 0; mode < 2; ++mode) {
        double *st = mode == 0 ? h_state : d_state_fg;
        unsigned long long *go = mode == 0 ? h_go : d_go_fg;
        if (mode == 0) { *h_go = 0; } else { CK(hipMemset(d_go_fg, 0, 64)); CK(hipDeviceSynchronize()); }
        double *d_st = st; unsigned long long *d_go = go; double *d_out;
        if (mode == 0) { CK(hipHostGetDevicePointer((void **)&d_st, h_state, 0)); CK(hipHostGetDevicePointer((void **)&d_go, h_go, 0)); }
        CK(hipHostGetDevicePointer((void **)&d_out, h_out, 0));
        const unsigned long long SENT = 0x7ff8dead00000000ull;
        for (int r = 0; r < rows * 64; ++r) ((unsigned long long *)h_out)[r] = SENT;
        k_server<<<1, 64>>>(d_st, d_go, d_out, iters, rows);
        int stale = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= iters; ++i) {
            for (int r = 0; r < rows; ++r) ((unsigned long long *)h_out)[(long)r * 64 + OUT - 4] = SENT;   // arm (last sector)
            for (int k = 0; k < rows * ROW; ++k) st[k] = i + k * 1e-3;      // the new state rows (host stores; B: through the BAR)
            _mm_sfence();
            __atomic_store_n(go, (unsigned long long)i, __ATOMIC_RELEASE);
            for (int r = 0; r < rows; ++r)
                while (__atomic_load_n((unsigned long long *)h_out + (long)r * 64 + OUT - 4, __ATOMIC_ACQUIRE) == SENT) _mm_pause();
            // the wave must have seen THIS iteration's rows: out[lane] = sum over rows of (st[row][lane] + st[row][64 + lane]) + i
            for (int lane = 0; lane < OUT; lane += 17) {
                double want = 0.0;
                for (int r = 0; r < rows; ++r) want += (i + (r * ROW + lane) * 1e-3) + (i + (r * ROW + 64 + lane) * 1e-3);
                want += i;
                const double got = h_out[lane];
                if (!(got > want - 1e-6 && got < want + 1e-6)) { if (stale++ < 5) printf("  stale/wrong data at iteration %d lane %d: got %.6f want %.6f\n", i, lane, got, want); }
            }
        }
        printf("  data check: %d mismatches\n", stale);
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("%s: %.2f us per round trip (%d rows of %d doubles in, %d of %d out)\n", mode == 0 ? "A host-pinned state + go (GPU polls/reads over PCIe)"
               : "B device fine-grained state + go (host writes through the BAR)", std::chrono::duration<double>(t1 - t0).count() / iters * 1e6, rows, ROW, rows, OUT);
    }
    return 0;
}
