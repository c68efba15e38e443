// Wave-issued system-scope stores of torque rows into pinned host memory, by row stride (52 doubles per row as the engine's
// [N][nu] array, or padded to 56 = whole 64-byte lines).   hipcc --offload-arch=gfx950 -O3 pcie_write_probe.hip -o pcie_write_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_write(double *rows, int n_rows, int ld, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int it = 0; it < iters; ++it) {
        for (int r = wave; r < n_rows; r += n_waves)
            if (lane >= 6 && lane < 58)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(rows + (long)r * ld + lane - 6), (unsigned long long)(it + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __builtin_amdgcn_s_barrier();
    }
}
int main() {
    double *h, *d;
    const int max_rows = 2048;
    CK(hipHostMalloc((void **)&h, (size_t)max_rows * 64 * 8, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    for (int n_rows : {512, 1024})
        for (int ld : {52, 56, 64}) {
            const int iters = 200;
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            k_write<<<n_rows / 4, 256>>>(d, n_rows, ld, 5); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            k_write<<<n_rows / 4, 256>>>(d, n_rows, ld, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%4d rows of 52 doubles, row stride %2d doubles: %7.2f us per pass, %6.1f GB/s of payload\n", n_rows, ld, ms * 1e3 / iters, (double)n_rows * 52 * 8 * iters / (ms * 1e-3) / 1e9);
        }
    return 0;
}
