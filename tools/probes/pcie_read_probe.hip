// How fast do waves pull state rows out of pinned host memory, by access shape?  (the full-activity env-step is bound by it:
// 1 024 envs x 176 doubles per substep.)   hipcc --offload-arch=gfx950 -O3 pcie_read_probe.hip -o pcie_read_probe.bin
//   A  K1's shape: one wave per row, lane = dof, three 8-byte loads per lane (qpos | qvel | bias segments of 58 doubles)
//   B  one wave per row, 16 bytes per lane over the contiguous row (2 instructions for 1 408 bytes)
//   C  one wave per row, 8 bytes per lane over the contiguous row (3 instructions)
//   D  the whole buffer as one contiguous stream, 16 bytes per lane, grid-stride (upper bound for wave-issued reads)
//   E  hipMemcpyAsync host -> device of the same bytes (the copy engine)
// Every kernel re-reads the rows `iters` times (host memory is not cached on the device: each pass crosses PCIe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int LD = 176;
typedef double d2 __attribute__((ext_vector_type(2)));
// system-scope loads (sc0 sc1): what a kernel must use to see rows the host has just written -- they bypass the device's caches
__device__ __forceinline__ double ld8(const double *p) { double v; asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ d2 ld16(const d2 *p) { d2 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
#define WAIT_() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
template <int MODE>
__global__ void k_read(const double *__restrict__ rows, int n_rows, int iters, double *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        int zero = 0;
        asm volatile("" : "+v"(zero));          // an offset the compiler cannot see through: every pass reloads
        const double *rows_ = rows + zero;
        if (MODE == 0) {
            for (int r = wave; r < n_rows; r += n_waves) {
                const double *p = rows_ + (long)r * LD;
                double a = 0, b = 0, c = 0;
                if (lane < 58) { a = ld8(p + lane); b = ld8(p + 59 + lane); c = ld8(p + 117 + lane); }
                WAIT_();
                acc += a + b + c;
            }
        } else if (MODE == 1) {
            for (int r = wave; r < n_rows; r += n_waves) {
                const d2 *p = reinterpret_cast<const d2 *>(rows_ + (long)r * LD);
                d2 a = ld16(p + lane), b = {0, 0};
                if (lane < 24) b = ld16(p + 64 + lane);
                WAIT_();
                acc += a.x + a.y + b.x + b.y;
            }
        } else if (MODE == 2) {
            for (int r = wave; r < n_rows; r += n_waves) {
                const double *p = rows_ + (long)r * LD;
                double a = ld8(p + lane), b = ld8(p + 64 + lane), c = 0;
                if (lane < 48) c = ld8(p + 128 + lane);
                WAIT_();
                acc += a + b + c;
            }
        } else {
            const d2 *p = reinterpret_cast<const d2 *>(rows_);
            const long n2 = (long)n_rows * LD / 2;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
                d2 a = ld16(p + i);
                WAIT_();
                acc += a.x + a.y;
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <int MODE>
static void run(const char *name, const double *d_rows, int n_rows, int blocks, double *sink) {
    const int iters = 200;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_read<MODE><<<blocks, 256>>>(d_rows, n_rows, 5, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_read<MODE><<<blocks, 256>>>(d_rows, n_rows, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)n_rows * LD * 8 * iters;
    printf("%-58s %4d rows, %3d blocks: %7.2f us per pass, %6.1f GB/s\n", name, n_rows, blocks, ms * 1e3 / iters, bytes / (ms * 1e-3) / 1e9);
}
int main() {
    double *h, *d, *sink, *dev_copy;
    const int max_rows = 2048;
    CK(hipHostMalloc((void **)&h, (size_t)max_rows * LD * 8, hipHostMallocDefault));
    for (long i = 0; i < (long)max_rows * LD; ++i) h[i] = i * 1e-6;
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    CK(hipMalloc((void **)&sink, 64)); CK(hipMalloc((void **)&dev_copy, (size_t)max_rows * LD * 8));
    for (int n_rows : {512, 1024, 2048}) {
        const int blocks = n_rows / 4;               // one wave per row, 4 waves per block as K1
        run<0>("A lane = dof, 3 x 8 B per lane (K1 today)", d, n_rows, blocks, sink);
        run<1>("B contiguous row, 16 B per lane", d, n_rows, blocks, sink);
        run<2>("C contiguous row, 8 B per lane", d, n_rows, blocks, sink);
        run<3>("D one contiguous stream, 16 B per lane", d, n_rows, blocks, sink);
        run<3>("D' the same with 1 024 blocks", d, n_rows, 1024, sink);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipMemcpyAsync(dev_copy, h, (size_t)n_rows * LD * 8, hipMemcpyHostToDevice, 0)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 50; ++i) CK(hipMemcpyAsync(dev_copy, h, (size_t)n_rows * LD * 8, hipMemcpyHostToDevice, 0));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-58s %4d rows             : %7.2f us per copy, %6.1f GB/s\n", "E hipMemcpyAsync (copy engine), back to back", n_rows, ms * 1e3 / 50, (double)n_rows * LD * 8 * 50 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
