// Issue rate of the float32 MFMA forms the LSTM recurrences could use (gfx950): cycles per instruction with two
// independent accumulator chains per wave, one wave per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void k(float *out, long long *cyc, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    f32x16 d0 = {0}, d1 = {0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0); }
            if (KIND == 1) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0); }
            if (KIND == 2) { d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + d0[0] + d1[3];
}
int main() {
    float *out; long long *cyc, h[4];
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const int iters = 2000;
    const char *names[3] = {"v_mfma_f32_4x4x1_16b_f32 (256 MAC)", "v_mfma_f32_16x16x4_f32 (1024 MAC)", "v_mfma_f32_32x32x2_f32 (2048 MAC)"};
    const double macs[3] = {256, 1024, 2048};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int kind = 0; kind < 3; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) k<0><<<1, 64 * waves * 4>>>(out, cyc, iters);
                if (kind == 1) k<1><<<1, 64 * waves * 4>>>(out, cyc, iters);
                if (kind == 2) k<2><<<1, 64 * waves * 4>>>(out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
            // s_memtime / readcyclecounter counts at 100 MHz on gfx9: report per-instruction time from wall-clock via events instead
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kind == 0) k<0><<<256, 64 * waves * 4>>>(out, cyc, iters);
            if (kind == 1) k<1><<<256, 64 * waves * 4>>>(out, cyc, iters);
            if (kind == 2) k<2><<<256, 64 * waves * 4>>>(out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n_inst = (double)iters * 32;            // per wave
            const double ns_per_inst_per_simd = ms * 1e6 / (n_inst * waves);
            printf("%-36s %d wave(s)/SIMD: %.2f ns per instruction and SIMD  -> %.1f MAC/ns/SIMD, chip %.1f TFLOP/s\n", names[kind], waves,
                   ns_per_inst_per_simd, macs[kind] / ns_per_inst_per_simd, macs[kind] / ns_per_inst_per_simd * 2 * 4 * 256 / 1e3);
        }
    return 0;
}
