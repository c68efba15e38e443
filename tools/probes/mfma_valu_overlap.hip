// Do the matrix pipe and the vector ALU of a SIMD overlap when they are fed by DIFFERENT waves (one wave of back-to-back
// v_mfma_f32_32x32x16_bf16, one wave of independent v_and / v_sub / v_perm chains -- the consumer / producer pair of k_gemm_ws)?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap.bin tools/probes/mfma_valu_overlap.hip ; run: ./mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mfma_wave && (mode & 1)) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        for (int it = 0; it < iters; ++it) {           // 4 independent accumulators: the pipe stays full
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
        }
        sink[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    }
    if (!mfma_wave && (mode & 2)) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1.37f + i;
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {           // per iteration: 8 x (and, sub, and, sub) + 12 perms = 44 VALU ops, 8 chains
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const unsigned u0 = __float_as_uint(x[i]), u1 = __float_as_uint(x[i + 1]);
                const float h0 = __uint_as_float(u0 & 0xffff0000u), h1 = __uint_as_float(u1 & 0xffff0000u);
                const float r0 = x[i] - h0, r1 = x[i + 1] - h1;
                const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
                const float m0 = __uint_as_float(v0 & 0xffff0000u), m1 = __uint_as_float(v1 & 0xffff0000u);
                const float l0 = r0 - m0, l1 = r1 - m1;
                acc ^= __builtin_amdgcn_perm(u1, u0, 0x07060302u) ^ __builtin_amdgcn_perm(v1, v0, 0x07060302u) ^
                       __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
                x[i] = l0 + 1.0001f * x[i]; x[i + 1] = l1 + 0.9999f * x[i + 1];
            }
        }
        sink[threadIdx.x] = __uint_as_float(acc) + x[0];
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    long long *out; float *sink;
    hipMalloc(&out, 8 * 8 * 256); hipMalloc(&sink, 4096);
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, sink);
            hipDeviceSynchronize();
        }
        long long h[8];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d (%s): mfma wave %lld cycles (%.1f per MFMA), valu wave %lld cycles (%.2f per VALU op)\n", mode,
               mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", h[0], (double)h[0] / (4.0 * iters), h[4], (double)h[4] / (52.0 * iters));
    }
    return 0;
}
