// Probe of the 64-bit DPP forms on gfx950 used by the grid-layout stable-PD kernel: semantics of
// v_fmac_f64_dpp row_newbcast (with row_mask) and its issue rate next to the plain v_fmac_f64.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp64_probe tools/probes/dpp64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_sem(double *o) {
    const int l = threadIdx.x;
    double x = 100.0 + l, y = 2.0, acc = 0.5, acc2 = 0.5, m = -1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xb bank_mask:0xf" : "+v"(acc2) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(x));
    o[l] = acc; o[64 + l] = acc2; o[128 + l] = m;
}

template <bool DPP>
__global__ void k_rate(double *o, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9;
    for (int i = 0; i < iters; ++i) {
        if (DPP)
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        else
            asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                         "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    double *d; CK(hipMalloc(&d, 1 << 24));
    k_sem<<<1, 64>>>(d);
    std::vector<double> h(192);
    CK(hipMemcpy(h.data(), d, 192 * 8, hipMemcpyDeviceToHost));
    for (int t = 0; t < 3; ++t) { printf("%s:", t == 0 ? "fmac bcast3" : t == 1 ? "fmac bcast5 rowmask b" : "mov bcast7"); for (int l = 0; l < 64; l += 1) printf(" %g", h[t * 64 + l]); printf("\n"); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int wpb : {1, 2, 4}) for (int dpp = 0; dpp < 2; ++dpp) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (dpp) k_rate<true><<<256 * 4, 64 * wpb>>>(d, iters); else k_rate<false><<<256 * 4, 64 * wpb>>>(d, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        // waves per SIMD = wpb (1024 blocks over 256 CUs x 4 SIMDs, wpb waves each -> wpb waves per SIMD)
        printf("%s waves/SIMD %d: %.3f ms -> %.2f ns per instruction per wave-slot\n", dpp ? "fmac_f64_dpp" : "fmac_f64    ", wpb, best, best * 1e6 / (iters * 8.0 * wpb));
    }
    return 0;
}
