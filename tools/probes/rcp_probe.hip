// Accuracy of v_rcp_f64 with 0 / 1 / 2 Newton steps and with one third-order step against the correctly rounded quotient (gfx950).
// build: hipcc --offload-arch=gfx950 -O2 -o rcp_probe.bin tools/probes/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double e = fma(-v, r0, 1.0);
    double r1 = fma(r0, e, r0);
    e = fma(-v, r1, 1.0);
    double r2 = fma(r1, e, r1);
    e = fma(-v, r0, 1.0);
    const double r3 = fma(r0, fma(e, e, e), r0);
    o[4 * i] = r0; o[4 * i + 1] = r1; o[4 * i + 2] = r2; o[4 * i + 3] = r3;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), out(4 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(out.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    double worst[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double ex = 1.0L / (long double)h[i];
        for (int c = 0; c < 4; ++c) {
            const double ulp = ldexp(1.0, ilogb((double)ex) - 52);
            const double err = (double)fabsl((long double)out[4 * i + c] - ex) / ulp;
            if (err > worst[c]) worst[c] = err;
        }
    }
    printf("max error in ulp: v_rcp_f64 %.3g, +1 Newton %.3g, +2 Newton %.3g, one third-order step %.3g\n", worst[0], worst[1], worst[2], worst[3]);
    return 0;
}
