// Cycle stamps (s_memtime) of one wave of one workgroup of egp_gemm_f32 at an update shape: where a k-tile's time goes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEGP_GEMM_TRACE=1500 -o gemm_trace.bin tools/probes/gemm_trace.hip
// run:   ./gemm_trace.bin [terms] [M] [K] [N]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../egopose_amd/csrc/egp_internal.hpp"
namespace egp { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
#include "../../egopose_amd/csrc/egp_gemm.hip"

int main(int argc, char **argv) {
    const int terms = argc > 1 ? atoi(argv[1]) : 6, M = argc > 2 ? atoi(argv[2]) : 139264, K = argc > 3 ? atoi(argv[3]) : 243, N = argc > 4 ? atoi(argv[4]) : 300;
    float *x, *w, *b, *y;
    hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&y, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), N * 4, hipMemcpyHostToDevice);
    egp_gemm_desc d{};
    d.M = M; d.N = N; d.K = K; d.A = x; d.lda = K; d.a_kcontig = 1; d.B = w; d.ldb = K; d.b_kcontig = 1; d.C = y; d.ldc = N;
    d.bias = b; d.relu = 1; d.terms = terms; d.splits = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        int zero = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace_n), &zero, sizeof(int));
        int zero2[2] = {0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace_n2), zero2, sizeof(zero2));
        hipEventRecord(e0);
        if (egp_gemm_f32(&d, nullptr) != 0) return 1;
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us\n", rep, ms * 1e3);
    }
    int n; std::vector<long long> tr(1024);
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_gemm_trace_n), sizeof(int));
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_gemm_trace), 1024 * 8);
    int n2[2];
    hipMemcpyFromSymbol(n2, HIP_SYMBOL(g_gemm_trace_n2), 2 * sizeof(int));
    for (int who = 0; who < 2; ++who) {
        const long long *q = tr.data() + who * 512;
        const int lim = n2[who] < 40 ? n2[who] : 40;
        for (int i = 0; i < lim; ++i)
            printf("%s tag %2lld  +%6lld cycles (since first %7lld)\n", who ? "producer" : "consumer", q[2 * i], i ? q[2 * i + 1] - q[2 * i - 1] : 0, q[2 * i + 1] - q[1]);
    }
    for (int i = 0; i < n; ++i) printf("tag %2lld  +%6lld cycles (since first %7lld)\n", tr[2 * i], i ? tr[2 * i + 1] - tr[2 * i - 1] : 0, tr[2 * i + 1] - tr[1]);
    return 0;
}
