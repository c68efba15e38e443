// Where a step of the grouped forward LSTM sweep goes, under the full launch: per-phase cycle sums (s_memtime) of wave 0 of one
// workgroup of the last problem (runs every step). build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEGP_LSTM_TRACE=7 -o lstm_trace.bin tools/probes/lstm_trace.hip
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../egopose_amd/csrc/egp_internal.hpp"
namespace egp { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
#include "../../egopose_amd/csrc/egp_lstm.hip"
int main(int argc, char **argv) {
    const int T = 220, B = argc > 1 ? atoi(argv[1]) : 1280, H = 64, P = 4, train = argc > 2 ? atoi(argv[2]) : 1;
    float *gx, *w, *h, *cells;
    const size_t ng = (size_t)T * B * P * 4 * H;
    hipMalloc(&gx, ng * 4); hipMalloc(&w, (size_t)P * 4 * H * H * 4); hipMalloc(&h, (size_t)P * (T + 2) * B * H * 4); hipMalloc(&cells, (size_t)P * T * B * H * 4);
    std::vector<float> hv(ng);
    for (size_t i = 0; i < ng; ++i) hv[i] = (float)((i * 2654435761u) % 1000) / 2000.f - 0.25f;
    hipMemcpy(gx, hv.data(), ng * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hv.data(), (size_t)P * 4 * H * H * 4, hipMemcpyHostToDevice);
    hipMemset(h, 0, (size_t)P * (T + 2) * B * H * 4);
    float *hp[4];
    for (int p = 0; p < P; ++p) hp[p] = h + ((size_t)p * (T + 2) + 1) * B * H;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (egp_lstm_group_fwd_f32(gx, w, T, B, H, P, 0xC, hp, H, train ? gx : nullptr, train ? cells : nullptr, nullptr) != 0) return 1;
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us (%d workgroups)\n", rep, ms * 1e3, (B / 4) * P);
    }
    long long tr[8];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_lstm_trace), sizeof(tr));
    const char *names[5] = {"between steps (prefetch issue, copies, wait for the input tile)", "hidden-tile LDS reads issued", "products (wait for LDS + 64 MFMAs)",
                            "cell update + stores issued", "barrier"};
    long long tot = 0;
    for (int i = 0; i < 5; ++i) tot += tr[i];
    for (int i = 0; i < 5; ++i) printf("%-68s %7.0f cycles per step\n", names[i], (double)tr[i] / (double)tr[6]);
    printf("%-68s %7.0f cycles per step over %lld steps\n", "total", (double)tot / (double)tr[6], tr[6]);
    return 0;
}
