// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on? Every workgroup records its XCC and HW_ID.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_where(unsigned *out, long long spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID, 4 bits
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

static void run(const char *name, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int nb = 4096;
    unsigned *d, *h = (unsigned *)malloc(nb * 8);
    CK(hipMalloc(&d, nb * 8));
    k_where<<<nb, 64, 0, s>>>(d, 2000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, d, nb * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < nb; ++i) {
        const unsigned hw = h[2 * i], cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[h[2 * i + 1] & 0xf].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%-28s:", name);
    for (auto &kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  -> %d CUs\n", total);
    if (mask.size() <= 2) {
        for (auto &kv : per_xcc) { printf("    xcc%u:", kv.first); for (unsigned c : kv.second) printf(" se%u.cu%u", c >> 8, c & 0xf); printf("\n"); }
    }
    CK(hipFree(d)); free(h);
    CK(hipStreamDestroy(s));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> all(words, 0xffffffffu);
    run("all ones", all);
    std::vector<uint32_t> m(words, 0);
    m[0] = 0xff; run("bits 0-7", m);
    m[0] = 0xffff; run("bits 0-15", m);
    m[0] = 0xffffffffu; run("bits 0-31", m);
    m[1] = 0xffffffffu; run("bits 0-63", m);
    std::vector<uint32_t> hi(words, 0xffffffffu);
    hi[0] = 0; hi[1] = 0; run("all but bits 0-63", hi);
    std::vector<uint32_t> ev(words, 0x55555555u); run("even bits", ev);
    std::vector<uint32_t> q(words, 0x11111111u); run("every 4th bit", q);
    return 0;
}
