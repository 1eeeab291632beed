// probe_pinned.hip — how fast does the CPU write / read the kinds of host memory the runtime offers for DMA?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pinned.hip -o /tmp/probe_pinned && /tmp/probe_pinned
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void run(const char *name, char *dst, const char *src, size_t n) {
    memcpy(dst, src, n);
    double t0 = now();
    for (int r = 0; r < 4; r++) memcpy(dst, src, n);
    double w = 4.0 * n / (now() - t0) / 1e9;
    char *tmp = (char *)malloc(n);
    memcpy(tmp, dst, n);
    t0 = now();
    for (int r = 0; r < 4; r++) memcpy(tmp, dst, n);
    double rd = 4.0 * n / (now() - t0) / 1e9;
    // small pieces, as a row sink writes them
    t0 = now();
    for (int r = 0; r < 4; r++)
        for (size_t o = 0; o + 36864 <= n; o += 36864) memcpy(dst + o, src + o, 36864);
    double ws = 4.0 * (n / 36864 * 36864) / (now() - t0) / 1e9;
    printf("%-44s CPU write %6.2f GB/s (36 kB pieces %6.2f), CPU read %6.2f GB/s\n", name, w, ws, rd);
    free(tmp);
}
int main() {
    const size_t n = 64u << 20;
    char *src = (char *)malloc(n);
    memset(src, 1, n);
    char *a = (char *)malloc(n);
    run("malloc", a, src, n);
    char *b = nullptr;
    if (hipHostMalloc((void **)&b, n, hipHostMallocDefault) == hipSuccess) run("hipHostMalloc(Default)", b, src, n);
    char *c = nullptr;
    if (hipHostMalloc((void **)&c, n, hipHostMallocNonCoherent) == hipSuccess) run("hipHostMalloc(NonCoherent)", c, src, n);
    char *d = nullptr;
    if (hipHostMalloc((void **)&d, n, hipHostMallocCoherent) == hipSuccess) run("hipHostMalloc(Coherent)", d, src, n);
    char *e = (char *)aligned_alloc(4096, n);
    memset(e, 0, n);
    if (hipHostRegister(e, n, hipHostRegisterDefault) == hipSuccess) run("malloc + hipHostRegister", e, src, n);
    // H2D speed from each
    char *dev = nullptr;
    hipMalloc((void **)&dev, n);
    for (auto p : {std::make_pair("malloc (pageable)", a), std::make_pair("hipHostMalloc(Default)", b), std::make_pair("hipHostMalloc(NonCoherent)", c), std::make_pair("registered", e)}) {
        if (!p.second) continue;
        hipMemcpy(dev, p.second, n, hipMemcpyHostToDevice);
        double t0 = now();
        for (int r = 0; r < 4; r++) hipMemcpy(dev, p.second, n, hipMemcpyHostToDevice);
        printf("H2D from %-32s %6.2f GB/s\n", p.first, 4.0 * n / (now() - t0) / 1e9);
    }
    return 0;
}
