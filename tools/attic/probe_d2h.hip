// probe_d2h.hip — what limits device-to-host copies of the pipeline's size (round 5: to_host_4096 reached 33 GB/s where one pinned 1-GB
// copy of torch's reached 57)?  Copies of 800 MB (a sub-batch of 128 1080p images) out of HBM into pinned host memory: one stream or two
// side by side, default / non-coherent / NUMA-user host allocations, each buffer written once (fresh pages) or reused.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_d2h.hip -o /tmp/probe_d2h && /tmp/probe_d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const size_t chunk = 759u << 20;
    const int n_chunks = argc > 1 ? atoi(argv[1]) : 8;  // (32: the footprint of a 4,096-file call that downloads its pixels, 24 GB)
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, chunk * 2));
    CK(hipMemset(dev, 1, chunk * 2));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    struct Kind { const char *name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                                 {"hipHostMallocNumaUser", hipHostMallocNumaUser}, {"hipHostMallocPortable", hipHostMallocPortable}};
    const int n_kinds = argc > 2 ? atoi(argv[2]) : 4;
    for (int ki = 0; ki < n_kinds && ki < 4; ki++) {
        const Kind &k = kinds[ki];
        std::vector<char *> h(n_chunks, nullptr);
        bool ok = true;
        const double a0 = now();
        for (auto &p : h) ok = ok && hipHostMalloc((void **)&p, chunk, k.flags) == hipSuccess;
        const double alloc_s = now() - a0;
        if (!ok) { printf("%-28s allocation failed\n", k.name); (void)hipGetLastError(); continue; }
        for (int streams = 1; streams <= 2; streams++)
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int c = 0; c < n_chunks; c++) CK(hipMemcpyAsync(h[c], dev + (c & 1) * chunk, chunk, hipMemcpyDeviceToHost, s[c % streams]));
                CK(hipStreamSynchronize(s[0]));
                CK(hipStreamSynchronize(s[1]));
                const double dt = now() - t0;
                printf("%-28s %d stream(s) pass %d: %6.2f GB/s (%d x %zu MB in %.1f ms; allocation %.2f s)\n", k.name, streams, rep, n_chunks * (double)chunk / dt / 1e9, n_chunks, chunk >> 20, dt * 1e3, alloc_s);
            }
        // split copies: 8 MB pieces on one stream
        {
            CK(hipDeviceSynchronize());
            const size_t piece = 8u << 20;
            const double t0 = now();
            for (int c = 0; c < n_chunks; c++)
                for (size_t o = 0; o < chunk; o += piece) CK(hipMemcpyAsync(h[c] + o, dev + (c & 1) * chunk + o, std::min(piece, chunk - o), hipMemcpyDeviceToHost, s[0]));
            CK(hipStreamSynchronize(s[0]));
            const double dt = now() - t0;
            printf("%-28s 1 stream, 8 MB pieces: %6.2f GB/s\n", k.name, n_chunks * (double)chunk / dt / 1e9);
        }
        for (auto p : h) (void)hipHostFree(p);
    }
    return 0;
}
