// probe_d2h2.hip — round 5, second look at the pipeline's downloads (33 GB/s inside jpgpu_pipeline_decode, 57 GB/s for the same copies alone:
// probe_d2h.hip): the copies as the pipeline issues them — each behind an event recorded on a compute stream that has just run kernels —
// with host-to-device traffic beside them, and the alternative: a copy KERNEL that writes pinned host memory itself.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_d2h2.hip -o /tmp/probe_d2h2 && /tmp/probe_d2h2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void busy_kernel(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 3u + 1u;
}
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_to_host_kernel(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) __builtin_nontemporal_store(src[i], dst + i);
}
int main(int argc, char **argv) {
    const size_t chunk = 759u << 20;
    const int n_chunks = argc > 1 ? atoi(argv[1]) : 16;
    char *dev = nullptr, *dev_up = nullptr, *h_up = nullptr;
    CK(hipMalloc((void **)&dev, chunk * 2));
    CK(hipMemset(dev, 1, chunk * 2));
    CK(hipMalloc((void **)&dev_up, 64u << 20));
    CK(hipHostMalloc((void **)&h_up, 64u << 20, hipHostMallocDefault));
    hipStream_t d2h[2], comp[4], up;
    for (auto &s : d2h) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : comp) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(n_chunks);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<char *> h(n_chunks, nullptr);
    for (auto &p : h) CK(hipHostMalloc((void **)&p, chunk, hipHostMallocDefault));
    for (int variant = 0; variant < 6; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int c = 0; c < n_chunks; c++) {
                hipStream_t ds = d2h[c % 2];
                if (variant >= 1 && variant <= 3) {  // a kernel on a compute stream, its event, the copy behind the event
                    busy_kernel<<<dim3(4096), dim3(256), 0, comp[c % 4]>>>((uint32_t *)dev, 1u << 20);
                    CK(hipEventRecord(ev[c], comp[c % 4]));
                    CK(hipStreamWaitEvent(ds, ev[c], 0));
                }
                if (variant == 2 || variant == 3) CK(hipMemcpyAsync(dev_up, h_up, 48u << 20, hipMemcpyHostToDevice, up));  // uploads beside it
                if (variant == 3) ds = d2h[0];  // one download stream
                if (variant <= 3) CK(hipMemcpyAsync(h[c], dev + (c & 1) * chunk, chunk, hipMemcpyDeviceToHost, ds));
                else {  // the copy kernel: 64 (variant 4) or 256 (variant 5) workgroups
                    void *mapped = nullptr;
                    CK(hipHostGetDevicePointer(&mapped, h[c], 0));
                    copy_to_host_kernel<<<dim3(variant == 4 ? 64 : 256), dim3(256), 0, ds>>>((v4u *)mapped, (const v4u *)(dev + (c & 1) * chunk), chunk / 16);
                }
            }
            CK(hipStreamSynchronize(d2h[0]));
            CK(hipStreamSynchronize(d2h[1]));
            const double dt = now() - t0;
            CK(hipDeviceSynchronize());
            static const char *names[] = {"plain copies, two streams", "each behind a compute stream's event", "... with uploads beside them", "... on ONE download stream", "copy kernel, 64 workgroups", "copy kernel, 256 workgroups"};
            printf("%-40s pass %d: %6.2f GB/s (%d x %zu MB in %.1f ms)\n", names[variant], rep, n_chunks * (double)chunk / dt / 1e9, n_chunks, chunk >> 20, dt * 1e3);
        }
    }
    return 0;
}
