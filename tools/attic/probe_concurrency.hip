// Do kernels of different streams run side by side on this device / runtime?  A kernel that keeps `blocks` workgroups of 256 lanes busy
// for ~1 ms with dependent arithmetic (no memory), launched once on each of S streams: all S concurrent -> ~1 ms, serial -> ~S ms.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_concurrency.bin tools/probe_concurrency.hip && GPU_MAX_HW_QUEUES=24 /tmp/probe_concurrency.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ __launch_bounds__(256) void spin(uint32_t *out, uint32_t iters) {
    uint32_t x = threadIdx.x + 1u;
    for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) out[0] = x;
}
// the shape of a sync pass: 22.5 kB of LDS per workgroup, a chain of dependent LDS reads (latency-bound: a wave is parked most of the time)
__global__ __launch_bounds__(256) void chase(uint32_t *out, uint32_t iters) {
    __shared__ uint32_t t[5760];
    for (uint32_t i = threadIdx.x; i < 5760u; i += 256u) t[i] = (i * 2654435761u) % 5760u;
    __syncthreads();
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) x = t[x] ^ (i & 1u);
    if (x == 0xdeadbeefu) out[0] = x;
}
template <bool CHASE>
static double run(int n_streams, int blocks, uint32_t iters, uint32_t *d) {
    std::vector<hipStream_t> s(n_streams);
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    for (auto &x : s) { if (CHASE) chase<<<blocks, 256, 0, x>>>(d, 1000); else spin<<<blocks, 256, 0, x>>>(d, 1000); }  // warm
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s[0]);
        for (auto &x : s) { if (CHASE) chase<<<blocks, 256, 0, x>>>(d, iters); else spin<<<blocks, 256, 0, x>>>(d, iters); }
        for (int i = 1; i < n_streams; i++) {  // s[0] waits for the others
            hipEvent_t e;
            hipEventCreateWithFlags(&e, hipEventDisableTiming);
            hipEventRecord(e, s[i]);
            hipStreamWaitEvent(s[0], e, 0);
            hipEventDestroy(e);
        }
        hipEventRecord(e1, s[0]);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    for (auto &x : s) hipStreamDestroy(x);
    return best;
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 4096);
    printf("GPU_MAX_HW_QUEUES=%s\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    for (int blocks : {256, 1024})
        for (int n : {1, 2, 4, 8}) printf("arithmetic chain  blocks %5d  streams %2d  %.3f ms\n", blocks, n, run<false>(n, blocks, 60000, d));
    for (int blocks : {640, 1280})  // 640 workgroups = the sync pass of a 128-file sub-batch
        for (int n : {1, 2, 3, 4, 6, 8, 12}) printf("LDS chase (22.5 kB) blocks %5d  streams %2d  %.3f ms\n", blocks, n, run<true>(n, blocks, 8000, d));
    return 0;
}
