// Micro-benchmark 2: what shapes the write rate?  All variants write 256 images x 1088 rows x 5760 B (1.6 GB) with
// global_store_dwordx4, 256-thread workgroups; they differ in which bytes a workgroup writes per step and in how many
// workgroups there are.  Reference points on the same device: torch fill_ 7.0 TB/s, elementwise add (R+W) 6.2 TB/s.
//   L  linear: workgroup w writes one contiguous piece of the arena
//   R  rows:   workgroup = (image, segment of rows): 16 full rows (5760 B each) per step
//   S  strips: workgroup = (image, strip of 1920 B, segment): 16 runs of 1920 B at pitch 5760 per step   (the 4:2:0 walk)
//   S' strips, one row per instruction round: a wave writes whole rows in turn instead of (row, piece) units dealt over lanes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define GP __attribute__((address_space(1)))
constexpr uint32_t PITCH = 5760u, ROWS = 1088u, IMGS = 256u;

template <int PAT>
__global__ __launch_bounds__(256) void k(uint8_t *out, uint32_t nseg) {
    const uint32_t tid = threadIdx.x;
    const v4u d = {tid, blockIdx.x, 3u, 4u};
    if (PAT == 0) {  // linear: gridDim.x pieces
        const size_t total = (size_t)IMGS * ROWS * PITCH, per = total / gridDim.x;
        GP uint8_t *o = (GP uint8_t *)out + (size_t)blockIdx.x * per;
        for (size_t off = (size_t)tid * 16u; off < per; off += 4096u) *(GP v4u *)(o + off) = d;
    } else if (PAT == 1) {  // rows: grid = IMGS * nseg
        const uint32_t img = blockIdx.x / nseg, seg = blockIdx.x % nseg, rows = ROWS / nseg;
        GP uint8_t *o = (GP uint8_t *)out + ((size_t)img * ROWS + (size_t)seg * rows) * PITCH;
        const size_t per = (size_t)rows * PITCH;
        for (size_t off = (size_t)tid * 16u; off < per; off += 4096u) *(GP v4u *)(o + off) = d;
    } else {  // strips: grid = IMGS * 3 * nseg
        const uint32_t strip = blockIdx.x % 3u, seg = (blockIdx.x / 3u) % nseg, img = blockIdx.x / (3u * nseg), rows = ROWS / nseg;
        GP uint8_t *base = (GP uint8_t *)out + ((size_t)img * ROWS + (size_t)seg * rows) * PITCH + strip * 1920u;
        for (uint32_t s = 0; s < rows / 16u; s++) {
            GP uint8_t *r0 = base + (size_t)s * 16u * PITCH;
            if (PAT == 2) {
                for (uint32_t u = tid; u < 1920u; u += 256u) {
                    const uint32_t row = u / 120u, pc = u % 120u;
                    *(GP v4u *)(r0 + row * PITCH + pc * 16u) = d;
                }
            } else {  // wave w: rows w, w+4, ...; its 64 lanes write 1024 B, then 896 B of the row
                const uint32_t wv = tid >> 6, ln = tid & 63u;
                for (uint32_t row = wv; row < 16u; row += 4u) {
                    GP uint8_t *o = r0 + row * PITCH;
                    *(GP v4u *)(o + ln * 16u) = d;
                    if (ln < 56u) *(GP v4u *)(o + 1024u + ln * 16u) = d;
                }
            }
        }
    }
}
template <int PAT>
void run(const char *name, uint8_t *d, uint32_t grid, uint32_t nseg) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PAT><<<grid, 256>>>(d, nseg);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        k<PAT><<<grid, 256>>>(d, nseg);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)IMGS * ROWS * PITCH;
    printf("%-46s grid %6u  %7.3f ms  %7.1f GB/s\n", name, grid, best, bytes / best / 1e6);
}
int main() {
    uint8_t *d;
    (void)hipMalloc(&d, (size_t)IMGS * PITCH * ROWS + 4096);
    for (uint32_t g : {1024u, 2048u, 4096u, 16384u}) run<0>("L linear", d, g, 1);
    for (uint32_t ns : {1u, 4u, 17u, 68u}) run<1>("R full rows, 1088/nseg rows per workgroup", d, IMGS * ns, ns);
    for (uint32_t ns : {1u, 2u, 4u, 17u, 68u}) run<2>("S strips (row, piece) over lanes", d, IMGS * 3 * ns, ns);
    for (uint32_t ns : {1u, 4u, 17u, 68u}) run<3>("S' strips, a wave writes whole rows", d, IMGS * 3 * ns, ns);
    return 0;
}
