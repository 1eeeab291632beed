// Micro-benchmark: how fast do the pixel-store patterns of the 4:2:0 kernels go on their own?
//   A: two global_store_dwordx3 per lane, lanes 24 B apart (what a lane holding an 8-pixel chunk does)
//   B: global_store_dwordx4, lanes 16 B apart (fully contiguous kilobytes per instruction)
//   C: three dwordx4 per lane, lanes 48 B apart (a lane holding 16 pixels)
// Each workgroup writes runs of 1920 B (a 40-MCU strip) on 16 consecutive rows of pitch 5760 B, like the strip kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v3u v3u_a4 __attribute__((aligned(4)));
#define GP __attribute__((address_space(1)))

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k(uint8_t *out, uint32_t steps) {
    const uint32_t strip = blockIdx.x % 3u, img = blockIdx.x / 3u, tid = threadIdx.x;
    GP uint8_t *base = (GP uint8_t *)out + (size_t)img * (5760u * 1088u) + strip * 1920u;
    for (uint32_t s = 0; s < steps; s++) {
        GP uint8_t *rows = base + (size_t)s * 16u * 5760u;
        const uint32_t v = s * 2654435761u + tid;
        if (PAT == 0) {  // 16 rows x 80 chunks = 1280 (row, chunk) -> 5 rounds of 256 lanes
            for (uint32_t u = tid; u < 1280u; u += 256u) {
                const uint32_t row = u / 80u, chk = u % 80u;
                GP uint8_t *o = rows + row * 5760u + chk * 24u;
                const v3u d = {v, v + 1, v + 2};
                if (NT) { __builtin_nontemporal_store(d, (GP v3u_a4 *)o); __builtin_nontemporal_store(d, (GP v3u_a4 *)(o + 12)); }
                else { *(GP v3u_a4 *)o = d; *(GP v3u_a4 *)(o + 12) = d; }
            }
        } else if (PAT == 1) {  // 16 rows x 120 pieces of 16 B = 1920 -> 7.5 rounds
            for (uint32_t u = tid; u < 1920u; u += 256u) {
                const uint32_t row = u / 120u, pc = u % 120u;
                GP uint8_t *o = rows + row * 5760u + pc * 16u;
                const v4u d = {v, v + 1, v + 2, v + 3};
                if (NT) __builtin_nontemporal_store(d, (GP v4u *)o); else *(GP v4u *)o = d;
            }
        } else {  // 16 rows x 40 pieces of 48 B = 640 -> 2.5 rounds
            for (uint32_t u = tid; u < 640u; u += 256u) {
                const uint32_t row = u / 40u, pc = u % 40u;
                GP uint8_t *o = rows + row * 5760u + pc * 48u;
                const v4u d = {v, v + 1, v + 2, v + 3};
                if (NT) { __builtin_nontemporal_store(d, (GP v4u *)o); __builtin_nontemporal_store(d, (GP v4u *)(o + 16)); __builtin_nontemporal_store(d, (GP v4u *)(o + 32)); }
                else { *(GP v4u *)o = d; *(GP v4u *)(o + 16) = d; *(GP v4u *)(o + 32) = d; }
            }
        }
    }
}
template <int PAT, bool NT>
void run(const char *name, uint8_t *d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t imgs = 256, steps = 68;
    k<PAT, NT><<<imgs * 3, 256>>>(d, steps);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        k<PAT, NT><<<imgs * 3, 256>>>(d, steps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)imgs * 3 * steps * 16 * 1920;
    printf("%-40s %7.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
}
int main() {
    uint8_t *d;
    (void)hipMalloc(&d, (size_t)256 * 5760 * 1088 + 4096);
    run<0, true>("A dwordx3 x2, 24 B apart, nontemporal", d);
    run<0, false>("A dwordx3 x2, 24 B apart", d);
    run<1, true>("B dwordx4 contiguous, nontemporal", d);
    run<1, false>("B dwordx4 contiguous", d);
    run<2, true>("C dwordx4 x3, 48 B apart, nontemporal", d);
    run<2, false>("C dwordx4 x3, 48 B apart", d);
    return 0;
}
