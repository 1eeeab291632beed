// Micro-benchmark 3: the memory skeleton of the 4:2:0 strip walk without any arithmetic — which structural element costs
// bandwidth?  256 images x 3 strips x 3 segments = 2304 workgroups of 256 threads, 33 KB of LDS each (4 per CU); per step a
// workgroup reads 32 KB of "coefficients" (contiguous runs, dwordx4 per lane) and writes 16 runs of 1920 B at pitch 5760.
//   V0  as the kernel: loads -> wait -> LDS -> barrier -> LDS -> stores (dwordx3 pairs, lanes 24 B apart) -> barrier
//   V1  V0 with the next step's loads issued before this step's stores (registers)
//   V2  V0 with dwordx4 stores, lanes 16 B apart
//   V3  V1 + V2
//   V4  no LDS, no barriers: every lane loads 128 B and stores 120 B per step, next loads issued before the stores
//   V5  V3 with 8 workgroups per CU (16.5 KB of LDS claimed)
// Reference on the same device: elementwise add over the same bytes 0.51 ms (6.2 TB/s); the kernel without arithmetic 0.636 ms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v3u v3u_a4 __attribute__((aligned(4)));
#define GP __attribute__((address_space(1)))
constexpr uint32_t PITCH = 5760u, ROWS = 1088u, IMGS = 256u, STEPS_PER_IMG = 68u;
constexpr size_t COEF_PER_IMG = (size_t)STEPS_PER_IMG * 3u * 32768u;  // 32 KB per (strip, step)

template <bool X4>
__device__ __forceinline__ void store_step(GP uint8_t *r0, const uint8_t *lds, uint32_t tid) {
    if (!X4) {
        for (uint32_t u = tid; u < 1280u; u += 256u) {  // 16 rows x 80 chunks of 24 B
            const uint32_t row = u / 80u, chk = u % 80u;
            const v4u a = *(const v4u *)(lds + (u * 16u & 32767u));
            GP uint8_t *o = r0 + row * PITCH + chk * 24u;
            *(GP v3u_a4 *)o = v3u{a.x, a.y, a.z};
            *(GP v3u_a4 *)(o + 12) = v3u{a.w, a.x, a.y};
        }
    } else {
        for (uint32_t u = tid; u < 1920u; u += 256u) {  // 16 rows x 120 pieces of 16 B
            const uint32_t row = u / 120u, pc = u % 120u;
            *(GP v4u *)(r0 + row * PITCH + pc * 16u) = *(const v4u *)(lds + (u * 16u & 32767u));
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void k(const uint8_t *coef, uint8_t *out, uint32_t nseg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t strip = blockIdx.x % 3u, seg = (blockIdx.x / 3u) % nseg, img = blockIdx.x / (3u * nseg), steps = STEPS_PER_IMG / nseg;
    GP uint8_t *obase = (GP uint8_t *)out + ((size_t)img * ROWS + (size_t)seg * steps * 16u) * PITCH + strip * 1920u;
    const GP uint8_t *cbase = (const GP uint8_t *)coef + (size_t)img * COEF_PER_IMG + ((size_t)(seg * steps) * 3u + strip) * 32768u;
    constexpr bool PRE = V == 1 || V == 3 || V == 5, X4 = V == 2 || V == 3 || V == 5;
    if (V == 4) {
        v4u a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = *(const GP v4u *)(cbase + (size_t)(tid + 256u * i) * 16u);
        for (uint32_t s = 0; s < steps; s++) {
            v4u b[8];
            const GP uint8_t *cn = cbase + (size_t)((s + 1u < steps ? s + 1u : s) * 3u) * 32768u;
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = *(const GP v4u *)(cn + (size_t)(tid + 256u * i) * 16u);
            GP uint8_t *r0 = obase + (size_t)s * 16u * PITCH;
#pragma unroll
            for (int i = 0; i < 7; i++) {  // 7 x 4 KB + 3 KB = 30 KB: piece u = tid + 256 i of 1920
                const uint32_t u = tid + 256u * i, row = u / 120u, pc = u % 120u;
                *(GP v4u *)(r0 + row * PITCH + pc * 16u) = a[i];
            }
            if (tid < 128u) {
                const uint32_t u = tid + 1792u, row = u / 120u, pc = u % 120u;
                *(GP v4u *)(r0 + row * PITCH + pc * 16u) = a[7];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = b[i];
        }
        return;
    }
    v4u a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = *(const GP v4u *)(cbase + (size_t)(tid + 256u * i) * 16u);
    for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 8; i++) *(v4u *)(lds + (tid + 256u * i) * 16u) = a[i];
        __syncthreads();
        const bool more = s + 1u < steps;
        const GP uint8_t *cn = cbase + (size_t)((more ? s + 1u : s) * 3u) * 32768u;
        if (PRE) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = *(const GP v4u *)(cn + (size_t)(tid + 256u * i) * 16u);
        }
        store_step<X4>(obase + (size_t)s * 16u * PITCH, lds, tid);
        __syncthreads();
        if (!PRE) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = *(const GP v4u *)(cn + (size_t)(tid + 256u * i) * 16u);
        }
    }
}
// V6: a workgroup owns full rows: per step it handles the three strips one after the other (grid = images x nseg)
template <bool X4>
__global__ __launch_bounds__(256) void k6(const uint8_t *coef, uint8_t *out, uint32_t nseg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t seg = blockIdx.x % nseg, img = blockIdx.x / nseg, steps = STEPS_PER_IMG / nseg;
    GP uint8_t *obase = (GP uint8_t *)out + ((size_t)img * ROWS + (size_t)seg * steps * 16u) * PITCH;
    const GP uint8_t *cbase = (const GP uint8_t *)coef + (size_t)img * COEF_PER_IMG + (size_t)(seg * steps) * 3u * 32768u;
    for (uint32_t s = 0; s < steps * 3u; s++) {  // (step, strip) pairs: coefficient runs are stored in this order
        v4u a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = *(const GP v4u *)(cbase + (size_t)s * 32768u + (size_t)(tid + 256u * i) * 16u);
#pragma unroll
        for (int i = 0; i < 8; i++) *(v4u *)(lds + (tid + 256u * i) * 16u) = a[i];
        __syncthreads();
        store_step<X4>(obase + (size_t)(s / 3u) * 16u * PITCH + (s % 3u) * 1920u, lds, tid);
        __syncthreads();
    }
}
template <bool X4>
void run6(const char *name, const uint8_t *c, uint8_t *d, size_t shm, uint32_t nseg) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t grid = IMGS * nseg;
    k6<X4><<<grid, 256, shm>>>(c, d, nseg);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        k6<X4><<<grid, 256, shm>>>(c, d, nseg);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)IMGS * (STEPS_PER_IMG / nseg * nseg) * 3 * (32768.0 + 16 * 1920.0);
    printf("%-62s %7.3f ms  %7.1f GB/s (read + write)\n", name, best, bytes / best / 1e6);
}
template <int V>
void run(const char *name, const uint8_t *c, uint8_t *d, size_t shm, uint32_t nseg = 3) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t grid = IMGS * 3 * nseg;
    k<V><<<grid, 256, shm>>>(c, d, nseg);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        k<V><<<grid, 256, shm>>>(c, d, nseg);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)IMGS * (STEPS_PER_IMG / nseg * nseg) * 3 * (32768.0 + 16 * 1920.0);
    printf("%-62s %7.3f ms  %7.1f GB/s (read + write)\n", name, best, bytes / best / 1e6);
}
int main() {
    uint8_t *c, *d;
    (void)hipMalloc(&c, IMGS * COEF_PER_IMG + 65536);
    (void)hipMalloc(&d, (size_t)IMGS * PITCH * ROWS + 4096);
    (void)hipMemset(c, 1, IMGS * COEF_PER_IMG + 65536);
    run<0>("V0 as the kernel (dwordx3 pairs, loads after the stores)", c, d, 33792);
    run<1>("V1 next loads before the stores", c, d, 33792);
    run<2>("V2 dwordx4 stores", c, d, 33792);
    run<3>("V3 next loads before the stores + dwordx4 stores", c, d, 33792);
    run<4>("V4 no LDS, no barriers, registers only", c, d, 0);
    run<5>("V5 = V3 with room for 8 workgroups per CU (LDS 16.5 KB claimed)", c, d, 32768);
    run<5>("V5' = V3 claiming 40 KB (3 workgroups per CU... 4 fit 160 KB)", c, d, 40960);
    run<3>("V3'' claiming 53 KB (3 workgroups per CU)", c, d, 54272);
    // how the work is dealt: segments of 68/nseg steps; consecutive workgroups = the three strips of a segment, then the next segment
    for (uint32_t ns : {1u, 2u, 4u, 17u, 34u, 68u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "V4 registers only, %u segments per strip (%u workgroups)", ns, IMGS * 3 * ns);
        run<4>(nm, c, d, 0, ns);
    }
    for (uint32_t ns : {3u, 68u})
        for (size_t shm : {(size_t)0, (size_t)20480, (size_t)33792, (size_t)54272}) {  // occupancy of the register-only variant: 8 / 7 / 4 / 2 workgroups per CU
            char nm[96];
            snprintf(nm, sizeof nm, "V4 registers only, %u segments, claiming %zu B of LDS", ns, shm);
            run<4>(nm, c, d, shm, ns);
        }
    for (uint32_t ns : {4u, 17u, 68u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "V6 full rows per workgroup (3 strips in turn), %u segments (%u wgs)", ns, IMGS * ns);
        run6<false>(nm, c, d, 33792, ns);
    }
    for (uint32_t ns : {1u, 4u, 17u, 68u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "V0 as the kernel, %u segments per strip (%u workgroups)", ns, IMGS * 3 * ns);
        run<0>(nm, c, d, 33792, ns);
    }
    return 0;
}
