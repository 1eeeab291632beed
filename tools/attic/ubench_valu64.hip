// Micro-benchmark: issue cost of the 64-bit integer operations the device bit reader is (was) made of, next to their 32-bit
// replacements — 16 independent chains per lane, 8 waves per SIMD, so the figure is the issue rate, not the latency.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu64.bin tools/ubench_valu64.hip && ./ubench_valu64.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define N_ACC 16
#define ITERS 2048

#define OPS(X)                                                              \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %2")                               \
    X(1, "v_lshlrev_b64 (variable)", "v_lshlrev_b64 %1, %2, %1")            \
    X(2, "v_lshrrev_b64 (variable)", "v_lshrrev_b64 %1, %2, %1")            \
    X(3, "v_lshlrev_b64 (constant 8)", "v_lshlrev_b64 %1, 8, %1")           \
    X(4, "v_lshl_add_u64", "v_lshl_add_u64 %1, %1, 2, %1")                  \
    X(5, "v_mad_u64_u32", "v_mad_u64_u32 %1, vcc, %0, %2, %1")              \
    X(6, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %2, %3")                 \
    X(7, "v_lshlrev_b32", "v_lshlrev_b32 %0, %2, %0")                       \
    X(8, "v_bfe_u32", "v_bfe_u32 %0, %0, %2, %3")                           \
    X(9, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3")                         \
    X(10, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %2, vcc")                 \
    X(11, "v_mov_b64", "v_mov_b64 %1, %1")                                  \
    X(12, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x30")        \
    X(13, "v_cmp_lt_u32 -> sgpr pair", "v_cmp_lt_u32 s[10:11], %0, %2")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[N_ACC];
    uint64_t w[N_ACC];
    uint32_t x = threadIdx.x * 2654435761u + seed, y = (x ^ 0x9e3779b9u) & 31u, z = 5u;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) a[i] = x + i, w[i] = ((uint64_t)x << 32) | (uint32_t)(i * 77 + 1);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
#define X(ID, NAME, ASM) \
    if (OP == ID) asm volatile(ASM : "+v"(a[i]), "+v"(w[i]) : "v"(y), "v"(z) : "vcc", "s10", "s11");
            OPS(X)
#undef X
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, uint32_t *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 x 256 threads per CU -> 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(d, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double insts_per_simd = (double)blocks * 4 /*waves*/ * ITERS * N_ACC / (256.0 * 4);
    printf("%-28s %8.3f ms  %6.3f ns/wave-instr/SIMD  (= %.2f cyc @2.4GHz)\n", name, best, best * 1e6 / insts_per_simd, best * 1e6 / insts_per_simd * 2.4);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(ID, NAME, ASM) run<ID>(NAME, d);
    OPS(X)
#undef X
    return 0;
}
