// Micro-benchmark (round 6): what ONE wave pays per instruction of a DEPENDENT chain — the walk of a progressive scan is one
// (csrc/huff_prog_wave.hpp) — for the instruction kinds such a walk can be made of.  One wave on the device (and one per SIMD:
// same figures), 4,096 x 64 dependent operations each, wall time by events.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_scalar_chain.bin tools/ubench_scalar_chain.hip && ./ubench_scalar_chain.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

// every chain: `OPS` operations per repetition of the asm text
#define CHAINS(X)                                                                                                                       \
    X(0, "s_add_u32 (dependent)", 1, "s_add_u32 %0, %0, 1\n")                                                                            \
    X(1, "s_lshl_b64 (dependent)", 1, "s_lshl_b64 %1, %1, 1\n")                                                                          \
    X(2, "s_bfe_u32 + s_add (dependent pair)", 2, "s_bfe_u32 s10, %0, 0x100004\n s_add_u32 %0, %0, s10\n")                               \
    X(3, "s_bcnt1_i32_b64 + s_ff1 + s_add + s_lshl_b64 + s_add (dependent)", 5, "s_bcnt1_i32_b64 s10, %1\n s_ff1_i32_b64 s11, %1\n s_add_u32 s10, s10, s11\n s_lshl_b64 %1, %1, 0\n s_add_u32 %0, %0, s10\n") \
    X(4, "v_add_u32 (dependent)", 1, "v_add_u32 %2, %2, 1\n")                                                                            \
    X(5, "v_readlane -> s_and -> v_readlane (lane select depends)", 2, "v_readlane_b32 s10, %2, %0\n s_and_b32 %0, s10, 63\n")            \
    X(6, "v_readlane -> s_add -> v_add(sgpr) -> v_readlane", 3, "v_readlane_b32 s10, %2, 5\n s_add_u32 s10, s10, 1\n v_add_u32 %2, %2, s10\n") \
    X(7, "s_cmp + taken s_cbranch", 2, "s_cmp_lg_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n")                               \
    X(8, "v_cmp(vcc) -> s_ff1_b64 -> v_add(sgpr) -> v_cmp", 3, "v_cmp_lt_u32 vcc, %2, %3\n s_ff1_i32_b64 s10, vcc\n v_add_u32 %2, %2, s10\n") \
    X(9, "s_mov_b64 + v_mbcnt_lo/hi + v_cmp_eq + s_and_b64 + s_ff1 + s_lshl_b64 (rank select)", 7, "s_mov_b64 s[12:13], %1\n v_mbcnt_lo_u32_b32 %3, s12, 0\n v_mbcnt_hi_u32_b32 %3, s13, %3\n v_cmp_eq_u32 vcc, 3, %3\n s_and_b64 vcc, vcc, s[12:13]\n s_ff1_i32_b64 s10, vcc\n s_lshl_b64 %1, %1, s10\n") \
    X(10, "s_add x2 (two independent chains)", 2, "s_add_u32 %0, %0, 1\n s_add_u32 s10, s10, 1\n")                                        \
    X(11, "s_add + v_add (independent, alternating)", 2, "s_add_u32 %0, %0, 1\n v_add_u32 %2, %2, 1\n")                                    \
    X(12, "s_nop 0", 1, "s_nop 0\n")

template <int ID>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, const uint32_t *table) {
    uint32_t s = seed & 63u, v = threadIdx.x + seed, v2 = 1000000u + threadIdx.x;
    uint64_t w = 0x0f0f0f0f0f0f0f0full | seed;
    s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
    for (int it = 0; it < ITERS; it++) {
#define X(NUM, NAME, OPS, ASM) \
    if (ID == NUM) asm volatile(R64(ASM) : "+s"(s), "+s"(w), "+v"(v), "+v"(v2) : : "s10", "s11", "s12", "s13", "vcc", "scc");
        CHAINS(X)
#undef X
    }
    out[threadIdx.x] = s + v + v2 + (uint32_t)w;
}
// LDS look-up chain and scalar-load chain (the table reads of a Huffman walk): the next address depends on what was read
__global__ __launch_bounds__(64) void k_lds(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t t[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += 64u) t[i] = (i * 37u + 11u) & 255u;
    __syncthreads();
    uint32_t a = seed & 255u;
    for (int it = 0; it < ITERS * 16; it++) a = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[a]);
    out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void k_sload(uint32_t *out, uint32_t seed, const uint32_t *__restrict__ table) {
    uint32_t a = seed & 255u;
    const __attribute__((address_space(4))) uint32_t *t = (const __attribute__((address_space(4))) uint32_t *)table;
    for (int it = 0; it < ITERS * 16; it++) a = t[a];
    out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void k_readlane_table(uint32_t *out, uint32_t seed) {
    uint32_t t0 = (threadIdx.x * 37u + 11u) & 63u, a = seed & 63u;  // lane i holds the next index: a look-up in a register
    a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    for (int it = 0; it < ITERS * 16; it++) a = (uint32_t)__builtin_amdgcn_readlane((int)t0, (int)a);
    out[threadIdx.x] = a;
}

template <class F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t *d, *tab;
    hipMalloc(&d, 4096);
    hipMalloc(&tab, 1024);
    uint32_t h[256];
    for (uint32_t i = 0; i < 256; i++) h[i] = (i * 37u + 11u) & 255u;
    hipMemcpy(tab, h, 1024, hipMemcpyHostToDevice);
    const float empty = time_ms([&] { k<12><<<1, 64>>>(d, 3, tab); });  // s_nop: the loop's own cost is in there too
    printf("(clock: s_nop 0 x %d: %.3f ms -> %.2f cycles each at 2.4 GHz)\n", ITERS * 64, empty, empty * 1e-3 * 2.4e9 / (ITERS * 64.0));
#define X(NUM, NAME, OPS, ASM)                                                                                        \
    {                                                                                                                 \
        const float one = time_ms([&] { k<NUM><<<1, 64>>>(d, 3, tab); });                                              \
        const float many = time_ms([&] { k<NUM><<<1024, 64>>>(d, 3, tab); });                                          \
        const double n = (double)ITERS * 64.0 * OPS;                                                                    \
        printf("%-78s one wave %7.3f ms = %6.2f cycles/instr | a wave per SIMD %7.3f ms = %6.2f\n", NAME, one, one * 1e-3 * 2.4e9 / n, many, many * 1e-3 * 2.4e9 / n); \
    }
    CHAINS(X)
#undef X
    const float l = time_ms([&] { k_lds<<<1, 64>>>(d, 3); });
    printf("%-78s one wave %7.3f ms = %6.2f cycles per look-up\n", "LDS table: ds_read (uniform address) -> readfirstlane -> next address", l, l * 1e-3 * 2.4e9 / (ITERS * 16.0));
    const float sl = time_ms([&] { k_sload<<<1, 64>>>(d, 3, tab); });
    printf("%-78s one wave %7.3f ms = %6.2f cycles per look-up\n", "scalar cache: s_load_dword -> next address", sl, sl * 1e-3 * 2.4e9 / (ITERS * 16.0));
    const float rl = time_ms([&] { k_readlane_table<<<1, 64>>>(d, 3); });
    printf("%-78s one wave %7.3f ms = %6.2f cycles per look-up\n", "register table: v_readlane (lane = previous result) -> next", rl, rl * 1e-3 * 2.4e9 / (ITERS * 16.0));
    return 0;
}
