// Micro-benchmark 2: issue cost (ns and cycles per wave64 instruction per SIMD) of candidate gfx950 VALU ops for the
// pixel pipeline, plus semantic probes (op_sel / SDWA destination halves).  Each op runs as 16 independent chains per
// lane, 8 waves per SIMD, so the figure is the issue rate, not the latency.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu2 tools/ubench_valu2.hip && ./ubench_valu2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define N_ACC 16
#define ITERS 2048

#define OPS(X)                                                                                             \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                             \
    X(1, "v_and_b32", "v_and_b32 %0, %0, %1")                                                             \
    X(2, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0")                                                      \
    X(3, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0")                                                      \
    X(4, "v_bfe_i32", "v_bfe_i32 %0, %0, 4, 12")                                                          \
    X(5, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                       \
    X(6, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2")                                                 \
    X(7, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1")                                                     \
    X(8, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %0, %1, %2")                                               \
    X(9, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2")                                                   \
    X(10, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1")                                                      \
    X(11, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1")                                                \
    X(12, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 1, %0")                                               \
    X(13, "v_ashr_pk_u8_i32", "v_ashr_pk_u8_i32 %0, %0, %1, 3")                                           \
    X(14, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 3")                                               \
    X(15, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                                                      \
    X(16, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1")                                               \
    X(17, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2")                                                        \
    X(18, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0")                                                  \
    X(19, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %0, %1, 1, %0")                                             \
    X(20, "v_cvt_f32_i32_sdwa", "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
    X(21, "v_lshlrev_b32_sdwa", "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(22, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                      \
    X(23, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2")                                                \
    X(24, "v_mad_i32_i16", "v_mad_i32_i16 %0, %0, %1, %2 op_sel:[1,0,0,0]")                               \
    X(25, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 8")                                               \
    X(26, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc")                                               \
    X(27, "v_max_i32", "v_max_i32 %0, %0, %1")                                                            \
    X(28, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2")                                                      \
    X(29, "v_mul_i32_i24_sdwa", "v_mul_i32_i24_sdwa %0, %1, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(30, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %0, %1, %2")                                                \
    X(31, "v_lerp_u8", "v_lerp_u8 %0, %0, %1, %2")                                                        \
    X(32, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 1, %0")                                               \
    X(33, "v_sat_pk_u8_i16", "v_sat_pk_u8_i16 %0, %0")                                                    \
    X(34, "v_ashrrev_i32_sdwa_w1", "v_ashrrev_i32_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD") \
    X(35, "v_mov_b32", "v_mov_b32 %0, %1")                                                                \
    X(36, "v_xor_b32", "v_xor_b32 %0, %0, %1")                                                            \
    X(37, "v_sub_u32", "v_sub_u32 %0, %0, %1")                                                            \
    X(38, "v_pk_mad_i16", "v_pk_mad_i16 %0, %0, %1, %2")                                                  \
    X(39, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                                                  \
    X(40, "v_bfe_u32", "v_bfe_u32 %0, %0, 4, 8")                                                          \
    X(41, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1")                                                      \
    X(42, "v_cvt_pk_i16_i32", "v_cvt_pk_i16_i32 %0, %0, %1")                                              \
    X(43, "v_add_u32_sdwa_b1", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[N_ACC];
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u, z = 0x05040100u;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) a[i] = x + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
#define X(ID, NAME, ASM) \
    if (OP == ID) asm volatile(ASM : "+v"(a[i]) : "v"(y), "v"(z));
            OPS(X)
#undef X
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) r ^= a[i];
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
    printf("%-24s %8.3f ms  %6.3f ns/wave-instr/SIMD  (= %.2f cyc @2.4GHz)\n", name, best, best * 1e6 / insts_per_simd,
           best * 1e6 / insts_per_simd * 2.4);
}

// ---- semantic probes --------------------------------------------------------------------------------
__global__ void probe(uint32_t *out) {
    uint32_t a = 0x12345678u << 4, b = 0x00000900u << 4, c = 0x0000a500u << 4, d = 0xfff00000u;  // >>12: 0x45678(sat 255), 0x9, 0xa5, negative(0)
    uint32_t r0 = 0xdeadbeefu, r1 = 0xdeadbeefu, r2 = 0xdeadbeefu, r3;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 12" : "+v"(r0) : "v"(b), "v"(c));                    // low half written; high half?
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 12 op_sel:[0,0,0,1]" : "+v"(r1) : "v"(b), "v"(c));  // high half written?
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 12\n\tv_ashr_pk_u8_i32 %0, %3, %4, 12 op_sel:[0,0,0,1]" : "+v"(r2) : "v"(b), "v"(c), "v"(a), "v"(d));
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    // SDWA shift into the high word, preserving the low word
    r3 = 0x0000beefu;
    uint32_t sh = 4, val = 0x00012340u;
    asm volatile("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r3) : "v"(sh), "v"(val));
    out[3] = r3;  // expect 0x1234beef
    // v_sat_pk_u8_i16: two i16 -> two u8 saturated
    uint32_t p = 0x01ffff80u, q;  // hi = 0x01ff (511 -> 255), lo = 0xff80 (-128 -> 0)
    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(q) : "v"(p));
    out[4] = q;
    // v_cvt_pk_u8_f32: truncation or rounding? 3.99 -> ?, -1.5 -> ?, 300 -> ?
    float f0 = 3.99f, f1 = -1.5f, f2 = 300.0f, f3 = 254.999f;
    uint32_t w = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0\n\tv_cvt_pk_u8_f32 %0, %2, 1, %0\n\tv_cvt_pk_u8_f32 %0, %3, 2, %0\n\tv_cvt_pk_u8_f32 %0, %4, 3, %0"
                 : "+v"(w) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    out[5] = w;
    // v_mad_i32_i16 op_sel: hi half of src0
    uint32_t pk = 0xfff60005u, kk = 1000u, acc = 7u, m0, m1;  // hi = -10, lo = 5
    asm volatile("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(m0) : "v"(pk), "v"(kk), "v"(acc));
    asm volatile("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(m1) : "v"(pk), "v"(kk), "v"(acc));
    out[6] = m0;  // 5007
    out[7] = m1;  // -9993
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(ID, NAME, ASM) run<ID>(NAME, d);
    OPS(X)
#undef X
    probe<<<1, 1>>>(d);
    uint32_t h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("probe ashr_pk lo-only (dst was deadbeef): %08x\nprobe ashr_pk op_sel hi (dst was deadbeef): %08x\nprobe ashr_pk lo then hi: %08x (want 00ff a509 -> 00ffa509?)\n", h[0], h[1], h[2]);
    printf("probe sdwa WORD_1 preserve: %08x (want 1234beef)\nprobe v_sat_pk_u8_i16(0x01ffff80): %08x\nprobe cvt_pk_u8_f32(3.99,-1.5,300,254.999): %08x\n", h[3], h[4], h[5]);
    printf("probe mad_i32_i16 lo: %d (want 5007), hi via op_sel: %d (want -9993)\n", (int)h[6], (int)h[7]);
    return 0;
}
