// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU ops the
// pixel pipeline is made of.  8 waves per SIMD, 16 independent chains per lane, 4096 iterations.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define N_ACC 16
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[N_ACC];
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) a[i] = x + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
            if (OP == 0) a[i] = a[i] + y;                                         // v_add_u32
            if (OP == 1) a[i] = (uint32_t)__mul24((int)a[i], 2217);               // v_mul_i32_i24
            if (OP == 2) a[i] = (uint32_t)__mul24((int)a[i], 2217) + y;           // v_mad_i32_i24
            if (OP == 3) a[i] = a[i] * 2217u;                                     // v_mul_lo_u32
            if (OP == 4) a[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, y), (int)a[i], false);  // v_dot2c_i32_i16
            if (OP == 5) { us2 t = {3, 3}; a[i] = __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a[i]) * t + __builtin_bit_cast(us2, y))); }  // v_pk_mad_u16
            if (OP == 6) a[i] = __builtin_amdgcn_perm(a[i], y, 0x05040100u);      // v_perm_b32
            if (OP == 7) a[i] = (a[i] << 12) + y;                                 // v_lshl_add_u32
            if (OP == 8) a[i] = a[i] + y + (uint32_t)it;                          // v_add3_u32
            if (OP == 9) a[i] = (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32((int)a[i], (int)y, 17);  // v_ashr_pk_u8_i32 (+and)
            if (OP == 10) a[i] = __builtin_amdgcn_alignbit(a[i], y, 16);          // v_alignbit_b32
            if (OP == 11) a[i] = (uint32_t)((int32_t)a[i] >> 10);                 // v_ashrrev_i32
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < N_ACC; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 x 256 threads per CU -> 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)blocks * 4 /*waves*/ * ITERS * N_ACC / (256.0 * 4);
    printf("%-22s %8.3f ms  -> %6.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, ms * 1e6 / insts_per_simd * 2.0);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_i32_i24", d); run<2>("v_mad_i32_i24", d); run<3>("v_mul_lo_u32", d);
    run<4>("v_dot2c_i32_i16", d); run<5>("v_pk_mad_u16", d); run<6>("v_perm_b32", d); run<7>("v_lshl_add_u32", d);
    run<8>("v_add3_u32", d); run<9>("v_ashr_pk_u8_i32+and", d); run<10>("v_alignbit_b32", d); run<11>("v_ashrrev_i32", d);
    return 0;
}
