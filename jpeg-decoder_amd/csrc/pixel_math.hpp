// pixel_math.hpp — gfx950 device functions for the JPEG pixel pipeline (exact integer path).
//
// Bit-exact restatement targets (image-rs/jpeg-decoder v0.3.2, scalar / platform_independent):
//   src/idct.rs:241-452,568-578   dequantize + 8x8 IDCT (stb_image derived, Wrapping<i32>)
//   src/idct.rs:456-565           reduced 4x4 / 2x2 / 1x1
//   src/decoder.rs:1486-1508      YCbCr -> RGB (20-bit fixed point BT.601)
// All arithmetic that can wrap is done on uint32_t (defined mod 2^32); arithmetic right
// shifts and clamps are done on int32_t, exactly as Wrapping<i32> behaves.
//
// The header is also compiled by g++ (tests/emu, -DJPGPU_HOST_EMULATION) so that kernel
// logic can be checked against the oracle on the CPU; the product only ever runs it on gfx950.
#pragma once
#ifndef JPGPU_HOST_EMULATION
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// Address spaces.  Pointers that reach a kernel inside a struct are "flat" to the compiler:
// flat_load/flat_store instead of global_*, no scalar loads, no wide stores.  The kernels
// therefore cast them once: JP_GLOBAL = global memory, JP_CONST = read-only global memory whose
// wave-uniform loads become s_load (used for the quantization tables, which then live in SGPRs).
#ifdef JPGPU_HOST_EMULATION
#define JP_GLOBAL
#define JP_CONST
#define JP_LDS
#else
#define JP_GLOBAL __attribute__((address_space(1)))
#define JP_CONST __attribute__((address_space(4)))
#define JP_LDS __attribute__((address_space(3)))  // ds_* instead of flat_* when a pointer to LDS crosses a function boundary
#endif

// A pointer into LDS as a value the compiler cannot look into (it stays an LDS pointer: ds_* instructions).  For addresses
// that are used inside a loop: a known one is kept as offset + segment start and put together at every use.
template <class T>
__device__ __forceinline__ T *opaque_lds(T *p) {
#ifdef JPGPU_HOST_EMULATION
    return p;
#else
    JP_LDS T *q = (JP_LDS T *)p;
    asm volatile("" : "+v"(q));
    return (T *)q;
#endif
}

namespace jpgpu {

// Plain vector types for memory access (the HIP uint2/uint4 classes cannot be assigned through
// address-space qualified pointers).  v3u_a4: 12 bytes at 4-byte alignment -> *_dwordx3.
#ifdef JPGPU_HOST_EMULATION
struct v2u { uint32_t x, y; };
struct v3u { uint32_t x, y, z; };
struct v4u { uint32_t x, y, z, w; };
typedef v3u v3u_a4;
#else
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v3u v3u_a4 __attribute__((aligned(4)));
#endif

typedef uint32_t w32;
// Streaming accesses: coefficients are read once and pixels written once per decode, so they carry the
// non-temporal hint (unless built with -DJPGPU_NO_STREAM_NT) and leave L2 / Infinity Cache to the data that is re-read (the 4:2:0 chroma
// planes between the two passes).  Measured on MI355X (256 x 1080p): 4:2:0 0.845 -> 0.812 ms, gray 0.299 -> 0.273 ms.
// (macros, not templates: template argument deduction would drop the 4-byte alignment of v3u_a4)
#if !defined(JPGPU_NO_STREAM_NT) && !defined(JPGPU_HOST_EMULATION)
#define stream_load(p) __builtin_nontemporal_load(p)
#else
#define stream_load(p) (*(p))
#endif
#if !defined(JPGPU_NO_STREAM_NT) && !defined(JPGPU_HOST_EMULATION)
#define stream_store(p, ...) __builtin_nontemporal_store((__VA_ARGS__), (p))
#else
#define stream_store(p, ...) (*(p) = (__VA_ARGS__))
#endif
// (What the hint does to PARTIAL lines — a lane writing 12-byte pieces 24 bytes apart fills half of each line per
// instruction: tools/attic/ubench_store.hip, profiles/round2/01_store_patterns.txt — alone 2.5 TB/s against 5.3 TB/s for plain
// stores.  The single-launch 4:2:0 kernel therefore stores plainly, PixelOps::row_pixels<.., NTS = false>.)

typedef const JP_CONST uint32_t *qtab_t;  // 64 u16 quantization values packed two per dword, 4-B aligned

__device__ __forceinline__ qtab_t as_qtab(const uint16_t *q) { return (qtab_t)q; }
__device__ __forceinline__ int32_t q_at(qtab_t q, int idx) {
    uint32_t d = q[idx >> 1];
    return (int32_t)((idx & 1) ? (d >> 16) : (d & 0xffffu));
}
__device__ __forceinline__ int32_t qw_at(const uint32_t (&qw)[32], int idx) {
    uint32_t d = qw[idx >> 1];
    return (int32_t)((idx & 1) ? (d >> 16) : (d & 0xffffu));
}

// stbi_f2f(x) = (x * 4096.0f + 0.5f) as i32, evaluated in f32 (src/idct.rs:572-574).
// Values as listed in SURVEY Appendix A.1; the oracle computes them with the f32 formula and
// tests/test_emulation.py checks both agree on every block.
constexpr int32_t F_0_5411961 = 2217;
constexpr int32_t F_N1_847759065 = -7567;
constexpr int32_t F_0_765366865 = 3135;
constexpr int32_t F_1_175875602 = 4816;
constexpr int32_t F_0_298631336 = 1223;
constexpr int32_t F_2_053119869 = 8410;
constexpr int32_t F_3_072711026 = 12586;
constexpr int32_t F_1_501321110 = 6149;
constexpr int32_t F_N0_899976223 = -3685;
constexpr int32_t F_N2_562915447 = -10497;
constexpr int32_t F_N1_961570560 = -8034;
constexpr int32_t F_N0_390180644 = -1597;

__device__ __forceinline__ w32 sar(w32 x, int n) { return (w32)((int32_t)x >> n); }

// ---- packed 2 x u16 helpers (v_pk_mad_u16 / v_pk_add_u16 / v_pk_lshrrev_b16, v_alignbit, v_perm) ----
#ifdef JPGPU_HOST_EMULATION
__device__ __forceinline__ uint32_t pk_mad3(uint32_t a, uint32_t b) {  // per 16-bit lane: 3*a + b
    uint32_t lo = (3u * (a & 0xffffu) + (b & 0xffffu)) & 0xffffu, hi = (3u * (a >> 16) + (b >> 16)) & 0xffffu;
    return lo | (hi << 16);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return (((a & 0xffffu) + (b & 0xffffu)) & 0xffffu) | ((((a >> 16) + (b >> 16)) & 0xffffu) << 16);
}
__device__ __forceinline__ uint32_t pk_shr(uint32_t a, int n) { return ((a & 0xffffu) >> n) | (((a >> 16) >> n) << 16); }
__device__ __forceinline__ uint32_t pk_sar2(uint32_t a) {  // per 16-bit lane: arithmetic shift right by 2
    return ((uint32_t)((int32_t)(int16_t)(a & 0xffffu) >> 2) & 0xffffu) | ((uint32_t)((int32_t)(int16_t)(a >> 16) >> 2) << 16);
}
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, int n) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> n);
}
__device__ __forceinline__ uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) {  // per 16-bit lane: low 16 bits of a*b
    return (((a & 0xffffu) * (b & 0xffffu)) & 0xffffu) | ((((a >> 16) * (b >> 16)) & 0xffffu) << 16);
}
// c + a.lo*b.lo + a.hi*b.hi with the halves read as signed 16-bit (wraps mod 2^32)
__device__ __forceinline__ w32 dot2_i16(uint32_t a, uint32_t b, w32 c) {
    int64_t r = (int64_t)(int16_t)(a & 0xffffu) * (int16_t)(b & 0xffffu) + (int64_t)(int16_t)(a >> 16) * (int16_t)(b >> 16);
    return (w32)((uint64_t)r + c);
}
__device__ __forceinline__ w32 dot2_i16_sc(uint32_t a, uint32_t b, w32 c) { return dot2_i16(a, b, c); }
// byte i of the result = byte sel[i] of {src0 (4..7), src1 (0..3)}; 0x0c = zero
__device__ __forceinline__ uint32_t perm_b32(uint32_t src0, uint32_t src1, uint32_t sel) {
    uint64_t all = ((uint64_t)src0 << 32) | src1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xffu;
        uint32_t b = s < 8 ? (uint32_t)((all >> (8 * s)) & 0xffu) : 0u;
        r |= b << (8 * i);
    }
    return r;
}
#else
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2_t as_us2(uint32_t x) { return __builtin_bit_cast(us2_t, x); }
__device__ __forceinline__ uint32_t us2_bits(us2_t x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_mad3(uint32_t a, uint32_t b) {
    const us2_t three = {3, 3};
    return us2_bits(as_us2(a) * three + as_us2(b));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return us2_bits(as_us2(a) + as_us2(b)); }
__device__ __forceinline__ uint32_t pk_shr(uint32_t a, int n) { return us2_bits(as_us2(a) >> (unsigned short)n); }
__device__ __forceinline__ uint32_t pk_sar2(uint32_t a) {  // v_pk_ashrrev_i16
    typedef short ss2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ss2_t, a) >> (short)2);
}
__device__ __forceinline__ uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) { return us2_bits(as_us2(a) * as_us2(b)); }
typedef short s2_t __attribute__((ext_vector_type(2)));
// v_dot2_i32_i16 in its three-operand (VOP3P) form.  Through the builtin hipcc picks the accumulating
// VOP2 form v_dot2c and pays a v_mov per chain to seed the accumulator with the rounding constant; gfx9
// allows one SGPR per VALU instruction, so the constant pair sits in a VGPR and the addend may be an
// SGPR / inline constant.
__device__ __forceinline__ w32 dot2_i16(uint32_t a, uint32_t b, w32 c) {
    uint32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ w32 dot2_i16_sc(uint32_t a, uint32_t b, w32 c_uniform) {  // addend wave-uniform (SGPR / inline)
    uint32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, int n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
__device__ __forceinline__ uint32_t perm_b32(uint32_t src0, uint32_t src1, uint32_t sel) { return __builtin_amdgcn_perm(src0, src1, sel); }
#endif

// 24-bit multiply: low 32 bits of sext24(a) * sext24(b).  Exact mod 2^32 whenever both true
// operands lie in [-2^23, 2^23) — v_mul_i32_i24 / v_mad_i32_i24 are full rate on gfx950 while
// v_mul_lo_u32 is not.
__device__ __forceinline__ w32 mul24(w32 a, int32_t c) {
#ifdef JPGPU_HOST_EMULATION
    int64_t x = (int64_t)((int32_t)(a << 8) >> 8) * (int64_t)((int32_t)((w32)c << 8) >> 8);
    return (w32)x;
#else
    return (w32)__mul24((int)a, c);
#endif
}

// multiply by an IDCT constant: SANE -> 24-bit path (caller guarantees the operand range),
// otherwise the full 32-bit wrapping multiply.
template <bool SANE>
__device__ __forceinline__ w32 mulc(w32 a, int32_t c) {
    if constexpr (SANE) return mul24(a, c);
    else return a * (w32)c;
}

// sat_u8(a >> n) | sat_u8(b >> n) << 8, upper 16 bits zero.
// gfx950 has v_ashr_pk_u8_i32 for exactly this.  NOTE (ROCm 7.2 / LLVM 22): when the compiler
// pattern-matches the instruction on its own from `clamp(a>>n) | clamp(b>>n)<<8 | c<<16 ...`
// it does not clear bits 31:16 of the result (found on the first GPU run: bytes 2,3 of every
// packed dword were corrupted).  So the instruction is always used through the builtin, whose
// 16-bit return type makes the compiler mask the upper half.
__device__ __forceinline__ uint32_t sar_sat_u8x2(w32 a, w32 b, int n) {
#ifdef JPGPU_HOST_EMULATION
    int32_t x = (int32_t)a >> n, y = (int32_t)b >> n;
    x = x < 0 ? 0 : (x > 255 ? 255 : x);
    y = y < 0 ? 0 : (y > 255 ? 255 : y);
    return (uint32_t)x | ((uint32_t)y << 8);
#else
    return (uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32((int)a, (int)b, n);
#endif
}
// Same, but bits 31:16 of the result are UNDEFINED (the raw instruction leaves garbage there).
// Only for consumers that read bytes 0,1 alone (v_perm_b32): saves the v_and the builtin costs.
template <int N>
__device__ __forceinline__ uint32_t sar_sat_u8x2_raw(w32 a, w32 b) {
#ifdef JPGPU_HOST_EMULATION
    return sar_sat_u8x2(a, b, N) | 0xdead0000u;  // poison the undefined half so misuse shows up in emulation
#else
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(N));
    return r;
#endif
}
// four values -> one dword, byte 0 = a.  The instruction writes ONE 16-bit half of its destination and leaves the other
// alone (that is what the "undefined" upper half above is: the register's previous content); op_sel[3] selects the
// upper half.  So the second pair goes straight into the upper half of the first pair's register — no v_perm_b32 to
// merge them (probe: profiles/round2/00_valu_issue_cost_ubench.txt; the instruction alone costs two issue slots).
template <int N>
__device__ __forceinline__ uint32_t sar_sat_u8x4(w32 a, w32 b, w32 c, w32 d) {
#ifdef JPGPU_HOST_EMULATION
    return (sar_sat_u8x2(a, b, N) & 0xffffu) | (sar_sat_u8x2(c, d, N) << 16);
#else
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(N));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d), "n"(N));
    return r;
#endif
}
__device__ __forceinline__ uint32_t clamp_u8(w32 x) {  // stbi_clamp, src/idct.rs:568-570
    return sar_sat_u8x2(x, 0u, 0) & 0xffu;
}

// One 8-point pass = kernel_x + kernel_t (src/idct.rs:377-447) followed by the butterfly
// of :318-325 / :361-368.  o[k] is the value *before* the final shift.
template <bool SANE>
__device__ __forceinline__ void idct_pass8(const w32 (&s)[8], w32 x_scale, w32 (&o)[8]) {
    // kernel_x (even part)
    w32 p1 = mulc<SANE>(s[2] + s[6], F_0_5411961);
    w32 t2 = p1 + mulc<SANE>(s[6], F_N1_847759065);
    w32 t3 = p1 + mulc<SANE>(s[2], F_0_765366865);
    w32 t0 = ((s[0] + s[4]) << 12) + x_scale;
    w32 t1 = ((s[0] - s[4]) << 12) + x_scale;
    w32 x0 = t0 + t3;
    w32 x3 = t0 - t3;
    w32 x1 = t1 + t2;
    w32 x2 = t1 - t2;
    // kernel_t (odd part)
    w32 u0 = s[7], u1 = s[5], u2 = s[3], u3 = s[1];
    w32 p3 = u0 + u2, p4 = u1 + u3, q1 = u0 + u3, q2 = u1 + u2;
    w32 p5 = mulc<SANE>(p3 + p4, F_1_175875602);
    u0 = mulc<SANE>(u0, F_0_298631336);
    u1 = mulc<SANE>(u1, F_2_053119869);
    u2 = mulc<SANE>(u2, F_3_072711026);
    u3 = mulc<SANE>(u3, F_1_501321110);
    q1 = p5 + mulc<SANE>(q1, F_N0_899976223);
    q2 = p5 + mulc<SANE>(q2, F_N2_562915447);
    p3 = mulc<SANE>(p3, F_N1_961570560);
    p4 = mulc<SANE>(p4, F_N0_390180644);
    u3 += q1 + p4;
    u2 += q2 + p3;
    u1 += q2 + p4;
    u0 += q1 + p3;
    o[0] = x0 + u3;
    o[7] = x0 - u3;
    o[1] = x1 + u2;
    o[6] = x1 - u2;
    o[2] = x2 + u1;
    o[5] = x2 - u1;
    o[3] = x3 + u0;
    o[4] = x3 - u0;
}

// coefficient (row r, column c) of a block held as 32 packed dwords (natural order,
// two i16 per dword, little endian): dword r*4 + c/2.
__device__ __forceinline__ int32_t coef_at(const uint32_t (&cw)[32], int r, int c) {
    uint32_t d = cw[r * 4 + (c >> 1)];
    return (c & 1) ? ((int32_t)d >> 16) : (int32_t)(int16_t)(d & 0xffffu);
}

// 8x8 dequantize + IDCT of one block held by ONE lane (src/idct.rs:278-369).
//   cw : 64 coefficients, packed as above
//   q  : 64 u16 quantization values (natural order)
//   out: 8 rows x 8 bytes, two dwords per row (byte 0 = leftmost sample)
// SANE == false: wrap-exact for any input.  Both DC-only short-cuts of the reference are
//   honoured: the column one is NOT value-neutral under wrap-around and is
//   selected per column; the row one is algebraically identical to the general formula.
// SANE == true : caller guarantees |c*q| < 2^15 for every coefficient.  Then every multiplicand
//   of both passes lies inside [-2^23, 2^23) (column outputs are bounded by 5.55*sum|s| < 2^21,
//   row multiplicands are sums of at most four of them), so 24-bit multiplies are exact, and the
//   column short-cut equals the general formula (s0 << 12 cannot wrap).  See DESIGN.md.
// (lo, hi) pair of signed 16-bit constants as one dword
constexpr uint32_t pk_i16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// One 8-point pass written as the exact integer linear map it is (mod 2^32), derived from
// kernel_x / kernel_t (src/idct.rs:377-447) by distributing the constants:
//   x0..x3 = Ex * (s0,s2,s4,s6) + x_scale        u0..u3 = Ou * (s1,s3,s5,s7)
//   Ex = [4096  5352  4096  2217]   Ou = [1131 -3218  4816 -5680]
//        [4096  2217 -4096 -5350]        [3219 -5681  1132  4816]
//        [4096 -2217 -4096  5350]        [4816 -1129 -5681 -3218]
//        [4096 -5352  4096 -2217]        [5683  4816  3219  1131]
// Inputs arrive as signed 16-bit pairs p04 = (s0,s4), p26 = (s2,s6), p13 = (s1,s3), p57 = (s5,s7).  The odd half is the
// 4x4 map Ou: two v_dot2_i32_i16 per row (2 MACs per instruction).  The even half keeps one butterfly level of
// kernel_x — e0/e1 = 4096*(s0 +- s4) + x_scale and t3/t2 = the (s2,s6) rotation, one dot2 each, then x0,x3 = e0 +- t3 and
// x1,x2 = e1 +- t2 — because additions issue at twice the rate of dot2 on gfx950 (profiles/round2/00_*): 12 dot2 +
// 12 add/sub per pass instead of 16 + 8.  The same integer linear map mod 2^32 either way.  Valid whenever every input fits i16.
// PLUS_SHL: o[0..3] (the sums) come out shifted left by that much — (a + b) << n is one instruction (v_add_lshl_u32),
// (a - b) << n is not.
template <int PLUS_SHL = 0>
__device__ __forceinline__ void idct_pass8_dot2(uint32_t p04, uint32_t p26, uint32_t p13, uint32_t p57, w32 x_scale,
                                                w32 (&o)[8]) {
    const w32 e0 = dot2_i16_sc(p04, pk_i16(4096, 4096), x_scale);
    const w32 e1 = dot2_i16_sc(p04, pk_i16(4096, -4096), x_scale);
    const w32 t3 = dot2_i16_sc(p26, pk_i16(5352, 2217), 0u);
    const w32 t2 = dot2_i16_sc(p26, pk_i16(2217, -5350), 0u);
    const w32 x0 = e0 + t3, x3 = e0 - t3, x1 = e1 + t2, x2 = e1 - t2;
    const w32 u0 = dot2_i16(p13, pk_i16(1131, -3218), dot2_i16_sc(p57, pk_i16(4816, -5680), 0u));
    const w32 u1 = dot2_i16(p13, pk_i16(3219, -5681), dot2_i16_sc(p57, pk_i16(1132, 4816), 0u));
    const w32 u2 = dot2_i16(p13, pk_i16(4816, -1129), dot2_i16_sc(p57, pk_i16(-5681, -3218), 0u));
    const w32 u3 = dot2_i16(p13, pk_i16(5683, 4816), dot2_i16_sc(p57, pk_i16(3219, 1131), 0u));
    o[0] = (x0 + u3) << PLUS_SHL;
    o[7] = x0 - u3;
    o[1] = (x1 + u2) << PLUS_SHL;
    o[6] = x1 - u2;
    o[2] = (x2 + u1) << PLUS_SHL;
    o[5] = x2 - u1;
    o[3] = (x3 + u0) << PLUS_SHL;
    o[4] = x3 - u0;
}

// Arithmetic variants of the 8x8 IDCT (all bit-exact with the reference on the inputs they accept):
enum : int {
    ARITH_EXACT = 0,  // any input: 32-bit wrapping multiplies + the per-column DC short-cut select
    ARITH_SANE = 1,   // every |c*q| < 2^15: packed 16-bit dequantization, dot2 column pass, 24-bit row pass
    ARITH_TIGHT = 2,  // additionally every column of every block has sum_k |c*q| <= 5900, so the column-pass
                      // outputs (<= (5683*5900 + 512) >> 10 < 2^15) fit i16 and the row pass runs on dot2 too
};

// The two passes on DEQUANTIZED coefficients that fit i16 (classes ARITH_SANE / ARITH_TIGHT): d[k*4+j] = (s[k][2j], s[k][2j+1]).
// Split from the dequantization so that a kernel may multiply while it copies a block out of LDS (S420::read_block).
template <int ARITH>
__device__ __forceinline__ void idct8x8_products(const uint32_t (&d)[32], uint32_t (&out)[16]) {
    static_assert(ARITH == ARITH_SANE || ARITH == ARITH_TIGHT, "the exact class works on 32-bit products");
    const w32 X_SCALE = 65536u + (128u << 17);
    if constexpr (ARITH == ARITH_TIGHT) {
        // Column-pass outputs fit i16: they are paired for the row pass — (0,4) (2,6) (1,3) (5,7) — as soon as both columns
        // of a pair exist, so 32 packed dwords are kept instead of 64 values (registers: one more wave per SIMD).
        // The row pass wants bits 10..25 of the sums as 16-bit halves: rows 0..3 leave the pass shifted left by 6 and the
        // pairing takes their HIGH halves (32 shifts less per block); rows 4..7 are shifted here.
        uint32_t tp[32];  // tp[r*4 + m]: row r, pair m of (0,4) (2,6) (1,3) (5,7)
        constexpr int PAIR[4][2] = {{0, 4}, {2, 6}, {1, 3}, {5, 7}};
#pragma unroll
        for (int m = 0; m < 4; m++) {
            w32 o[2][8];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = PAIR[m][h], j = i >> 1;
                const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;  // (lo.half, hi.half) of column i
                const uint32_t p04 = perm_b32(d[4 * 4 + j], d[0 * 4 + j], sel), p26 = perm_b32(d[6 * 4 + j], d[2 * 4 + j], sel);
                const uint32_t p13 = perm_b32(d[3 * 4 + j], d[1 * 4 + j], sel), p57 = perm_b32(d[7 * 4 + j], d[5 * 4 + j], sel);
                idct_pass8_dot2<6>(p04, p26, p13, p57, 512u, o[h]);
            }
#pragma unroll
            for (int r = 0; r < 8; r++)
                tp[r * 4 + m] = r < 4 ? perm_b32(o[1][r], o[0][r], 0x07060302u) : perm_b32(sar(o[1][r], 10), sar(o[0][r], 10), 0x05040100u);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            w32 o[8];
            idct_pass8_dot2(tp[r * 4 + 0], tp[r * 4 + 1], tp[r * 4 + 2], tp[r * 4 + 3], X_SCALE, o);
            out[r * 2] = sar_sat_u8x4<17>(o[0], o[1], o[2], o[3]);
            out[r * 2 + 1] = sar_sat_u8x4<17>(o[4], o[5], o[6], o[7]);
        }
    } else {
        w32 t[64];  // t[k*8+i] = column-pass output (row k, column i)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = i >> 1;
            const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;  // (lo.half, hi.half) of column i
            const uint32_t p04 = perm_b32(d[4 * 4 + j], d[0 * 4 + j], sel), p26 = perm_b32(d[6 * 4 + j], d[2 * 4 + j], sel);
            const uint32_t p13 = perm_b32(d[3 * 4 + j], d[1 * 4 + j], sel), p57 = perm_b32(d[7 * 4 + j], d[5 * 4 + j], sel);
            w32 o[8];
            if constexpr (ARITH == ARITH_TIGHT) {
                idct_pass8_dot2<6>(p04, p26, p13, p57, 512u, o);
#pragma unroll
                for (int k = 0; k < 8; k++) t[k * 8 + i] = k < 4 ? o[k] : sar(o[k], 10);
            } else {
                idct_pass8_dot2(p04, p26, p13, p57, 512u, o);
#pragma unroll
                for (int k = 0; k < 8; k++) t[k * 8 + i] = sar(o[k], 10);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            w32 o[8];
            if constexpr (ARITH == ARITH_TIGHT) {
                const uint32_t h16 = r < 4 ? 0x07060302u : 0x05040100u;
                idct_pass8_dot2(perm_b32(t[r * 8 + 4], t[r * 8 + 0], h16), perm_b32(t[r * 8 + 6], t[r * 8 + 2], h16),
                                perm_b32(t[r * 8 + 3], t[r * 8 + 1], h16), perm_b32(t[r * 8 + 7], t[r * 8 + 5], h16), X_SCALE, o);
            } else {
                w32 s[8];
#pragma unroll
                for (int k = 0; k < 8; k++) s[k] = t[r * 8 + k];
                idct_pass8<true>(s, X_SCALE, o);
            }
            out[r * 2] = sar_sat_u8x4<17>(o[0], o[1], o[2], o[3]);
            out[r * 2 + 1] = sar_sat_u8x4<17>(o[4], o[5], o[6], o[7]);
        }
    }
}

// ONE output row (ROW = 0 or 7) of idct8x8_products, the same values at about a fifth of the instructions: the column pass
// needs a single output per column — o[0] = x0 + u3 or o[7] = x0 - u3: four dot2 — and the row pass runs once.  For the seam
// rounds of the strip walks (fused_core.hpp): of the chroma blocks above and below a workgroup's rows only the sample row that
// touches them enters the fancy upsampler (src/upsampler.rs:200-206).
template <int ARITH, int ROW>
__device__ __forceinline__ void idct8x8_products_row(const uint32_t (&d)[32], uint32_t &lo, uint32_t &hi) {
    static_assert(ARITH == ARITH_SANE || ARITH == ARITH_TIGHT, "the exact class works on 32-bit products");
    static_assert(ROW == 0 || ROW == 7, "first or last sample row");
    const w32 X_SCALE = 65536u + (128u << 17);
    w32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = i >> 1;
        const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;  // (lo.half, hi.half) of column i
        const uint32_t p04 = perm_b32(d[4 * 4 + j], d[0 * 4 + j], sel), p26 = perm_b32(d[6 * 4 + j], d[2 * 4 + j], sel);
        const uint32_t p13 = perm_b32(d[3 * 4 + j], d[1 * 4 + j], sel), p57 = perm_b32(d[7 * 4 + j], d[5 * 4 + j], sel);
        const w32 x0 = dot2_i16_sc(p04, pk_i16(4096, 4096), 512u) + dot2_i16_sc(p26, pk_i16(5352, 2217), 0u);
        const w32 u3 = dot2_i16(p13, pk_i16(5683, 4816), dot2_i16_sc(p57, pk_i16(3219, 1131), 0u));
        t[i] = sar(ROW == 0 ? x0 + u3 : x0 - u3, 10);
    }
    w32 o[8];
    if constexpr (ARITH == ARITH_TIGHT) {  // (the column-pass outputs fit i16: the row pass on dot2, as in the full transform)
        idct_pass8_dot2(perm_b32(t[4], t[0], 0x05040100u), perm_b32(t[6], t[2], 0x05040100u), perm_b32(t[3], t[1], 0x05040100u),
                        perm_b32(t[7], t[5], 0x05040100u), X_SCALE, o);
    } else {
        idct_pass8<true>(t, X_SCALE, o);
    }
    lo = sar_sat_u8x4<17>(o[0], o[1], o[2], o[3]);
    hi = sar_sat_u8x4<17>(o[4], o[5], o[6], o[7]);
}

// qw: the 64 quantization values packed two per dword (natural order), as VALUES — in SGPRs when they
// were loaded through a wave-uniform table pointer, in VGPRs when lanes of a wave use different tables.
template <int ARITH>
__device__ __forceinline__ void idct8x8(const uint32_t (&cw)[32], const uint32_t (&qw)[32], uint32_t (&out)[16]) {
    const w32 X_SCALE = 65536u + (128u << 17);
    if constexpr (ARITH == ARITH_EXACT) {
        w32 t[64];
        // dequantize first (row by row, so the packed coefficients and the table die early and the
        // 64 products are the only long-lived values), then the column pass in place
        uint32_t acbits[4] = {0u, 0u, 0u, 0u};  // OR of the packed raw coefficients of rows 1..7
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) t[k * 8 + i] = mul24((w32)coef_at(cw, k, i), qw_at(qw, k * 8 + i));  // i16 x u16: always exact
            if (k >= 1) {
#pragma unroll
                for (int dd = 0; dd < 4; dd++) acbits[dd] |= cw[k * 4 + dd];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w32 s[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = t[k * 8 + i];
            idct_pass8<false>(s, 512u, o);
            // raw-coefficient test of :279-285 on the packed halves
            const uint32_t bits = acbits[i >> 1];
            const bool dc_only = ((i & 1) ? (bits >> 16) : (bits & 0xffffu)) == 0;
            const w32 dcterm = s[0] << 2;
#pragma unroll
            for (int k = 0; k < 8; k++) t[k * 8 + i] = dc_only ? dcterm : sar(o[k], 10);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            w32 s[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = t[r * 8 + k];
            idct_pass8<false>(s, X_SCALE, o);
            out[r * 2] = sar_sat_u8x4<17>(o[0], o[1], o[2], o[3]);
            out[r * 2 + 1] = sar_sat_u8x4<17>(o[4], o[5], o[6], o[7]);
        }
    } else {
        // |c*q| < 2^15: the products fit i16, so the dequantization is a packed 16-bit multiply
        // (two coefficients per instruction) and the column pass runs on dot2.
        uint32_t d[32];  // d[k*4+j] = (s[k][2j], s[k][2j+1])
#pragma unroll
        for (int i = 0; i < 32; i++) d[i] = pk_mul_lo_u16(cw[i], qw[i]);
        idct8x8_products<ARITH>(d, out);
    }
}

// table through a wave-uniform pointer: the 32 dwords are s_load'ed and stay in SGPRs
template <int ARITH>
__device__ __forceinline__ void idct8x8(const uint32_t (&cw)[32], qtab_t q, uint32_t (&out)[16]) {
    uint32_t qw[32];
#pragma unroll
    for (int i = 0; i < 32; i++) qw[i] = q[i];
    idct8x8<ARITH>(cw, qw, out);
}

// src/idct.rs:456-517; out: 4 rows x 4 bytes (one dword per row)
__device__ __forceinline__ void idct4x4_exact(const uint32_t (&cw)[32], qtab_t q, uint32_t (&out)[4]) {
    w32 temp[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w32 s0 = (w32)(coef_at(cw, 0, i) * q_at(q, i));
        w32 s1 = (w32)(coef_at(cw, 1, i) * q_at(q, 8 + i));
        w32 s2 = (w32)(coef_at(cw, 2, i) * q_at(q, 16 + i));
        w32 s3 = (w32)(coef_at(cw, 3, i) * q_at(q, 24 + i));
        w32 x0 = (s0 + s2) << 2;
        w32 x2 = (s0 - s2) << 2;
        w32 p1 = (s1 + s3) * (w32)F_0_5411961;
        w32 t0 = sar(p1 + s3 * (w32)F_N1_847759065 + 512u, 10);
        w32 t2 = sar(p1 + s1 * (w32)F_0_765366865 + 512u, 10);
        temp[i] = x0 + t2;
        temp[i + 12] = x0 - t2;
        temp[i + 4] = x2 + t0;
        temp[i + 8] = x2 - t0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w32 s0 = temp[i * 4], s1 = temp[i * 4 + 1], s2 = temp[i * 4 + 2], s3 = temp[i * 4 + 3];
        w32 x0 = ((s0 + s2) << 12) + (1u << 16) + (128u << 17);
        w32 x2 = ((s0 - s2) << 12) + (1u << 16) + (128u << 17);
        w32 p1 = (s1 + s3) * (w32)F_0_5411961;
        w32 t0 = p1 + s3 * (w32)F_N1_847759065;
        w32 t2 = p1 + s1 * (w32)F_0_765366865;
        out[i] = sar_sat_u8x4<17>(x0 + t2, x2 + t0, x2 - t0, x0 - t2);
    }
}

// src/idct.rs:519-553; out: row0 in bytes 0,1 and row1 in bytes 2,3
__device__ __forceinline__ uint32_t idct2x2_exact(const uint32_t (&cw)[32], qtab_t q) {
    w32 s00 = (w32)(coef_at(cw, 0, 0) * q_at(q, 0));
    w32 s10 = (w32)(coef_at(cw, 1, 0) * q_at(q, 8));
    w32 s01 = (w32)(coef_at(cw, 0, 1) * q_at(q, 1));
    w32 s11 = (w32)(coef_at(cw, 1, 1) * q_at(q, 9));
    w32 x0 = s00 + s10 + 4u + (128u << 3);
    w32 x2 = s00 - s10 + 4u + (128u << 3);
    w32 x1 = s01 + s11, x3 = s01 - s11;
    return sar_sat_u8x4<3>(x0 + x1, x0 - x1, x2 + x3, x2 - x3);
}

// src/idct.rs:555-565 — truncating division by 8 of the wrapped sum
__device__ __forceinline__ uint32_t idct1x1_exact(uint32_t c0_word, qtab_t q) {
    int32_t s0 = (int32_t)((w32)((int32_t)(int16_t)(c0_word & 0xffffu) * q_at(q, 0)) + 1024u);
    return clamp_u8((w32)(s0 / 8));
}

// ---- colour, src/decoder.rs:1486-1508 -----------------------------------------------------
// stbi_f2f(x) = (x * 2^20 + 0.5) as i32 in f32: 1.402 -> 1470104, 0.34414 -> 360857,
// 0.71414 -> 748830, 1.772 -> 1858077.
//   r = clamp((Y + CR_R*cr') >> 20), Y = y*2^20 + 2^19, cr' = cr - 128  ... etc.
// Restated with the -128 offsets folded into constants (pure integer re-association, no
// intermediate can overflow: every term is below 2^29):
//   r_raw = (y << 20) + KR + CR_R*cr,               KR = 2^19 - 128*CR_R
//   g_raw = (y << 20) + KG - CB_G*cb - CR_G*cr,     KG = 2^19 + 128*(CB_G + CR_G)
//   b_raw = (y << 20) + KB + CB_B*cb,               KB = 2^19 - 128*CB_B
// Returns r | g << 8 | b << 16.
constexpr int32_t CR_R = 1470104, CB_G = 360857, CR_G = 748830, CB_B = 1858077;
constexpr int32_t KR = (1 << 19) - 128 * CR_R;
constexpr int32_t KG = (1 << 19) + 128 * (CB_G + CR_G);
constexpr int32_t KB = (1 << 19) - 128 * CB_B;

// (byte k of d) << 20 in ONE instruction: a shift with an SDWA byte select on its operand
// (hipcc emits shift + mask for bytes 0..2).
template <int K>
__device__ __forceinline__ w32 byte_shl20(uint32_t d) {
#ifdef JPGPU_HOST_EMULATION
    return ((d >> (8 * K)) & 0xffu) << 20;
#else
    uint32_t r;
    if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(20u), "v"(d));
    else if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(20u), "v"(d));
    else if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(20u), "v"(d));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(20u), "v"(d));
    return r;
#endif
}

struct RawRgb {
    w32 r, g, b;  // 20-bit fixed point, before the >> 20 and the clamp
};
// yb = y << 20
__device__ __forceinline__ RawRgb ycbcr_raw_yb(w32 yb, uint32_t cb, uint32_t cr) {
    RawRgb o;
    o.r = yb + (mul24(cr, CR_R) + (w32)KR);
    o.g = yb + (mul24(cb, -CB_G) + (mul24(cr, -CR_G) + (w32)KG));
    o.b = yb + (mul24(cb, CB_B) + (w32)KB);
    return o;
}
// The same values from chroma samples that already had their 128 taken off (src/decoder.rs:1489-1491 as written): one
// rounding term for all three channels, so each channel is a chain of v_mad_i32_i24 that starts from it.
// (as an instruction: left to itself the compiler turns the two-step chain of the green channel into two multiplications
// and a three-operand addition)
__device__ __forceinline__ w32 mad24(w32 a, int32_t c_uniform, w32 acc) {
#ifdef JPGPU_HOST_EMULATION
    return mul24(a, c_uniform) + acc;
#else
    w32 r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c_uniform), "v"(acc));
    return r;
#endif
}
__device__ __forceinline__ RawRgb ycbcr_raw_centred(w32 yb, int32_t cb, int32_t cr) {
    const w32 yh = yb + (1u << 19);
    RawRgb o;
    o.r = mad24((w32)cr, CR_R, yh);
    o.g = mad24((w32)cb, -CB_G, mad24((w32)cr, -CR_G, yh));
    o.b = mad24((w32)cb, CB_B, yh);
    return o;
}
__device__ __forceinline__ RawRgb ycbcr_raw(uint32_t y, uint32_t cb, uint32_t cr) {
    const w32 yb = y << 20;
    RawRgb o;
    o.r = yb + (mul24(cr, CR_R) + (w32)KR);
    o.g = yb + (mul24(cb, -CB_G) + (mul24(cr, -CR_G) + (w32)KG));
    o.b = yb + (mul24(cb, CB_B) + (w32)KB);
    return o;
}
// Four pixels -> the 12 output bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3.  The shift-saturate-
// pack instruction takes two values at a time, so the pairs are chosen in output byte order and
// each dword is two of them, one per half (sar_sat_u8x4).
__device__ __forceinline__ void rgb4_to_12bytes(const RawRgb &p0, const RawRgb &p1, const RawRgb &p2, const RawRgb &p3,
                                                uint32_t &d0, uint32_t &d1, uint32_t &d2) {
    d0 = sar_sat_u8x4<20>(p0.r, p0.g, p0.b, p1.r);
    d1 = sar_sat_u8x4<20>(p1.g, p1.b, p2.r, p2.g);
    d2 = sar_sat_u8x4<20>(p2.b, p3.r, p3.g, p3.b);
}

__device__ __forceinline__ uint32_t ycbcr_to_rgb24(uint32_t y, uint32_t cb, uint32_t cr) {
    w32 yb = y << 20;
    w32 r = yb + (w32)KR + mul24(cr, CR_R);
    w32 g = yb + (w32)KG + mul24(cb, -CB_G) + mul24(cr, -CR_G);
    w32 b = yb + (w32)KB + mul24(cb, CB_B);
    uint32_t rg = sar_sat_u8x2(r, g, 20);
    uint32_t b8 = sar_sat_u8x2(b, 0u, 20);
    return rg | (b8 << 16);
}

// store n (<= 8) RGB24 pixels held as packed 24-bit values; off = byte offset in `out`.
// Full chunks at a 4-byte aligned offset go out as two 12-B stores (global_store_dwordx3).
__device__ __forceinline__ void store_rgb_run(JP_GLOBAL uint8_t *out, size_t off, const uint32_t (&px)[8], uint32_t n) {
    JP_GLOBAL uint8_t *o = out + off;
    if (n == 8 && (off & 3u) == 0) {
        const v3u lo = {px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8)};
        const v3u hi = {px[4] | (px[5] << 24), (px[5] >> 8) | (px[6] << 16), (px[6] >> 16) | (px[7] << 8)};
        stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(o), lo);
        stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(o + 12), hi);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++)
            if (k < n) {
                o[3 * k] = (uint8_t)px[k];
                o[3 * k + 1] = (uint8_t)(px[k] >> 8);
                o[3 * k + 2] = (uint8_t)(px[k] >> 16);
            }
    }
}

}  // namespace jpgpu
