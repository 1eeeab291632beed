// pixel_math.hpp — gfx950 device functions for the JPEG pixel pipeline (exact integer path).
//
// Bit-exact restatement targets (image-rs/jpeg-decoder v0.3.2, scalar / platform_independent):
//   src/idct.rs:241-452,568-578   dequantize + 8x8 IDCT (stb_image derived, Wrapping<i32>)
//   src/idct.rs:456-565           reduced 4x4 / 2x2 / 1x1
//   src/upsampler.rs:119-250      H1V1 / H2V1 / H1V2 / H2V2 / Generic
//   src/decoder.rs:1391-1508      colour conversion (20-bit fixed point BT.601)
// All arithmetic that can wrap is done on uint32_t (defined mod 2^32); arithmetic right
// shifts and clamps are done on int32_t, exactly as Wrapping<i32> behaves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpgpu {

typedef uint32_t w32;

// stbi_f2f(x) = (x * 4096.0f + 0.5f) as i32, evaluated in f32 (src/idct.rs:572-574).
// Values pinned by SURVEY Appendix A.1 (static_asserts in tests/test_constants via the oracle).
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
__device__ __forceinline__ w32 mulc(w32 a, int32_t c) { return a * (w32)c; }
__device__ __forceinline__ uint32_t clamp_u8(w32 x) {  // stbi_clamp, src/idct.rs:568-570
    int32_t v = (int32_t)x;
    return (uint32_t)min(max(v, 0), 255);
}

// One 8-point pass = kernel_x + kernel_t (src/idct.rs:377-447) followed by the butterfly
// of :318-325 / :361-368.  o[k] is the value *before* the final shift.
__device__ __forceinline__ void idct_pass8(const w32 (&s)[8], w32 x_scale, w32 (&o)[8]) {
    // kernel_x (even part)
    w32 p1 = mulc(s[2] + s[6], F_0_5411961);
    w32 t2 = p1 + mulc(s[6], F_N1_847759065);
    w32 t3 = p1 + mulc(s[2], F_0_765366865);
    w32 t0 = (s[0] + s[4]) << 12;
    w32 t1 = (s[0] - s[4]) << 12;
    w32 x0 = t0 + t3 + x_scale;
    w32 x3 = t0 - t3 + x_scale;
    w32 x1 = t1 + t2 + x_scale;
    w32 x2 = t1 - t2 + x_scale;
    // kernel_t (odd part)
    w32 u0 = s[7], u1 = s[5], u2 = s[3], u3 = s[1];
    w32 p3 = u0 + u2, p4 = u1 + u3, q1 = u0 + u3, q2 = u1 + u2;
    w32 p5 = mulc(p3 + p4, F_1_175875602);
    u0 = mulc(u0, F_0_298631336);
    u1 = mulc(u1, F_2_053119869);
    u2 = mulc(u2, F_3_072711026);
    u3 = mulc(u3, F_1_501321110);
    q1 = p5 + mulc(q1, F_N0_899976223);
    q2 = p5 + mulc(q2, F_N2_562915447);
    p3 = mulc(p3, F_N1_961570560);
    p4 = mulc(p4, F_N0_390180644);
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

// Exact 8x8 dequantize + IDCT of one block held by ONE lane.
//   cw : 64 coefficients, packed as above
//   q  : 64 u16 quantization values (natural order); wave-uniform pointer (scalar loads)
//   out: 8 rows x 8 bytes, two dwords per row (byte 0 = leftmost sample)
// Follows src/idct.rs:278-369 including both DC-only short-cuts: the column one is NOT
// value-neutral under wrap-around and is selected per column; the row one is
// algebraically identical to the general formula and is therefore not special-cased.
__device__ __forceinline__ void idct8x8_exact(const uint32_t (&cw)[32], const uint16_t *__restrict__ q,
                                              uint32_t (&out)[16]) {
    w32 temp[64];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        w32 s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = (w32)(coef_at(cw, k, i) * (int32_t)q[k * 8 + i]);
        // raw-coefficient test of :279-285 on the packed halves
        uint32_t acbits = 0;
#pragma unroll
        for (int k = 1; k < 8; k++) acbits |= cw[k * 4 + (i >> 1)];
        bool dc_only = ((i & 1) ? (acbits >> 16) : (acbits & 0xffffu)) == 0;
        w32 o[8];
        idct_pass8(s, 512u, o);
        w32 dcterm = s[0] << 2;
#pragma unroll
        for (int k = 0; k < 8; k++) temp[k * 8 + i] = dc_only ? dcterm : sar(o[k], 10);
    }
    const w32 X_SCALE = 65536u + (128u << 17);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        w32 s[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = temp[r * 8 + k];
        idct_pass8(s, X_SCALE, o);
        uint32_t b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) b[k] = clamp_u8(sar(o[k], 17));
        out[r * 2] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        out[r * 2 + 1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    }
}

// src/idct.rs:456-517; out: 4 rows x 4 bytes (one dword per row)
__device__ __forceinline__ void idct4x4_exact(const uint32_t (&cw)[32], const uint16_t *__restrict__ q,
                                              uint32_t (&out)[4]) {
    w32 temp[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w32 s0 = (w32)(coef_at(cw, 0, i) * (int32_t)q[i]);
        w32 s1 = (w32)(coef_at(cw, 1, i) * (int32_t)q[8 + i]);
        w32 s2 = (w32)(coef_at(cw, 2, i) * (int32_t)q[16 + i]);
        w32 s3 = (w32)(coef_at(cw, 3, i) * (int32_t)q[24 + i]);
        w32 x0 = (s0 + s2) << 2;
        w32 x2 = (s0 - s2) << 2;
        w32 p1 = mulc(s1 + s3, F_0_5411961);
        w32 t0 = sar(p1 + mulc(s3, F_N1_847759065) + 512u, 10);
        w32 t2 = sar(p1 + mulc(s1, F_0_765366865) + 512u, 10);
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
        w32 p1 = mulc(s1 + s3, F_0_5411961);
        w32 t0 = p1 + mulc(s3, F_N1_847759065);
        w32 t2 = p1 + mulc(s1, F_0_765366865);
        out[i] = clamp_u8(sar(x0 + t2, 17)) | (clamp_u8(sar(x2 + t0, 17)) << 8) |
                 (clamp_u8(sar(x2 - t0, 17)) << 16) | (clamp_u8(sar(x0 - t2, 17)) << 24);
    }
}

// src/idct.rs:519-553; out: 2 rows x 2 bytes packed as row0 | row1 << 16
__device__ __forceinline__ uint32_t idct2x2_exact(const uint32_t (&cw)[32], const uint16_t *__restrict__ q) {
    w32 s00 = (w32)(coef_at(cw, 0, 0) * (int32_t)q[0]);
    w32 s10 = (w32)(coef_at(cw, 1, 0) * (int32_t)q[8]);
    w32 s01 = (w32)(coef_at(cw, 0, 1) * (int32_t)q[1]);
    w32 s11 = (w32)(coef_at(cw, 1, 1) * (int32_t)q[9]);
    w32 x0 = s00 + s10 + 4u + (128u << 3);
    w32 x2 = s00 - s10 + 4u + (128u << 3);
    w32 x1 = s01 + s11, x3 = s01 - s11;
    return clamp_u8(sar(x0 + x1, 3)) | (clamp_u8(sar(x0 - x1, 3)) << 8) | (clamp_u8(sar(x2 + x3, 3)) << 16) |
           (clamp_u8(sar(x2 - x3, 3)) << 24);
}

// src/idct.rs:555-565 — truncating division by 8 of the wrapped sum
__device__ __forceinline__ uint32_t idct1x1_exact(uint32_t c0_word, const uint16_t *__restrict__ q) {
    int32_t s0 = (int32_t)((w32)((int32_t)(int16_t)(c0_word & 0xffffu) * (int32_t)q[0]) + 1024u);
    return clamp_u8((w32)(s0 / 8));
}

// ---- colour, src/decoder.rs:1486-1508 -----------------------------------------------------
// stbi_f2f(x) = (x * 2^20 + 0.5) as i32 in f32: 1.402 -> 1470104, 0.34414 -> 360857,
// 0.71414 -> 748830, 1.772 -> 1858077.
__device__ __forceinline__ uint32_t clamp_fixed20(int32_t v) { return (uint32_t)min(max(v >> 20, 0), 255); }
__device__ __forceinline__ void ycbcr_to_rgb(uint32_t y8, uint32_t cb8, uint32_t cr8, uint32_t &r, uint32_t &g,
                                             uint32_t &b) {
    int32_t y = (int32_t)y8 * (1 << 20) + (1 << 19);
    int32_t cb = (int32_t)cb8 - 128;
    int32_t cr = (int32_t)cr8 - 128;
    r = clamp_fixed20(y + 1470104 * cr);
    g = clamp_fixed20(y - 360857 * cb - 748830 * cr);
    b = clamp_fixed20(y + 1858077 * cb);
}

}  // namespace jpgpu
