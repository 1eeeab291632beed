/*
 * oracle_pixels.c — CPU ORACLE for the pixel pipeline (test infrastructure only).
 *
 * Restates, in plain C with explicit wrap-around arithmetic, the scalar path of
 * image-rs/jpeg-decoder v0.3.2:
 *   src/idct.rs            dequantize + IDCT 8x8 / 4x4 / 2x2 / 1x1
 *   src/worker/immediate.rs plane layout of append_row
 *   src/upsampler.rs       H1V1 / H2V1 / H1V2 / H2V2 / Generic
 *   src/decoder.rs:1300-1508 compute_image and the colour-convert line functions
 * Every function cites the lines it follows.  See jpeg_oracle.h for parity status.
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Wrapping<i32> (src/idct.rs:12,275): all adds / muls / left shifts are done on
 * uint32_t (defined wrap-around in C); arithmetic right shift on int32_t. */
typedef uint32_t w32;
static inline w32 W(int32_t x) { return (w32)x; }
static inline int32_t S(w32 x) { return (int32_t)x; }
static inline w32 sar(w32 x, int n) {
    /* arithmetic shift right of the i32 value, implementation-defined in C for
     * negatives but arithmetic on every compiler we use; made explicit anyway */
    int32_t v = (int32_t)x;
    return (w32)(v < 0 ? ~((~v) >> n) : (v >> n));
}

/* src/idct.rs:572-574  stbi_f2f: (x * 4096.0 + 0.5) as i32, evaluated in f32 */
static inline w32 f2f(float x) { return W((int32_t)(x * 4096.0f + 0.5f)); }
/* src/idct.rs:576-578 */
static inline w32 fsh(w32 x) { return x << 12; }
/* src/idct.rs:568-570 */
static inline uint8_t stbi_clamp(w32 x) {
    int32_t v = S(x);
    if (v < 0) v = 0;
    if (v > 255) v = 255;
    return (uint8_t)v;
}
/* src/idct.rs:450-452 */
static inline w32 dequantize(int16_t c, uint16_t q) { return W((int32_t)c * (int32_t)q); }

/* src/idct.rs:377-407 */
static void kernel_x(w32 s0, w32 s2, w32 s4, w32 s6, w32 x_scale, w32 xs[4]) {
    w32 t2, t3, t0, t1;
    {
        w32 p2 = s2, p3 = s6;
        w32 p1 = (p2 + p3) * f2f(0.5411961f);
        t2 = p1 + p3 * f2f(-1.847759065f);
        t3 = p1 + p2 * f2f(0.765366865f);
    }
    {
        w32 p2 = s0, p3 = s4;
        t0 = fsh(p2 + p3);
        t1 = fsh(p2 - p3);
    }
    w32 x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;
    xs[0] = x0 + x_scale;
    xs[1] = x1 + x_scale;
    xs[2] = x2 + x_scale;
    xs[3] = x3 + x_scale;
}

/* src/idct.rs:409-439 */
static void kernel_t(w32 s1, w32 s3, w32 s5, w32 s7, w32 ts[4]) {
    w32 t0 = s7, t1 = s5, t2 = s3, t3 = s1;
    w32 p3 = t0 + t2, p4 = t1 + t3, p1 = t0 + t3, p2 = t1 + t2;
    w32 p5 = (p3 + p4) * f2f(1.175875602f);
    t0 *= f2f(0.298631336f);
    t1 *= f2f(2.053119869f);
    t2 *= f2f(3.072711026f);
    t3 *= f2f(1.501321110f);
    p1 = p5 + p1 * f2f(-0.899976223f);
    p2 = p5 + p2 * f2f(-2.562915447f);
    p3 = p3 * f2f(-1.961570560f);
    p4 = p4 * f2f(-0.390180644f);
    t3 += p1 + p4;
    t2 += p2 + p3;
    t1 += p2 + p4;
    t0 += p1 + p3;
    ts[0] = t0;
    ts[1] = t1;
    ts[2] = t2;
    ts[3] = t3;
}

/* src/idct.rs:241-370 (dequantize_and_idct_block_8x8 + _inner, scalar path) */
static void idct_8x8(const int16_t *c, const uint16_t *q, size_t stride, uint8_t *out) {
    w32 temp[64];
    /* columns, :278-325 */
    for (int i = 0; i < 8; i++) {
        if (c[i + 8] == 0 && c[i + 16] == 0 && c[i + 24] == 0 && c[i + 32] == 0 &&
            c[i + 40] == 0 && c[i + 48] == 0 && c[i + 56] == 0) {
            w32 dcterm = dequantize(c[i], q[i]) << 2;
            for (int k = 0; k < 8; k++) temp[i + 8 * k] = dcterm;
        } else {
            w32 s[8], xs[4], ts[4];
            for (int k = 0; k < 8; k++) s[k] = dequantize(c[i + 8 * k], q[i + 8 * k]);
            kernel_x(s[0], s[2], s[4], s[6], W(512), xs);
            kernel_t(s[1], s[3], s[5], s[7], ts);
            temp[i] = sar(xs[0] + ts[3], 10);
            temp[i + 56] = sar(xs[0] - ts[3], 10);
            temp[i + 8] = sar(xs[1] + ts[2], 10);
            temp[i + 48] = sar(xs[1] - ts[2], 10);
            temp[i + 16] = sar(xs[2] + ts[1], 10);
            temp[i + 40] = sar(xs[2] - ts[1], 10);
            temp[i + 24] = sar(xs[3] + ts[0], 10);
            temp[i + 32] = sar(xs[3] - ts[0], 10);
        }
    }
    /* rows, :327-369 */
    const w32 X_SCALE = W(65536 + (128 << 17));
    for (int r = 0; r < 8; r++) {
        const w32 *chunk = temp + 8 * r;
        uint8_t *o = out + (size_t)r * stride;
        if (chunk[1] == 0 && chunk[2] == 0 && chunk[3] == 0 && chunk[4] == 0 && chunk[5] == 0 &&
            chunk[6] == 0 && chunk[7] == 0) {
            uint8_t dcterm = stbi_clamp(sar(fsh(chunk[0]) + X_SCALE, 17));
            for (int k = 0; k < 8; k++) o[k] = dcterm;
        } else {
            w32 xs[4], ts[4];
            kernel_x(chunk[0], chunk[2], chunk[4], chunk[6], X_SCALE, xs);
            kernel_t(chunk[1], chunk[3], chunk[5], chunk[7], ts);
            o[0] = stbi_clamp(sar(xs[0] + ts[3], 17));
            o[7] = stbi_clamp(sar(xs[0] - ts[3], 17));
            o[1] = stbi_clamp(sar(xs[1] + ts[2], 17));
            o[6] = stbi_clamp(sar(xs[1] - ts[2], 17));
            o[2] = stbi_clamp(sar(xs[2] + ts[1], 17));
            o[5] = stbi_clamp(sar(xs[2] - ts[1], 17));
            o[3] = stbi_clamp(sar(xs[3] + ts[0], 17));
            o[4] = stbi_clamp(sar(xs[3] - ts[0], 17));
        }
    }
}

/* src/idct.rs:456-517 */
static void idct_4x4(const int16_t *c, const uint16_t *q, size_t stride, uint8_t *out) {
    w32 temp[16];
    for (int i = 0; i < 4; i++) {
        w32 s0 = dequantize(c[i], q[i]);
        w32 s1 = dequantize(c[i + 8], q[i + 8]);
        w32 s2 = dequantize(c[i + 16], q[i + 16]);
        w32 s3 = dequantize(c[i + 24], q[i + 24]);
        w32 x0 = (s0 + s2) << 2;
        w32 x2 = (s0 - s2) << 2;
        w32 p1 = (s1 + s3) * f2f(0.541196100f);
        w32 t0 = sar(p1 + s3 * f2f(-1.847759065f) + W(512), 10);
        w32 t2 = sar(p1 + s1 * f2f(0.765366865f) + W(512), 10);
        temp[i] = x0 + t2;
        temp[i + 12] = x0 - t2;
        temp[i + 4] = x2 + t0;
        temp[i + 8] = x2 - t0;
    }
    for (int i = 0; i < 4; i++) {
        w32 s0 = temp[i * 4], s1 = temp[i * 4 + 1], s2 = temp[i * 4 + 2], s3 = temp[i * 4 + 3];
        w32 x0 = (s0 + s2) << 12;
        w32 x2 = (s0 - s2) << 12;
        w32 p1 = (s1 + s3) * f2f(0.541196100f);
        w32 t0 = p1 + s3 * f2f(-1.847759065f);
        w32 t2 = p1 + s1 * f2f(0.765366865f);
        x0 = x0 + W(1 << 16) + W(128 << 17);
        x2 = x2 + W(1 << 16) + W(128 << 17);
        uint8_t *o = out + (size_t)i * stride;
        o[0] = stbi_clamp(sar(x0 + t2, 17));
        o[3] = stbi_clamp(sar(x0 - t2, 17));
        o[1] = stbi_clamp(sar(x2 + t0, 17));
        o[2] = stbi_clamp(sar(x2 - t0, 17));
    }
}

/* src/idct.rs:519-553 */
static void idct_2x2(const int16_t *c, const uint16_t *q, size_t stride, uint8_t *out) {
    w32 s00 = dequantize(c[0], q[0]);
    w32 s10 = dequantize(c[8], q[8]);
    w32 x0 = s00 + s10, x2 = s00 - s10;
    w32 s01 = dequantize(c[1], q[1]);
    w32 s11 = dequantize(c[9], q[9]);
    w32 x1 = s01 + s11, x3 = s01 - s11;
    x0 = x0 + W(1 << 2) + W(128 << 3);
    x2 = x2 + W(1 << 2) + W(128 << 3);
    out[0] = stbi_clamp(sar(x0 + x1, 3));
    out[1] = stbi_clamp(sar(x0 - x1, 3));
    out[stride + 0] = stbi_clamp(sar(x2 + x3, 3));
    out[stride + 1] = stbi_clamp(sar(x2 - x3, 3));
}

/* src/idct.rs:555-565 — note Wrapping<i32> `/`: truncating division */
static void idct_1x1(const int16_t *c, const uint16_t *q, uint8_t *out) {
    int32_t s0 = S(dequantize(c[0], q[0]) + W(128 * 8)) / 8;
    out[0] = stbi_clamp(W(s0));
}

/* src/idct.rs:205-239 */
void orc_dequantize_and_idct_block(int scale, const int16_t coefficients[64],
                                   const uint16_t quantization_table[64],
                                   size_t output_linestride, uint8_t *output) {
    switch (scale) {
    case 8: idct_8x8(coefficients, quantization_table, output_linestride, output); break;
    case 4: idct_4x4(coefficients, quantization_table, output_linestride, output); break;
    case 2: idct_2x2(coefficients, quantization_table, output_linestride, output); break;
    case 1: idct_1x1(coefficients, quantization_table, output); break;
    default: abort(); /* panic!("Unsupported IDCT scale") */
    }
}

/* src/idct.rs:14-28 */
int orc_choose_idct_size(uint16_t full_w, uint16_t full_h, uint16_t req_w, uint16_t req_h) {
    static const int scales[3] = {1, 2, 4};
    for (int k = 0; k < 3; k++) {
        uint32_t sc = (uint32_t)scales[k];
        uint16_t sw = (uint16_t)(((uint32_t)full_w * sc - 1) / 8 + 1);
        uint16_t sh = (uint16_t)(((uint32_t)full_h * sc - 1) / 8 + 1);
        if (sw >= req_w || sh >= req_h) return scales[k];
    }
    return 8;
}

/* src/parser.rs:282-290 */
static int ceil_div(uint32_t x, uint32_t y, uint16_t *out) {
    if (x == 0 || y == 0) return ORC_ERR_FORMAT;
    *out = (uint16_t)(1 + ((x - 1) / y));
    return ORC_OK;
}

/* src/parser.rs:292-310 */
int orc_update_component_sizes(uint16_t width, uint16_t height, orc_component *components,
                               int ncomp, uint16_t *mcu_w, uint16_t *mcu_h) {
    uint32_t h_max = 0, v_max = 0;
    for (int i = 0; i < ncomp; i++) {
        if (components[i].h > h_max) h_max = components[i].h;
        if (components[i].v > v_max) v_max = components[i].v;
    }
    uint16_t mw, mh;
    if (ceil_div(width, h_max * 8, &mw) || ceil_div(height, v_max * 8, &mh)) return ORC_ERR_FORMAT;
    for (int i = 0; i < ncomp; i++) {
        orc_component *c = &components[i];
        if (ceil_div((uint32_t)width * c->h * c->dct_scale, h_max * 8, &c->size_w)) return ORC_ERR_FORMAT;
        if (ceil_div((uint32_t)height * c->v * c->dct_scale, v_max * 8, &c->size_h)) return ORC_ERR_FORMAT;
        c->block_w = (uint16_t)(mw * c->h);
        c->block_h = (uint16_t)(mh * c->v);
    }
    *mcu_w = mw;
    *mcu_h = mh;
    return ORC_OK;
}

/* src/worker/immediate.rs:30-37 */
size_t orc_plane_bytes(const orc_component *c) {
    return (size_t)c->block_w * c->block_h * c->dct_scale * c->dct_scale;
}

/* src/worker/immediate.rs:39-60 (== src/worker/rayon.rs:71-112) */
void orc_append_rows(const orc_component *c, const uint16_t qt[64], const int16_t *coefs,
                     size_t first_mcu_row, size_t n_mcu_rows, uint8_t *plane) {
    size_t block_count = (size_t)c->block_w * c->v;
    size_t line_stride = (size_t)c->block_w * c->dct_scale;
    size_t row_bytes = block_count * c->dct_scale * c->dct_scale; /* offsets[index] += ... */
    for (size_t r = 0; r < n_mcu_rows; r++) {
        size_t offset = (first_mcu_row + r) * row_bytes;
        const int16_t *data = coefs + r * block_count * 64;
        for (size_t i = 0; i < block_count; i++) {
            size_t x = (i % c->block_w) * c->dct_scale;
            size_t y = (i / c->block_w) * c->dct_scale;
            orc_dequantize_and_idct_block((int)c->dct_scale, data + i * 64, qt, line_stride,
                                          plane + offset + y * line_stride + x);
        }
    }
}

/* ---- upsampling, src/upsampler.rs ---------------------------------------- */

enum { UP_H1V1, UP_H2V1, UP_H1V2, UP_H2V2, UP_GENERIC };

typedef struct {
    int kind;
    int hf, vf; /* Generic: horizontal/vertical scaling factors */
    size_t width, height, row_stride;
} up_component;

/* src/upsampler.rs:76-105 */
static int choose_upsampler(int h, int v, int h_max, int v_max, uint16_t out_w, uint16_t out_h,
                            up_component *u) {
    int h1 = (h == h_max) || out_w == 1;
    int v1 = (v == v_max) || out_h == 1;
    int h2 = h * 2 == h_max;
    int v2 = v * 2 == v_max;
    u->hf = u->vf = 1;
    if (h1 && v1) u->kind = UP_H1V1;
    else if (h2 && v1) u->kind = UP_H2V1;
    else if (h1 && v2) u->kind = UP_H1V2;
    else if (h2 && v2) u->kind = UP_H2V2;
    else if (h_max % h != 0 || v_max % v != 0) return ORC_ERR_UNSUPPORTED;
    else {
        u->kind = UP_GENERIC;
        u->hf = h_max / h;
        u->vf = v_max / v;
    }
    return ORC_OK;
}

/* `as usize` on an f32: saturating, NaN -> 0 */
static size_t f32_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    return (size_t)f;
}

/* src/upsampler.rs:174-180 / 200-206 */
static void near_far(size_t row, size_t input_height, size_t *near, size_t *far) {
    float row_near = (float)row / 2.0f;
    float fract = row_near - truncf(row_near);
    float row_far = row_near + fract * 3.0f - 0.25f;
    float lim = (float)(input_height - 1);
    if (lim < row_far) row_far = lim; /* f32::min */
    *near = f32_as_usize(row_near);
    *far = f32_as_usize(row_far);
}

/* One output row of one component; returns 0 or ORC_ERR_INTERNAL when the
 * reference would index out of bounds (a Rust panic). `in_len` = plane bytes. */
static int upsample_row(const up_component *u, const uint8_t *input, size_t in_len, size_t row,
                        size_t output_width, uint8_t *output, size_t out_len) {
    size_t W = u->width, stride = u->row_stride;
    switch (u->kind) {
    case UP_H1V1: { /* :119-132 */
        size_t off = row * stride;
        if (off + output_width > in_len || output_width > out_len) return ORC_ERR_INTERNAL;
        memcpy(output, input + off, output_width);
        return ORC_OK;
    }
    case UP_H2V1: { /* :134-163 */
        size_t off = row * stride;
        if (off + W > in_len || 2 * W > out_len) return ORC_ERR_INTERNAL;
        const uint8_t *in = input + off;
        if (W == 1) {
            output[0] = in[0];
            output[1] = in[0];
            return ORC_OK;
        }
        output[0] = in[0];
        output[1] = (uint8_t)(((uint32_t)in[0] * 3 + in[1] + 2) >> 2);
        for (size_t i = 1; i + 1 < W; i++) {
            uint32_t sample = 3 * (uint32_t)in[i] + 2;
            output[i * 2] = (uint8_t)((sample + in[i - 1]) >> 2);
            output[i * 2 + 1] = (uint8_t)((sample + in[i + 1]) >> 2);
        }
        output[(W - 1) * 2] = (uint8_t)(((uint32_t)in[W - 1] * 3 + in[W - 2] + 2) >> 2);
        output[(W - 1) * 2 + 1] = in[W - 1];
        return ORC_OK;
    }
    case UP_H1V2: { /* :165-189 */
        size_t near, far;
        near_far(row, u->height, &near, &far);
        if (near * stride + output_width > in_len || far * stride + output_width > in_len ||
            output_width > out_len)
            return ORC_ERR_INTERNAL;
        const uint8_t *in_near = input + near * stride, *in_far = input + far * stride;
        for (size_t i = 0; i < output_width; i++)
            output[i] = (uint8_t)((3 * (uint32_t)in_near[i] + in_far[i] + 2) >> 2);
        return ORC_OK;
    }
    case UP_H2V2: { /* :191-228 */
        size_t near, far;
        near_far(row, u->height, &near, &far);
        if (near * stride + W > in_len || far * stride + W > in_len || 2 * W > out_len)
            return ORC_ERR_INTERNAL;
        const uint8_t *in_near = input + near * stride, *in_far = input + far * stride;
        if (W == 1) {
            uint8_t value = (uint8_t)((3 * (uint32_t)in_near[0] + in_far[0] + 2) >> 2);
            output[0] = value;
            output[1] = value;
            return ORC_OK;
        }
        uint32_t t1 = 3 * (uint32_t)in_near[0] + in_far[0];
        output[0] = (uint8_t)((t1 + 2) >> 2);
        for (size_t i = 1; i < W; i++) {
            uint32_t t0 = t1;
            t1 = 3 * (uint32_t)in_near[i] + in_far[i];
            output[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            output[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        output[W * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return ORC_OK;
    }
    default: { /* Generic, :230-250 */
        size_t start = (row / (size_t)u->vf) * stride;
        if (start + W > in_len || W * (size_t)u->hf > out_len) return ORC_ERR_INTERNAL;
        size_t index = 0;
        for (size_t i = 0; i < W; i++)
            for (int k = 0; k < u->hf; k++) output[index++] = input[start + i];
        return ORC_OK;
    }
    }
}

/* ---- colour conversion, src/decoder.rs:1391-1508 ------------------------ */

static int32_t f2f20(float x) { return (int32_t)(x * (float)(1 << 20) + 0.5f); } /* :1502-1504 */
static uint8_t clamp_fixed_point(int32_t v) {                                     /* :1506-1508 */
    v >>= 20;
    if (v > 255) v = 255;
    if (v < 0) v = 0;
    return (uint8_t)v;
}
/* :1486-1500 */
void orc_ycbcr_to_rgb(uint8_t y8, uint8_t cb8, uint8_t cr8, uint8_t rgb[3]) {
    int32_t y = (int32_t)y8 * (1 << 20) + (1 << 19);
    int32_t cb = (int32_t)cb8 - 128;
    int32_t cr = (int32_t)cr8 - 128;
    rgb[0] = clamp_fixed_point(y + f2f20(1.40200f) * cr);
    rgb[1] = clamp_fixed_point(y - f2f20(0.34414f) * cb - f2f20(0.71414f) * cr);
    rgb[2] = clamp_fixed_point(y + f2f20(1.77200f) * cb);
}

enum { CC_NONE, CC_RGB, CC_YCBCR, CC_CMYK, CC_YCCK };

/* src/decoder.rs:1339-1389 */
static int choose_color_convert(int ncomp, int ct, int *fn, char *msg) {
    const char *m = NULL;
    int code = ORC_ERR_FORMAT;
    if (ncomp != 3 && ncomp != 4) abort(); /* panic!() */
    switch (ct) {
    case ORC_CT_NONE: *fn = CC_NONE; return ORC_OK;
    case ORC_CT_GRAYSCALE: m = "Invalid number of channels for Grayscale data"; break;
    case ORC_CT_RGB:
        if (ncomp == 3) { *fn = CC_RGB; return ORC_OK; }
        m = "Invalid number of channels (4) for RGB data"; break;
    case ORC_CT_YCBCR:
        if (ncomp == 3) { *fn = CC_YCBCR; return ORC_OK; }
        m = "Invalid number of channels (4) for YCbCr data"; break;
    case ORC_CT_CMYK:
        if (ncomp == 4) { *fn = CC_CMYK; return ORC_OK; }
        m = "Invalid number of channels (3) for CMYK data"; break;
    case ORC_CT_YCCK:
        if (ncomp == 4) { *fn = CC_YCCK; return ORC_OK; }
        m = "Invalid number of channels (3) for YCCK data"; break;
    case ORC_CT_JCS_BG_YCC:
    case ORC_CT_JCS_BG_RGB: m = "unsupported colour transform"; code = ORC_ERR_UNSUPPORTED; break;
    default: m = "Unknown colour transform"; break;
    }
    if (msg) snprintf(msg, 128, "%s", m);
    return code;
}

/* src/decoder.rs:1300-1336 and src/worker/mod.rs:97-128 */
int orc_compute_image(const orc_component *comps, int ncomp, uint8_t *const *planes,
                      uint16_t out_w, uint16_t out_h, int color_transform, uint8_t *out,
                      char *msg) {
    if (msg) msg[0] = 0;
    if (ncomp == 1) {
        /* :1310-1332 — copy_within compaction (done here as a strided copy) */
        const orc_component *c = &comps[0];
        size_t width = c->size_w, height = c->size_h;
        size_t line_stride = (size_t)c->block_w * c->dct_scale;
        if ((size_t)out_w != line_stride) {
            for (size_t y = 0; y < height; y++) memmove(out + y * width, planes[0] + y * line_stride, width);
        } else {
            /* decoded.resize(size, 0) — plane is at least width*height long when width==stride */
            size_t have = orc_plane_bytes(c), want = width * height;
            memcpy(out, planes[0], have < want ? have : want);
            if (have < want) memset(out + have, 0, want - have);
        }
        return ORC_OK;
    }
    int fn;
    int rc = choose_color_convert(ncomp, color_transform, &fn, msg);
    if (rc) return rc;

    /* Upsampler::new, src/upsampler.rs:20-45 */
    int h_max = 0, v_max = 0;
    size_t max_w = 0;
    for (int i = 0; i < ncomp; i++) {
        if (comps[i].h > h_max) h_max = comps[i].h;
        if (comps[i].v > v_max) v_max = comps[i].v;
        if (comps[i].size_w > max_w) max_w = comps[i].size_w;
    }
    up_component ups[4];
    for (int i = 0; i < ncomp; i++) {
        rc = choose_upsampler(comps[i].h, comps[i].v, h_max, v_max, out_w, out_h, &ups[i]);
        if (rc) {
            if (msg) snprintf(msg, 128, "NonIntegerSubsamplingRatio");
            return rc;
        }
        ups[i].width = comps[i].size_w;
        ups[i].height = comps[i].size_h;
        ups[i].row_stride = (size_t)comps[i].block_w * comps[i].dct_scale;
    }
    size_t line_buffer_size = max_w * (size_t)h_max;
    uint8_t *lines = (uint8_t *)calloc((size_t)ncomp, line_buffer_size ? line_buffer_size : 1);
    if (!lines) return ORC_ERR_INTERNAL;
    size_t line_size = (size_t)out_w * (size_t)ncomp;

    for (size_t row = 0; row < out_h; row++) {
        uint8_t *o = out + row * line_size;
        /* upsample_and_interleave_row, src/upsampler.rs:47-63 (fresh zeroed buffers) */
        memset(lines, 0, (size_t)ncomp * line_buffer_size);
        for (int i = 0; i < ncomp; i++) {
            rc = upsample_row(&ups[i], planes[i], orc_plane_bytes(&comps[i]), row, out_w,
                              lines + (size_t)i * line_buffer_size, line_buffer_size);
            if (rc) {
                if (msg) snprintf(msg, 128, "reference would panic: upsample out of bounds");
                free(lines);
                return rc;
            }
        }
        const uint8_t *l0 = lines, *l1 = lines + line_buffer_size, *l2 = lines + 2 * line_buffer_size,
                      *l3 = lines + 3 * line_buffer_size;
        size_t npx = out_w < line_buffer_size ? out_w : line_buffer_size; /* zip() */
        switch (fn) {
        case CC_RGB: /* :1391-1404 */
            for (size_t x = 0; x < npx; x++) { o[3*x] = l0[x]; o[3*x+1] = l1[x]; o[3*x+2] = l2[x]; }
            break;
        case CC_YCBCR: /* :1406-1437 */
            for (size_t x = 0; x < npx; x++) orc_ycbcr_to_rgb(l0[x], l1[x], l2[x], o + 3 * x);
            break;
        case CC_YCCK: /* :1439-1456 */
            for (size_t x = 0; x < npx; x++) {
                orc_ycbcr_to_rgb(l0[x], l1[x], l2[x], o + 4 * x);
                o[4 * x + 3] = (uint8_t)(255 - l3[x]);
            }
            break;
        case CC_CMYK: /* :1458-1474 */
            for (size_t x = 0; x < npx; x++) {
                o[4*x] = (uint8_t)(255 - l0[x]); o[4*x+1] = (uint8_t)(255 - l1[x]);
                o[4*x+2] = (uint8_t)(255 - l2[x]); o[4*x+3] = (uint8_t)(255 - l3[x]);
            }
            break;
        default: /* color_no_convert :1476-1484 — walks whole line buffers; output_iter.next().unwrap()
                    panics when ncomp*line_buffer_size > line_size */
            if ((size_t)ncomp * line_buffer_size > line_size) {
                if (msg) snprintf(msg, 128, "reference would panic: color_no_convert overruns the row");
                free(lines);
                return ORC_ERR_INTERNAL;
            }
            memcpy(o, lines, (size_t)ncomp * line_buffer_size);
            break;
        }
    }
    free(lines);
    return ORC_OK;
}
