// upsample_color_body.hpp — lane body of the generic upsample + colour-convert kernel (kernels.hip): rows a8-a15 of
// SURVEY §8a for every sampling / colour / scale combination the reference supports (src/upsampler.rs:47-250,
// src/decoder.rs:1300-1484).  Also compiled by g++ for tests/emu.
//
// A lane owns 8 consecutive output pixels of one row.  When the job is `fast8` (all planes at dct_scale 8, fixed
// upsamplers) it reads the planes as aligned dwords — 2 loads for a full-resolution component, 3 (6 with a far
// row) for a horizontally subsampled one — and stores 24 / 32 contiguous bytes; anything else takes the
// byte-granular path below, four pixels at a time.
#pragma once
#include "jobs.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

// src/upsampler.rs:174-180,200-206: row_near = row/2 (f32), row_far = min(row_near +
// fract*3 - 0.25, height-1), both `as usize` (saturating) == the integer forms below.
__device__ __forceinline__ void near_far(uint32_t row, uint32_t height, uint32_t &near, uint32_t &far) {
    near = row >> 1;
    if (row & 1u) far = min(near + 1u, height - 1u);
    else far = near > 0u ? near - 1u : 0u;
}

__device__ __forceinline__ uint32_t up_sample(const UpComp &u, uint32_t x, uint32_t row) {
    const uint8_t *__restrict__ p = u.plane;
    switch (u.kind) {
    case UP_H1V1:  // :119-132
        return p[(size_t)row * u.stride + x];
    case UP_H2V1: {  // :134-163
        const uint8_t *in = p + (size_t)row * u.stride;
        uint32_t W = u.width, i = x >> 1;
        if (x == 0u) return in[0];
        if (x == 2u * W - 1u) return in[W - 1u];
        uint32_t a = in[i], b = (x & 1u) ? in[i + 1u] : in[i - 1u];
        return (3u * a + b + 2u) >> 2;
    }
    case UP_H1V2: {  // :165-189
        uint32_t near, far;
        near_far(row, u.height, near, far);
        return (3u * p[(size_t)near * u.stride + x] + p[(size_t)far * u.stride + x] + 2u) >> 2;
    }
    case UP_H2V2: {  // :191-228
        uint32_t near, far;
        near_far(row, u.height, near, far);
        const uint8_t *n = p + (size_t)near * u.stride, *f = p + (size_t)far * u.stride;
        uint32_t W = u.width, j = x >> 1;
        uint32_t tj = 3u * n[j] + f[j];
        if (x == 0u || x == 2u * W - 1u) return (tj + 2u) >> 2;
        uint32_t o = (x & 1u) ? j + 1u : j - 1u;
        uint32_t to = 3u * n[o] + f[o];
        return (3u * tj + to + 8u) >> 4;
    }
    default:  // Generic :230-250
        return p[(size_t)(row / u.vf) * u.stride + x / u.hf];
    }
}

__device__ __forceinline__ void upsample_color_body(const ImageJob &job, uint32_t x0, uint32_t row) {
    const uint32_t nc = job.ncomp;
    if (job.color_fn == CC_GRAY) {
        // compute_image 1-component compaction, src/decoder.rs:1310-1332
        const UpComp &u = job.comp[0];
        if (row >= u.height || x0 >= u.width) return;
        const uint32_t m = min(4u, u.width - x0);
        for (uint32_t k = 0; k < m; k++)
            job.out[(size_t)row * u.width + x0 + k] = u.plane[(size_t)row * u.stride + x0 + k];
        return;
    }
    if (row >= job.out_h || x0 >= job.out_w) return;
    const uint32_t n = min(4u, job.out_w - x0);
    uint32_t s[4][4];
    for (uint32_t c = 0; c < nc; c++)
        for (uint32_t k = 0; k < 4; k++) s[c][k] = k < n ? up_sample(job.comp[c], x0 + k, row) : 0u;

    if (job.color_fn == CC_NONE) {
        // color_no_convert, src/decoder.rs:1476-1484 (planar within the row; host guarantees
        // line_buffer_size == out_w, otherwise the reference panics and so do we, earlier)
        for (uint32_t c = 0; c < nc; c++)
            for (uint32_t k = 0; k < n; k++)
                job.out[(size_t)row * job.out_w * nc + (size_t)c * job.out_w + x0 + k] = (uint8_t)s[c][k];
        return;
    }
    // px[k] = byte 0..ncomp-1 of output pixel k
    uint32_t px[4];
    for (uint32_t k = 0; k < 4; k++) {
        switch (job.color_fn) {
        case CC_RGB:  // :1391-1404
            px[k] = s[0][k] | (s[1][k] << 8) | (s[2][k] << 16);
            break;
        case CC_YCBCR:  // :1406-1437
            px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]);
            break;
        case CC_YCCK:  // :1439-1456
            px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]) | ((255u - s[3][k]) << 24);
            break;
        default:  // CC_CMYK :1458-1474
            px[k] = (255u - s[0][k]) | ((255u - s[1][k]) << 8) | ((255u - s[2][k]) << 16) | ((255u - s[3][k]) << 24);
            break;
        }
    }
    const size_t off = ((size_t)row * job.out_w + x0) * nc;
    uint8_t *o = job.out + off;
    if (nc == 4) {
        for (uint32_t k = 0; k < n; k++) reinterpret_cast<uint32_t *>(o)[k] = px[k];
    } else if (n == 4 && ((reinterpret_cast<uintptr_t>(o) & 3u) == 0)) {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
        o32[0] = px[0] | (px[1] << 24);
        o32[1] = (px[1] >> 8) | (px[2] << 16);
        o32[2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        for (uint32_t k = 0; k < n; k++) {
            o[3 * k] = (uint8_t)px[k]; o[3 * k + 1] = (uint8_t)(px[k] >> 8); o[3 * k + 2] = (uint8_t)(px[k] >> 16);
        }
    }
}


// ---- fast path: 8 pixels per lane on aligned dwords -----------------------------------------------------
// aligned dword of a plane row at column `col`, clamped into the row (clamped values are never used: they feed only
// the first / last column, which the reference computes without neighbours, or pixels past the image)
__device__ __forceinline__ uint32_t row_dword(const JP_GLOBAL uint8_t *row, int32_t col, uint32_t stride) {
    col = min(max(col, 0), (int32_t)stride - 4);
    return *reinterpret_cast<const JP_GLOBAL uint32_t *>(row + col);
}
// source samples s[-1..4] around j0 = x0/2 (s[0] = column j0-1)
__device__ __forceinline__ void fetch6(const JP_GLOBAL uint8_t *row, uint32_t j0, uint32_t stride, uint32_t (&s)[6]) {
    const uint32_t a = row_dword(row, (int32_t)j0 - 4, stride), b = row_dword(row, (int32_t)j0, stride),
                   c = row_dword(row, (int32_t)j0 + 4, stride);
    s[0] = a >> 24;
    s[1] = b & 0xffu;
    s[2] = (b >> 8) & 0xffu;
    s[3] = (b >> 16) & 0xffu;
    s[4] = b >> 24;
    s[5] = c & 0xffu;
}
__device__ __forceinline__ void fetch8(const JP_GLOBAL uint8_t *row, uint32_t x0, uint32_t (&s)[8]) {
    const v2u d = *reinterpret_cast<const JP_GLOBAL v2u *>(row + x0);  // x0 % 8 == 0, stride % 8 == 0
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) s[k] = ((k < 4 ? d.x : d.y) >> (8u * (k & 3u))) & 0xffu;
}

__device__ __forceinline__ void up_sample8(const UpComp &u, uint32_t x0, uint32_t row, uint32_t (&out)[8]) {
    const JP_GLOBAL uint8_t *p = (const JP_GLOBAL uint8_t *)u.plane;
    const uint32_t W = u.width;
    if (u.kind == UP_H1V1) {  // src/upsampler.rs:119-132
        fetch8(p + (size_t)row * u.stride, x0, out);
    } else if (u.kind == UP_H1V2) {  // :165-189
        uint32_t near, far, n[8], f[8];
        near_far(row, u.height, near, far);
        fetch8(p + (size_t)near * u.stride, x0, n);
        fetch8(p + (size_t)far * u.stride, x0, f);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) out[k] = (3u * n[k] + f[k] + 2u) >> 2;
    } else if (u.kind == UP_H2V1) {  // :134-163
        uint32_t s[6];
        fetch6(p + (size_t)row * u.stride, x0 >> 1, u.stride, s);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t x = x0 + k, m = s[1u + (k >> 1)], o = (k & 1u) ? s[2u + (k >> 1)] : s[k >> 1];
            out[k] = (x == 0u || x == 2u * W - 1u) ? m : (3u * m + o + 2u) >> 2;
        }
    } else {  // UP_H2V2 :191-228
        uint32_t near, far, n[6], f[6], t[6];
        near_far(row, u.height, near, far);
        fetch6(p + (size_t)near * u.stride, x0 >> 1, u.stride, n);
        fetch6(p + (size_t)far * u.stride, x0 >> 1, u.stride, f);
#pragma unroll
        for (uint32_t m = 0; m < 6; m++) t[m] = 3u * n[m] + f[m];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t x = x0 + k, tm = t[1u + (k >> 1)], to = (k & 1u) ? t[2u + (k >> 1)] : t[k >> 1];
            out[k] = (x == 0u || x == 2u * W - 1u) ? (tm + 2u) >> 2 : (3u * tm + to + 8u) >> 4;
        }
    }
}

#ifdef JPGPU_HOST_EMULATION
typedef v4u v4u_a4;
#else
typedef v4u v4u_a4 __attribute__((aligned(4)));
#endif

// interleaved colour functions only (RGB / YCbCr / CMYK / YCCK); x0 % 8 == 0
__device__ __forceinline__ void upsample_color_fast8(const ImageJob &job, uint32_t x0, uint32_t row) {
    if (row >= job.out_h || x0 >= job.out_w) return;
    const uint32_t nc = job.ncomp, n = min(8u, job.out_w - x0);
    uint32_t s[4][8];
#pragma unroll
    for (uint32_t c = 0; c < 4; c++)
        if (c < nc) up_sample8(job.comp[c], x0, row, s[c]);
    uint32_t px[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        switch (job.color_fn) {
        case CC_RGB: px[k] = s[0][k] | (s[1][k] << 8) | (s[2][k] << 16); break;
        case CC_YCBCR: px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]); break;
        case CC_YCCK: px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]) | ((255u - s[3][k]) << 24); break;
        default: px[k] = (255u - s[0][k]) | ((255u - s[1][k]) << 8) | ((255u - s[2][k]) << 16) | ((255u - s[3][k]) << 24); break;
        }
    }
    JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)job.out;
    const size_t off = ((size_t)row * job.out_w + x0) * nc;
    if (nc == 4) {
        if (n == 8u) {
            *reinterpret_cast<JP_GLOBAL v4u_a4 *>(out + off) = v4u{px[0], px[1], px[2], px[3]};
            *reinterpret_cast<JP_GLOBAL v4u_a4 *>(out + off + 16) = v4u{px[4], px[5], px[6], px[7]};
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
                if (k < n) *reinterpret_cast<JP_GLOBAL uint32_t *>(out + off + 4u * k) = px[k];
        }
    } else {
        store_rgb_run(out, off, px, n);
    }
}

__device__ __forceinline__ void upsample_color_lane(const ImageJob &job, uint32_t x0, uint32_t row) {
    if (job.fast8) {
        upsample_color_fast8(job, x0, row);
    } else {
        upsample_color_body(job, x0, row);
        upsample_color_body(job, x0 + 4u, row);
    }
}

}  // namespace jpgpu
