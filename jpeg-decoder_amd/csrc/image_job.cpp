// image_job.cpp — host-only (no HIP): error strings and the ImageJob builder of the generic path, i.e. Upsampler::new +
// choose_upsampler + choose_color_convert_func of the reference.  Separate translation unit so that tests/emu can
// link the same planner into the CPU emulation of the kernels.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "host_common.hpp"

#include "compact.hpp"

namespace jpgpu {

int set_err(std::string &dst, int code, const char *fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    dst = buf;
    return code;
}

size_t plane_bytes(const jpgpu_component &c) {
    return (size_t)c.block_width * c.block_height * c.dct_scale * c.dct_scale;
}

// choose_color_convert_func, src/decoder.rs:1339-1389
int choose_color_fn(uint32_t ncomp, int ct, uint32_t &fn, std::string &err) {
    if (ncomp == 1) {
        fn = CC_GRAY;
        return JPGPU_OK;
    }
    if (ncomp != 3 && ncomp != 4) return set_err(err, JPGPU_ERR_INTERNAL, "reference would panic: %u components", ncomp);
    switch (ct) {
    case JPGPU_CT_NONE: fn = CC_NONE; return JPGPU_OK;
    case JPGPU_CT_GRAYSCALE: return set_err(err, JPGPU_ERR_FORMAT, "Invalid number of channels (%u) for Grayscale data", ncomp);
    case JPGPU_CT_RGB:
        if (ncomp == 3) { fn = CC_RGB; return JPGPU_OK; }
        return set_err(err, JPGPU_ERR_FORMAT, "Invalid number of channels (4) for RGB data");
    case JPGPU_CT_YCBCR:
        if (ncomp == 3) { fn = CC_YCBCR; return JPGPU_OK; }
        return set_err(err, JPGPU_ERR_FORMAT, "Invalid number of channels (4) for YCbCr data");
    case JPGPU_CT_CMYK:
        if (ncomp == 4) { fn = CC_CMYK; return JPGPU_OK; }
        return set_err(err, JPGPU_ERR_FORMAT, "Invalid number of channels (3) for CMYK data");
    case JPGPU_CT_YCCK:
        if (ncomp == 4) { fn = CC_YCCK; return JPGPU_OK; }
        return set_err(err, JPGPU_ERR_FORMAT, "Invalid number of channels (3) for YCCK data");
    case JPGPU_CT_JCS_BG_YCC:
    case JPGPU_CT_JCS_BG_RGB: return set_err(err, JPGPU_ERR_UNSUPPORTED, "ColorTransform(%d)", ct);
    default: return set_err(err, JPGPU_ERR_FORMAT, "Unknown colour transform");
    }
}

// Upsampler::new + choose_upsampler (src/upsampler.rs:20-45,76-105) and the bounds the
// reference's slice indexing would enforce with a panic (checked for the last output row,
// which maximises every index).
int build_image_job(const jpgpu_component *comps, uint32_t ncomp, uint8_t *const *d_planes, uint16_t out_w,
                    uint16_t out_h, int color_transform, uint8_t *d_out, ImageJob &job, size_t &out_len,
                    std::string &err) {
    memset(&job, 0, sizeof(job));
    if (ncomp == 0 || ncomp > 4) return set_err(err, JPGPU_ERR_FORMAT, "not all components have data");
    uint32_t fn;
    int rc = choose_color_fn(ncomp, color_transform, fn, err);
    if (rc) return rc;
    job.ncomp = ncomp;
    job.out_w = out_w;
    job.out_h = out_h;
    job.out = d_out;
    job.color_fn = fn;
    uint32_t h_max = 0, v_max = 0, max_w = 0;
    for (uint32_t i = 0; i < ncomp; i++) {
        const jpgpu_component &c = comps[i];
        if (c.horizontal_sampling_factor == 0 || c.vertical_sampling_factor == 0 ||
            !(c.dct_scale == 8 || c.dct_scale == 4 || c.dct_scale == 2 || c.dct_scale == 1))
            return set_err(err, JPGPU_ERR_FORMAT, "invalid component %u", i);
        h_max = std::max<uint32_t>(h_max, c.horizontal_sampling_factor);
        v_max = std::max<uint32_t>(v_max, c.vertical_sampling_factor);
        max_w = std::max<uint32_t>(max_w, c.size_width);
    }
    for (uint32_t i = 0; i < ncomp; i++) {
        const jpgpu_component &c = comps[i];
        UpComp &u = job.comp[i];
        u.plane = d_planes[i];
        u.width = c.size_width;
        u.height = c.size_height;
        u.stride = (uint32_t)c.block_width * c.dct_scale;
        u.hf = u.vf = 1;
        if (fn == CC_GRAY) {
            u.kind = UP_H1V1;
            continue;
        }
        uint32_t h = c.horizontal_sampling_factor, v = c.vertical_sampling_factor;
        bool h1 = h == h_max || out_w == 1, v1 = v == v_max || out_h == 1;
        bool h2 = h * 2 == h_max, v2 = v * 2 == v_max;
        if (h1 && v1) u.kind = UP_H1V1;
        else if (h2 && v1) u.kind = UP_H2V1;
        else if (h1 && v2) u.kind = UP_H1V2;
        else if (h2 && v2) u.kind = UP_H2V2;
        else if (h_max % h != 0 || v_max % v != 0) return set_err(err, JPGPU_ERR_UNSUPPORTED, "NonIntegerSubsamplingRatio");
        else {
            u.kind = UP_GENERIC;
            u.hf = h_max / h;
            u.vf = v_max / v;
        }
    }
    if (fn == CC_GRAY) {
        out_len = (size_t)comps[0].size_width * comps[0].size_height;
        size_t need = (size_t)(comps[0].size_height ? comps[0].size_height - 1 : 0) * job.comp[0].stride + comps[0].size_width;
        if (need > plane_bytes(comps[0])) return set_err(err, JPGPU_ERR_INTERNAL, "reference would panic: plane smaller than the image");
        return JPGPU_OK;
    }
    out_len = (size_t)out_w * out_h * ncomp;
    const size_t line_buffer_size = (size_t)max_w * h_max;
    if (out_w > line_buffer_size) return set_err(err, JPGPU_ERR_INTERNAL, "line buffer shorter than the output row");
    if (fn == CC_NONE && (size_t)ncomp * line_buffer_size > (size_t)out_w * ncomp)
        return set_err(err, JPGPU_ERR_INTERNAL, "reference would panic: color_no_convert overruns the row");
    if (out_h == 0 || out_w == 0) return JPGPU_OK;
    // dword path of the kernel (upsample_color_body.hpp): interleaved output, every plane under one of the four fixed
    // upsamplers, 8-byte aligned, and its rows as well where a lane reads eight samples at once (H1V1 / H1V2: stride % 8 == 0)
    // or aligned dwords around column x / 2 (H2V1 / H2V2: stride % 4 == 0).  Planes at dct_scale 8 always qualify; reduced-size
    // planes (Decoder::scale, stride = block_width * dct_scale) when their block count makes the stride so — 1080p does at
    // every scale (round 3: such frames ran the byte-granular path, 1.15 of the 1.5 ms of a 256 x 1080p decode at scale 4).
    job.fast8 = (fn == CC_RGB || fn == CC_YCBCR || fn == CC_CMYK || fn == CC_YCCK) ? 1u : 0u;
    for (uint32_t i = 0; i < ncomp; i++) {
        const UpComp &u = job.comp[i];
        const uint32_t align = (u.kind == UP_H1V1 || u.kind == UP_H1V2) ? 8u : 4u;
        if (u.kind > UP_H2V2 || ((uintptr_t)u.plane & 7u) || (u.stride % align) != 0u || u.stride < 8u) job.fast8 = 0u;
    }
    for (uint32_t i = 0; i < ncomp; i++) {
        const UpComp &u = job.comp[i];
        const size_t len = plane_bytes(comps[i]);
        const size_t row = (size_t)out_h - 1;
        size_t need = 0;
        switch (u.kind) {
        case UP_H1V1: need = row * u.stride + out_w; break;
        case UP_H2V1:
            need = row * u.stride + u.width;
            if (2 * (size_t)u.width > line_buffer_size) need = (size_t)-1;
            break;
        case UP_H1V2:
        case UP_H2V2: {
            if (u.height == 0) return set_err(err, JPGPU_ERR_INTERNAL, "reference would panic: empty component");
            size_t near = row >> 1;
            size_t far = (row & 1) ? std::min<size_t>(near + 1, u.height - 1) : (near ? near - 1 : 0);
            size_t w = u.kind == UP_H1V2 ? out_w : u.width;
            need = std::max(near, far) * u.stride + w;
            break;
        }
        default: need = (row / u.vf) * u.stride + u.width; break;
        }
        if (need > len) return set_err(err, JPGPU_ERR_INTERNAL, "reference would panic: upsample out of bounds (component %u)", i);
    }
    return JPGPU_OK;
}

}  // namespace jpgpu

// ---- compact coefficient transport: host encoder (include/jpgpu.h) ---------------------------------------------
extern "C" {

// range scan (part of H2D staging): per-position |c|*q must stay below 2^15 for the 24-bit / packed paths to be
// exact, and block-column sums below 5900 for the dot2 row pass (pixel_math.hpp idct8x8<ARITH>)
int jpgpu_range_class(const int16_t *coefficients, size_t len, const uint16_t q[64]) {
    if (!coefficients || !q) return 0;
    int32_t qq[64];
    for (int k = 0; k < 64; k++) qq[k] = q[k];
    int32_t max_abs = 0, max_col = 0;
    const size_t nblk = len / 64;
    for (size_t blk = 0; blk < nblk; blk++) {
        const int16_t *p = coefficients + blk * 64;
        int64_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (eight products of up to 2^31 each: 64-bit sums)
        for (int k = 0; k < 64; k++) {
            int32_t v = (int32_t)p[k] * qq[k];  // |i16 x u16| <= 2^31 - 2^15: fits
            v = v < 0 ? -v : v;
            max_abs = v > max_abs ? v : max_abs;
            col[k & 7] += v;
        }
        for (int i = 0; i < 8; i++) max_col = col[i] > max_col ? (int32_t)std::min<int64_t>(col[i], INT32_MAX) : max_col;
    }
    if (max_abs < (1 << 15)) return (max_col <= 5900) ? 3 : 1;
    return 0;
}

size_t jpgpu_compact_max_bytes(size_t n_blocks) { return jpgpu::compact_max_bytes(n_blocks); }

size_t jpgpu_compact_encode(const int16_t *coefficients, size_t n_blocks, const uint16_t q[64], void *dst, int *range_class) {
    if (!coefficients || !dst) return 0;
    jpgpu::CompactWriter w(dst, n_blocks, q);
    w.add_blocks(coefficients, n_blocks);
    return w.finish(range_class);
}

}  // extern "C"
