// kernels.hip — generic gfx950 kernels of the pixel pipeline (full-coverage path).
//
//   idct_planes_kernel     SURVEY §8a rows a3/a5/a6: dequantize + IDCT of every block of a
//                          component's coefficient plane into its u8 sample plane
//                          (src/worker/rayon.rs:71-112, src/idct.rs:205-565).
//   upsample_color_kernel  rows a8-a15: per output row upsample every component and colour
//                          convert into interleaved pixels (src/worker/mod.rs:97-128,
//                          src/upsampler.rs:47-250, src/decoder.rs:1300-1484).
//
// These two kernels cover every sampling / colour / scale combination the reference
// supports.  Same-geometry 4:2:0 / 4:4:4 / gray batches take the fused kernels in
// fused.hip instead; this file is the path everything else (and every odd edge) runs on.
#include "kernels.hpp"
#include "pixel_math.hpp"
#include "idct_plane_body.hpp"

namespace jpgpu {

// ------------------------------------------------------------------------------------------
// IDCT: one lane per 8x8 block, 256 blocks per workgroup.
// Coefficients are fetched with fully coalesced 16-B loads (lane j of the workgroup reads
// chunk j of the workgroup's contiguous 32 KiB) and staged in LDS so that each lane can then
// pull its own 128-B block with 8 ds_read_b128.  LDS slot of (block b, row k):
//     b*8 + (k ^ ((b >> 1) & 7))
// which makes both sides conflict-free on gfx950: a ds_write_b128 8-lane group covers one
// block = 8 consecutive 16-B slots; a ds_read_b128 16-lane group ({0-3,12-15,20-27}, ...) hits
// 16 distinct slots mod 16 (MI355X_MICROARCH.md §LDS).
// ------------------------------------------------------------------------------------------
template <int SCALE>
__global__ __launch_bounds__(256) void idct_planes_kernel(const PlaneJob *__restrict__ jobs) {
    __shared__ v4u lds[256 * 8];
    const PlaneJob job = jobs[blockIdx.y];
    if (job.scale != SCALE) return;
    idct_planes_body<SCALE>(job, blockIdx.x, lds);
}

template <int SCALE>
__global__ __launch_bounds__(256) void idct_plane_one_kernel(PlaneJob job) {
    __shared__ v4u lds[256 * 8];
    idct_planes_body<SCALE>(job, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// Upsample + colour convert: one lane per 4 consecutive output pixels of one row.
// ------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void upsample_color_kernel(const ImageJob *__restrict__ jobs) {
    const ImageJob &job = jobs[blockIdx.z];
    upsample_color_body(job, (blockIdx.x * 256u + threadIdx.x) * 4u, blockIdx.y);
}
__global__ __launch_bounds__(256) void upsample_color_one_kernel(ImageJob job) {
    upsample_color_body(job, (blockIdx.x * 256u + threadIdx.x) * 4u, blockIdx.y);
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_idct_planes(const PlaneJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t scale,
                              hipStream_t stream) {
    if (n_jobs == 0 || max_blocks == 0) return hipSuccess;
    dim3 grid((max_blocks + 255u) / 256u, n_jobs), block(256);
    switch (scale) {
    case 8: idct_planes_kernel<8><<<grid, block, 0, stream>>>(d_jobs); break;
    case 4: idct_planes_kernel<4><<<grid, block, 0, stream>>>(d_jobs); break;
    case 2: idct_planes_kernel<2><<<grid, block, 0, stream>>>(d_jobs); break;
    case 1: idct_planes_kernel<1><<<grid, block, 0, stream>>>(d_jobs); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_idct_plane_one(const PlaneJob &job, hipStream_t stream) {
    if (job.n_blocks == 0) return hipSuccess;
    dim3 grid((job.n_blocks + 255u) / 256u), block(256);
    switch (job.scale) {
    case 8: idct_plane_one_kernel<8><<<grid, block, 0, stream>>>(job); break;
    case 4: idct_plane_one_kernel<4><<<grid, block, 0, stream>>>(job); break;
    case 2: idct_plane_one_kernel<2><<<grid, block, 0, stream>>>(job); break;
    case 1: idct_plane_one_kernel<1><<<grid, block, 0, stream>>>(job); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_upsample_color(const ImageJob *d_jobs, uint32_t n_jobs, uint32_t max_w, uint32_t max_h,
                                 hipStream_t stream) {
    if (n_jobs == 0 || max_w == 0 || max_h == 0) return hipSuccess;
    dim3 grid(((max_w + 3u) / 4u + 255u) / 256u, max_h, n_jobs), block(256);
    upsample_color_kernel<<<grid, block, 0, stream>>>(d_jobs);
    return hipGetLastError();
}

hipError_t launch_upsample_color_one(const ImageJob &job, hipStream_t stream) {
    uint32_t w = job.color_fn == CC_GRAY ? job.comp[0].width : job.out_w;
    uint32_t h = job.color_fn == CC_GRAY ? job.comp[0].height : job.out_h;
    if (w == 0 || h == 0) return hipSuccess;
    dim3 grid(((w + 3u) / 4u + 255u) / 256u, h), block(256);
    upsample_color_one_kernel<<<grid, block, 0, stream>>>(job);
    return hipGetLastError();
}

}  // namespace jpgpu
