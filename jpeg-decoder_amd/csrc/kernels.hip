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
#include "compact.hpp"
#include "pixel_math.hpp"
#include "idct_plane_body.hpp"
#include "upsample_color_body.hpp"

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

// Upsample + colour convert: upsample_color_body.hpp (one lane per 8 consecutive output pixels of one row)
__global__ __launch_bounds__(256) void upsample_color_kernel(const ImageJob *__restrict__ jobs) {
    const ImageJob &job = jobs[blockIdx.z];
    upsample_color_lane(job, (blockIdx.x * 256u + threadIdx.x) * 8u, blockIdx.y);
}
__global__ __launch_bounds__(256) void upsample_color_one_kernel(ImageJob job) {
    upsample_color_lane(job, (blockIdx.x * 256u + threadIdx.x) * 8u, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// Compact transport -> dense coefficient arena (compact.hpp).  Eight lanes per block, one per row of eight
// coefficients: a lane finds its values at index[block] + popcount(bitmap bits below its row) and writes its 16-B row,
// so a wave writes 1 KiB of consecutive arena bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_compact_kernel(const ExpandJob *__restrict__ jobs) {
    const ExpandJob job = jobs[blockIdx.y];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, b = t >> 3, r = t & 7u;
    if (b >= job.n_blocks) return;
    const JP_GLOBAL uint64_t *bitmaps = (const JP_GLOBAL uint64_t *)job.compact;
    const JP_GLOBAL uint32_t *first = (const JP_GLOBAL uint32_t *)(job.compact + (size_t)job.n_blocks * 8u);
    const JP_GLOBAL int16_t *values = (const JP_GLOBAL int16_t *)(job.compact + (size_t)job.n_blocks * 12u);
    const uint64_t bm = bitmaps[b];
    const uint32_t bits = (uint32_t)(bm >> (8u * r)) & 0xffu;
    uint32_t idx = first[b] + (uint32_t)__popcll(bm & ((1ull << (8u * r)) - 1ull));
    uint32_t v[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        v[k] = 0u;
        if (bits & (1u << k)) v[k] = (uint16_t)values[idx++];
    }
    const v4u row = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    *reinterpret_cast<JP_GLOBAL v4u *>((JP_GLOBAL uint8_t *)job.dense + (size_t)b * 128u + r * 16u) = row;
}

// Progressive accumulation on the device: coefficient[index] += delta for the changes one scan made to
// one component plane.  One lane per entry; a scan touches a coefficient at most once (host front-end, RowSink::scan_deltas),
// so no atomics — launches of consecutive scans are ordered by their stream.  i16 wrapping add: the sum of all deltas is the
// coefficient the host accumulated, which fits.
__global__ __launch_bounds__(256) void delta_add_kernel(const jpgpu_coef_delta *__restrict__ d, uint32_t n, int16_t *__restrict__ plane,
                                                        uint32_t plane_coefficients) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const jpgpu_coef_delta e = d[i];
    if (e.index >= plane_coefficients) return;  // (checked on the host as well)
    plane[e.index] = (int16_t)(uint16_t)((uint32_t)(uint16_t)plane[e.index] + (uint32_t)e.delta);
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_delta_add(const jpgpu_coef_delta *d_entries, uint32_t n, int16_t *d_plane, uint32_t plane_coefficients, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    delta_add_kernel<<<dim3((n + 255u) / 256u), dim3(256), 0, stream>>>(d_entries, n, d_plane, plane_coefficients);
    return hipGetLastError();
}

hipError_t launch_expand_compact(const ExpandJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, hipStream_t stream) {
    if (n_jobs == 0 || max_blocks == 0) return hipSuccess;
    dim3 grid((max_blocks * 8u + 255u) / 256u, n_jobs), block(256);
    expand_compact_kernel<<<grid, block, 0, stream>>>(d_jobs);
    return hipGetLastError();
}

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
    dim3 grid(((max_w + 7u) / 8u + 255u) / 256u, max_h, n_jobs), block(256);
    upsample_color_kernel<<<grid, block, 0, stream>>>(d_jobs);
    return hipGetLastError();
}

hipError_t launch_upsample_color_one(const ImageJob &job, hipStream_t stream) {
    uint32_t w = job.color_fn == CC_GRAY ? job.comp[0].width : job.out_w;
    uint32_t h = job.color_fn == CC_GRAY ? job.comp[0].height : job.out_h;
    if (w == 0 || h == 0) return hipSuccess;
    dim3 grid(((w + 7u) / 8u + 255u) / 256u, h), block(256);
    upsample_color_one_kernel<<<grid, block, 0, stream>>>(job);
    return hipGetLastError();
}

}  // namespace jpgpu
