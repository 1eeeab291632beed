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
#include "fused_scaled.hpp"
#include "range_stats.hpp"

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

// Upsample + colour convert: upsample_color_body.hpp (one lane per 8 consecutive output pixels of one row).  The lanes of a
// launch walk the (row, chunk) pairs of an image in row-major order, `cpr` chunks per row: a workgroup spans several rows of a
// narrow image (round 3: one workgroup per row left 136 of 256 lanes idle on the 960-pixel rows of a 1080p decode at scale 4).
__global__ __launch_bounds__(256) void upsample_color_kernel(const ImageJob *__restrict__ jobs, uint32_t cpr, uint32_t rows) {
    const ImageJob &job = jobs[blockIdx.z];
    const uint32_t l = blockIdx.x * 256u + threadIdx.x, row = l / cpr;
    if (row < rows) upsample_color_lane(job, (l - row * cpr) * 8u, row);
}
__global__ __launch_bounds__(256) void upsample_color_one_kernel(ImageJob job, uint32_t cpr, uint32_t rows) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x, row = l / cpr;
    if (row < rows) upsample_color_lane(job, (l - row * cpr) * 8u, row);
}

// Reduced-size decodes in one launch (fused_scaled.hpp).  A 1-D grid, numbered for the XCDs: workgroups are handed to the 8 XCDs
// round-robin in launch order and every XCD has an L2 of its own, so the workgroups that read the same coefficients — a tile and
// the tiles above and below it, whose rings of neighbour blocks overlap it — must be EIGHT apart in launch order to meet in one
// L2.  Column s = (image, tile across) goes to XCD s mod 8, its MCU rows in consecutive slots of that XCD:
//     id = 8 * ((s / 8) * rows + y) + s % 8.
// (First version: grid (tiles, rows, images), two tiles across a 1080p image: vertical neighbours on different XCDs, every ring
// row fetched again — 3.37 GB of L2 misses for 2.00 GB algorithmic, 0.90 ms per 256 x 1080p at scale 4.)  A workgroup beyond its
// own image's grid leaves at once.
template <int SCALE>
__global__ __launch_bounds__(FS_NT) void scaled_fused_kernel(const ScaledGeom *__restrict__ geoms, const ImageJob *__restrict__ jobs,
                                                             const PlaneJob *__restrict__ planes, uint32_t max_tiles_x, uint32_t max_bands, uint32_t n_images) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, col = (slot / max_bands) * 8u + xcd, band = slot % max_bands;
    const uint32_t image = col / max_tiles_x, tile = col - image * max_tiles_x;
    if (image >= n_images) return;
    const ScaledGeom &g = geoms[image];
    if (g.scale != (uint32_t)SCALE || tile >= g.tiles_x || band >= g.bands) return;  // (uniform)
    typedef FScaled<SCALE> K;
    K::transform(g, planes + g.first_plane_job, tile, band, threadIdx.x, lds_raw);
    __syncthreads();
    K::pixels(g, jobs[image], tile, band, threadIdx.x, lds_raw);
}

// ------------------------------------------------------------------------------------------
// Compact transport -> dense coefficient arena (compact.hpp).  Eight lanes per block, one per row of eight
// coefficients: a lane finds its values at index[block] + popcount(bitmap bits below its row) and writes its 16-B row,
// so a wave writes 1 KiB of consecutive arena bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_compact_kernel(const ExpandJob *__restrict__ jobs) {
    const ExpandJob job = jobs[blockIdx.y];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, b = t >> 3, r = t & 7u;
    if (blockIdx.x * 32u >= job.n_blocks) return;  // (whole workgroup beyond the plane)
    const bool valid = b < job.n_blocks;            // (lanes beyond it stay for the wave reduction of the statistics)
    uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        const JP_GLOBAL uint64_t *bitmaps = (const JP_GLOBAL uint64_t *)job.compact;
        const JP_GLOBAL uint32_t *first = (const JP_GLOBAL uint32_t *)(job.compact + (size_t)job.n_blocks * 8u);
        const JP_GLOBAL int16_t *values = (const JP_GLOBAL int16_t *)(job.compact + (size_t)job.n_blocks * 12u);
        const uint64_t bm = bitmaps[b];
        const uint32_t bits = (uint32_t)(bm >> (8u * r)) & 0xffu;
        uint32_t idx = first[b] + (uint32_t)__popcll(bm & ((1ull << (8u * r)) - 1ull));
#pragma unroll
        for (uint32_t k = 0; k < 8; k++)
            if (bits & (1u << k)) v[k] = (uint16_t)values[idx++];
        const v4u row = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
        *reinterpret_cast<JP_GLOBAL v4u *>((JP_GLOBAL uint8_t *)job.dense + (size_t)b * 128u + r * 16u) = row;
    }
    if (job.stats) {  // (uniform per job) nobody classified these coefficients: their range, while they are in registers
        const v4u qv = ((const JP_GLOBAL v4u *)job.qt)[r];
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
        uint32_t max_dc = 0, max_ac = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const int32_t c = (int16_t)(uint16_t)v[k];
            const uint32_t p = (uint32_t)(c < 0 ? -c : c) * ((qw[k >> 1] >> (16u * (k & 1u))) & 0xffffu);
            if (k == 0u && r == 0u) max_dc = p;
            else max_ac = max(max_ac, p);
        }
        stat_publish_wave(job.stats, max_dc, max_ac);
    }
}

// Device-side classes for the generic path: a plane job's class bits from the device statistics of its image (kept per image)
// or from the class the host knows (fused kernels: class_finalize_fused_kernel, fused.hip).
__global__ __launch_bounds__(256) void class_finalize_planes_kernel(PlaneJob *__restrict__ jobs, const uint32_t *__restrict__ slot, uint32_t n,
                                                                    const uint32_t *__restrict__ stats, const uint8_t *__restrict__ host_cls) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint32_t sl = slot[j], h = host_cls[sl];
    const uint32_t *st = stats + (size_t)(sl >> 2) * RS_WORDS;
    const uint32_t cls = h == CLS_FROM_DEVICE ? range_class_from_stats(st[RS_MAX_DC], st[RS_MAX_AC], st[RS_MAX_COL], st[RS_COL_EXACT]) : h;
    jobs[j].flags = (cls & 1u) ? (cls & 3u) : 0u;
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_class_finalize_planes(PlaneJob *d_jobs, const uint32_t *d_slot, uint32_t n_jobs, const uint32_t *d_stats, const uint8_t *d_host_cls,
                                        hipStream_t stream) {
    if (n_jobs == 0 || !d_jobs || !d_slot || !d_stats || !d_host_cls) return hipSuccess;
    class_finalize_planes_kernel<<<dim3((n_jobs + 255u) / 256u), dim3(256), 0, stream>>>(d_jobs, d_slot, n_jobs, d_stats, d_host_cls);
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
    const uint32_t cpr = (max_w + 7u) / 8u;
    const uint64_t lanes = (uint64_t)cpr * max_h;  // <= 8192 * 65535
    dim3 grid((uint32_t)((lanes + 255u) / 256u), 1, n_jobs), block(256);
    upsample_color_kernel<<<grid, block, 0, stream>>>(d_jobs, cpr, max_h);
    return hipGetLastError();
}

hipError_t launch_scaled_fused(const ScaledGeom *d_geoms, const ImageJob *d_jobs, const PlaneJob *d_planes, uint32_t n_images, uint32_t max_tiles_x,
                               uint32_t max_bands, uint32_t lds_bytes, const bool (&scales)[9], hipStream_t stream) {
    if (n_images == 0 || max_tiles_x == 0 || max_bands == 0) return hipSuccess;
    const uint64_t cols = (uint64_t)n_images * max_tiles_x, wgs = ((cols + 7u) / 8u) * 8u * max_bands;
    if (wgs > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)wgs), block(FS_NT);
    if (scales[4]) scaled_fused_kernel<4><<<grid, block, lds_bytes, stream>>>(d_geoms, d_jobs, d_planes, max_tiles_x, max_bands, n_images);
    if (scales[2]) scaled_fused_kernel<2><<<grid, block, lds_bytes, stream>>>(d_geoms, d_jobs, d_planes, max_tiles_x, max_bands, n_images);
    if (scales[1]) scaled_fused_kernel<1><<<grid, block, lds_bytes, stream>>>(d_geoms, d_jobs, d_planes, max_tiles_x, max_bands, n_images);
    return hipGetLastError();
}

hipError_t launch_upsample_color_one(const ImageJob &job, hipStream_t stream) {
    uint32_t w = job.color_fn == CC_GRAY ? job.comp[0].width : job.out_w;
    uint32_t h = job.color_fn == CC_GRAY ? job.comp[0].height : job.out_h;
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t cpr = (w + 7u) / 8u;
    dim3 grid((uint32_t)(((uint64_t)cpr * h + 255u) / 256u)), block(256);
    upsample_color_one_kernel<<<grid, block, 0, stream>>>(job, cpr, h);
    return hipGetLastError();
}

}  // namespace jpgpu
