// huff.hip — device entropy decoding of restart-marker streams (huff_core.hpp) and the range scan that classifies the
// coefficients it produced (the host never sees them).
#include <hip/hip_runtime.h>

#include "huff.hpp"
#include "huff_core.hpp"

namespace jpgpu {

// grid = (ceil(max segments / 64), scan jobs), one wave per workgroup: lanes diverge (every lane walks its own bit
// stream), so small workgroups spread the segments over as many SIMDs as possible
// The scan's job record and Huffman tables are copied to LDS first: every symbol costs dependent table reads.
__global__ __launch_bounds__(64) void huff_segments_kernel(const HuffScanJob *__restrict__ jobs) {
    __shared__ HuffLds L;
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)&jobs[blockIdx.y];
        uint32_t *dst = reinterpret_cast<uint32_t *>(&L.job);
        if (threadIdx.x < sizeof(HuffScanJob) / 4u) dst[threadIdx.x] = src[threadIdx.x];
    }
    huff_fill_unzigzag((JP_LDS uint8_t *)L.unzig, threadIdx.x);
    __syncthreads();
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)L.job.tables;
        uint32_t *dst = reinterpret_cast<uint32_t *>(L.tables);
        for (uint32_t i = threadIdx.x; i < 8u * sizeof(DevHuffTable) / 4u; i += 64u) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t seg = blockIdx.x * 64u + threadIdx.x;
    if (seg >= L.job.n_seg) return;
    huff_decode_segment(*(JP_LDS HuffLds *)&L, seg);
}

// one lane per block: max |c*q| and the largest block-column sum of |c*q| (the two quantities behind the range classes of
// include/jpgpu.h), reduced per wave and merged with atomicMax
__global__ __launch_bounds__(256) void range_scan_kernel(const RangeJob *__restrict__ jobs, uint32_t *__restrict__ stats) {
    const RangeJob &job = jobs[blockIdx.y];
    const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
    uint32_t max_abs = 0, max_col = 0;
    if (blk < job.n_blocks) {
        const JP_GLOBAL v4u *p = (const JP_GLOBAL v4u *)(job.coefs + (size_t)blk * 64u);
        uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const v4u v = p[r];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const int32_t c = (int16_t)(uint16_t)(w[k >> 1] >> (16u * (k & 1u)));
                const uint32_t a = (uint32_t)(c < 0 ? -c : c) * (uint32_t)job.q[r * 8u + k];  // <= 32768 * 65535 < 2^31
                max_abs = a > max_abs ? a : max_abs;
                col[k] += a < 0x00ffffffu ? a : 0x00ffffffu;  // saturate the addends: 8 of them cannot wrap
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) max_col = col[k] > max_col ? col[k] : max_col;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_abs = max(max_abs, (uint32_t)__shfl_xor((int)max_abs, off));
        max_col = max(max_col, (uint32_t)__shfl_xor((int)max_col, off));
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicMax(&stats[2u * job.slot], max_abs);
        atomicMax(&stats[2u * job.slot + 1u], max_col);
    }
}

hipError_t launch_huff_segments(const HuffScanJob *d_jobs, uint32_t n_jobs, uint32_t max_segments, hipStream_t stream) {
    if (n_jobs == 0 || max_segments == 0) return hipSuccess;
    huff_segments_kernel<<<dim3((max_segments + 63u) / 64u, n_jobs), dim3(64), 0, stream>>>(d_jobs);
    return hipGetLastError();
}

hipError_t launch_range_scan(const RangeJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t *d_stats, hipStream_t stream) {
    if (n_jobs == 0 || max_blocks == 0) return hipSuccess;
    range_scan_kernel<<<dim3((max_blocks + 255u) / 256u, n_jobs), dim3(256), 0, stream>>>(d_jobs, d_stats);
    return hipGetLastError();
}

}  // namespace jpgpu
