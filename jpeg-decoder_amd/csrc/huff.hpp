// huff.hpp — launch interface of huff.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "huff_job.hpp"

namespace jpgpu {

struct RangeJob {  // one component plane to classify
    const int16_t *coefs;
    uint32_t n_blocks;
    uint32_t slot;    // stats[2*slot] = max |c*q|, stats[2*slot+1] = max block-column sum of |c*q|
    uint16_t q[64];
};

hipError_t launch_huff_segments(const HuffSyncJob *d_jobs, uint32_t n_jobs, uint32_t max_segments, hipStream_t stream);
hipError_t launch_huff_sync(const HuffSyncJob *d_jobs, uint32_t n_jobs, uint32_t max_chunks, uint32_t launches, uint32_t iters, hipStream_t stream);
hipError_t launch_range_scan(const RangeJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t *d_stats, hipStream_t stream);

}  // namespace jpgpu
