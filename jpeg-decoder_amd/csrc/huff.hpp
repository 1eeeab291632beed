// huff.hpp — launch interface of huff.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "huff_job.hpp"
#include "huff_prog_job.hpp"
#include "huff_unstuff_core.hpp"

namespace jpgpu {

struct RangeJob {  // one component plane to classify
    const int16_t *coefs;
    uint32_t n_blocks;
    uint32_t slot;    // stats[RS_WORDS*slot + RS_MAX_AC] = max |c*q|, [.. + RS_MAX_COL] = max block-column sum of |c*q| (range_stats.hpp);
                      // several jobs may share a slot (the planes of one image)
    uint16_t q[64];
};

// Entry lists read by the pixel kernel itself (fused_entries.hpp): instead of huff_expand_kernel, per (MCU row, strip) of the image's
// 4:2:0 walk the chunk and the entry at which the run of that row's MCUs starts.
struct EntryIndexJob {
    uint32_t job;             // which HuffSyncJob of the launch
    uint32_t tx, tiles_x, rows;  // strip width in MCUs, strips per MCU row, MCU rows
    uint32_t *tab;            // rows x tiles_x x {chunk, entry}
};
// Sync passes (with speculative emission) + block numbering | expansion of the entry lists + DC sums of `uniform` scans.
// after_sync (optional): recorded between the two (phase timing)
// low_table_ids: every job's components use Huffman table ids 0 and 1 only — the sync passes run with four table slots in LDS
// d_index (n_index jobs, at most max_index_items (row, strip) pairs each): those jobs' lists stay lists (HuffSyncJob::keep_lists:
// huff_expand_kernel leaves them alone) and get their strip index instead
hipError_t launch_huff_sync(const HuffSyncJob *d_jobs, uint32_t n_jobs, uint32_t max_chunks, uint32_t launches, uint32_t iters, hipStream_t stream,
                            hipEvent_t after_sync = nullptr, bool low_table_ids = false, const EntryIndexJob *d_index = nullptr, uint32_t n_index = 0,
                            uint32_t max_index_items = 0);
// progressive frames: one wave per scan, coefficients accumulated in the arena (huff_prog_wave.hpp)
hipError_t launch_huff_progw(const ProgTrack *d_tracks, uint32_t n_tracks, hipStream_t stream);  // round 6: a wave per scan
// n words from device memory into pinned host memory (dst: the DEVICE address of a hipHostMalloc'ed block), by a kernel
hipError_t launch_copy_words_to_host(uint32_t *dst_host_mapped, const uint32_t *d_src, uint32_t n, hipStream_t stream);
// `bytes` from device memory into pinned host memory (dst: the DEVICE address of a hipHostMalloc'ed block, 16-byte aligned), by a kernel
hipError_t launch_copy_to_host(void *dst_host_mapped, const void *d_src, size_t bytes, hipStream_t stream);
// "host light": the staging pass of scans that went up as the file holds them — marker check, unstuffing, the job records' lengths — in front of launch_huff_sync
hipError_t launch_huff_unstuff(const UnstuffJob *d_jobs, uint32_t n_jobs, uint32_t max_pieces, hipStream_t stream);
hipError_t launch_copy_from_host(void *d_dst, const void *src_host_mapped, size_t bytes, hipStream_t stream);  // bytes rounded up to 16
hipError_t launch_range_scan(const RangeJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t *d_stats, hipStream_t stream);
// one plane whose quantization table sits in device memory; raises the RS_WORDS statistics words at d_stats
hipError_t launch_range_scan_one(const int16_t *d_coefs, uint32_t n_blocks, const uint16_t *d_q, uint32_t *d_stats, hipStream_t stream);

}  // namespace jpgpu
