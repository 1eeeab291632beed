// huff_unstuff_core.hpp — the staging pass of a sequential scan ON THE DEVICE ("host light", include/jpgpu_decoder.h): what
// huff_stage_segment (huff_job.hpp) does on a host thread — copy the scan without its stuffing zeros (0xFF00 -> 0xFF), refuse it if a
// 0xFF inside is followed by anything else (a marker, a fill byte: src/huffman.rs:123-160 would stop there) — for scans that went up as
// the file holds them.  Three kernels per launch (huff.hip): count what each 4-kB piece keeps | prefix sums per scan, which also fill
// in what the staging task used to tell the job record (unstuffed length, bits, chunks) | compaction.  This header: the job record and
// the per-16-bytes rule, shared with tests/emu.
#pragma once
#include <stdint.h>

#include "huff_job.hpp"

namespace jpgpu {

constexpr uint32_t UNSTUFF_PIECE = 4096u;  // bytes of the (16-byte aligned) raw scan one workgroup of 256 lanes looks at

struct UnstuffJob {
    const uint8_t *raw;    // the scan's entropy-coded bytes in device memory, any alignment (the 16-byte pieces around it are readable)
    uint32_t raw_bytes;
    uint32_t n_pieces;     // ceil((raw % 16 + raw_bytes) / UNSTUFF_PIECE)
    uint8_t *dst;          // the scan's slot (16-byte aligned, huff_slot_bytes(raw_bytes) long): the unstuffed bytes go here
    uint32_t *piece_kept;  // n_pieces + 1 words of work space: bytes each piece keeps, then (in place) where its bytes go
    HuffSyncJob *job;      // the scan's job record in DEVICE memory: n_bits, n_chunks, data_dwords are filled in here
    uint32_t *status;      // the image's status word
};

// One aligned 16-byte piece of the raw scan: which of its bytes are kept, and whether it holds a 0xFF that is not followed by its
// stuffing zero.  w[0..3]: the piece (little endian dwords); prev: the byte in front of it (0 if there is none inside the scan);
// next: the byte behind it (0 if the scan ends with this piece — a 0xFF as the scan's LAST byte has no zero behind it: refused, like
// huff_stage_segment); first / last: the bytes [first, last) of the piece belong to the scan.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t
unstuff_piece_flags(const uint32_t w[4], uint32_t prev, uint32_t next, bool has_next, uint32_t first, uint32_t last, bool &bad) {
    uint32_t keep = 0;
    for (uint32_t j = 0; j < 16u; j++) {
        const uint32_t b = (w[j >> 2] >> (8u * (j & 3u))) & 0xffu;
        const uint32_t nb = j < 15u ? (w[(j + 1u) >> 2] >> (8u * ((j + 1u) & 3u))) & 0xffu : next;
        if (j >= first && j < last) {
            const bool stuffing = b == 0u && prev == 0xffu;  // (prev: the raw byte in front, inside the scan)
            if (!stuffing) keep |= 1u << j;
            if (b == 0xffu) {
                const bool follower_in_scan = j + 1u < last || (j == 15u && has_next);
                if (!follower_in_scan || nb != 0u) bad = true;
            }
            prev = b;
        } else {
            prev = 0u;  // (bytes outside the scan are nobody's 0xFF)
        }
    }
    return keep;
}

}  // namespace jpgpu
