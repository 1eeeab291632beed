// huff_job.hpp — job descriptors of the device entropy decoder (huff_core.hpp); no HIP dependency: the host front-end
// fills them (csrc/host/frontend.cpp, plan_device_scans).
#pragma once
#include <stdint.h>

namespace jpgpu {

constexpr int HUFF_LUT_BITS = 10;  // == kLutBits of the host front-end

struct DevHuffTable {
    uint16_t lut[1 << HUFF_LUT_BITS];  // per prefix: symbol | code length << 8 (length 0: not resolved within the lookahead)
    int32_t maxcode[16], delta[16];
    uint8_t values[256];
    int32_t nvalues;
};

struct HuffScanComp {
    int16_t *dst;       // the component's coefficient plane in the arena (zero-filled before the launch)
    uint32_t block_w;   // blocks per plane row
    uint32_t h, v;      // blocks per MCU (1, 1 in a single-component scan)
    uint32_t dc, ac;    // table slots: dc in tables[0..3], ac in tables[4..7]
};

struct HuffScanJob {            // one scan of one image
    const uint8_t *data;        // staged segments (huff_stage_segment): unstuffed, each in a 16-byte aligned slot, zero padded
    const uint32_t *seg_off;    // 2 * n_seg words: segment s starts at data + seg_off[2s] and has seg_off[2s+1] unstuffed bytes
    const DevHuffTable *tables; // 8 tables of this scan
    uint32_t *status;           // the image's status word: bit 0 set = decode this image on the host instead
    uint32_t n_seg, ri;         // restart interval in MCUs
    uint32_t cols, n_mcu;       // MCUs per row / in the scan
    uint32_t ncomp, _pad;
    HuffScanComp comp[4];
};

// Host side of the staging: copy one restart segment (markers excluded, 0xFF00 pairs inside) without its stuffing
// zeros, then zero bytes up to the next 16-byte boundary plus 16 (the device reader fetches aligned 16-byte chunks ahead
// and treats what follows a segment as zero bits).  Returns the unstuffed length.  Slot size: huff_slot_bytes(n).
inline uint32_t huff_slot_bytes(uint32_t stuffed_bytes) { return ((stuffed_bytes + 15u) & ~15u) + 32u; }
inline uint32_t huff_stage_segment(uint8_t *dst, const uint8_t *src, uint32_t n) {
    uint32_t o = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t v = src[i];
        dst[o++] = v;
        if (v == 0xFF && i + 1 < n && src[i + 1] == 0) i++;
    }
    const uint32_t slot = huff_slot_bytes(n);
    for (uint32_t z = o; z < slot; z++) dst[z] = 0;
    return o;
}

}  // namespace jpgpu
