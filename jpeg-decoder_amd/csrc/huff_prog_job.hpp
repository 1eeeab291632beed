// huff_prog_job.hpp — job descriptors of the device decoder for PROGRESSIVE frames (huff_prog_wave.hpp); no HIP dependency: the
// host front-end fills them (csrc/host/frontend.cpp, plan_progressive_scans).  SURVEY §8f n3 / BASELINE configs[3]: "multi-scan
// coefficient accumulation on device".
//
// What the reference does (src/decoder.rs:1086-1298): every scan of a progressive frame refines a band of coefficients of one
// component (or the DC coefficient of several) in planes that persist from scan to scan (`coefficients`, :400-412); a refinement scan's
// bits mean what they mean only given which coefficients of a block are non-zero ALREADY (refine_non_zeroes, :1260-1298) — the scans of
// a band are a chain, and a scan is one long dependent walk.  What is independent: different images, and inside an image the bands that
// share no coefficient — the DC coefficients of all components on one side, the AC band of each component on the other.
//
// So: one WAVE per scan (huff_prog_wave.hpp; a track = the scans of one image that transitively share a coefficient, and a scan stays
// behind the scans of its track it depends on, chunk by chunk); the planes live in the batch's coefficient arena from the start
// (zero-filled), every scan works on them in place:
//   * first scans (Ah = 0) store the coefficients they decode (2-byte stores, nothing read),
//   * refinement scans never read a coefficient either: per block they need only WHICH coefficients are non-zero and their signs —
//     two 64-bit masks per block, kept beside the planes — and change a coefficient with a no-return atomic add on its dword
//     (a correction of +-(1 << Al) cannot carry into the neighbouring half: |c| stays below 2^14, checked by the first scans),
//   * DC refinement is an atomic OR of one bit.
#pragma once
#include <stdint.h>
#include <string.h>

namespace jpgpu {

// One Huffman table in the form the progressive walk uses: the reference's own two-step procedure (src/huffman.rs:31-58) — an 8-bit
// lookup, then the maxcode walk from length 9.  912 bytes in memory; a wave keeps what the six bits at the head of its stream mean in
// ONE vector register (built from the lookup when the scan starts), reads codes of seven and eight bits from the lookup through the
// scalar cache and walks maxcode for longer ones.
struct alignas(16) ProgHuffTable {
    uint16_t lut[256];  // per 8-bit prefix: symbol | code length << 8 (length 0: not a code of up to 8 bits — the walk decides)
    int32_t maxcode[16], delta[16];
    uint8_t values[256];
    int32_t nvalues;
    int32_t pad_[3];
};
static_assert(sizeof(ProgHuffTable) == 912, "layout");
constexpr uint32_t PROG_TABLE_DWORDS = sizeof(ProgHuffTable) / 4u;  // 228
constexpr uint32_t PROG_LANE_DWORDS = 129u;                          // a lane's LDS region: the 8-bit lookup of its scan's table (or two DC lookups of a byte per entry), skewed by one dword so that lane l starts in bank l

struct ProgScanComp {
    int16_t *coefs;      // the component's plane in the coefficient arena (block-raster, natural order inside a block)
    uint64_t *masks;     // per block two words: which coefficients (zig-zag positions) are non-zero, which of those are negative
    uint32_t block_w;    // blocks per plane row
    uint32_t h, v;       // blocks per MCU (1, 1 in a single-component scan)
    uint32_t table;      // DC first scans: which of the scan's tables (0..3) decodes this component's differences
};

struct ProgScan {
    const uint8_t *data;  // the scan's entropy-coded bytes: unstuffed, 16-byte aligned, zero padded (huff_stage_segment)
    uint32_t n_bytes;     // unstuffed length; what follows counts as zero bits (the reference feeds zeros once it has met the
                          // marker that ends the scan: src/huffman.rs:123-160)
    uint8_t ss, se, ah, al;  // spectral band [ss, se] (se inclusive), successive approximation high / low
    uint32_t ncomp, cols, rows;  // components; MCUs per row / rows the scan walks (decode_scan's loops, src/decoder.rs:871-1000)
    ProgScanComp comp[4];
    const ProgHuffTable *table[4];  // DC first: the distinct DC tables of the scan (comp[].table indexes them); AC scans: table[0]
    // Scans of a track PIPELINED over lanes (one lane per scan instead of one per track): a scan may work on block b as soon as the
    // scans it depends on — for every coefficient it covers, the last earlier scan that covered it — have completed block b.
    uint32_t *progress;       // blocks (in walk order) this scan has completed, published per chunk; PROG_DONE at its end.  nullptr: nobody waits
    const uint32_t *wait[3];  // the progress words of the scans this one stays behind (nullptr: none)
    uint32_t wait_whole;      // bit i: wait[i] walks its blocks in another order: it must have ENDED before this scan starts
    uint32_t pad_;
    // what the wave that walked the scan reports (JPGPU_PROG_TIMES=1 prints it for the first frame): [0] its time in 10-ns units,
    // refinement scans: [1] calls of the hand-scheduled loop, [2] symbols it handed to the portable path, [3] window switches
    uint32_t report[4];
};
constexpr uint32_t PROG_DONE = 0xffffffffu;

struct ProgTrack {
    const ProgScan *scans;  // in stream order
    uint32_t n_scans;
    uint32_t *status;       // the image's status word (bit 0: decode on the host instead)
};

// status bits of a progressive image (bit 0 set with every one of them: the host decodes the image)
constexpr uint32_t PROG_ST_HOST = 1u, PROG_ST_BAD_CODE = 2u, PROG_ST_BAD_DC = 4u, PROG_ST_BAND = 8u, PROG_ST_STAGING = 16u, PROG_ST_RANGE = 32u,
                   PROG_ST_REPLACED = 64u /* a refinement scan puts a new value where a coefficient is non-zero already (damaged streams only) */,
                   PROG_ST_REFINE_SYMBOL = 128u, PROG_ST_WAIT = 512u /* a wave gave up waiting for the scan it depends on */;

}  // namespace jpgpu
