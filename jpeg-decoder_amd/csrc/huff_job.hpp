// huff_job.hpp — job descriptors of the device entropy decoder (huff_core.hpp); no HIP dependency: the host front-end
// fills them (csrc/host/frontend.cpp, plan_device_scans).
#pragma once
#include <stdint.h>
#include <string.h>

namespace jpgpu {

#ifndef JPGPU_DEV_LUT_BITS  // (A/B builds: -DJPGPU_DEV_LUT_BITS=9 — half the LDS per table, more symbols through the bit-serial walk)
#define JPGPU_DEV_LUT_BITS 10
#endif
constexpr int HUFF_LUT_BITS = JPGPU_DEV_LUT_BITS;  // <= kLutBits of the host front-end (10), whose table the device table is cut from

// What a Huffman symbol means to the decoding loop, 12 bits: how many magnitude bits follow, how far the coefficient index moves
// (DC: to 1; AC coefficient: run + 1; ZRL: 16; EOB: to 64, which ends the block), whether it is a coefficient, whether it cannot
// occur in a sequential scan.  The wide table holds it per prefix together with the code length, so that the loop's one table
// read per symbol says everything (round 3: symbol from the table, then its class from a second table — a dependent LDS round
// trip per step in a loop that is a chain of them).
constexpr uint32_t SYM_NREAD = 0x000fu, SYM_ADV_SHIFT = 4, SYM_ADV_MASK = 0x3fu /* advance - 1 */, SYM_COEF = 0x0400u, SYM_BAD = 0x0800u;
constexpr uint32_t SYM_LEN_SHIFT = 12;  // table entries: code length here (0: the prefix is not resolved within the lookahead)
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t
    huff_sym_info(uint32_t ac, uint32_t sym) {
    const uint32_t r = sym >> 4, sz = sym & 15u;
    if (!ac) return sym > 11u ? SYM_BAD : ((0u << SYM_ADV_SHIFT) | sym);  // "invalid DC difference magnitude category"
    if (sz) return SYM_COEF | (r << SYM_ADV_SHIFT) | sz;
    if (r == 15u) return 15u << SYM_ADV_SHIFT;                             // ZRL
    return r == 0u ? (63u << SYM_ADV_SHIFT) : SYM_BAD;                     // EOB; an EOBn run is for the host
}

// Codes longer than the lookahead: a second table per unresolved prefix, indexed by the six bits that follow it (a code has at
// most 16).  Its entries hold the code length minus one.  Up to HUFF_SUB_TABLES prefixes per table get one (Annex K's AC tables have
// 7-8 such prefixes); the others — and the prefixes of malformed tables that the reference's procedure rejects — say
// HUFF_SUB_NONE and the maxcode walk decides at decode time, as in rounds 1-3 (≈45 instructions that some lane of a wave needed
// in a quarter of the steps).
constexpr int HUFF_SUB_TABLES = 12, HUFF_SUB_BITS = 16 - HUFF_LUT_BITS;
constexpr uint32_t HUFF_SUB_NONE = 0x0fffu;

struct alignas(16) DevHuffTable {  // (maxcode[8..15] are read as two 16-byte words)
    uint16_t lut[1 << HUFF_LUT_BITS];  // per prefix: huff_sym_info of the symbol | code length << 12; length 0: not resolved within the
                                       // lookahead — the entry is the number of the prefix's second-level table, or HUFF_SUB_NONE
    uint16_t lut2[HUFF_SUB_TABLES][1 << HUFF_SUB_BITS];  // huff_sym_info | (code length - 1) << 12
    int32_t maxcode[16], delta[16];
    uint8_t values[256];
    int32_t nvalues;
};
static_assert(sizeof(DevHuffTable) % 16 == 0 && (sizeof(uint16_t) << HUFF_LUT_BITS) % 16 == 0, "16-byte reads of maxcode");

// Where table `id` (0..3) of class `ac` (0: DC, 1: AC) sits among a scan's eight device tables: ids 0 and 1 — all that most files use —
// in the first four slots.
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t huff_table_slot(uint32_t ac, uint32_t id) { return 2u * id + ac; }
struct HuffScanComp {
    int16_t *dst;       // the component's coefficient plane in the arena (zero-filled before the launch)
    uint32_t block_w;   // blocks per plane row
    uint32_t h, v;      // blocks per MCU (1, 1 in a single-component scan)
    uint32_t dc, ac;    // table ids 0..3 of the scan's DC / AC table (slot in `tables`: huff_table_slot)
};

// Host side of the staging: copy one restart segment (markers excluded, 0xFF00 pairs inside) without its stuffing
// zeros, then zero bytes up to the next 16-byte boundary plus 144 (the device reader fetches aligned 16-byte chunks ahead
// and treats what follows a segment as zero bits).  Returns the unstuffed length.  Slot size: huff_slot_bytes(n).
inline uint32_t huff_slot_bytes(uint32_t stuffed_bytes) { return ((stuffed_bytes + 15u) & ~15u) + 144u; }  // (the LDS ring reads up to 8 pieces ahead)
// `clean` (optional) is cleared if a 0xFF inside the segment is not followed by its stuffing zero — a marker or a fill byte: the
// planner that takes the short way for scans without restart markers leaves that check to this pass over the same bytes.
// The portable form: runs between 0xFF bytes (one per ~256 bytes of entropy-coded data) go through memchr + memcpy, ~5x a byte loop.
inline uint32_t huff_unstuff_portable(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t o, uint32_t i, bool *clean) {
    while (i < n) {
        const uint8_t *ff = static_cast<const uint8_t *>(memchr(src + i, 0xFF, n - i));
        const uint32_t run = ff ? (uint32_t)(ff - (src + i)) + 1u : n - i;  // up to and including the 0xFF
        memcpy(dst + o, src + i, run);
        o += run;
        i += run;
        if (ff) {
            if (i < n && src[i] == 0) i++;  // its stuffing zero
            else if (clean) *clean = false;
        }
    }
    return o;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// 32 bytes at a time where the host has AVX2 (checked once at run time): a vector without a 0xFF — seven of eight — is one load, one
// compare and one store; one with a 0xFF is stored whole as well (what lies behind the 0xFF is overwritten by the next store: the
// slot is at least 144 bytes longer than the data) and the walk goes on behind its stuffing zero.  Per thread 2-3 x the portable form
// (tools/host_stage_bench.cpp), which matters where a call waits for its first sub-batch to be staged.
__attribute__((target("avx2"))) inline uint32_t huff_unstuff_avx2(uint8_t *dst, const uint8_t *src, uint32_t n, bool *clean, uint32_t &i_out) {
    typedef char v32c __attribute__((vector_size(32), aligned(1), may_alias));
    uint32_t o = 0, i = 0;
    while (i + 32u <= n) {
        const v32c v = *reinterpret_cast<const v32c *>(src + i);
        const uint32_t m = (uint32_t)__builtin_ia32_pmovmskb256((v32c)(v == (char)0xFF));
        *reinterpret_cast<v32c *>(dst + o) = v;
        if (m == 0u) {
            o += 32u;
            i += 32u;
            continue;
        }
        const uint32_t t = (uint32_t)__builtin_ctz(m) + 1u;  // up to and including the first 0xFF
        o += t;
        i += t;
        if (i < n && src[i] == 0) i++;  // its stuffing zero
        else if (clean) *clean = false;
    }
    i_out = i;
    return o;
}
#endif
inline uint32_t huff_stage_segment(uint8_t *dst, const uint8_t *src, uint32_t n, bool *clean = nullptr) {
    uint32_t o = 0, i = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    static const bool avx2 = __builtin_cpu_supports("avx2") > 0;
    if (avx2) o = huff_unstuff_avx2(dst, src, n, clean, i);
#endif
    o = huff_unstuff_portable(dst, src, n, o, i, clean);
    const uint32_t slot = huff_slot_bytes(n);
    memset(dst + o, 0, slot - o);
    return o;
}

// ---- one scan for the device decoders (huff_sync_core.hpp): restart segments, or the self-synchronising chunk decoder ------

struct HuffSyncJob {             // one scan of one image, for either device decoder:
                                 //  - ri == 0: no restart interval in force, the scan is one staged slot cut into chunks (all fields);
                                 //  - ri > 0: restart segments (seg_off, n_seg; the per-chunk arrays and chunk fields are unused)
    const uint8_t *data;         // staged scan: unstuffed, 16-byte aligned, zero padded (ri > 0: base of the batch's data area)
    const uint32_t *seg_off;     // ri > 0: 2 * n_seg words, segment s starts at data + seg_off[2s] and has seg_off[2s+1] unstuffed bytes
    uint32_t n_seg, ri;          // restart interval in MCUs
    const DevHuffTable *tables;  // 8 tables of this scan (slots: huff_table_slot)
    uint32_t *status;            // the image's status word (bit 0: decode on the host instead)
    uint32_t *changed;           // per job: 3 counters of published states, used in rotation by consecutive launches (huff.hip)
    // per chunk (n_chunks entries each)
    uint32_t *in_pos, *in_qk;    // start state last decoded from          (qk = block-within-MCU << 8 | coefficient index)
    uint32_t *out_pos, *out_qk;  // published end state
    uint32_t *n_blocks;          // blocks completed by the chunk; after the scan: number of its first block
    uint32_t *blk_end;           // bit position at which the last block the chunk completed ended (emitting jobs; 0: none completed)
    uint32_t *dc_sum;            // 2 words per chunk: the sums of its DC differences per component, 16 bits each (component 0 in the
                                 // low half of word 0); after the scan: the DC predictors its first block starts from.  Unused
                                 // when `uniform` (which component a block belongs to is unknown until the blocks are numbered)
    uint32_t n_bits, n_chunks;
    uint32_t cols, n_mcu;        // MCUs per row / in the scan
    uint32_t ncomp, bpm;         // components, blocks per MCU
    uint32_t uniform, chunk_shift; // chunk = 1 << chunk_shift bits (huff_sync_chunk_shift); uniform: every component uses the same pair of tables: the block-within-MCU index does not influence
                                 // decoding and stays out of the sync state (else such scans — RGB, CMYK files — settle one chunk
                                 // per pass: a lane synchronised in position but one block off would hand the error on forever)
    HuffScanComp comp[4];
    uint8_t q_comp[16], q_sub[16];  // block-within-MCU -> component of the scan, block inside that component's part of the MCU
    // Range statistics as a by-product of the passes that write coefficients (the write pass, the restart-segment decoder, the
    // DC sums of `uniform` scans): the image's RangeStats words (range_stats.hpp) in device memory, and the quantization
    // tables of the scan's components in natural order — the writer sees every non-zero coefficient and its position anyway.
    uint32_t *stats;
    uint16_t q[4][64];
    // Speculative emission (huff_sync_core.hpp, "one pass less"): from the second sync pass on every lane leaves what it decodes
    // as a list of entries in stream order in its chunk's own buffer; once the segmentation has settled the lists ARE the
    // scan, and huff_expand_kernel turns them into whole blocks (no write pass, no zero fill).  nullptr: write pass instead.
    uint32_t *emit;         // n_chunks buffers of emit_stride entries: value & 0xffff | zig-zag index << 16 (0: a DC value, the first entry of a block) | scan component << 22
    uint32_t *emit_cnt;     // per chunk: entries | entries before the first block start << 16 (HUFF_EMIT_OVERFLOW: see there)
    uint32_t emit_stride;
    uint32_t pass0_skip;    // bits of its chunk every lane but the first leaves out in sync pass 0 (huff_sync_chunk)
    // Restart-marker streams through the chunk decoder (round 3): the segments are independent scans in miniature — every one
    // starts at a byte boundary with the predictors at zero — so each gets `seg_chunks` chunk slots of its own (chunk i = slot
    // i % seg_chunks of segment i / seg_chunks; slots behind a segment's data stay empty), its first lane starts from the truth,
    // block numbers and DC sums restart per segment.  n_seg <= 1: the one-slot scan of a stream without restart markers.
    uint32_t seg_chunks;
    uint32_t late_pass;     // from this sync pass on a lane stores its entries one by one (huff_sync_core.hpp, huff_sync_run<2>)
    // The weave (round 4, huff_weave_*): what the sync passes read instead of `data` — the dwords of 64 neighbouring chunks side by
    // side, so that a wave's 64 stream fetches of a step fall into a few cache lines instead of 64.  Device-only work space.
    const uint32_t *weave;  // huff_weave_dwords(n_chunks, chunk_shift) dwords
    uint32_t data_dwords;   // dwords that may be read from `data` (the scan's slots); what lies beyond counts as zeros
    uint32_t keep_lists;    // 1: the pixel kernel reads the entry lists itself (fused_entries.hpp) — no expansion into the arena
};
// Where chunk i lies: bits [start, end) of the job's data, whether a segment starts there, which segment it belongs to.
struct HuffChunkSpan {
    uint32_t start, end, seg;
    bool first;
};
template <class Job>
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    HuffChunkSpan
    huff_chunk_span(const Job &job, uint32_t i) {
    HuffChunkSpan sp;
    if (job.n_seg <= 1u) {
        sp.seg = 0u;
        sp.first = i == 0u;
        sp.start = i << job.chunk_shift;
        const uint32_t e = (i + 1u) << job.chunk_shift;
        sp.end = e < job.n_bits ? e : job.n_bits;
        if (sp.end < sp.start) sp.end = sp.start;
        return sp;
    }
    sp.seg = i / job.seg_chunks;
    const uint32_t j = i - sp.seg * job.seg_chunks, seg_start = job.seg_off[2u * sp.seg] * 8u, seg_bits = job.seg_off[2u * sp.seg + 1u] * 8u;
    const uint32_t lo = j << job.chunk_shift, hi = (j + 1u) << job.chunk_shift;
    sp.first = j == 0u;
    sp.start = seg_start + (lo < seg_bits ? lo : seg_bits);
    sp.end = seg_start + (hi < seg_bits ? hi : seg_bits);
    return sp;
}
// The weave.  A lane of a sync pass walks its chunk a dword at a time, and a wave's 64 lanes walk 64 neighbouring chunks at about
// the same pace: in the staged scan those 64 fetches of a step are 64 cache lines, each of which the lane comes back to 32 times —
// 270,000 lanes of a 256-image call keep 35 MB of such lines open, more than the L2s hold, and nearly every fetch was a miss that a
// whole wave waited for (sync passes of 256 1080p files: 1.5 GB fetched for 0.1 GB of scans).  huff_weave_kernel therefore lays the
// chunks of a job out side by side before the first pass — tile t holds chunks 64 t .. 64 t + 63, row r of a tile the r-th dword of
// each of them (256 bytes, two cache lines) — and every chunk's column goes on HUFF_WEAVE_EXTRA dwords into what follows the
// chunk, so a lane never leaves its column: it reads at most dword C + 1 of it (C = dwords per chunk; huff_sync_run stops at the
// first symbol boundary at or behind the chunk's end and holds no more than 64 bits + one dword ahead).
constexpr uint32_t HUFF_WEAVE_LANES = 64u, HUFF_WEAVE_EXTRA = 4u;
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t huff_weave_height(uint32_t chunk_shift) { return (1u << (chunk_shift - 5u)) + HUFF_WEAVE_EXTRA; }  // dwords per column
inline size_t huff_weave_dwords(uint32_t n_chunks, uint32_t chunk_shift) {
    return (size_t)((n_chunks + HUFF_WEAVE_LANES - 1u) / HUFF_WEAVE_LANES) * HUFF_WEAVE_LANES * huff_weave_height(chunk_shift);
}
// where row r of chunk i's column sits in the weave (dwords)
#ifdef __HIPCC__
__host__ __device__
#endif
inline size_t huff_weave_at(uint32_t chunk_shift, uint32_t i, uint32_t r) {
    return ((size_t)(i / HUFF_WEAVE_LANES) * huff_weave_height(chunk_shift) + r) * HUFF_WEAVE_LANES + i % HUFF_WEAVE_LANES;
}
// ... and what it holds: dword (first dword of the chunk) + r of the staged scan, zero beyond the scan's slots
template <class Job>
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t
    huff_weave_value(const Job &job, uint32_t first_dword, uint32_t r) {
    const uint32_t w = first_dword + r;
    return w < job.data_dwords ? reinterpret_cast<const uint32_t *>(job.data)[w] : 0u;
}
// blocks segment `seg` must hold, and the number of its first block
template <class Job>
inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    void
    huff_segment_blocks(const Job &job, uint32_t seg, uint32_t &first_block, uint32_t &n_blocks) {
    const uint32_t total = job.n_mcu * job.bpm;
    if (job.n_seg <= 1u) {
        first_block = 0u;
        n_blocks = total;
        return;
    }
    const uint32_t per = job.ri * job.bpm;
    first_block = seg * per < total ? seg * per : total;
    n_blocks = total - first_block < per ? total - first_block : per;
}
// Does the scan write every block of its components' planes?  (An interleaved scan does; a scan of one component of several
// leaves out the blocks that pad the plane to whole MCUs of the frame.)  block_h[c]: rows of blocks of scan component c's plane.
inline bool huff_scan_covers_planes(const HuffSyncJob &j, const uint32_t block_h[4]) {
    if (j.cols == 0 || j.n_mcu % j.cols != 0) return false;
    const uint32_t rows = j.n_mcu / j.cols;
    for (uint32_t c = 0; c < j.ncomp; c++)
        if (j.cols * j.comp[c].h != j.comp[c].block_w || rows * j.comp[c].v != block_h[c]) return false;
    return true;
}
constexpr uint32_t HUFF_EMIT_OVERFLOW = 0xffffffffu, HUFF_LATE_PASS = 2u;  // (HuffSyncJob::late_pass by default)
// An entry with zig-zag index 0 is a DC value — the first entry of its block (AC entries have indices 1..63).
#ifdef __HIPCC__
__host__ __device__
#endif
inline bool huff_entry_is_dc(uint32_t ent) { return (ent & 0x3f0000u) == 0u; }
// Entries a chunk can produce: a DC entry takes at least 1 bit and is followed by an end-of-block code (>= 1 bit) or 63
// coefficients, an AC entry at least 2 bits (code + magnitude) — at most 64 entries per 127 bits — and the symbols a lane decodes
// start inside its chunk (the last one may end up to 31 bits beyond it).  Multiple of 4 entries: buffers stay 16-byte aligned.
inline uint32_t huff_emit_stride(uint32_t chunk_shift) {
    const uint32_t bits = 1u << chunk_shift;
    return (bits / 2u + bits / 128u + 32u + 3u) & ~3u;  // (a multiple of 4: the lists are written 16 bytes at a time, huff_emit_entry_if)
}

// Chunk size.  A lane that starts at a wrong place finds the symbol boundaries within a few symbols, the block boundaries at
// the next end-of-block — and the position inside the MCU (which tables apply) only by luck, one try per re-synchronisation,
// so what matters is the number of BLOCKS in a chunk: ~48 of them (measured: 15 blocks per chunk settle 63 % of the lanes per
// pass, 60 blocks 98 %), between 1,024 and 32,768 bits, from the stream's average (the stuffed length serves: an upper bound
// taken before the staging copy).
inline uint32_t huff_sync_chunk_shift(uint32_t stuffed_bytes, uint32_t total_blocks, uint32_t blocks_per_chunk = 48u, uint32_t min_shift = 10u) {
    const uint64_t target = (uint64_t)stuffed_bytes * 8u * blocks_per_chunk / (total_blocks ? total_blocks : 1u);
    uint32_t shift = min_shift;
    while (shift < 15u && (1ull << shift) * 1414u / 1000u < target) shift++;  // nearest power of two (in the log domain)
    return shift;
}
inline uint32_t huff_sync_chunks(uint32_t bytes, uint32_t chunk_shift) {
    const uint32_t n = (uint32_t)(((uint64_t)bytes * 8u + (1u << chunk_shift) - 1u) >> chunk_shift);
    return n ? n : 1u;
}
// comp[0..ncomp) filled in -> bpm, the block-within-MCU maps and `uniform`
inline void huff_sync_finish_job(HuffSyncJob &j) {
    uint32_t bpm = 0;
    for (uint32_t c = 0; c < j.ncomp; c++)
        for (uint32_t b = 0; b < j.comp[c].h * j.comp[c].v && bpm < 16u; b++) {
            j.q_comp[bpm] = (uint8_t)c;
            j.q_sub[bpm] = (uint8_t)b;
            bpm++;
        }
    j.bpm = bpm;
    j.uniform = j.ncomp > 1u ? 1u : 0u;  // (one component: every block is its block — nothing to be unsure about)
    for (uint32_t c = 1; c < j.ncomp; c++)
        if (j.comp[c].dc != j.comp[0].dc || j.comp[c].ac != j.comp[0].ac) j.uniform = 0;
}

}  // namespace jpgpu
