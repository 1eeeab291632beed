// huff_sync_core.hpp — entropy decoding ON THE DEVICE for sequential Huffman scans: the self-synchronising chunked decoder
// (Klein & Wiseman 2003; Weissenberger & Schmidt 2018/2021 for JPEG) with speculative emission.
//
// A Huffman bit stream decoded from a wrong position re-synchronises with the true symbol boundaries after a few
// symbols with high probability.  The scan (unstuffed by the host, huff_stage_segment) is cut into chunks of 512 to 32,768
// bits (huff_sync_chunk_shift), one lane per chunk; a restart segment is a scan in miniature with chunk slots of its own:
//   1. sync passes (huff_sync_chunk): lane i decodes from its current start state to the first symbol boundary at or beyond
//      the end of its chunk and publishes that state for lane i+1.  State = (bit position, block-within-MCU, coefficient
//      index) — which tables apply and where a block ends; DC predictors and absolute block numbers are additive and come
//      later.  Pass 0 starts every lane at a guess inside its chunk with state (block 0, DC next); pass t > 0 re-decodes the
//      lanes whose predecessor published something new.  The first lane of a scan / segment always starts from the truth,
//      so a pass that changes nothing has reached the unique consistent — the true — segmentation.  Every pass but the first
//      leaves what it decodes as a list of entries per chunk (HuffEmit): once nothing changes, the lists ARE the scan;
//   2. an exclusive scan of the blocks completed per chunk gives every chunk its first block number, and the sums of DC
//      differences per chunk and component (i16 wrapping adds, src/decoder.rs:1095-1099) become the predictors each chunk
//      starts from (huff_sync_scan_kernel; per restart segment: huff_emit_segment_scan);
//   3. huff_expand_kernel turns the lists into whole 128-byte blocks.  (Scans whose components all share their tables —
//      `uniform` — do not know the component of a block before step 2: their entries hold the differences and
//      huff_dc_prefix_kernel runs the sums afterwards, one scattered read-modify-write per block.)
// Speculative decoding may run into impossible codes; only what the settled segmentation says counts: an undecodable code,
// an EOBn run (legal only in progressive scans), data that ends before the last block — and a segmentation that has not
// settled after the allotted passes — raise the image's status word and the host decodes that image.
// The per-symbol step is decode_block (src/decoder.rs:1020-1107) as a table-driven state machine (huff_sym_info).
// (Rounds 1-3 had a write pass — every chunk decoded once more, coefficients stored one by one into a zero-filled arena — and a
// one-lane-per-restart-segment decoder; both lost to emission + expansion and were deleted in round 4.)
#pragma once
#include "huff_core.hpp"

namespace jpgpu {

constexpr uint32_t HUFF_POS_INVALID = 0xffffffffu;  // published by a lane whose speculative decode hit an impossible code

// (what a symbol means to the loop: huff_sym_info, huff_job.hpp — in the wide tables next to the code length, and in a 512-entry
// LDS table for the symbols the slow path finds)
struct alignas(16) HuffBlockDst {  // block-within-MCU -> where its coefficients go: base + my * row_stride + mx * mcu_stride (bytes)
    uint64_t base;
    uint32_t row_stride, mcu_stride;
};
__device__ __forceinline__ void huff_fill_block_dst(const JP_LDS HuffSyncJob &job, JP_LDS HuffBlockDst *q_dst, uint32_t lane) {  // lane < 16
    const uint32_t c = job.q_comp[lane < job.bpm ? lane : 0u];
    const JP_LDS HuffScanComp &sc = job.comp[c];
    const uint32_t sub = job.q_sub[lane < job.bpm ? lane : 0u], h = sc.h ? sc.h : 1u, vp = sub / h, hp = sub - vp * h;
    q_dst[lane].base = (uint64_t)(uintptr_t)sc.dst + ((uint64_t)vp * sc.block_w + hp) * 128u;
    q_dst[lane].row_stride = sc.v * sc.block_w * 128u;
    q_dst[lane].mcu_stride = sc.h * 128u;
}

struct HuffSyncLds {
    HuffSyncJob job;
    uint16_t sym_info[2][256];  // [DC | AC][symbol]
    uint32_t q_tables[16];      // block-within-MCU -> byte offset of its DC table in `tables` | its AC table << 16
    uint32_t dc[256][4];        // per lane and component: sum of DC differences
    // Last, so that a kernel whose jobs use table ids 0 and 1 only can do with the first four slots (HuffSyncLdsCompact below): slot
    // 2 * id is DC table id, slot 2 * id + 1 AC table id (huff_table_slot).
    DevHuffTable tables[8];
};
// The same without the slots of table ids 2 and 3: the sync pass kernel of calls whose scans all use ids 0 and 1 (what encoders write
// for YCbCr and gray) — 24 kB of LDS instead of 40, six workgroups per CU instead of four.  Never touch tables[4..7] through it.
constexpr uint32_t HUFF_SYNC_LDS_COMPACT_BYTES = (uint32_t)(sizeof(HuffSyncLds) - 4u * sizeof(DevHuffTable));
constexpr uint32_t HUFF_SYNC_LANES = 256;  // lanes per workgroup (dc[] slots)
// after job and tables are in place; every lane of the workgroup calls it (lane < 512 does something), then a barrier
__device__ __forceinline__ void huff_sync_fill_lds(JP_LDS HuffSyncLds &L, uint32_t lane) {
    if (lane < 512u) L.sym_info[lane >> 8][lane & 255u] = (uint16_t)huff_sym_info(lane >> 8, lane & 255u);
    if (lane < 16u) {
        const uint32_t c = L.job.q_comp[lane < L.job.bpm ? lane : 0u];
        L.q_tables[lane] = (uint32_t)(huff_table_slot(0u, L.job.comp[c].dc) * sizeof(DevHuffTable)) | ((uint32_t)(huff_table_slot(1u, L.job.comp[c].ac) * sizeof(DevHuffTable)) << 16);
    }
}

// column: row 0 of the chunk's column in the weave, first_dword: the dword of the scan that row holds
__device__ __forceinline__ void huff_open_at(DevBits &b, const uint32_t *column, uint32_t first_dword, uint32_t bit_pos) {
    b.g = (const JP_GLOBAL uint32_t *)((uintptr_t)column - (uintptr_t)first_dword * (HUFF_WEAVE_LANES * 4u));
    b.wpos = bit_pos >> 5;
    b.ahead = b.g[(size_t)b.wpos * HUFF_WEAVE_LANES];
    b.bits = 0;
    b.nbits = 0;
    b.bad = false;
    huff_refill(b);
    huff_consume(b, bit_pos & 31u);
}
__device__ __forceinline__ uint32_t huff_bit_pos(const DevBits &b) { return b.wpos * 32u - b.nbits; }

__device__ __forceinline__ bool huff_sync_state_plausible(const JP_LDS HuffSyncJob &job, uint32_t first, uint32_t pos, uint32_t q, uint32_t k) {
    return pos >= first && pos - first <= 32u && q < job.bpm && k < 64u;
}

// Largest |coefficient * quantization value| among the DC / the AC coefficients a lane has written (range_stats.hpp): the
// writer's by-product that spares the pixel kernels' feeder a second pass over the arena.
struct HuffRange {
    uint32_t dc = 0, ac = 0;
};
// Speculative emission of a sync pass (HuffSyncJob::emit): the lane's chunk buffer and what it has put there.  An entry per DC
// value (always, zero or not: it marks the start of a block; the value is the running sum of the chunk's differences for its
// component — the chunk's predictor is added by the expansion — or the difference itself in a `uniform` scan) and per
// non-zero AC coefficient, in stream order.  Bits 22-23: the component of the scan the block belongs to — as far as the lane knows
// it: in a `uniform` scan it does not (the expansion derives it from the block number there).
// How the entries reach the buffer: four gathered in registers, two such groups per store round (the first waits in registers for
// the second), two 16-byte stores one after the other.  On gfx9 a wait for a load is a wait for every store issued before it, and
// the loop waits for its stream fetch in nearly every step; what was measured before settling on this (sync passes per 256 1080p
// images; 2.72 ms without emission; profiles/round3/14_emission_path.txt, tools/gpu_r3m.sh):
//   one 4-byte store per entry                                   3.85 ms
//   no stores at all (what the bookkeeping alone costs)          2.87 ms
//   one 16-byte store per four entries                           3.20 ms
//   two groups of four per store round (this)                    3.20 ms, fabric traffic -25 %: half as many partial-line writes meet a line that
//       has left the L2 in between (200 k lanes x one open 128-byte line each is more than the L2 holds); with the select-only loop
//       of the round's end 2.25 ms against 2.35 for one group per store
//   the stream read through an LDS ring                          5.16 ms (60 kB of LDS: two workgroups per CU)
//   entries collected in LDS, written every eighth step by all lanes at once   3.48-3.54 ms (the stores' cost is not the waits behind them)
struct HuffEmit {
    JP_GLOBAL uint32_t *buf = nullptr;  // nullptr: this run emits nothing
    uint32_t n = 0, cap = 0, lead = 0xffffffffu;  // entries so far (counts on past `cap`: overflow), capacity (a multiple of 4), entries before the first DC
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // the last entries, youngest in s3, not yet stored
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;      // the complete group of four before them, waiting for its neighbour
};
// One step of a lane: an entry, or none (`put`) — register selects, not a divergent region (a wave would enter it in every step).
__device__ __forceinline__ void huff_emit_entry_if(HuffEmit &em, uint32_t e, bool put) {
    em.s0 = put ? em.s1 : em.s0;
    em.s1 = put ? em.s2 : em.s1;
    em.s2 = put ? em.s3 : em.s2;
    em.s3 = put ? e : em.s3;
    const bool first_full = put && (em.n & 7u) == 3u;
    em.t0 = first_full ? em.s0 : em.t0;
    em.t1 = first_full ? em.s1 : em.t1;
    em.t2 = first_full ? em.s2 : em.t2;
    em.t3 = first_full ? em.s3 : em.t3;
    if (put && (em.n & 7u) == 7u && em.n < em.cap) {  // two groups, one after the other: the second store finds the line where the first left it
        JP_GLOBAL v4u *dst = (JP_GLOBAL v4u *)(em.buf + (em.n - 7u));  // (one address, the second store at offset 16)
        dst[0] = v4u{em.t0, em.t1, em.t2, em.t3};
        dst[1] = v4u{em.s0, em.s1, em.s2, em.s3};
    }
    em.n += put ? 1u : 0u;
}
// the entries of an incomplete round, at the end of a run
__device__ __forceinline__ void huff_emit_finish(HuffEmit &em) {
    if (em.buf && em.n <= em.cap) {
        uint32_t first = em.n & ~7u;
        if (em.n & 4u) {  // a complete group waits in t
            *(JP_GLOBAL v4u *)(em.buf + first) = v4u{em.t0, em.t1, em.t2, em.t3};
            first += 4u;
        }
        const uint32_t r = em.n & 3u;  // (branches, not selects: the compiler turned selects into a table in scratch memory)
        if (r == 1u) {
            em.buf[first] = em.s3;
        } else if (r == 2u) {
            em.buf[first] = em.s2;
            em.buf[first + 1u] = em.s3;
        } else if (r == 3u) {
            em.buf[first] = em.s1;
            em.buf[first + 1u] = em.s2;
            em.buf[first + 2u] = em.s3;
        }
    }
}

// The decoding loop of a sync pass: one step per Huffman symbol, arranged for the instruction streams
// the compiler makes of it — a wave's step is as many SCALAR instructions (lane-mask bookkeeping around every divergent region) as
// vector ones, and the two issue at the same rate, so regions count.  DC and AC entries leave through ONE emission site; whether
// a pass emits is a template parameter (it is the same for every lane of a launch's iteration), `bad` is a number in a vector
// register, not a lane mask carried round the loop.  From (pos, q, k) until the bit position reaches `limit`; returns the
// position reached, q, k, nblk (blocks completed) updated.
template <int EMIT>  // 0: no entries; 1: entries in rounds of eight (a pass of every lane); 2: entry by entry (a late pass: few lanes, the wave's step is what counts)
__device__ __forceinline__ uint32_t huff_sync_run(JP_LDS HuffSyncLds &L, const uint32_t *column, uint32_t first_dword, uint32_t pos, uint32_t limit, uint32_t &q, uint32_t &k,
                                                  uint32_t &nblk, JP_LDS uint32_t *dc, bool dc_sums, bool &bad_out, HuffEmit &em, uint32_t &last_block_end) {
    const JP_LDS HuffSyncJob &job = L.job;
    DevBits b;
    huff_open_at(b, column, first_dword, pos);
    uint32_t c = job.q_comp[q];  // component of block q
    const JP_LDS uint8_t *tbase = (const JP_LDS uint8_t *)L.tables;
    uint32_t qt = L.q_tables[q];  // table offsets of block q
    uint32_t badv = 0;
    const uint32_t bpm = job.bpm;  // (in a register: the loop's LDS writes keep the compiler from hoisting the read itself)
    do {  // (the caller has checked pos < limit; one exit, at the bottom: the compiler keeps one set of registers for the loop's values)
        huff_refill(b);
        const uint32_t ac = k != 0u ? 1u : 0u;
        const JP_LDS DevHuffTable &t = *(const JP_LDS DevHuffTable *)(tbase + (ac ? qt >> 16 : qt & 0xffffu));
        const uint32_t e = t.lut[huff_peek(b, HUFF_LUT_BITS)];
        uint32_t info = e, csz = e >> SYM_LEN_SHIFT, raw;
        if (csz == 0u && e != HUFF_SUB_NONE) {  // a code longer than the lookahead: the prefix's second-level table
            info = t.lut2[0][e * (1u << HUFF_SUB_BITS) + (huff_peek(b, 16) & ((1u << HUFF_SUB_BITS) - 1u))];
            csz = (info >> SYM_LEN_SHIFT) + 1u;
        }
        if (csz) {  // code and magnitude bits leave the reader together (<= 16 + 15 of the > 32 bits it holds)
            const uint32_t nr = info & SYM_NREAD;
            raw = huff_peek(b, csz + nr) & ((1u << nr) - 1u);
            huff_consume(b, csz + nr);
        } else {
            const uint32_t sym = huff_walk(b, t);
            badv |= b.bad ? 1u : 0u;
            info = (uint32_t)L.sym_info[ac][sym];
            raw = huff_peek(b, info & SYM_NREAD);
            huff_consume(b, info & SYM_NREAD);
        }
        const uint32_t nread = info & SYM_NREAD;
        const bool isdc = k == 0u, coef = (info & SYM_COEF) != 0u;
        k += ((info >> SYM_ADV_SHIFT) & SYM_ADV_MASK) + 1u;
        // (a coefficient beyond index 63: the reference breaks out of the block in a way that depends on its own table layout,
        // src/decoder.rs:1045-1075 — only broken streams have it, the host decides)
        badv |= (info & SYM_BAD) | ((coef && k > 64u) ? 1u : 0u);
        // From here on: selects, not regions (every divergent region is three scalar instructions and a set of register copies
        // where it joins); the regions left are the DC sums, the store of a full round and the end of a block — turning those into
        // selects as well (sums as packed 16-bit adds in two registers, table offsets re-read in every step) costs more vector
        // instructions than it saves scalar ones: sync passes of 256 files 2.44 ms against 2.28.
        int32_t val = huff_extend(raw, nread);
        if (isdc && dc_sums && badv == 0u) {  // the chunk's sum of differences so far, per component
            dc[c] += (uint32_t)val;
            val = (int16_t)(uint16_t)dc[c];
        }
        if (EMIT) {
            const bool put = (isdc || coef) && badv == 0u;
            em.lead = (put && isdc && em.lead == 0xffffffffu) ? em.n : em.lead;
            // (k - 1: the zig-zag index of the coefficient just read, 0 for a DC value; the expansion turns it into the natural position)
            const uint32_t ent = (((k - 1u) & 63u) << 16) | (c << 22) | (uint32_t)(uint16_t)val;
            if (EMIT == 2) {
                if (put && em.n < em.cap) em.buf[em.n] = ent;
                em.n += put ? 1u : 0u;
            } else {
                huff_emit_entry_if(em, ent, put);
            }
        }
        if (k >= 64u && badv == 0u) {  // end of the block
            k = 0u;
            nblk++;
            last_block_end = huff_bit_pos(b);
            q = q + 1u == bpm ? 0u : q + 1u;
            qt = L.q_tables[q];
            c = job.q_comp[q];
        }
    } while (badv == 0u && huff_bit_pos(b) < limit);
    bad_out = badv != 0u;
    return huff_bit_pos(b);
}

// One chunk in one sync pass (`pass` = its number); returns whether the lane published a new state (the caller counts those per
// job: one atomic per workgroup, not per lane — a quarter of a million lanes adding to a few hundred neighbouring counters took
// 18 ms per pass).  The pass also leaves the chunk's entries — every pass but the first, whose start states are guesses.  A lane
// that has decoded from a state WITHOUT emitting still has work when the same state comes round again (QK_EMITTED in in_qk tells).
constexpr uint32_t QK_EMITTED = 0x80000000u;
// From pass HuffSyncJob::late_pass on a lane stores its entries one by one (huff_sync_run<2>): few lanes are left, what a late pass
// costs is the wave's step — 16 vector instructions shorter without the rounds' register shuffling — times the symbols of a chunk
// (sync passes of 256 files alone 2.24 -> 2.15-2.18 ms with 2; from pass 3 on: 2.19).  One image through Decoder.decode() measures the
// same with 1 and with 2 (1080p 1.40-1.50 ms): 2 everywhere (HUFF_LATE_PASS, huff_job.hpp; JPGPU_SYNC_LATE_PASS pins another).
__device__ __forceinline__ bool huff_emit_in_pass(const JP_LDS HuffSyncJob &job, uint32_t pass) { return job.emit != nullptr && pass > 0u; }

// dc_slot: which of the workgroup's HUFF_SYNC_LANES slots for DC sums the lane uses (default: the chunk's own; the late launches, whose
// lanes take chunks from anywhere in a span of several blocks, pass their thread index)
__device__ __forceinline__ bool huff_sync_chunk(JP_LDS HuffSyncLds &L, uint32_t i, uint32_t pass, uint32_t dc_slot = 0xffffffffu) {
    const JP_LDS HuffSyncJob &job = L.job;
    // start state
    const HuffChunkSpan span = huff_chunk_span(job, i);
    uint32_t pos, q, k;
    if (pass == 0u) {
        // The first pass is there to find where the chunks END, from guessed start states; a lane that starts at a guess
        // finds the true segmentation within ~15 blocks on average (the misses decay exponentially): it need not walk the
        // whole chunk for that.  Every lane decodes its chunk again from a real state in pass 1 anyway — the first lane too,
        // whose true start is known (one lane walking a whole chunk would keep the launch waiting for it).
        pos = span.start + job.pass0_skip;
        q = 0u;
        k = 0u;
    } else if (span.first) {  // the start of the scan, or of a restart segment: the truth
        pos = span.start;
        q = 0u;
        k = 0u;
    } else {
        pos = huff_load_shared(job.out_pos + (i - 1u));
        const uint32_t qk = huff_load_shared(job.out_qk + (i - 1u));
        q = qk >> 8;
        k = qk & 0xffu;
    }
    if (job.uniform) q = 0u;
    // A state published by the lane to the left lies within one symbol (16 code bits + 15 more) after the end of ITS chunk,
    // which is the start of ours.  Anything else is not a state of this launch sequence — that lane belongs to a workgroup
    // which has not run yet, and the words are what an earlier batch left there — and must not be decoded from (it could
    // mean walking half the scan) nor handed on (it would travel down the scan, one lane per pass, keeping the job unsettled).
    if (!span.first && pass != 0u && !huff_sync_state_plausible(job, span.start, pos, q, k)) return false;  // nothing to offer yet: keep what we have
    const bool emit = huff_emit_in_pass(job, pass);
    const uint32_t qk_in = (q << 8) | k | (emit ? QK_EMITTED : 0u);
    if (pass > 0u && pos == job.in_pos[i] && qk_in == job.in_qk[i]) return false;  // same start as last time
    job.in_pos[i] = pos;
    job.in_qk[i] = qk_in;
    const uint32_t limit = span.end;
    uint32_t nblk = 0;
    bool bad = false;
    const bool dc_sums = !job.uniform;
    JP_LDS uint32_t *dc = L.dc[dc_slot == 0xffffffffu ? i % HUFF_SYNC_LANES : dc_slot];
    if (dc_sums) dc[0] = dc[1] = dc[2] = dc[3] = 0u;
    uint32_t last_block_end = 0;
    HuffEmit em;
    if (emit) {
        em.buf = (JP_GLOBAL uint32_t *)(job.emit + (size_t)i * job.emit_stride);
        em.cap = job.emit_stride;
    }
    const bool late = emit && pass >= job.late_pass;
    if (pos < limit) {
        const uint32_t *column = job.weave + huff_weave_at(job.chunk_shift, i, 0u);
        const uint32_t w0 = span.start >> 5;
        if (late) pos = huff_sync_run<2>(L, column, w0, pos, limit, q, k, nblk, dc, dc_sums, bad, em, last_block_end);
        else if (emit) pos = huff_sync_run<1>(L, column, w0, pos, limit, q, k, nblk, dc, dc_sums, bad, em, last_block_end);
        // (a pass without emission — pass 0, whose start states are guesses — is there for the end states: its sums of DC differences
        // would be overwritten by pass 1, which every lane runs; without them the loop loses a divergent region with an LDS
        // read-modify-write in it and the sign extension of every value: sync passes of 256 files alone 2.25 -> 2.17 ms.  The sums in
        // REGISTERS instead of LDS for the emitting passes — four selects each way — measured 2.26 against 2.18: the loop is bound by the
        // vector instructions it issues, not by that round trip)
        else pos = huff_sync_run<0>(L, column, w0, pos, limit, q, k, nblk, dc, false, bad, em, last_block_end);
    }
    if (job.emit != nullptr) {
        job.blk_end[i] = last_block_end;
        if (!late) huff_emit_finish(em);  // (a late pass has stored every entry already)
        // (pass 0 leaves an empty list behind: the word is never what an earlier batch left there)
        job.emit_cnt[i] = !emit ? 0u : (em.n > em.cap ? HUFF_EMIT_OVERFLOW : (em.n | (min(em.lead, em.n) << 16)));
    }
    if (dc_sums) {
        job.dc_sum[2u * i] = (dc[0] & 0xffffu) | (dc[1] << 16);
        job.dc_sum[2u * i + 1u] = (dc[2] & 0xffffu) | (dc[3] << 16);
    }
    const uint32_t np = bad ? HUFF_POS_INVALID : pos, nqk = bad ? 0u : (((job.uniform ? 0u : q) << 8) | k);
    if (pass == 0u || np != job.out_pos[i] || nqk != job.out_qk[i] || nblk != job.n_blocks[i]) {
        huff_store_shared(job.out_qk + i, nqk);
        huff_store_shared(job.out_pos + i, np);
        job.n_blocks[i] = nblk;
        return true;
    }
    return false;
}

// What only the settled segmentation can say (block numbering, huff_sync_scan_kernel): a chunk
// whose last run met an impossible code (or overran its buffer) before the scan's last block was complete ...
__device__ __forceinline__ uint32_t huff_emit_chunk_status(const HuffSyncJob &job, uint32_t i, uint32_t blocks_through_chunk) {
    const uint32_t total_blocks = job.n_mcu * job.bpm;
    if (blocks_through_chunk >= total_blocks) return 0u;  // (what follows the last block is nobody's business: src/decoder.rs stops there)
    if (job.emit_cnt[i] == HUFF_EMIT_OVERFLOW) return 1u | 128u;
    return job.out_pos[i] == HUFF_POS_INVALID ? (1u | 2u) : 0u;
}
// ... and data that ends before the last block does (`blocks` = what all chunks completed), or whose last block took bits from
// beyond the end: only symbols that start inside the data are decoded, so that is the last symbol of the last chunk completing
// block number total - 1.
__device__ __forceinline__ uint32_t huff_emit_final_status(const HuffSyncJob &job, uint32_t blocks) {
    const uint32_t total_blocks = job.n_mcu * job.bpm;
    if (blocks < total_blocks || job.n_chunks == 0u) return 1u | 8u;
    const uint32_t p = job.out_pos[job.n_chunks - 1u];
    if (blocks == total_blocks && p != HUFF_POS_INVALID && p > job.n_bits && (job.out_qk[job.n_chunks - 1u] & 0xffu) == 0u) return 1u | 8u;
    return 0u;
}

// The same for a job with restart markers (HuffSyncJob::seg_chunks), one segment at a time — by one thread, a segment has a
// handful of chunks: number its chunks' first blocks from the segment's own first block, turn their sums of DC differences
// into the predictors they start from (zero at the segment start, src/decoder.rs:928-931), and return what a decoder that walks
// the segment from its start would flag: an impossible code or a full buffer before the segment's blocks are
// complete, fewer or more blocks than the restart interval holds, a last block that took bits from beyond the segment, or more
// than 56 bits between its end and the marker (the reference would not find the marker there: src/huffman.rs:103-160).
__device__ __forceinline__ uint32_t huff_emit_segment_scan(const HuffSyncJob &job, uint32_t seg) {
    uint32_t first_block, expected;
    huff_segment_blocks(job, seg, first_block, expected);
    const uint32_t c0 = seg * job.seg_chunks, seg_bits = job.seg_off[2u * seg + 1u] * 8u, seg_end = job.seg_off[2u * seg] * 8u + seg_bits;
    uint32_t run = 0, status = 0, end_pos = 0, acc[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < job.seg_chunks; j++) {
        const uint32_t i = c0 + j, nb = job.n_blocks[i];
        job.n_blocks[i] = first_block + run;
        if (!job.uniform) {
            const uint32_t w0 = job.dc_sum[2u * i], w1 = job.dc_sum[2u * i + 1u];
            job.dc_sum[2u * i] = (acc[0] & 0xffffu) | ((acc[1] & 0xffffu) << 16);
            job.dc_sum[2u * i + 1u] = (acc[2] & 0xffffu) | ((acc[3] & 0xffffu) << 16);
            acc[0] += w0 & 0xffffu, acc[1] += w0 >> 16, acc[2] += w1 & 0xffffu, acc[3] += w1 >> 16;
        }
        if (run < expected) {  // (what a chunk does once the segment's blocks are complete is nobody's business)
            if (job.emit_cnt[i] == HUFF_EMIT_OVERFLOW) status |= 1u | 128u;
            if (run + nb < expected && (j << job.chunk_shift) < seg_bits && job.out_pos[i] == HUFF_POS_INVALID) status |= 1u | 2u;
            if (run + nb == expected) end_pos = job.blk_end[i];
        }
        run += nb;
    }
    if (run < expected) return status | 1u | 8u;
    if (run > expected) return status | 1u | 4u;
    if (expected) {
        if (end_pos > seg_end) status |= 1u | 8u;
        else if (seg_end - end_pos > 56u) status |= 1u | 4u;
    }
    return status;
}

}  // namespace jpgpu
