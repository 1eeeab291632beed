// huff_sync_core.hpp — entropy decoding ON THE DEVICE for sequential Huffman scans WITHOUT restart markers: the
// self-synchronising chunked decoder (Klein & Wiseman 2003; Weissenberger & Schmidt 2018/2021 for JPEG).
//
// A Huffman bit stream decoded from a wrong position re-synchronises with the true symbol boundaries after a few
// symbols with high probability.  The scan (unstuffed by the host, huff_stage_segment) is cut into chunks of 1,024 to 8,192
// bits (huff_sync_chunk_shift), one lane per chunk:
//   1. sync passes (huff_sync_chunk<false>): lane i decodes from its current start state to the first symbol boundary at
//      or beyond the end of its chunk and publishes that state for lane i+1.  State = (bit position, block-within-MCU,
//      coefficient index) — which tables apply and where a block ends; DC predictors and absolute block numbers are
//      additive and come later.  Pass 0 starts every lane but the first at its chunk boundary with state (block 0,
//      DC next); pass t > 0 re-decodes the lanes whose predecessor published something new.  Lane 0 always starts from the
//      truth, so a pass that changes nothing has reached the unique consistent — the true — segmentation.
//   2. an exclusive scan of the blocks completed per chunk gives every chunk its first block number;
//   3. the write pass (huff_sync_chunk<true>) decodes every chunk once more from its final start state and writes
//      coefficients into the zero-filled arena, DC positions holding the decoded DIFFERENCE;
//   4. a scan per component along the order the blocks have in the stream turns the differences into DC values
//      (i16 wrapping adds, src/decoder.rs:1095-1099).
// Speculative decoding may run into impossible codes; only what the write pass sees counts: an undecodable code, an
// EOBn run (legal only in progressive scans), data that ends before the last block — and a segmentation that has not
// settled after the allotted passes — raise the image's status word and the host decodes that image.
// The per-symbol step is the one of huff_core.hpp (decode_block as a select-based state machine).
#pragma once
#include "huff_core.hpp"

namespace jpgpu {

constexpr uint32_t HUFF_POS_INVALID = 0xffffffffu;  // published by a lane whose speculative decode hit an impossible code

struct HuffSyncLds {
    DevHuffTable tables[8];
    HuffSyncJob job;
    uint8_t unzig[64];
};

__device__ __forceinline__ void huff_open_at(DevBits &b, const uint8_t *slot, uint32_t bit_pos) {
    b.g = reinterpret_cast<const v4u *>(slot);
    b.wpos = bit_pos >> 5;
    b.cur = b.g[b.wpos >> 2];
    b.nxt = b.g[(b.wpos >> 2) + 1u];
    b.bits = 0;
    b.nbits = 0;
    b.bad = false;
    huff_refill(b);
    huff_consume(b, bit_pos & 31u);
}
__device__ __forceinline__ uint32_t huff_bit_pos(const DevBits &b) { return b.wpos * 32u - b.nbits; }

__device__ __forceinline__ bool huff_sync_state_plausible(const JP_LDS HuffSyncJob &job, uint32_t i, uint32_t pos, uint32_t q, uint32_t k) {
    const uint32_t first = i << job.chunk_shift;
    return pos >= first && pos - first <= 32u && q < job.bpm && k < 64u;
}

// One chunk.  WRITE = false: a sync pass (`pass` = its number), returns whether the lane published a new state (the caller
// counts those per job: one atomic per workgroup, not per lane — a quarter of a million lanes adding to a few hundred
// neighbouring counters took 18 ms per pass); WRITE = true: the write pass.
template <bool WRITE>
__device__ __forceinline__ bool huff_sync_chunk(JP_LDS HuffSyncLds &L, uint32_t i, uint32_t pass) {
    const JP_LDS HuffSyncJob &job = L.job;
    // start state
    uint32_t pos, q, k;
    if (i == 0u) {
        pos = 0u;
        q = 0u;
        k = 0u;
    } else if (!WRITE && pass == 0u) {
        pos = i << job.chunk_shift;
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
    if (i > 0u && !huff_sync_state_plausible(job, i, pos, q, k)) pos = HUFF_POS_INVALID;
    if (!WRITE) {
        if (pos == HUFF_POS_INVALID) return false;  // the predecessor has nothing to offer yet: keep what we have
        if (pass > 0u && pos == job.in_pos[i] && ((q << 8) | k) == job.in_qk[i]) return false;  // same start as last time
        job.in_pos[i] = pos;
        job.in_qk[i] = (q << 8) | k;
    } else if (pos == HUFF_POS_INVALID) {
        atomicOr_status(job.status, 1u | 32u);
        return false;
    }
    const uint32_t limit = min((i + 1u) << job.chunk_shift, job.n_bits);
    uint32_t nblk = 0;
    const uint32_t total_blocks = job.n_mcu * job.bpm;
    uint32_t blkno = WRITE ? job.n_blocks[i] : 0u;  // number of the block being decoded (write pass)
    if (WRITE) q = blkno % job.bpm;                 // (what the settled state says anyway; the only source when `uniform`)
    bool bad = false;
    if (pos < limit) {
        DevBits b;
        huff_open_at(b, job.data, pos);
        // fields of the component of block q
        uint32_t c = job.q_comp[q];
        uint32_t c_dc = job.comp[c].dc, c_ac = 4u + job.comp[c].ac;
        JP_GLOBAL int16_t *blk = nullptr;
        auto locate = [&]() {  // arena address of block `blkno` (write pass)
            const uint32_t m = blkno / job.bpm, my = m / job.cols, mx = m - my * job.cols;
            const JP_LDS HuffScanComp &sc = job.comp[c];
            const uint32_t sub = job.q_sub[q], vp = sub / sc.h, hp = sub - vp * sc.h;
            blk = (JP_GLOBAL int16_t *)(sc.dst + ((size_t)(my * sc.v + vp) * sc.block_w + (mx * sc.h + hp)) * 64u);
        };
        if (WRITE && blkno < total_blocks) locate();
        while (huff_bit_pos(b) < limit && !(WRITE && blkno >= total_blocks)) {
            huff_refill(b);
            const bool is_dc = k == 0u;
            const JP_LDS DevHuffTable &t = L.tables[is_dc ? c_dc : c_ac];
            const uint32_t e = t.lut[huff_peek(b, HUFF_LUT_BITS)], csz = e >> 8;
            uint32_t sym = e & 0xffu;
            if (csz) {
                huff_consume(b, csz);
            } else {
                sym = huff_walk(b, t);
                if (b.bad) {
                    bad = true;
                    break;
                }
            }
            const uint32_t r = sym >> 4, sz = sym & 15u;
            if (is_dc && sym > 11u) {
                bad = true;
                break;
            }
            const bool is_coef = !is_dc && sz != 0u, is_zrl = !is_dc && sz == 0u && r == 15u, is_eob = !is_dc && sz == 0u && r != 15u;
            const uint32_t knew = is_dc ? 0u : k + (is_zrl ? 16u : (is_coef ? r : 0u));
            const bool over = is_coef && knew >= 64u;
            const bool fused = csz > 0u && csz <= 8u && csz + sz <= 8u;
            const uint32_t nread = is_dc ? sym : (is_coef ? ((!over || fused) ? sz : 0u) : (is_eob ? r : 0u));
            const uint32_t raw = huff_peek(b, nread);
            huff_consume(b, nread);
            if (is_eob && r != 0u) {  // an end-of-band RUN: the blocks it covers are not in this chunk's state — host
                bad = true;
                break;
            }
            if (WRITE) {
                const int32_t val = huff_extend(raw, nread);
                if (is_dc) {
                    if (val) blk[0] = (int16_t)val;  // the difference; pass 4 accumulates
                } else if (is_coef && !over) {
                    blk[L.unzig[knew]] = (int16_t)val;
                }
            }
            bool done;
            if (is_dc) {
                done = false;
                k = 1u;
            } else {
                k = is_coef ? knew + 1u : knew;
                done = is_eob || over || k >= 64u;
            }
            if (done) {
                k = 0u;
                nblk++;
                q++;
                if (q == job.bpm) q = 0u;
                c = job.q_comp[q];
                c_dc = job.comp[c].dc;
                c_ac = 4u + job.comp[c].ac;
                if (WRITE) {
                    blkno++;
                    if (blkno < total_blocks) locate();
                }
            }
        }
        pos = huff_bit_pos(b);
    }
    if (!WRITE) {
        const uint32_t np = bad ? HUFF_POS_INVALID : pos, nqk = bad ? 0u : (((job.uniform ? 0u : q) << 8) | k);
        if (pass == 0u || np != job.out_pos[i] || nqk != job.out_qk[i] || nblk != job.n_blocks[i]) {
            huff_store_shared(job.out_qk + i, nqk);
            huff_store_shared(job.out_pos + i, np);
            job.n_blocks[i] = nblk;
            return true;
        }
    } else {
        if (bad) atomicOr_status(job.status, 1u | 2u);
        // the last chunk: every block must be complete before the data runs out, and none may have used bits from beyond
        // its end (the reference would decode zero bits there: the host decides)
        if (i + 1u == job.n_chunks && (blkno < total_blocks || pos > job.n_bits)) atomicOr_status(job.status, 1u | 8u);
    }
    return false;
}

}  // namespace jpgpu
