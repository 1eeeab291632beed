// fused_entries.hpp — the 4:2:0 strip walk fed by the device entropy decoder's ENTRY LISTS (round 6).
//
// The chunk decoder (huff_sync_core.hpp) leaves a scan as lists of entries in stream order — value | zig-zag index << 16 |
// component << 22, one buffer per chunk of the bit stream — and through round 5 huff_expand_kernel turned them into whole
// coefficient blocks in the arena (6.2 MB per 1080p image written, and read again by the pixel kernel).  Here the pixel kernel's
// staging step reads the lists itself: per MCU row of its strip it clears the LDS staging area and scatters the entries of the MCUs
// [x0m - 1, x0m + te] (the strip and one halo MCU either side, whose chroma blocks the upsampler needs) straight into it; read_block /
// transform / colour are the S420 walk's, unchanged.  Nothing of the scan touches the coefficient arena.
//   * Which entries: interleaved 4:2:0 streams hold the MCUs in raster order, so a strip's MCU row is ONE contiguous run of blocks
//     of the stream.  huff_strip_index_kernel (huff.hip) finds, per (MCU row, strip), the chunk and the entry at which the run's
//     first block starts (a search over the chunks' first-block numbers, then a count of DC entries in one list).
//   * Which block an entry belongs to: block numbers count the DC entries (a ballot + v_mbcnt per 64 entries), starting from the
//     chunk's first block number (HuffSyncJob::n_blocks after the numbering pass); the entries in front of a chunk's first DC entry
//     belong to the block the chunk before left open.  The four waves of the workgroup take the run's chunks in turn.
//   * Where it lands: a table per workgroup maps (block of the run) -> staging block (= the lane that transforms it) and its bank
//     swizzle; a second one maps (component, zig-zag index) -> byte inside an unswizzled block and the quantization value.
//   * DC values are the chunk's running sums + what the chunks before add up to (HuffSyncJob::dc_sum after the numbering pass).
//   * Arithmetic class: the walk runs the "sane" body (every |coefficient x quantization value| < 2^15, pixel_math.hpp) — true of
//     every legal 8-bit stream.  The scatter sees every coefficient anyway and checks: an image that breaks the bound is flagged in its
//     status word (bit 9) and the host decodes it, like any stream the device decoder refuses.
// Only for scans with per-component tables (`uniform` == 0: the entries then carry their component), with or without restart
// segments; everything else keeps the expansion kernel.
#pragma once
#include "fused_core.hpp"
#include "huff_job.hpp"

namespace jpgpu {

struct EntrySrc {                // per batch image
    const HuffSyncJob *job;      // nullptr: the image's coefficients are in the arena (the dense kernels' business)
    const uint32_t *tab;         // per (MCU row, strip): chunk, entry — where the run of the strip's MCU row starts (huff_strip_index_kernel)
};
constexpr uint32_t ENTRY_ST_RANGE = 512u;  // status bit: a coefficient outside the sane class (with bit 0: the host decodes the image)

#if defined(__HIPCC__) && !defined(JPGPU_HOST_EMULATION)

struct S420ELds {                // behind S420Lds::total_bytes(tx)
    uint32_t *blk[3];            // block of the run -> staging byte offset | (swizzle key << 4) << 16 | invalid << 31; [0] a step, [1] seam above, [2] seam below
    uint32_t *zq;                // [3][64]: by component and zig-zag index: quantization value | byte inside an unswizzled block << 16
    uint32_t *rg;                // per wave: largest |coefficient x quantization value| seen; behind them a dump slot for the lanes that store nothing
    static __device__ __host__ __forceinline__ uint32_t run_blocks(uint32_t tx) { return 6u * (tx + 2u); }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx) { return 3u * run_blocks(tx) * 4u + 3u * 64u * 4u + 32u; }
    static __device__ __forceinline__ S420ELds make(uint8_t *base, uint32_t tx) {
        S420ELds l;
        const uint32_t nb = run_blocks(tx);
        l.blk[0] = reinterpret_cast<uint32_t *>(base);
        l.blk[1] = l.blk[0] + nb;
        l.blk[2] = l.blk[1] + nb;
        l.zq = l.blk[2] + nb;
        l.rg = l.zq + 3u * 64u;
        return l;
    }
};

__device__ __forceinline__ uint32_t e_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class T>
__device__ __forceinline__ T *e_uniform(T *p) {  // a pointer every lane holds the same value of -> scalar registers (loads through it: SGPR base + 32-bit lane offset)
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return (T *)(uintptr_t)(((uint64_t)e_rfl((uint32_t)(v >> 32)) << 32) | e_rfl((uint32_t)v));
}
__device__ __forceinline__ uint32_t e_lane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

struct S420E {
    typedef S420<ARITH_SANE, 256> K;
    static constexpr uint32_t NT = 256;
    static constexpr uint32_t R = 8;  // rounds of 64 entries requested before the first is used

    // tables of the workgroup (te: the strip's width in MCUs; the same for all its MCU rows)
    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t te, uint32_t tid, const S420ELds &e) {
        static const uint8_t unzig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                          15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        const uint32_t nb = 6u * (te + 2u);
        for (uint32_t d = tid; d < nb; d += NT) {
            const uint32_t m = d / 6u, q = d - 6u * m;  // MCU of the run (0 and te + 1: the halo), block within the MCU
            uint32_t lb = 0, seam_a = 0, seam_b = 0;
            bool ok = true;
            if (q < 4u) {  // luma: the strip's own MCUs only; staging block = row (q >> 1) * 2 te + 2 (m - 1) + (q & 1)
                ok = m >= 1u && m <= te;
                lb = (q >> 1) * 2u * te + 2u * (m - 1u) + (q & 1u);
            } else {       // chroma: block column m of the tile (S420::lane_block)
                lb = 4u * te + (q - 4u) * (te + 2u) + m;
                seam_a = (q - 4u) * (te + 2u) + m;        // S420::seam_stage: Cb above, Cr above, Cb below, Cr below
                seam_b = (2u + q - 4u) * (te + 2u) + m;
            }
            // coef_slot(lb, row) = lb * 8 + (row ^ key), key = (lb >> 1) & 7, in 16-byte units: the key goes in as (key << 4), XORed onto the byte offset
            auto pack = [](uint32_t b, bool valid) { return valid ? (b * 128u) | (((b >> 1) & 7u) << 20) : 0x80000000u; };
            e.blk[0][d] = pack(lb, ok);
            e.blk[1][d] = pack(seam_a, q >= 4u);
            e.blk[2][d] = pack(seam_b, q >= 4u);
        }
        {   // (one wave per component: a scalar choice of the table — a per-lane one would put the image struct in scratch)
            const uint32_t c = e_rfl(tid >> 6);
            const uint16_t *q0 = img.qt[0], *q1 = img.qt[1], *q2 = img.qt[2];
            const JP_GLOBAL uint16_t *q = (const JP_GLOBAL uint16_t *)(c == 0u ? q0 : (c == 1u ? q1 : q2));
            const uint32_t z = unzig[tid & 63u];
            if (c < 3u) e.zq[tid] = (uint32_t)q[z] | ((((z >> 3) << 4) | ((z & 7u) << 1)) << 16);
        }
        if (tid < 4u) e.rg[tid] = 0u;
    }

    static __device__ __forceinline__ void clear_stage(const S420Lds &lds, uint32_t nblocks, uint32_t tid) {
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
#pragma unroll
        for (uint32_t i = 0; i < 9u; i++)  // (6 * 42 + 4 = 256 blocks x 8 slots = 8 per lane; seam rounds of one-MCU strips: 12 blocks)
            if (tid + NT * i < nblocks * 8u) dst[tid + NT * i] = v4u{0u, 0u, 0u, 0u};
    }

    // The entries of the MCUs [a, b) of MCU row k -> the staging area, through table `blk`.  Every wave of the workgroup calls it;
    // wave w takes the run's chunks c0 + w, c0 + w + 4, ...   (c0, e0: the run's place in the lists, from the strip index.)  The largest
    // |value x quantization value| its lanes saw goes into e.rg[wave] (an LDS maximum: a register kept across the transform would
    // spill there).
    // Latencies: what a wave must know of its chunks comes in ONE round of vector loads (lane j: chunk c0 + j; all four waves read the
    // same lines), a chunk's entries in rounds of R x 64 requested before the first is used, and the three table reads of an entry
    // (block, zig-zag / quantization) do not depend on one another.
    struct Meta {  // what a wave must know of the chunks c .. c + 63, one chunk per lane
        uint32_t cw, nb, qk;
        v2u w;
    };
    static __device__ __forceinline__ Meta request_meta(const HuffSyncJob *__restrict__ job, uint32_t cbase, uint32_t tid) {
        const uint32_t n_chunks = e_rfl(job->n_chunks), cl = min(cbase + (tid & 63u), n_chunks - 1u);
        const JP_GLOBAL uint32_t *emit_cnt = (const JP_GLOBAL uint32_t *)e_uniform(job->emit_cnt), *n_blocks = (const JP_GLOBAL uint32_t *)e_uniform(job->n_blocks),
                                 *out_qk = (const JP_GLOBAL uint32_t *)e_uniform(job->out_qk);
        const JP_GLOBAL v2u *dc_sum = (const JP_GLOBAL v2u *)e_uniform(job->dc_sum);
        Meta m;
        m.cw = emit_cnt[cl], m.nb = n_blocks[cl], m.qk = cl ? out_qk[cl - 1u] : 0u;
        m.w = dc_sum[cl];
        // restart segments: a segment's first chunk continues nothing (huff_chunk_span(..).first)
        const uint32_t seg_chunks = e_rfl(job->n_seg) > 1u ? e_rfl(job->seg_chunks) : 0u;
        if (seg_chunks && cl % seg_chunks == 0u) m.qk = 0u;
        return m;
    }
    // `pre`: request_meta(job, c0, tid), asked for earlier (the walk asks before the colour phase of the row before: one round trip less
    // between the barrier and the first entry)
    template <bool PRE>
    static __device__ __forceinline__ void scatter_row(const HuffSyncJob *__restrict__ job, uint32_t c0, uint32_t e0, const FusedGeom &g, uint32_t strip,
                                                       uint32_t k, uint32_t tid, const S420Lds &lds, const uint32_t *blk, const S420ELds &e, const Meta &pre) {
        const uint32_t lane = tid & 63u, wave = e_rfl(tid >> 6);
        const uint32_t x0m = strip * g.tx, te = K::txe(g, strip);
        const uint32_t a = x0m ? x0m - 1u : 0u, b = min(x0m + te + 1u, g.mcu_w);
        const uint32_t B0 = 6u * (k * g.mcu_w + a), nB = 6u * (b - a), shift = x0m ? 0u : 6u;  // (no halo MCU in front of the first strip)
        const uint32_t n_chunks = e_rfl(job->n_chunks), stride = e_rfl(job->emit_stride);
        // restart segments (HuffSyncJob::seg_chunks): chunk c belongs to segment c / seg_chunks, whose blocks end at (segment + 1) x restart
        // interval; what its last chunk decodes out of the bits behind the segment's last block is nobody's (huff_expand_kernel: `total`)
        const uint32_t seg_chunks = e_rfl(job->n_seg) > 1u ? e_rfl(job->seg_chunks) : 0u, seg_blocks = e_rfl(job->ri) * 6u, all_blocks = e_rfl(job->n_mcu) * 6u;
        const JP_GLOBAL uint32_t *emit = (const JP_GLOBAL uint32_t *)e_uniform(job->emit);
        uint8_t *stage = lds.stage;
        const uint32_t dump = (uint32_t)(reinterpret_cast<uint8_t *>(e.rg + 4) - stage);  // (two bytes nobody reads)
        uint32_t rg = 0;
        bool done = false;
        for (uint32_t cbase = c0; !done && cbase < n_chunks; cbase += 64u) {
            Meta mt = pre;
            if (!PRE || cbase != c0) mt = request_meta(job, cbase, tid);
            const uint32_t m_cw = mt.cw, m_nb = mt.nb, m_qk = mt.qk;
            const v2u m_w = mt.w;
            for (uint32_t j = wave; j < 64u; j += 4u) {
                const uint32_t c = cbase + j;
                if (c >= n_chunks) {
                    done = true;
                    break;
                }
                const uint32_t cw = e_lane(m_cw, j), nblk = e_lane(m_nb, j), qk = e_lane(m_qk, j);
                uint32_t S = nblk + ((qk & 0xffu) ? 1u : 0u);  // number of the first block that starts in the chunk
                uint32_t nBc = nB;                              // blocks of the run this chunk may write: those of its own segment
                if (seg_chunks) {
                    const uint32_t seg_end = min((c / seg_chunks + 1u) * seg_blocks, all_blocks);
                    S = min(S, seg_end);  // (huff_strip_index_kernel: the state behind a segment's last block)
                    nBc = seg_end > B0 ? min(nB, seg_end - B0) : 0u;
                }
                if (c != c0 && S >= B0 + nB + 1u) {             // even the block it continues lies behind the run
                    done = true;
                    break;
                }
                const uint32_t w0 = e_lane(m_w.x, j), w1 = e_lane(m_w.y, j);
                const uint32_t wy = w0 & 0xffffu, wcb = w0 >> 16, wcr = w1 & 0xffffu;
                const uint32_t cnt = min(cw & 0xffffu, stride);
                const JP_GLOBAL uint8_t *buf = (const JP_GLOBAL uint8_t *)(emit + (size_t)c * stride);  // (a scalar base, 32-bit byte offsets per lane)
                // (in c0 the walk starts AT the DC entry of block B0: what lies in front of it belongs to the run before)
                const int32_t dbase = c == c0 ? 0 : (int32_t)(S - B0);
                uint32_t started = 0;
                bool behind = false;
                for (uint32_t eb = c == c0 ? e0 : 0u; !behind && eb < cnt; eb += 64u * R) {
                    uint32_t ent[R];
#pragma unroll
                    for (uint32_t r = 0; r < R; r++)  // clamped: unconditional loads
                        ent[r] = stream_load(reinterpret_cast<const JP_GLOBAL uint32_t *>(buf + 4u * min(eb + 64u * r + lane, cnt - 1u)));
                    // (all R requests leave before anything is used: left alone the compiler moves each load behind the early exit of the round
                    // before it — one exposed round trip per 64 entries, measured 1.37 ms against 0.99 per 256 x 1080p)
                    asm volatile("" : "+v"(ent[0]), "+v"(ent[1]), "+v"(ent[2]), "+v"(ent[3]), "+v"(ent[4]), "+v"(ent[5]), "+v"(ent[6]), "+v"(ent[7]));
                    static_assert(R == 8, "the asm statement above names eight registers");
#pragma unroll
                    for (uint32_t r = 0; r < R; r++) {
                        if (eb + 64u * r >= cnt) break;
                        const bool valid = eb + 64u * r + lane < cnt;
                        const bool flag = valid && huff_entry_is_dc(ent[r]);
                        const uint64_t m = __ballot(flag);
                        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, started));
                        const int32_t d = dbase + (int32_t)before + (flag ? 0 : -1);  // the entry's block, counted from B0
                        const uint32_t czz = (ent[r] >> 16) & 0xffu;                   // component * 64 + zig-zag index
                        const uint32_t t = blk[min((uint32_t)d, nB - 1u) + shift], zq = e.zq[min(czz, 191u)];
                        const uint32_t comp = czz >> 6;
                        const uint32_t v = (ent[r] + (flag ? (comp == 0u ? wy : (comp == 1u ? wcb : wcr)) : 0u)) & 0xffffu;  // DC: + what the chunks before add up to
                        const bool ok = valid && (uint32_t)d < nBc && !(t >> 31);
                        const int32_t sv = (int16_t)(uint16_t)v;
                        rg = max(rg, (uint32_t)(sv < 0 ? -sv : sv) * (ok ? zq & 0xffffu : 0u));
                        const uint32_t addr = (t & 0xffffu) + ((zq ^ t) >> 16 & 0x7eu);
                        *reinterpret_cast<uint16_t *>(stage + (ok ? addr : dump)) = (uint16_t)v;
                        started += (uint32_t)__popcll(m);
                        if (dbase + (int32_t)started >= (int32_t)nB + 1) {  // the block now open lies behind the run
                            behind = true;
                            break;
                        }
                    }
                }
            }
        }
        if (rg) atomicMax(&e.rg[wave], rg);
    }
};

#endif

}  // namespace jpgpu
