// huff_core.hpp — entropy decoding ON THE DEVICE for sequential Huffman scans that carry restart markers
// ("DRI segments are independently decodable", src/decoder.rs:920-956).  One lane decodes one restart
// segment: `ri` MCUs, DC predictors starting at 0 (src/decoder.rs:928-931), straight into the zero-filled dense
// coefficient arena (natural order, block raster per component — what the host front-end would have appended row by
// row).  The decoding procedure is the reference's (src/huffman.rs:31-96, src/decoder.rs:1086-1172) on the same wide
// tables the host front-end uses (csrc/host/frontend.cpp: an exact cache of the 8-bit LUT + maxcode walk).  Anything
// unexpected — an undecodable code, a segment that is not consumed the way the reference would accept — raises the
// image's status flag and the caller re-decodes that image on the host, whose behaviour on damaged streams is the
// pinned one.  Compiled by hipcc for the kernel (huff.hip) and by g++ for tests/emu.
//
// Shape of the code.  The lanes of a wave walk unrelated bit streams: whatever any lane does, the wave executes, and a
// wave has nothing to overlap with (one wave per SIMD at best), so every dependent instruction costs its full latency
// (~12 cycles).  The run time is therefore the number of instructions on the UNION of the lanes' paths.  Hence:
//   * the host removes the 0xFF00 stuffing and aligns every segment (huff_stage_segment): the bit reader appends one
//     aligned dword when it holds <= 32 bits — no byte loop, no marker logic on the device;
//   * decode_block is unrolled into ONE step per Huffman symbol (k == 0: the DC symbol of the next block; k >= 1: an AC
//     symbol) computed with selects; only the rare table miss and the end of a block are branches;
//   * per-step memory operations are LDS reads through address-space-3 pointers (generic pointers made them flat_*
//     operations at several hundred cycles each) and one 2-byte store per non-zero coefficient.
// History (MI355X, 68 one-MCU-row segments per 1080p image): block-structured decoder with a byte-wise reader 22 ms for 64
// images — the same 22 ms for 256 (latency-bound); this form 13.4 ms for 1,024 images.
#pragma once
#include "huff_job.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

struct DevBits {
    uint64_t bits;   // unread bits, left-aligned
    uint32_t nbits;
    uint32_t wpos;   // dwords taken from the segment slot so far
    const v4u *g;    // the slot (16-byte aligned)
    v4u cur, nxt;    // chunk wpos / 4 and the one after it
    bool bad;
};

__device__ __forceinline__ void huff_open(DevBits &b, const uint8_t *slot) {
    b.g = reinterpret_cast<const v4u *>(slot);
    b.cur = b.g[0];
    b.nxt = b.g[1];
    b.bits = 0;
    b.nbits = 0;
    b.wpos = 0;
    b.bad = false;
}
// at most once per step: afterwards more than 32 bits are available (a step reads <= 16 + 15)
__device__ __forceinline__ void huff_refill(DevBits &b) {
    if (b.nbits <= 32u) {
        const uint32_t w = b.wpos & 3u;
        const uint32_t x = w == 0u ? b.cur.x : (w == 1u ? b.cur.y : (w == 2u ? b.cur.z : b.cur.w));
        b.bits |= (uint64_t)__builtin_bswap32(x) << (32u - b.nbits);
        b.nbits += 32u;
        b.wpos++;
        if ((b.wpos & 3u) == 0u) {
            b.cur = b.nxt;
            b.nxt = b.g[(b.wpos >> 2) + 1u];
        }
    }
}
__device__ __forceinline__ uint32_t huff_peek(const DevBits &b, uint32_t n) { return n ? (uint32_t)(b.bits >> (64u - n)) : 0u; }
__device__ __forceinline__ void huff_consume(DevBits &b, uint32_t n) {
    b.bits <<= n;
    b.nbits -= n;
}
__device__ __forceinline__ int32_t huff_extend(uint32_t v, uint32_t n) {  // src/huffman.rs:98-101 (n == 0 -> 0)
    const int32_t vt = n ? 1 << (n - 1u) : 0;
    return (int32_t)v < vt ? (int32_t)v + (int32_t)(0xffffffffu << n) + 1 : (int32_t)v;
}

__device__ __forceinline__ void atomicOr_status(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    atomicOr(p, v);
#endif
}
// state words other lanes (possibly of other workgroups, in the same launch) read while we run: straight to/from L2
__device__ __forceinline__ uint32_t huff_load_shared(const uint32_t *p) {
#ifdef JPGPU_HOST_EMULATION
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void huff_store_shared(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// The slow tail of src/huffman.rs:31-58 for prefixes the wide table does not resolve: the maxcode walk from length 9 —
// first length whose code does not exceed that length's largest code.  All eight comparisons at once (two 16-byte LDS
// reads) and a find-first-set instead of a loop: with 64 lanes per wave some lane nearly always needs the tail, and as a
// loop it cost every lane up to eight dependent LDS round trips per symbol.
__device__ __forceinline__ uint32_t huff_walk(DevBits &b, const JP_LDS DevHuffTable &t) {
    const uint32_t b16 = huff_peek(b, 16);
    const JP_LDS v4u *mc = (const JP_LDS v4u *)&t.maxcode[8];
    const v4u m0 = mc[0], m1 = mc[1];
    const int32_t m[8] = {(int32_t)m0.x, (int32_t)m0.y, (int32_t)m0.z, (int32_t)m0.w, (int32_t)m1.x, (int32_t)m1.y, (int32_t)m1.z, (int32_t)m1.w};
    uint32_t hits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) hits |= ((int32_t)(b16 >> (7 - j)) <= m[j] ? 1u : 0u) << j;
    if (hits == 0u) {
        b.bad = true;
        return 0u;
    }
    const uint32_t i = 8u + (uint32_t)__builtin_ctz(hits);
    const int32_t code = (int32_t)(b16 >> (15u - i));
    huff_consume(b, i + 1u);
    const int32_t index = code + t.delta[i];
    if (index < 0 || index >= t.nvalues) {
        b.bad = true;
        return 0u;
    }
    return t.values[index];
}

// What a workgroup keeps in LDS: the scan's job record and tables and the zig-zag table.
struct HuffLds {
    DevHuffTable tables[8];
    HuffScanJob job;
    uint8_t unzig[64];
};
// zig-zag -> natural order (src/decoder.rs:27-36), written to LDS once per workgroup
__device__ __forceinline__ void huff_fill_unzigzag(JP_LDS uint8_t *dst, uint32_t lane) {
    static const uint8_t unzig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                      15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (lane < 64u) dst[lane] = unzig[lane];
}

// One restart segment.  Returns false (and has raised the status bit) if the image must go to the host.
__device__ __forceinline__ bool huff_decode_segment(JP_LDS HuffLds &L, uint32_t seg) {
    const JP_LDS HuffScanJob &job = L.job;
    DevBits b;
    huff_open(b, job.data + job.seg_off[2u * seg]);
    const uint32_t seg_bits = job.seg_off[2u * seg + 1u] * 8u;
    int32_t pred0 = 0, pred1 = 0, pred2 = 0, pred3 = 0;  // (named scalars: a runtime-indexed array would live in scratch)
    uint32_t eob_run = 0;
    uint32_t m = seg * job.ri;                        // MCU
    const uint32_t m1 = min(m + job.ri, job.n_mcu), cols = job.cols, ncomp = job.ncomp;
    uint32_t c = 0, sub = 0;                          // component of the scan, block of the component inside the MCU
    uint32_t k = 0;                                   // 0: DC symbol next, else position of the next AC coefficient
    // fields of component c, reloaded when c changes
    uint32_t c_h = job.comp[0].h, c_hv = job.comp[0].h * job.comp[0].v, c_v = job.comp[0].v, c_bw = job.comp[0].block_w;
    uint32_t c_dc = job.comp[0].dc, c_ac = 4u + job.comp[0].ac;
    int16_t *c_dst = job.comp[0].dst;
    JP_GLOBAL int16_t *blk;  // the current block in the arena
    {
        const uint32_t my = m / cols, mx = m - my * cols;
        blk = (JP_GLOBAL int16_t *)(c_dst + ((size_t)(my * c_v) * c_bw + mx * c_h) * 64u);
    }
    while (m < m1 && !b.bad) {
        huff_refill(b);
        const bool is_dc = k == 0u;
        const JP_LDS DevHuffTable &t = L.tables[is_dc ? c_dc : c_ac];
        const uint32_t e = t.lut[huff_peek(b, HUFF_LUT_BITS)], csz = e >> 8;
        uint32_t sym = e & 0xffu;
        if (csz) {
            huff_consume(b, csz);
        } else {
            sym = huff_walk(b, t);
            if (b.bad) break;
        }
        const uint32_t r = sym >> 4, sz = sym & 15u;
        if (is_dc && sym > 11u) {  // "invalid DC difference magnitude category"
            b.bad = true;
            break;
        }
        // what the symbol is, and where the coefficient index stands after its run
        const bool is_coef = !is_dc && sz != 0u, is_zrl = !is_dc && sz == 0u && r == 15u, is_eob = !is_dc && sz == 0u && r != 15u;
        const uint32_t knew = is_dc ? 0u : k + (is_zrl ? 16u : (is_coef ? r : 0u));
        const bool over = is_coef && knew >= 64u;
        // (invalid stream, `over`) the reference's fused (run, size, value) table — code resolved by the 8-bit LUT, code +
        // magnitude within 8 bits — has taken the magnitude bits by now, its symbol-then-magnitude path has not
        const bool fused = csz > 0u && csz <= 8u && csz + sz <= 8u;
        const uint32_t nread = is_dc ? sym : (is_coef ? ((!over || fused) ? sz : 0u) : (is_eob ? r : 0u));
        const uint32_t raw = huff_peek(b, nread);
        huff_consume(b, nread);
        const int32_t val = huff_extend(raw, nread);
        if (is_dc) {
            int32_t pr = c == 0u ? pred0 : (c == 1u ? pred1 : (c == 2u ? pred2 : pred3));
            pr = (int16_t)(uint16_t)((uint32_t)pr + (uint32_t)val);  // i16 wrapping_add
            pred0 = c == 0u ? pr : pred0;
            pred1 = c == 1u ? pr : pred1;
            pred2 = c == 2u ? pr : pred2;
            pred3 = c == 3u ? pr : pred3;
            if (pr) blk[0] = (int16_t)pr;
        } else if (is_coef && !over) {
            blk[L.unzig[knew]] = (int16_t)val;
        }
        if (is_eob) eob_run = ((1u << r) - 1u + raw) & 0xffffu;
        // end of the block?  DC inside an end-of-band run (src/decoder.rs:1101-1104); EOB; index past 63
        bool done;
        if (is_dc) {
            done = eob_run > 0u;
            eob_run -= done ? 1u : 0u;
            k = 1u;
        } else {
            k = is_coef ? knew + 1u : knew;
            done = is_eob || over || k >= 64u;
        }
        if (done) {
            k = 0u;
            sub++;
            if (sub == c_hv) {
                sub = 0u;
                c++;
                if (c == ncomp) {
                    c = 0u;
                    m++;
                }
                const JP_LDS HuffScanComp &sc = job.comp[c];
                c_h = sc.h;
                c_v = sc.v;
                c_hv = sc.h * sc.v;
                c_bw = sc.block_w;
                c_dc = sc.dc;
                c_ac = 4u + sc.ac;
                c_dst = sc.dst;
            }
            const uint32_t my = m / cols, mx = m - my * cols;
            const uint32_t vp = sub / c_h, hp = sub - vp * c_h;
            blk = (JP_GLOBAL int16_t *)(c_dst + ((size_t)(my * c_v + vp) * c_bw + (mx * c_h + hp)) * 64u);
        }
    }
    // What the reference does at a restart (take_marker, src/huffman.rs:103-105, then reset): it keeps reading until it
    // meets the marker, which works iff the unread rest of the segment fits its 64-bit buffer; left-over bits are
    // dropped.  A segment that ran dry (bits taken from beyond its end — the reference would have fed zeros as well) is
    // left to the host to be safe.
    const int64_t consumed = (int64_t)b.wpos * 32 - (int64_t)b.nbits, left = (int64_t)seg_bits - consumed;
    if (b.bad || left < 0 || left > 64) {
        // bit 0 = re-decode on the host; bits 1..3 say why (diagnostics)
        const uint32_t why = 1u | (b.bad ? 2u : 0u) | (left > 64 ? 4u : 0u) | (left < 0 ? 8u : 0u);
#ifdef JPGPU_HOST_EMULATION
        *job.status |= why;
#else
        atomicOr(job.status, why);
#endif
        return false;
    }
    return true;
}

}  // namespace jpgpu
