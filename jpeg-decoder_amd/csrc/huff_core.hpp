// huff_core.hpp — entropy decoding ON THE DEVICE for sequential Huffman scans that carry restart markers
// ("DRI segments are independently decodable", src/decoder.rs:920-956).  One lane decodes one restart
// segment: `ri` MCUs, DC predictors starting at 0 (src/decoder.rs:928-931), straight into the dense coefficient arena
// (natural order, block raster per component — what the host front-end would have appended row by row).
// The decoding procedure is the reference's (src/huffman.rs:31-96, src/decoder.rs:1086-1172) on the same wide tables
// the host front-end uses (csrc/host/frontend.cpp: an exact cache of the 8-bit LUT + maxcode walk).  Anything
// unexpected — an undecodable code, a segment that is not consumed exactly — raises the image's status flag and the
// caller re-decodes that image on the host, whose behaviour on damaged streams is the pinned one.
// Compiled by hipcc for the kernel (huff.hip) and by g++ for tests/emu.
#pragma once
#include "huff_job.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

// Bit reader of one lane.  The segment is read in aligned 16-byte chunks, the next chunk is requested as soon as the
// current one is entered: with one lane per segment and only a few waves in flight a byte-at-a-time reader spent
// ~2 us per symbol waiting for memory (measured: 25 ms per 64 images).  Reads may run up to 31 bytes past the segment:
// the staging block is padded.
struct DevBits {
    uint64_t bits;
    uint32_t nbits;
    int32_t pad_bits;  // zero bits appended after the end of the segment that are still in `bits` (what the reference
                       // feeds after it has seen the marker, src/huffman.rs:123-160)
    uint32_t pos, end; // byte offsets from the 16-byte aligned address `g` (pos = next byte to read)
    const v4u *g;
    v4u cur, nxt;
    bool bad;
};

__device__ __forceinline__ void huff_open(DevBits &b, const uint8_t *data, uint32_t first, uint32_t last) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(data) + first;
    b.g = reinterpret_cast<const v4u *>(a & ~(uintptr_t)15);
    b.pos = (uint32_t)(a & 15u);
    b.end = b.pos + (last - first);
    b.cur = b.g[0];
    b.nxt = b.g[1];
    b.bits = 0;
    b.nbits = 0;
    b.pad_bits = 0;
    b.bad = false;
}
__device__ __forceinline__ uint32_t huff_dword(const DevBits &b, uint32_t w) {  // dword w (0..3) of the current chunk
    return w == 0u ? b.cur.x : (w == 1u ? b.cur.y : (w == 2u ? b.cur.z : b.cur.w));
}
__device__ __forceinline__ void huff_advance(DevBits &b, uint32_t n) {  // consume n bytes (never across more than one chunk edge)
    const uint32_t before = b.pos >> 4;
    b.pos += n;
    if ((b.pos >> 4) != before) {
        b.cur = b.nxt;
        b.nxt = b.g[(b.pos >> 4) + 1u];
    }
}
__device__ __forceinline__ uint32_t huff_byte(DevBits &b) {
    const uint32_t v = (huff_dword(b, (b.pos >> 2) & 3u) >> (8u * (b.pos & 3u))) & 0xffu;
    huff_advance(b, 1u);
    return v;
}

__device__ __forceinline__ void huff_refill(DevBits &b) {
    // four bytes at once when they are ordinary entropy-coded data: aligned in the chunk, inside the segment, no 0xFF
    if (b.nbits <= 32u && (b.pos & 3u) == 0u && b.pos + 4u <= b.end) {
        const uint32_t x = huff_dword(b, (b.pos >> 2) & 3u), nx = ~x;
        if (((nx - 0x01010101u) & ~nx & 0x80808080u) == 0u) {
            const uint32_t be = __builtin_bswap32(x);
            b.bits |= (uint64_t)be << (32u - b.nbits);
            b.nbits += 32u;
            huff_advance(b, 4u);
        }
    }
    while (b.nbits <= 56u) {
        uint32_t byte = 0;
        if (b.pos < b.end) {
            byte = huff_byte(b);
            if (byte == 0xFFu) {  // inside a segment only stuffed 0xFF00 pairs occur (the host cut the segments at markers)
                if (b.pos < b.end && huff_byte(b) == 0u) {
                } else {
                    b.bad = true;
                }
            }
        } else {
            b.pad_bits += 8;
        }
        b.bits |= (uint64_t)byte << (56u - b.nbits);
        b.nbits += 8u;
    }
}
__device__ __forceinline__ uint32_t huff_peek(const DevBits &b, uint32_t n) { return n ? (uint32_t)(b.bits >> (64u - n)) : 0u; }
__device__ __forceinline__ void huff_consume(DevBits &b, uint32_t n) {
    b.bits <<= n;
    b.nbits -= n;
}
__device__ __forceinline__ int32_t huff_extend(uint32_t v, uint32_t n) {  // :98-101
    const int32_t vt = 1 << (n - 1u);
    return (int32_t)v < vt ? (int32_t)v + (int32_t)(0xffffffffu << n) + 1 : (int32_t)v;
}

// The slow tail of src/huffman.rs:31-58 for prefixes the wide table does not resolve: the maxcode walk from length 9.
__device__ __forceinline__ uint32_t huff_walk(DevBits &b, const JP_LDS DevHuffTable &t) {
    const uint32_t b16 = huff_peek(b, 16);
    for (int i = 8; i < 16; i++) {
        const int32_t code = (int32_t)(b16 >> (15 - i));
        if (code <= t.maxcode[i]) {
            huff_consume(b, (uint32_t)i + 1u);
            const int32_t index = code + t.delta[i];
            if (index < 0 || index >= t.nvalues) {
                b.bad = true;
                return 0u;
            }
            return t.values[index];
        }
    }
    b.bad = true;
    return 0u;
}

// What a workgroup keeps in LDS: the scan's job record and tables, the zig-zag table, one 128-byte block buffer per lane.
struct HuffLds {
    DevHuffTable tables[8];
    HuffScanJob job;
    uint8_t unzig[64];
    v4u blocks[64 * 8];
};
// zig-zag -> natural order (src/decoder.rs:27-36), written to LDS once per workgroup
__device__ __forceinline__ void huff_fill_unzigzag(JP_LDS uint8_t *dst, uint32_t lane) {
    static const uint8_t unzig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                      15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (lane < 64u) dst[lane] = unzig[lane];
}

// One restart segment.  A block is assembled in the lane's LDS buffer and leaves as eight 16-byte stores
// (coefficient-by-coefficient 2-byte stores were partial-line writes by the million).
//
// The lanes of a wave walk unrelated bit streams, so the decoder is written as ONE step that every lane executes
// per Huffman symbol — decode_block of a sequential scan (ss = 0, se = 63, ah = al = 0, src/decoder.rs:1086-1172)
// unrolled into a state machine (k == 0: the DC symbol of the next block; k >= 1: an AC symbol).  A wave has nothing to
// overlap with (one wave per SIMD at best), so every dependent instruction costs its full latency: per-step memory
// operations are LDS reads through address-space-3 pointers (generic pointers made them flat_* operations at several
// hundred cycles each), the per-component fields are cached in registers.  Measured on MI355X: ~2 us per symbol and lane
// (64 divergent lanes cost ~220 VALU + ~130 SALU wave-instructions per step at ~12 cycles each), i.e. 22 ms for the 68
// one-MCU-row segments of a 1080p image — the same 22 ms for 256 images at once (one wave per SIMD), 55 ms for 1024.
// Writing coefficients straight to a zero-filled arena instead of through the block buffer made no difference.
// Returns false (and has raised the status bit) if the image must go to the host.
__device__ __forceinline__ bool huff_decode_segment(JP_LDS HuffLds &L, uint32_t seg, uint32_t lane) {
    const JP_LDS HuffScanJob &job = L.job;
    DevBits b;
    huff_open(b, job.data, job.seg_off[2u * seg], job.seg_off[2u * seg + 1u]);
    JP_LDS v4u *lb = &L.blocks[lane * 8u];
    JP_LDS int16_t *blk = reinterpret_cast<JP_LDS int16_t *>(lb);
#pragma unroll
    for (int r = 0; r < 8; r++) lb[r] = v4u{0u, 0u, 0u, 0u};
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
    while (m < m1 && !b.bad) {
        // every read of this step fits 32 bits: code <= 16, then <= 15 magnitude / run-length bits
        if (b.nbits < 32u) huff_refill(b);
        const bool is_dc = k == 0u;
        const JP_LDS DevHuffTable &t = L.tables[is_dc ? c_dc : c_ac];
        const uint32_t e = t.lut[huff_peek(b, HUFF_LUT_BITS)], csz = e >> 8;
        uint32_t sym = e & 0xffu;
        if (csz) huff_consume(b, csz);
        else sym = huff_walk(b, t);
        if (b.bad) break;
        const uint32_t r = sym >> 4, sz = sym & 15u;
        bool done = false;  // block finished by this symbol
        if (is_dc) {
            if (sym > 11u) {
                b.bad = true;
                break;
            }
            int32_t diff = 0;
            if (sym) {
                diff = huff_extend(huff_peek(b, sym), sym);
                huff_consume(b, sym);
            }
            int32_t pr = c == 0u ? pred0 : (c == 1u ? pred1 : (c == 2u ? pred2 : pred3));
            pr = (int16_t)(uint16_t)((uint32_t)pr + (uint32_t)diff);  // i16 wrapping_add
            pred0 = c == 0u ? pr : pred0;
            pred1 = c == 1u ? pr : pred1;
            pred2 = c == 2u ? pr : pred2;
            pred3 = c == 3u ? pr : pred3;
            blk[0] = (int16_t)pr;
            k = 1u;
            if (eob_run > 0u) {  // inside an end-of-band run: the block has no AC symbols (src/decoder.rs:1101-1104)
                eob_run--;
                done = true;
            }
        } else if (sz == 0u) {
            if (r == 15u) {  // ZRL
                k += 16u;
                done = k >= 64u;
            } else {  // EOBn
                eob_run = (1u << r) - 1u;
                if (r) {
                    eob_run += huff_peek(b, r);
                    huff_consume(b, r);
                }
                eob_run &= 0xffffu;
                done = true;
            }
        } else {
            k += r;
            if (k >= 64u) {
                // (invalid stream) the reference's fused (run, size, value) table — code resolved by the 8-bit LUT, code +
                // magnitude within 8 bits — has taken the magnitude bits by now, its symbol-then-magnitude path has not
                if (csz > 0u && csz <= 8u && csz + sz <= 8u) huff_consume(b, sz);
                done = true;
            } else {
                blk[L.unzig[k]] = (int16_t)huff_extend(huff_peek(b, sz), sz);
                huff_consume(b, sz);
                k++;
                done = k >= 64u;
            }
        }
        if (done) {
            const uint32_t my = m / cols, mx = m - my * cols;
            const uint32_t vp = sub / c_h, hp = sub - vp * c_h;
            const size_t block = (size_t)(my * c_v + vp) * c_bw + (mx * c_h + hp);
            JP_GLOBAL v4u *gd = (JP_GLOBAL v4u *)(c_dst + block * 64u);
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                gd[rr] = lb[rr];
                lb[rr] = v4u{0u, 0u, 0u, 0u};
            }
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
        }
    }
    // What the reference does at a restart (take_marker, src/huffman.rs:103-105, then reset): it keeps reading until it
    // meets the marker, which works iff the unread rest of the segment fits its 64-bit buffer; left-over bits are
    // dropped.  Same rule here: one more refill must reach the end of the segment.  A segment that ran dry (bits taken
    // from beyond its end — the reference would have fed zeros as well) is left to the host to be safe.
    huff_refill(b);
    const int32_t real_left = (int32_t)b.nbits - b.pad_bits;
    if (b.bad || b.pos != b.end || real_left < 0) {
        // bit 0 = re-decode on the host; bits 1..3 say why (diagnostics)
        const uint32_t why = 1u | (b.bad ? 2u : 0u) | (b.pos != b.end ? 4u : 0u) | (real_left < 0 ? 8u : 0u);
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
