// fused_x4.hpp — fused kernel for FOUR-component frames with subsampled components (round 3; SURVEY §8a rows a11 + a15):
//   components 0 and (K_FULL) 3 at full size, components 1, 2 and (!K_FULL) 3 at half size in both directions (H2V2,
//   src/upsampler.rs:191-228), colour function CMYK (255 - x on all four, src/decoder.rs:1458-1474) or YCCK (YCbCr -> RGB on the
//   first three, 255 - k, src/decoder.rs:1439-1456) -> CMYK32.
// The layouts of the reference's own fixture tests/reftest/images/mozilla/jpg-cmyk-2.jpg (sampling 22 11 11 11: the M, Y, K planes
// go through UpsamplerH2V2) and of YCCK files as Photoshop writes them (22 11 11 22).  They ran the generic kernel pair (planes
// through HBM, 25-28 % of the roofline).
//
// Shape: a ROW kernel, not a walk.  A workgroup owns tx MCUs of ONE MCU row and transforms everything that row's pixels need:
//   * its own blocks in full — 4 per MCU of every full-size component, 1 per MCU (+ one halo block either side: the fancy upsampler
//     reads +-1 sample) of every half-size component;
//   * of the half-size components' block rows ABOVE and BELOW, only the one sample row the vertical filter touches
//     (src/upsampler.rs:200-206: rows y/2 and y/2 -+ 1) — pixel_math.hpp idct8x8_products_row, a fifth of a transform's instructions.
// No carry between steps, no segments, no seams to agree on: workgroups are independent and dispatched in address order.  The price is
// the neighbour rows' coefficient reads (from L2 / MALL: the workgroups above and below read them at about the same time) and the
// partial transforms (one lane per block, most of them cheap).  For plain 4:2:0 that price is too high (VALU-bound: DESIGN.md 5.0);
// for these layouts it replaces a plane round trip through HBM and a second kernel.
// One lane per block: 4*NL*tx + 3*NH*(tx + 2) <= 256 lanes (NL full-size, NH half-size components).
// Phases (barriers between): stage (all coefficients of the tile -> LDS) | read (block -> registers, dequantized) | transform
// (samples -> tiles that alias the staging area; the half-size tiles repeat their first / last sample in the column outside the
// image, as in S420) | colour (upsample, convert, store: a unit = 8 pixels x two output rows that share two rows of the half-size tiles).
#pragma once
#include "fused_core.hpp"

namespace jpgpu {

#ifdef JPGPU_HOST_EMULATION
typedef v4u r4_v4u_a4;
#else
typedef v4u r4_v4u_a4 __attribute__((aligned(4)));  // a 4-byte pixel is all the alignment an output row has
#endif

struct R4Lds {
    uint8_t *stage;  // one 128-B slot per block (swizzled: coef_slot); later the tiles:
    uint8_t *ytile;  //   NL x 16 rows x ypitch
    uint8_t *ctile;  //   NH x 10 rows x cpitch: row 0 = plane row 8k-1, rows 1..8 the MCU row's own, row 9 = plane row 8k+8;
                     //   column lc <-> plane column 8*(x0m - 1) + lc
    uint8_t *qtab;   // 4 x 128 B
    uint32_t ypitch, cpitch;
    static __device__ __host__ __forceinline__ uint32_t blocks(uint32_t tx, uint32_t nl, uint32_t nh) { return 4u * nl * tx + 3u * nh * (tx + 2u); }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx, uint32_t nl, uint32_t nh) { return blocks(tx, nl, nh) * 128u + 512u; }
    static __device__ __forceinline__ R4Lds make(uint8_t *base, uint32_t tx, uint32_t nl, uint32_t nh) {
        R4Lds l;
        l.ypitch = 16u * tx;
        l.cpitch = 8u * (tx + 2u);
        l.stage = base;
        l.ytile = base;
        l.ctile = base + nl * 16u * l.ypitch;  // nl*256*tx + nh*80*(tx+2) <= (4*nl*tx + 3*nh*(tx+2)) * 128
        l.qtab = base + blocks(tx, nl, nh) * 128u;
        return l;
    }
};
// 14*tx + 12 / 13*tx + 18 <= 256 lanes allow 17 / 18 MCUs; 16 make the pixel phase exactly one unit per lane (8 slots x 32 chunks)
constexpr uint32_t r4_tx_max(bool) { return 16u; }

template <int ARITH, bool K_FULL>
struct R4 {
    typedef R4Lds Lds;
    typedef S420<ARITH, 256> W;        // fetch_block / transform_block (they look at lds.stage and lds.qtab only)
    typedef PixelOps<ARITH> P;        // ChromaEO / load_eo
    static constexpr uint32_t NT = 256, NL = K_FULL ? 2u : 1u, NH = K_FULL ? 2u : 3u;
    static constexpr uint32_t LY = 1u;  // 16*tx <= 256 chunks per full-size block row: one load per lane and run
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile) { return min(g.tx, g.mcu_w - tile * g.tx); }
    // frame component of full-size slot l / half-size slot h
    static __device__ __forceinline__ uint32_t lcomp(uint32_t l) { return l == 0u ? 0u : 3u; }
    static __device__ __forceinline__ uint32_t hcomp(uint32_t h) { return 1u + h; }

    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t tid, const Lds &lds) {
        if (tid < 32u) {
            uint32_t *d = reinterpret_cast<uint32_t *>(lds.qtab);
            d[tid] = ((const JP_GLOBAL uint32_t *)img.qt[0])[tid];
            d[32u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[1])[tid];
            d[64u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[2])[tid];
            d[96u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[3])[tid];
        }
    }

    // Block index = lane.  [0, 4*te*NL): full-size components, per component 2*te blocks of block row 2k then 2*te of row 2k+1;
    // then per half-size component te+2 own blocks; then, same order, the blocks of the row above, then those of the row below.
    struct Blk {
        uint32_t comp;   // frame component
        uint32_t slot;   // index among the full-size / half-size components
        uint32_t ry, cx; // full-size: block row within the MCU row, block column within the tile; half-size: cx = LDS block column
        uint32_t part;   // 0 own, 1 above (its last sample row), 2 below (its first sample row)
        bool full_size, valid;
    };
    static __device__ __forceinline__ Blk lane_block(const FusedGeom &g, uint32_t tile, uint32_t k, uint32_t tid) {
        const uint32_t x0m = tile * g.tx, te = txe(g, tile), nb = te + 2u;
        Blk b{};
        if (tid < 4u * te * NL) {
            b.full_size = true;
            b.slot = tid >= 4u * te ? 1u : 0u;
            const uint32_t t = tid - b.slot * 4u * te;
            b.ry = t >= 2u * te ? 1u : 0u;
            b.cx = t - b.ry * 2u * te;
            b.comp = lcomp(b.slot);
            b.valid = true;
            return b;
        }
        const uint32_t t = tid - 4u * te * NL;
        if (t >= 3u * NH * nb) return b;
        b.part = t / (NH * nb);
        const uint32_t u = t - b.part * NH * nb;
        b.slot = u / nb;
        b.cx = u - b.slot * nb;
        b.comp = hcomp(b.slot);
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)b.cx;
        b.valid = bx >= 0 && bx < (int32_t)g.bwc && (b.part == 0u || (b.part == 1u ? k > 0u : k + 1u < g.mcu_h));
        return b;
    }

    struct Pre {
        v4u y[2 * NL][LY], c[3 * NH];
    };
    static __device__ __forceinline__ void stage_load(const FusedGeom &g, const FusedImage &img, uint32_t tile, uint32_t k, uint32_t tid, Pre &pre) {
        const uint32_t x0m = tile * g.tx, te = txe(g, tile), nl = 16u * te, ncc = 8u * (te + 2u);
        auto at = [](const JP_GLOBAL v4u *base, uint32_t chunk) -> v4u {
            return *reinterpret_cast<const JP_GLOBAL v4u *>(reinterpret_cast<const JP_GLOBAL uint8_t *>(base) + chunk * 16u);
        };
#pragma unroll
        for (uint32_t l = 0; l < NL; l++) {
            const JP_GLOBAL v4u *y0 = (const JP_GLOBAL v4u *)img.coefs[lcomp(l)] + ((size_t)(2u * k) * g.bw0 + 2u * x0m) * 8u;
            const JP_GLOBAL v4u *y1 = y0 + (size_t)g.bw0 * 8u;
#pragma unroll
            for (uint32_t i = 0; i < LY; i++) {
                const uint32_t j = min(tid + NT * i, nl - 1u);  // clamped: unconditional loads
                pre.y[2 * l][i] = at(y0, j);
                pre.y[2 * l + 1][i] = at(y1, j);
            }
        }
        // half-size components: block rows k-1, k, k+1 (clamped into the plane: rows outside it are never transformed), one halo
        // block either side (clamped onto valid chunks likewise)
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
        const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid, ncc - 1u), 0), cmax);
        const uint32_t rows[3] = {k, k > 0u ? k - 1u : 0u, k + 1u < g.mcu_h ? k + 1u : k};
#pragma unroll
        for (uint32_t part = 0; part < 3; part++)
#pragma unroll
            for (uint32_t h = 0; h < NH; h++)
                pre.c[part * NH + h] = at((const JP_GLOBAL v4u *)img.coefs[hcomp(h)] + (size_t)rows[part] * g.bwc * 8u, e);
    }
    static __device__ __forceinline__ void stage_store(const FusedGeom &g, uint32_t tile, uint32_t tid, const Lds &lds, const Pre &pre) {
        const uint32_t te = txe(g, tile), nl = 16u * te, nb = te + 2u, ncc = 8u * nb;
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
        const uint32_t row = tid & 7u, b = tid >> 3;
#pragma unroll
        for (uint32_t l = 0; l < NL; l++)
#pragma unroll
            for (uint32_t i = 0; i < LY; i++)
                if (tid + NT * i < nl) {
                    dst[coef_slot(l * 4u * te + b + (NT / 8u) * i, row)] = pre.y[2 * l][i];
                    dst[coef_slot(l * 4u * te + 2u * te + b + (NT / 8u) * i, row)] = pre.y[2 * l + 1][i];
                }
        if (tid < ncc) {
#pragma unroll
            for (uint32_t j = 0; j < 3u * NH; j++) dst[coef_slot(4u * te * NL + j * nb + b, row)] = pre.c[j];
        }
    }

    static __device__ __forceinline__ void read_block(const FusedGeom &g, uint32_t tile, uint32_t k, uint32_t tid, const Lds &lds, S420Regs &r) {
        const Blk b = lane_block(g, tile, k, tid);
        if (!b.valid) return;
        W::fetch_block(lds, tid, b.comp, r.cw);
    }

    // samples -> tiles.  Half-size tiles repeat their first / last sample in the column outside the image (W::edge_fix), so the
    // pixel phase needs no edge cases: (3t + t) >> 4 == t >> 2.
    static __device__ __forceinline__ void transform(const FusedGeom &g, uint32_t tile, uint32_t k, uint32_t tid, const Lds &lds, S420Regs &r) {
        const Blk b = lane_block(g, tile, k, tid);
        if (!b.valid) return;
        const uint32_t x0m = tile * g.tx;
        if (b.full_size) {
            uint32_t out[16];
            W::transform_block(lds, b.comp, r.cw, out);
            uint8_t *base = lds.ytile + (b.slot * 16u + b.ry * 8u) * lds.ypitch + b.cx * 8u;
#pragma unroll
            for (int row = 0; row < 8; row++) *reinterpret_cast<v2u *>(base + (uint32_t)row * lds.ypitch) = v2u{out[2 * row], out[2 * row + 1]};
            return;
        }
        uint8_t *tile0 = lds.ctile + b.slot * 10u * lds.cpitch + b.cx * 8u;
        if (b.part == 0u) {
            uint32_t out[16];
            W::transform_block(lds, b.comp, r.cw, out);
            const typename W::EdgeFix ef = W::edge_fix(g, x0m, b.cx, out);
#pragma unroll
            for (int row = 0; row < 8; row++) {
                uint8_t *p = tile0 + (1u + (uint32_t)row) * lds.cpitch;
                *reinterpret_cast<v2u *>(p) = v2u{out[2 * row], out[2 * row + 1]};
                W::edge_bytes(ef, p, out[2 * row], out[2 * row + 1]);
            }
            return;
        }
        uint32_t row[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        if constexpr (ARITH != ARITH_EXACT) {
            if (b.part == 1u) idct8x8_products_row<ARITH, 7>(r.cw, row[0], row[1]);
            else idct8x8_products_row<ARITH, 0>(r.cw, row[0], row[1]);
        } else {
            uint32_t out[16];
            W::transform_block(lds, b.comp, r.cw, out);
            row[0] = b.part == 1u ? out[14] : out[0];
            row[1] = b.part == 1u ? out[15] : out[1];
        }
        const typename W::EdgeFix ef = W::template edge_fix<1>(g, x0m, b.cx, row);
        uint8_t *p = tile0 + (b.part == 1u ? 0u : 9u) * lds.cpitch;
        *reinterpret_cast<v2u *>(p) = v2u{row[0], row[1]};
        W::edge_bytes(ef, p, row[0], row[1]);
    }

    // t' = 3*near + far + 2 per 16-bit lane (src/upsampler.rs:209,217).  CENTRED = false: as it is (the sample is wanted);
    // true: minus 512, so that the horizontal step yields the sample minus 128 (what the colour conversion wants: PixelOps::tprime)
    template <bool CENTRED>
    static __device__ __forceinline__ typename P::TPrime tprime(const typename P::ChromaEO &n, const typename P::ChromaEO &f) {
        const uint32_t two = CENTRED ? 0xfe02fe02u : 0x00020002u;
        typename P::TPrime t;
        t.tE1 = pk_add(pk_mad3(n.E1, f.E1), two);
        t.tO1 = pk_add(pk_mad3(n.O1, f.O1), two);
        t.tOm = pk_add(pk_mad3(n.Om, f.Om), two);
        t.tEp = pk_add(pk_mad3(n.Ep, f.Ep), two);
        return t;
    }
    // the horizontal step for the eight pixels of a chunk: 16-bit lanes (px4,px0) (px5,px1) (px6,px2) (px7,px3) holding
    // 3*t'main + t'other = sample << 4 (+ fraction; the 8 of the rounding is 3 * 2 + 2, already in the t') (src/upsampler.rs:219-224)
    static __device__ __forceinline__ void hstep(const typename P::TPrime &q, uint32_t (&m)[4]) {
        m[0] = pk_mad3(q.tE1, q.tOm), m[1] = pk_mad3(q.tE1, q.tO1), m[2] = pk_mad3(q.tO1, q.tE1), m[3] = pk_mad3(q.tO1, q.tEp);
    }

    // One output row of one 8-pixel chunk -> 32 bytes.  full[l] = the eight samples of full-size component slot l (two dwords),
    // half[h][i] = hstep lanes of half-size component slot h (CMYK: plain samples; YCCK: Cb, Cr centred, K plain).
    static __device__ __forceinline__ void emit_row(const FusedGeom &g, JP_GLOBAL uint8_t *o, const v2u (&full)[NL], const uint32_t (&half)[NH][4], uint32_t n) {
        uint32_t px[8];
        if (g.color == FCOLOR_CMYK) {  // src/decoder.rs:1458-1474: 255 - x, four times
            // bytes (px i, 0, px i+4, 0) per half-size component; pixel = (c0, c1, c2, c3) by byte permutes
            uint32_t hb[NH][4];
#pragma unroll
            for (uint32_t h = 0; h < NH; h++)
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) hb[h][i] = pk_shr(half[h][i], 4);
#pragma unroll
            for (uint32_t kk = 0; kk < 8; kk++) {
                const uint32_t i = kk & 3u, hsel = kk < 4 ? 0x00u : 0x02u;  // byte of hb[..][i] that holds pixel kk
                const uint32_t y0 = kk < 4 ? full[0].x : full[0].y;
                // low half: (c0, c1): c0 = byte i of y0, c1 = hb[0][i].byte(hsel)
                const uint32_t lo = perm_b32(hb[0][i], y0, 0x0c0c0000u | ((4u + hsel) << 8) | i);  // (.., .., c1, c0)
                uint32_t hi;
                if constexpr (K_FULL) {
                    const uint32_t y1 = kk < 4 ? full[1].x : full[1].y;
                    hi = perm_b32(y1, hb[1][i], 0x0c0c0000u | ((4u + i) << 8) | hsel);  // (.., .., c3 = byte i of y1, c2)
                } else {
                    hi = perm_b32(hb[2][i], hb[1][i], 0x0c0c0000u | ((4u + hsel) << 8) | hsel);  // (.., .., c3, c2)
                }
                px[kk] = ~(lo | (hi << 16));
            }
        } else {  // YCCK, src/decoder.rs:1439-1456: YCbCr -> RGB on the first three, 255 - k
            const w32 yb[8] = {byte_shl20<0>(full[0].x), byte_shl20<1>(full[0].x), byte_shl20<2>(full[0].x), byte_shl20<3>(full[0].x),
                               byte_shl20<0>(full[0].y), byte_shl20<1>(full[0].y), byte_shl20<2>(full[0].y), byte_shl20<3>(full[0].y)};
#pragma unroll
            for (uint32_t kk = 0; kk < 8; kk++) {
                const uint32_t i = kk & 3u;
                const int32_t cb = kk < 4 ? ((int32_t)(half[0][i] << 16) >> 20) : ((int32_t)half[0][i] >> 20);
                const int32_t cr = kk < 4 ? ((int32_t)(half[1][i] << 16) >> 20) : ((int32_t)half[1][i] >> 20);
                const RawRgb p = ycbcr_raw_centred(yb[kk], cb, cr);
                uint32_t kv;
                if constexpr (K_FULL) kv = byte_of(kk < 4 ? full[1].x : full[1].y, i);
                else kv = kk < 4 ? ((half[2][i] >> 4) & 0xffu) : ((half[2][i] >> 20) & 0xffu);
                px[kk] = sar_sat_u8x4<20>(p.r, p.g, p.b, (w32)((255u - kv) << 20));
            }
        }
        if (n == 8u) {
            *reinterpret_cast<JP_GLOBAL r4_v4u_a4 *>(o) = v4u{px[0], px[1], px[2], px[3]};
            *reinterpret_cast<JP_GLOBAL r4_v4u_a4 *>(o + 16) = v4u{px[4], px[5], px[6], px[7]};
        } else {
#pragma unroll
            for (uint32_t kk = 0; kk < 8; kk++)
                if (kk < n) reinterpret_cast<JP_GLOBAL uint32_t *>(o)[kk] = px[kk];
        }
    }

    // Output rows 16k .. 16k+15: slot p (0..7) emits rows 2p and 2p+1, which share their NEAR row of the half-size tiles —
    // plane row 8k+p = tile row p+1 — and take tile row p / p+2 as the far one (src/upsampler.rs:200-206: far = near -+ 1,
    // clamped into the plane).  8 slots x 2*te chunks: tiles of 16 MCUs make exactly 256 units, one per lane.
    static __device__ __forceinline__ void colour(const FusedGeom &g, const FusedImage &img, uint32_t tile, uint32_t k, uint32_t tid, const Lds &lds) {
        const uint32_t x0m = tile * g.tx, te = txe(g, tile);
        const uint32_t nch = 2u * te, nunits = 8u * nch;
        const uint32_t magic = 0xffffffffu / nch + 1u;  // mul_hi(u, magic) == u / nch for u < 65536
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)img.out;
        const size_t pitch = (size_t)g.out_w * 4u;
        const bool ycck = g.color != FCOLOR_CMYK;  // (uniform)
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t slot = __umulhi(u, magic), chk = u - slot * nch;
            const uint32_t oya = 16u * k + 2u * slot, oyb = oya + 1u;
            const uint32_t ox0 = 16u * x0m + 8u * chk;
            if (oya >= g.out_h || ox0 >= g.out_w) continue;
            const bool vb = oyb < g.out_h;
            const uint32_t n = min(8u, g.out_w - ox0);
            const uint32_t near = 8u * k + slot;  // plane row; tile row slot + 1
            const uint32_t N = slot + 1u, U = near > 0u ? slot : N, D = near + 1u <= g.ch - 1u ? slot + 2u : N;
            const uint32_t coff = 4u * chk + 4u;  // tile column of plane column j0 - 4
            uint32_t ha[NH][4], hb[NH][4];
#pragma unroll
            for (uint32_t h = 0; h < NH; h++) {
                const uint8_t *t0 = lds.ctile + h * 10u * lds.cpitch + coff;
                const typename P::ChromaEO en = P::load_eo(t0 + N * lds.cpitch), eu = P::load_eo(t0 + U * lds.cpitch),
                                           ed = P::load_eo(t0 + D * lds.cpitch);
                if (ycck && h < 2u) {
                    hstep(tprime<true>(en, eu), ha[h]);
                    hstep(tprime<true>(en, ed), hb[h]);
                } else {
                    hstep(tprime<false>(en, eu), ha[h]);
                    hstep(tprime<false>(en, ed), hb[h]);
                }
            }
            v2u fa[NL], fb[NL];
#pragma unroll
            for (uint32_t l = 0; l < NL; l++) {
                const uint8_t *py = lds.ytile + (l * 16u + 2u * slot) * lds.ypitch + 8u * chk;
                fa[l] = *reinterpret_cast<const v2u *>(py);
                fb[l] = *reinterpret_cast<const v2u *>(py + lds.ypitch);
            }
            emit_row(g, out + (size_t)oya * pitch + (size_t)ox0 * 4u, fa, ha, n);
            if (vb) emit_row(g, out + (size_t)oyb * pitch + (size_t)ox0 * 4u, fb, hb, n);
        }
    }
};

}  // namespace jpgpu
