// fused_x4.hpp — fused kernel for FOUR-component frames with subsampled components (SURVEY §8a rows a11 + a15):
//   components 0 and (K_FULL) 3 at full size, components 1, 2 and (!K_FULL) 3 at half size in both directions (H2V2,
//   src/upsampler.rs:191-228), colour function CMYK (255 - x on all four, src/decoder.rs:1458-1474) or YCCK (YCbCr -> RGB on the
//   first three, 255 - k, src/decoder.rs:1439-1456) -> CMYK32.
// The layouts of the reference's own fixture tests/reftest/images/mozilla/jpg-cmyk-2.jpg (sampling 22 11 11 11: the M, Y, K planes
// go through UpsamplerH2V2) and of YCCK files as Photoshop writes them (22 11 11 22).
// History: the generic kernel pair (planes through HBM, 25-28 % of the roofline); rounds 3-4 a ROW kernel — a workgroup owned tx
// MCUs of ONE MCU row and transformed, besides its own blocks, one sample row of every half-size block above and below (0.46 / 0.52
// of the roofline, 1.85 M vector instructions per 1080p image: profiles/round4/13_*); round 5 the strip walk below (0.62 / 0.62:
// profiles/round5/17_*), the row kernel is in the git history.
#pragma once
#include "fused_core.hpp"

namespace jpgpu {

#ifdef JPGPU_HOST_EMULATION
typedef v4u x4_v4u_a4;
#else
typedef v4u x4_v4u_a4 __attribute__((aligned(4)));  // a 4-byte pixel is all the alignment an output row has
#endif

// Which components are which, and the colour function of one output row of one 8-pixel chunk.
template <int ARITH, bool K_FULL>
struct X4Colour {
    typedef PixelOps<ARITH> P;        // ChromaEO / load_eo
    static constexpr uint32_t NL = K_FULL ? 2u : 1u, NH = K_FULL ? 2u : 3u;
    // frame component of full-size slot l / half-size slot h
    static __device__ __forceinline__ uint32_t lcomp(uint32_t l) { return l == 0u ? 0u : 3u; }
    static __device__ __forceinline__ uint32_t hcomp(uint32_t h) { return 1u + h; }

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
            *reinterpret_cast<JP_GLOBAL x4_v4u_a4 *>(o) = v4u{px[0], px[1], px[2], px[3]};
            *reinterpret_cast<JP_GLOBAL x4_v4u_a4 *>(o + 16) = v4u{px[4], px[5], px[6], px[7]};
        } else {
#pragma unroll
            for (uint32_t kk = 0; kk < 8; kk++)
                if (kk < n) reinterpret_cast<JP_GLOBAL uint32_t *>(o)[kk] = px[kk];
        }
    }
};

// =============================================================================================
// W4: a STRIP WALK — S420's shape (fused_core.hpp) with NL full-size and NH half-size components.
// A workgroup owns a strip of tx MCU columns and walks the MCU rows [k0, k1) of it top to bottom; per MCU row it transforms the
// 4*NL*te full-size blocks and the NH*(te+2) half-size blocks under them (one halo block either side), ONE lane per block, and keeps
// the samples in LDS tiles; the half-size components' neighbour rows come from the step before (carry rows) instead of partial
// transforms of the block rows above and below — what made the row kernel pay 1.85 M vector instructions per 1080p image where the
// 4:2:0 walk pays 1.05 M.  Step k emits output rows 16k-1 .. 16k+14; a segment that starts below the
// image's top / ends above its bottom gets the one sample row it needs of the block rows k0-1 / k1 in a seam round (S420's).
// Lanes: 4*NL*te + NH*(te+2) <= 256 — te <= 35 (C at full size, M Y K at half: jpg-cmyk-2.jpg's layout), te <= 25 (Y and K full).
// =============================================================================================
template <uint32_t NL, uint32_t NH>
struct W4Lds {
    uint8_t *stage;   // one 128-B slot per block; later the tiles:
    uint8_t *ytile;   //   NL x 17 rows x ypitch: row 0 = full-size row 16k-1 (carry), rows 1..16 the step's own
    uint8_t *ctile;   //   NH x 9 rows x cpitch: row 0 = half-size row 8k-1 (carry), rows 1..8 the step's own; column lc <-> plane column 8*(x0m-1) + lc
    uint8_t *carry;   // NL*ypitch + NH*cpitch: the last rows of the step before (half-size part: the seam row at a segment start)
    uint8_t *bnd;     // NH*cpitch: half-size row 8*k1 (the seam below the segment)
    uint8_t *qtab;    // 4 x 128 B
    uint32_t ypitch, cpitch;
    static __device__ __host__ __forceinline__ uint32_t blocks(uint32_t tx) { return 4u * NL * tx + NH * (tx + 2u); }
    // (the seam round stages 2*NH*(tx+2) blocks: more than a step's when the strip is narrow)
    static __device__ __host__ __forceinline__ uint32_t stage_bytes(uint32_t tx) {
        const uint32_t a = blocks(tx), b = 2u * NH * (tx + 2u);
        return (a > b ? a : b) * 128u;
    }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx) {
        return stage_bytes(tx) + (NL * 16u * tx + NH * 8u * (tx + 2u)) + NH * 8u * (tx + 2u) + 512u;
    }
    static __device__ __forceinline__ W4Lds make(uint8_t *base, uint32_t tx) {
        W4Lds l;
        l.ypitch = 16u * tx;
        l.cpitch = 8u * (tx + 2u);
        l.stage = base;
        l.ytile = base;
        l.ctile = base + NL * 17u * l.ypitch;  // NL*272*tx + NH*72*(tx+2) <= (4*NL*tx + NH*(tx+2)) * 128
        l.carry = base + stage_bytes(tx);
        l.bnd = l.carry + NL * l.ypitch + NH * l.cpitch;
        l.qtab = l.bnd + NH * l.cpitch;
        return l;
    }
};
// (30, not the 35 the lanes would allow: one load round less per plane and the body stays inside 128 registers — with 35 two spilled
// registers gave the kernel a scratch segment, and a kernel with scratch is dispatched more slowly even where no wave touches it)
constexpr uint32_t w4_tx_max(bool k_full) { return k_full ? 25u : 30u; }

template <int ARITH, bool K_FULL>
struct W4 {
    typedef X4Colour<ARITH, K_FULL> C4;  // lcomp / hcomp, emit_row / hstep / tprime: the colour function
    typedef S420<ARITH, 256> W;        // fetch_block / transform_block / edge_fix (they look at lds.stage and lds.qtab only)
    typedef PixelOps<ARITH> P;
    static constexpr uint32_t NT = 256, NL = C4::NL, NH = C4::NH, TXM = w4_tx_max(K_FULL);
    typedef W4Lds<NL, NH> Lds;
    static constexpr uint32_t LY = (16u * TXM + NT - 1u) / NT;         // 3 / 2
    static constexpr uint32_t LC = (8u * (TXM + 2u) + NT - 1u) / NT;   // 2 / 1
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t strip) { return min(g.tx, g.mcu_w - strip * g.tx); }

    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t tid, const Lds &lds) {
        if (tid < 32u) {
            uint32_t *d = reinterpret_cast<uint32_t *>(lds.qtab);
            d[tid] = ((const JP_GLOBAL uint32_t *)img.qt[0])[tid];
            d[32u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[1])[tid];
            d[64u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[2])[tid];
            d[96u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[3])[tid];
        }
    }

    // Staging block index = lane that transforms it: per full-size component 2*te blocks of block row 2k, then 2*te of row 2k+1;
    // then per half-size component its te+2 blocks.
    struct Pre {
        v4u y[2 * NL][LY], c[NH][LC];
    };
    static __device__ __forceinline__ void stage_load(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k, uint32_t tid, Pre &pre) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nl = 16u * te, ncc = 8u * (te + 2u);
        auto at = [](const JP_GLOBAL v4u *base, uint32_t chunk) -> v4u {
            return *reinterpret_cast<const JP_GLOBAL v4u *>(reinterpret_cast<const JP_GLOBAL uint8_t *>(base) + chunk * 16u);
        };
#pragma unroll
        for (uint32_t l = 0; l < NL; l++) {
            const JP_GLOBAL v4u *y0 = (const JP_GLOBAL v4u *)img.coefs[C4::lcomp(l)] + ((size_t)(2u * k) * g.bw0 + 2u * x0m) * 8u;
            const JP_GLOBAL v4u *y1 = y0 + (size_t)g.bw0 * 8u;
#pragma unroll
            for (uint32_t i = 0; i < LY; i++) {
                const uint32_t j = min(tid + NT * i, nl - 1u);  // clamped: unconditional loads
                pre.y[2 * l][i] = at(y0, j);
                pre.y[2 * l + 1][i] = at(y1, j);
            }
        }
        // halo blocks outside the plane (image edges) are never transformed: clamp them onto valid chunks
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
#pragma unroll
        for (uint32_t h = 0; h < NH; h++) {
            const JP_GLOBAL v4u *c = (const JP_GLOBAL v4u *)img.coefs[C4::hcomp(h)] + (size_t)k * g.bwc * 8u;
#pragma unroll
            for (uint32_t i = 0; i < LC; i++) {
                const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid + NT * i, ncc - 1u), 0), cmax);
                pre.c[h][i] = at(c, e);
            }
        }
    }
    static __device__ __forceinline__ void stage_store(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, const Pre &pre) {
        const uint32_t te = txe(g, strip), nl = 16u * te, nb = te + 2u, ncc = 8u * nb;
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
#pragma unroll
        for (uint32_t h = 0; h < NH; h++)
#pragma unroll
            for (uint32_t i = 0; i < LC; i++)
                if (tid + NT * i < ncc) dst[coef_slot(4u * te * NL + h * nb + b + (NT / 8u) * i, row)] = pre.c[h][i];
    }

    struct Blk {
        uint32_t comp;  // frame component
        uint32_t slot;  // index among the full-size / half-size components
        uint32_t ry, cx;
        bool full_size, valid;
    };
    static __device__ __forceinline__ Blk lane_block(const FusedGeom &g, uint32_t strip, uint32_t tid) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nb = te + 2u;
        Blk b{};
        if (tid < 4u * te * NL) {
            b.full_size = true;
            b.slot = tid >= 4u * te ? 1u : 0u;
            const uint32_t t = tid - b.slot * 4u * te;
            b.ry = t >= 2u * te ? 1u : 0u;
            b.cx = t - b.ry * 2u * te;
            b.comp = C4::lcomp(b.slot);
            b.valid = true;
            return b;
        }
        const uint32_t t = tid - 4u * te * NL;
        if (t >= NH * nb) return b;
        b.slot = t >= nb ? (t >= 2u * nb ? 2u : 1u) : 0u;
        b.cx = t - b.slot * nb;
        b.comp = C4::hcomp(b.slot);
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)b.cx;
        b.valid = bx >= 0 && bx < (int32_t)g.bwc;
        return b;
    }

    // where byte o of the carry rows lies in row 0 of the tiles (o < NL*ypitch + NH*cpitch)
    static __device__ __forceinline__ uint8_t *row0_of(const Lds &lds, uint32_t o) {
        if (o < NL * lds.ypitch) {
            const uint32_t l = (NL > 1u && o >= lds.ypitch) ? 1u : 0u;
            return lds.ytile + l * 17u * lds.ypitch + (o - l * lds.ypitch);
        }
        const uint32_t oc = o - NL * lds.ypitch;
        const uint32_t h = oc >= lds.cpitch ? (oc >= 2u * lds.cpitch ? 2u : 1u) : 0u;
        return lds.ctile + h * 9u * lds.cpitch + (oc - h * lds.cpitch);
    }

    static __device__ __forceinline__ void read_block(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, S420Regs &r) {
        if (tid * 8u < NL * lds.ypitch + NH * lds.cpitch) r.carry = *reinterpret_cast<const v2u *>(lds.carry + tid * 8u);
        const Blk b = lane_block(g, strip, tid);
        if (!b.valid) return;
        W::fetch_block(lds, tid, b.comp, r.cw);
    }

    static __device__ __forceinline__ void transform(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, S420Regs &r) {
        if (tid * 8u < NL * lds.ypitch + NH * lds.cpitch) *reinterpret_cast<v2u *>(row0_of(lds, tid * 8u)) = r.carry;  // carry rows -> row 0 of the tiles
        const Blk b = lane_block(g, strip, tid);
        if (!b.valid) return;
        uint32_t out[16];
        W::transform_block(lds, b.comp, r.cw, out);
        uint8_t *base = b.full_size ? lds.ytile + (b.slot * 17u + 1u + b.ry * 8u) * lds.ypitch + b.cx * 8u
                                    : lds.ctile + (b.slot * 9u + 1u) * lds.cpitch + b.cx * 8u;
        const uint32_t pitch = b.full_size ? lds.ypitch : lds.cpitch;
        const typename W::EdgeFix ef = b.full_size ? typename W::EdgeFix{false, false} : W::edge_fix(g, strip * g.tx, b.cx, out);
#pragma unroll
        for (int row = 0; row < 8; row++) *reinterpret_cast<v2u *>(base + (uint32_t)row * pitch) = v2u{out[2 * row], out[2 * row + 1]};
        uint8_t *cy = b.full_size ? lds.carry + b.slot * lds.ypitch + b.cx * 8u : lds.carry + NL * lds.ypitch + b.slot * lds.cpitch + b.cx * 8u;
        if (!b.full_size || b.ry == 1u) *reinterpret_cast<v2u *>(cy) = v2u{out[14], out[15]};  // what the next step finds in front of its own rows
        if (ef.before) {
#pragma unroll
            for (int row = 0; row < 8; row++) (base + (uint32_t)row * pitch)[-1] = (uint8_t)out[2 * row];
            cy[-1] = (uint8_t)out[14];
        }
        if (ef.after) {
#pragma unroll
            for (int row = 0; row < 8; row++) (base + (uint32_t)row * pitch)[8] = (uint8_t)(out[2 * row + 1] >> 24);
            cy[8] = (uint8_t)(out[15] >> 24);
        }
    }

    // ---- segment seams: the half-size blocks of block rows k0-1 (their last sample row -> carry) and k1 (their first sample row
    // -> bnd).  Staging block index = lane: per half-size component te+2 blocks above, then the same below.
    static __device__ __forceinline__ void seam_stage(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k0, uint32_t k1, uint32_t tid,
                                                      const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nb = te + 2u, ncc = 8u * nb;
        const bool above = k0 > 0u, below = k1 < g.mcu_h;
        const size_t ra = (size_t)(above ? k0 - 1u : 0u) * g.bwc * 8u, rb = (size_t)(below ? k1 : 0u) * g.bwc * 8u;
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
        v4u v[2 * NH][LC];
#pragma unroll
        for (uint32_t w = 0; w < 2u * NH; w++) {
            const JP_GLOBAL v4u *run = (const JP_GLOBAL v4u *)img.coefs[C4::hcomp(w % NH)] + (w < NH ? ra : rb);
#pragma unroll
            for (uint32_t i = 0; i < LC; i++) {
                const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid + NT * i, ncc - 1u), 0), cmax);
                v[w][i] = run[e];
            }
        }
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
        const uint32_t row = tid & 7u, b = tid >> 3;
#pragma unroll
        for (uint32_t w = 0; w < 2u * NH; w++)
#pragma unroll
            for (uint32_t i = 0; i < LC; i++)
                if (tid + NT * i < ncc) dst[coef_slot(w * nb + b + (NT / 8u) * i, row)] = v[w][i];
    }
    static __device__ __forceinline__ void seam_transform(const FusedGeom &g, uint32_t strip, uint32_t k0, uint32_t k1, uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nb = te + 2u;
        if (tid >= 2u * NH * nb) return;
        uint32_t which = 0;
#pragma unroll
        for (uint32_t w = 1; w < 2u * NH; w++) which += tid >= w * nb ? 1u : 0u;
        const uint32_t cx = tid - which * nb, h = which >= NH ? which - NH : which;
        const bool below = which >= NH;
        if (below ? !(k1 < g.mcu_h) : !(k0 > 0u)) return;
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)cx;
        if (bx < 0 || bx >= (int32_t)g.bwc) return;
        uint32_t cw[32];
        W::fetch_block(lds, tid, C4::hcomp(h), cw);
        uint8_t *dst = below ? lds.bnd + h * lds.cpitch + cx * 8u : lds.carry + NL * lds.ypitch + h * lds.cpitch + cx * 8u;
        uint32_t lo, hi;
        typename W::EdgeFix ef;
        if constexpr (ARITH != ARITH_EXACT) {
            if (below) idct8x8_products_row<ARITH, 0>(cw, lo, hi);
            else idct8x8_products_row<ARITH, 7>(cw, lo, hi);
            uint32_t row[16] = {lo, hi, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            ef = W::template edge_fix<1>(g, x0m, cx, row);
            lo = row[0], hi = row[1];
        } else {
            uint32_t out[16];
            W::transform_block(lds, C4::hcomp(h), cw, out);
            ef = W::edge_fix(g, x0m, cx, out);
            lo = below ? out[0] : out[14], hi = below ? out[1] : out[15];
        }
        *reinterpret_cast<v2u *>(dst) = v2u{lo, hi};
        W::edge_bytes(ef, dst, lo, hi);
    }
    // after the last step: the carry rows and the seam row below become rows 0 / 1 of the tiles for the closing call
    static __device__ __forceinline__ void closing_tiles(uint32_t tid, const Lds &lds) {
        const uint32_t o = tid * 8u;
        if (o < NL * lds.ypitch + NH * lds.cpitch) *reinterpret_cast<v2u *>(row0_of(lds, o)) = *reinterpret_cast<const v2u *>(lds.carry + o);
        if (o < NH * lds.cpitch) {
            const uint32_t h = o >= lds.cpitch ? (o >= 2u * lds.cpitch ? 2u : 1u) : 0u, x = o - h * lds.cpitch;
            *reinterpret_cast<v2u *>(lds.ctile + (h * 9u + 1u) * lds.cpitch + x) = *reinterpret_cast<const v2u *>(lds.bnd + o);
        }
    }

    // Output rows 16k-1 .. 16k+14 of the strip: slot p (0..7) pairs half-size tile rows (p, p+1) = plane rows 8k-1+p, 8k+p and emits
    // full-size tile rows 2p (near = the upper half-size row) and 2p+1 (near = the lower one).  Rows above `row_lo` belong to the
    // workgroup of the segment above.  closing: only slot 0 (the segment's last output row, k = k1).
    static __device__ __forceinline__ void colour(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k, uint32_t row_lo, bool closing,
                                                  uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip);
        const uint32_t nch = 2u * te, nunits = (closing ? 1u : 8u) * nch;
        const uint32_t magic = 0xffffffffu / nch + 1u;  // mul_hi(u, magic) == u / nch for u < 65536
        const size_t pitch = (size_t)g.out_w * 4u;
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)img.out;
        const bool ycck = g.color != FCOLOR_CMYK;  // (uniform)
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t slot = __umulhi(u, magic), chk = u - slot * nch;
            const int32_t oya = 16 * (int32_t)k - 1 + 2 * (int32_t)slot;
            const uint32_t oyb = (uint32_t)(oya + 1);
            const bool va = oya >= (int32_t)row_lo && (uint32_t)oya < g.out_h, vb = !closing && oyb < g.out_h;
            const uint32_t ox0 = 16u * x0m + 8u * chk;
            if ((!va && !vb) || ox0 >= g.out_w) continue;
            const uint32_t n = min(8u, g.out_w - ox0);
            const int32_t cu = 8 * (int32_t)k - 1 + (int32_t)slot;  // plane row of the slot's upper half-size row
            // row a: near U, far min(near+1, ch-1);  row b: near L, far max(near-1, 0) (src/upsampler.rs:200-206)
            const bool clamp_a = cu + 1 > (int32_t)g.ch - 1, clamp_b = cu < 0;
            const uint32_t U = clamp_b ? slot + 1u : slot, L = clamp_a ? slot : slot + 1u;
            const uint32_t coff = 4u * chk + 4u;  // tile column of plane column j0 - 4
            uint32_t ha[NH][4], hb[NH][4];
#pragma unroll
            for (uint32_t h = 0; h < NH; h++) {
                const uint8_t *t0 = lds.ctile + h * 9u * lds.cpitch + coff;
                const typename P::ChromaEO eu = P::load_eo(t0 + U * lds.cpitch), el = P::load_eo(t0 + L * lds.cpitch);
                if (ycck && h < 2u) {
                    C4::hstep(C4::template tprime<true>(eu, el), ha[h]);
                    C4::hstep(C4::template tprime<true>(el, eu), hb[h]);
                } else {
                    C4::hstep(C4::template tprime<false>(eu, el), ha[h]);
                    C4::hstep(C4::template tprime<false>(el, eu), hb[h]);
                }
            }
            v2u fa[NL], fb[NL];
#pragma unroll
            for (uint32_t l = 0; l < NL; l++) {
                const uint8_t *py = lds.ytile + (l * 17u + 2u * slot) * lds.ypitch + 8u * chk;
                fa[l] = *reinterpret_cast<const v2u *>(py);
                fb[l] = *reinterpret_cast<const v2u *>(py + lds.ypitch);
            }
            if (va) C4::emit_row(g, out + (size_t)oya * pitch + (size_t)ox0 * 4u, fa, ha, n);
            if (vb) C4::emit_row(g, out + (size_t)oyb * pitch + (size_t)ox0 * 4u, fb, hb, n);
        }
    }
};

}  // namespace jpgpu
