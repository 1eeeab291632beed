// fused_core.hpp — bodies of the fused fast-path kernels (dct_scale 8): coefficients in, interleaved pixels out, ONE launch per
// kind, nothing but those two arenas in HBM.
//
//   S420  (FUSED_420) : 4:2:0 YCbCr (H2V2 / H1V1 / H1V1) -> RGB24, a strip walk (below).
//   S440  (FUSED_440) : 4:4:0 YCbCr (UpsamplerH1V2), the same walk without a horizontal halo.
//   FGen  (FUSED_GEN) : the UpsamplerGeneric layouts (4:1:1, 4:1:0, 1x4, 2x4, 4x4), tile = tx MCUs of one MCU row.
//   F422  (FUSED_422) : 4:2:2 YCbCr (UpsamplerH2V1), tile kernel, one halo block either side.
//   F444  (FUSED_444) : 4:4:4 YCbCr or RGB -> RGB24, CMYK / YCCK at 4:4:4 -> 32-bit pixels; no halo.
//   FGray (FUSED_GRAY): 1 component -> L8, IDCT written straight to the output rows (compute_image's stride compaction,
//                       src/decoder.rs:1310-1332, is just the output pitch).
//   (fused_x4.hpp: four components with half-size ones; fused_scaled.hpp: reduced-size decodes.)
//   PixelOps<ARITH>   : upsample + colour + store helpers shared by the 4:2:0 and 4:2:2 kernels (src/upsampler.rs:134-228,
//                       src/decoder.rs:1406-1437).
// (Round 1's 4:2:0 was two launches — a chroma pass into u8 planes and a main pass that read their neighbourhood back, 4.11 GB
// of traffic against 3.26 — kept through round 3 as the A/B partner of the walk and deleted in round 4.)
//
// Each kernel is a sequence of barrier-separated phases written as plain functions of
// (geometry, image, tile, tid, LDS, per-lane registers) so that tests/emu can run the very same
// code on the CPU (g++ -DJPGPU_HOST_EMULATION) against the oracle.  On the GPU the phases are
// called back to back from the __global__ wrappers in fused.hip with __syncthreads() between.
#pragma once
#include "pixel_math.hpp"

namespace jpgpu {

constexpr uint32_t FUSED_NT = 256;       // threads per workgroup
constexpr uint32_t F444_TX_MAX = 64;     // one wave per component, one lane per block
constexpr uint32_t F422_TX_MAX = 62;     // 2 luma waves (124 blocks), one wave per chroma component (62 + 2 halo blocks)
constexpr uint32_t FGRAY_TX_MAX = 256;   // 1 block per MCU
constexpr uint32_t FUSED_COEF_LDS = 256 * 128;          // staging area (bytes), aliased by the sample tiles

enum : uint32_t { FCOLOR_YCBCR = 0, FCOLOR_RGB = 1, FCOLOR_CMYK = 2, FCOLOR_YCCK = 3 };  // the last two: four components
enum FusedKind : int { FUSED_NONE = 0, FUSED_420 = 1, FUSED_444 = 2, FUSED_GRAY = 3, FUSED_422 = 4, FUSED_440 = 5, FUSED_GEN = 6,
                       FUSED_420X4 = 7 };  // four components, some of them at half size (fused_x4.hpp)

struct FusedGeom {
    uint32_t kind;
    uint32_t out_w, out_h;
    uint32_t mcu_w, mcu_h;
    uint32_t tx;       // MCUs per tile
    uint32_t tiles_x;  // ceil(mcu_w / tx)
    uint32_t bw0;      // block_width of component 0
    uint32_t bwc;      // block_width of the chroma components (420) / all components (444)
    uint32_t cw, ch;   // chroma component size (420)
    uint32_t color;    // FCOLOR_* (444)
    uint32_t strip;    // 1: a strip walk (S420, S440): work items are (strip, MCU rows [k0, k1))
    uint32_t seg_rows; // S420: MCU rows per workgroup
    uint32_t n_seg;    // S420: ceil(mcu_h / seg_rows)
    uint32_t hs, vs;   // FGen: log2 of the luma sampling factors (H x V luma blocks per MCU)
    uint32_t k_full;   // W4 (fused_x4.hpp): component 3 is at full size (sampling 22 11 11 22) instead of half size (22 11 11 11)
};

// One work item of a fused launch: which image, and which of its tiles (meaning of a / b / c per kernel, fused.hip).
// Tile kernels: a = tile within the MCU row, b = MCU row.  Strip walks: a = strip, MCU rows [b, c) of it.
struct alignas(16) FusedWork {
    uint32_t image, a, b, c;
};

struct FusedImage {
    const int16_t *coefs[4];
    const uint16_t *qt[4];
    uint8_t *out;
    uint32_t flags;    // bit0: every component "sane" (|c*q| < 2^15), bit1: "tight" (column sums <= 5900) — pixel_math.hpp
    uint32_t _pad;
};

struct alignas(16) FusedLdsSmall {          // kernels without a chroma neighbourhood
    uint8_t coef[FUSED_COEF_LDS];
};

struct FusedRegs {
    uint32_t out[16];  // one IDCT'd block: 8 rows x 2 dwords
};

// wave-uniform value -> SGPR (lets the compiler keep per-wave pointers / tables scalar)
__device__ __forceinline__ uint32_t uniform(uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    return v;
#else
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// LDS slot (in 16-B units) of row k of local block lb: conflict-free for both the 8-lane
// ds_write_b128 groups (one block = 8 consecutive slots) and the 16-lane ds_read_b128 groups
// (MI355X_MICROARCH.md §LDS): lb*8 + (k ^ ((lb >> 1) & 7)).
__device__ __forceinline__ uint32_t coef_slot(uint32_t lb, uint32_t k) { return lb * 8u + (k ^ ((lb >> 1) & 7u)); }

__device__ __forceinline__ void load_block_from_lds(const uint8_t *coef_lds, uint32_t lb, uint32_t (&cw)[32]) {
    const v4u *p = reinterpret_cast<const v4u *>(coef_lds);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v4u v = p[coef_slot(lb, (uint32_t)k)];
        cw[k * 4 + 0] = v.x;
        cw[k * 4 + 1] = v.y;
        cw[k * 4 + 2] = v.z;
        cw[k * 4 + 3] = v.w;
    }
}

__device__ __forceinline__ uint32_t byte_of(uint32_t d, uint32_t i) { return (d >> (8u * i)) & 0xffu; }

// Stage up to 8*NT 16-B coefficient chunks of a tile into LDS (NT = threads per workgroup).  `addr(j)` maps the tile-local
// chunk index to a global pointer.  All eight loads of a lane are issued before its first LDS
// store (one exposed memory latency, not eight).
template <uint32_t NT, class AddrFn>
__device__ __forceinline__ void stage_coefficients(uint8_t *coef_lds, uint32_t nchunks, uint32_t tid, AddrFn addr) {
    v4u *dst = reinterpret_cast<v4u *>(coef_lds);
    const uint32_t lastc = nchunks - 1u;
    v4u v[8];  // (plain vector type: an array of HIP's uint4 class would be kept in scratch by hipcc)
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) v[i] = stream_load(addr(min(tid + NT * i, lastc)));  // clamped: unconditional loads
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t j = tid + NT * i;
        if (j <= lastc) dst[coef_slot(j >> 3, j & 7u)] = v[i];
    }
}

// The same for a tile made of several contiguous runs of chunks (block rows, components): every load instruction
// reads from ONE run — wave-uniform base pointer, the lane's chunk index as 32-bit offset — so the address math is
// scalar (a per-lane choice between run bases costs ~6 VALU per load).  load_run / store_run are split so that a
// caller can issue the loads of all its runs before the first LDS store.
template <uint32_t NT, uint32_t LOADS>
__device__ __forceinline__ void load_run(v4u (&v)[LOADS], const JP_GLOBAL v4u *run, uint32_t nchunks, uint32_t tid) {
#pragma unroll
    for (uint32_t i = 0; i < LOADS; i++) v[i] = stream_load(run + min(tid + NT * i, nchunks - 1u));  // clamped: unconditional
}
// first_block: LDS block index of the run's first block
template <uint32_t NT, uint32_t LOADS>
__device__ __forceinline__ void store_run(uint8_t *coef_lds, const v4u (&v)[LOADS], uint32_t nchunks, uint32_t first_block,
                                          uint32_t tid) {
    v4u *dst = reinterpret_cast<v4u *>(coef_lds);
#pragma unroll
    for (uint32_t i = 0; i < LOADS; i++)
        if (tid + NT * i < nchunks) dst[coef_slot(first_block + (tid >> 3) + (NT / 8u) * i, tid & 7u)] = v[i];
}

// =============================================================================================
// Pixel helpers of the kernels with horizontally subsampled chroma (4:2:0, 4:2:2)
// =============================================================================================
// SWAR helpers: two 16-bit lanes per dword (values stay below 2^12, so plain 32-bit adds and
// shifts never carry between lanes).
__device__ __forceinline__ uint32_t swar_even(uint32_t d) { return d & 0x00ff00ffu; }          // bytes 0,2
__device__ __forceinline__ uint32_t swar_odd(uint32_t d) { return (d >> 8) & 0x00ff00ffu; }    // bytes 1,3
__device__ __forceinline__ uint32_t swar_3a_b(uint32_t a, uint32_t b) { return (a << 1) + a + b; }

template <int ARITH>
struct PixelOps {
    // Chroma columns of one 8-pixel chunk as packed 16-bit lane pairs (hi, lo).  Plane column
    // j0 + i (j0 = ox0/2) is sample s_i; an output row needs s_-1 .. s_4, as the main samples of the even / odd pixel
    // pairs and their outer neighbours:
    //   E1 = (s2, s0)   O1 = (s3, s1)   Om = (s1, s_-1)   Ep = (s4, s2)
    // (the two shifted pairs are byte permutes of the loaded dwords: cheaper here, once per chroma row, than as funnel
    // shifts of the t' values, once per output row)
    struct ChromaEO {
        uint32_t E1, O1, Om, Ep;
    };
    static __device__ __forceinline__ ChromaEO load_eo(const uint8_t *row) {
        const uint32_t *d = reinterpret_cast<const uint32_t *>(row);  // dwords: cols j0-4.., j0.., j0+4..
        ChromaEO c;
        c.E1 = d[1] & 0x00ff00ffu;
        c.O1 = pk_shr(d[1], 8);
        c.Om = perm_b32(d[1], d[0], 0x0c050c03u);  // (d1.byte1, d0.byte3)
        c.Ep = perm_b32(d[2], d[1], 0x0c040c02u);  // (d2.byte0, d1.byte2)
        return c;
    }
    // t' = 3*near + far + 2 (src/upsampler.rs:209,217), kept as t'' = t' - 512 (mod 2^16): the horizontal step then yields
    // 3*t''main + t''other = (3*t'main + t'other) - 2048, whose arithmetic shift by 4 is the chroma sample MINUS 128 — the
    // form the colour conversion wants (src/decoder.rs:1489-1491), for the price of another constant in the same addition.
    //   tE1 = (s2, s0)  tO1 = (s3, s1)  tOm = (s1, s_-1)  tEp = (s4, s2)
    struct TPrime {
        uint32_t tE1, tO1, tOm, tEp;
    };
    static __device__ __forceinline__ TPrime tprime(const ChromaEO &n, const ChromaEO &f) {
        const uint32_t two = 0xfe02fe02u;  // 2 - 512 per lane
        TPrime t;
        t.tE1 = pk_add(pk_mad3(n.E1, f.E1), two);
        t.tO1 = pk_add(pk_mad3(n.O1, f.O1), two);
        t.tOm = pk_add(pk_mad3(n.Om, f.Om), two);
        t.tEp = pk_add(pk_mad3(n.Ep, f.Ep), two);
        return t;
    }

    // One output row of one 8-pixel chunk (src/upsampler.rs:191-228 + src/decoder.rs:1406-1437).
    //   pixel k: main sample s_(k>>1), other tap s_(k>>1)+-1:  c = (3*t'main + t'other) >> 4
    //   first / last column of the image: c = t'main >> 2
    // `o` = address of the chunk's first output byte (scanline start + 3*ox0; callers keep the scanline start wave-uniform
    // where they can -> scalar address math), `row_al4` = it is 4-byte aligned
    // H2V1 = true: `t` holds raw samples and only the horizontal step of UpsamplerH2V1 is applied
    // (src/upsampler.rs:134-163): c = (3*s_main + s_other + 2) >> 2, first / last column c = s_main.
    // Either way a 16-bit lane of pk[][] ends up holding (c - 128) << SH (plus fraction bits below SH): the two values of
    // a dword are taken apart, shifted and sign-extended by one v_bfe_i32 / v_ashrrev_i32 each.
    // EDGES = false: the caller's chroma tile already holds the first / last sample of a row once more in the column
    // outside the image; then the general formula IS the edge formula — (3*t + t) >> 4 == t >> 2, (3*s + s + 2) >> 2 == s —
    // and no lane needs the fix-up below.
    // FULL = true: the caller knows the chunk has all 8 pixels and a 4-byte aligned address (no byte-wise store path).
    // NTS: non-temporal stores — they pay in the two-pass kernels, whose chroma planes want the L2 for themselves, and cost
    // in the single-launch one (partial lines: pixel_math.hpp stream_store).
    template <bool H2V1 = false, bool EDGES = true, bool FULL = false, bool NTS = true>
    static __device__ __forceinline__ void row_pixels(const FusedGeom &g, JP_GLOBAL uint8_t *o, bool row_al4,
                                                      const TPrime (&t)[2], v2u yy, uint32_t ox0) {
        // pk[comp][0..3] = (px4,px0) (px5,px1) (px6,px2) (px7,px3) as (hi, lo) lanes
        constexpr uint32_t SH = H2V1 ? 2u : 4u;
        uint32_t pk[2][4];
#pragma unroll
        for (uint32_t comp = 0; comp < 2; comp++) {
            const TPrime &q = t[comp];
            uint32_t m[4] = {pk_mad3(q.tE1, q.tOm), pk_mad3(q.tE1, q.tO1), pk_mad3(q.tO1, q.tE1), pk_mad3(q.tO1, q.tEp)};
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) pk[comp][i] = H2V1 ? pk_add(m[i], 0xfe02fe02u) : m[i];
        }
        const uint32_t last_x = 2u * g.cw - 1u;
        if (EDGES && (ox0 == 0u || last_x - ox0 < 8u)) {  // rare: first / last image column
            // the lane of c = t'main >> 2 (tm = t''main as 16 bits), of c = s_main (tm = the sample) with H2V1
            auto edge_lane = [](uint32_t tm) -> uint32_t { return H2V1 ? ((tm << 2) - 512u) & 0xffffu : (tm << 2) & 0xfff0u; };
#pragma unroll
            for (uint32_t comp = 0; comp < 2; comp++) {
                if (ox0 == 0u)  // src/upsampler.rs:213-214: px0 = t'(s0) >> 2
                    pk[comp][0] = (pk[comp][0] & 0xffff0000u) | edge_lane(t[comp].tE1 & 0xffffu);
                if (last_x - ox0 < 8u) {  // src/upsampler.rs:226: last column (odd k): t'(s_(k>>1)) >> 2
                    const uint32_t k = last_x - ox0;
                    const uint32_t tm = k == 1u ? (t[comp].tE1 & 0xffffu) : k == 3u ? (t[comp].tO1 & 0xffffu)
                                        : k == 5u ? (t[comp].tE1 >> 16) : (t[comp].tO1 >> 16);
                    const uint32_t v = edge_lane(tm);
                    if (k == 1u) pk[comp][1] = (pk[comp][1] & 0xffff0000u) | v;
                    if (k == 3u) pk[comp][3] = (pk[comp][3] & 0xffff0000u) | v;
                    if (k == 5u) pk[comp][1] = (pk[comp][1] & 0x0000ffffu) | (v << 16);
                    if (k == 7u) pk[comp][3] = (pk[comp][3] & 0x0000ffffu) | (v << 16);
                }
            }
        }
        RawRgb p[8];
        const w32 yb[8] = {byte_shl20<0>(yy.x), byte_shl20<1>(yy.x), byte_shl20<2>(yy.x), byte_shl20<3>(yy.x),
                           byte_shl20<0>(yy.y), byte_shl20<1>(yy.y), byte_shl20<2>(yy.y), byte_shl20<3>(yy.y)};
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const int32_t cb = (k < 4) ? ((int32_t)(pk[0][k & 3u] << 16) >> (16u + SH)) : ((int32_t)pk[0][k & 3u] >> (16u + SH));
            const int32_t cr = (k < 4) ? ((int32_t)(pk[1][k & 3u] << 16) >> (16u + SH)) : ((int32_t)pk[1][k & 3u] >> (16u + SH));
            p[k] = ycbcr_raw_centred(yb[k], cb, cr);
        }
        const uint32_t n = FULL ? 8u : min(8u, g.out_w - ox0);
        if (FULL || (n == 8u && row_al4)) {
            uint32_t d0, d1, d2, d3, d4, d5;
            rgb4_to_12bytes(p[0], p[1], p[2], p[3], d0, d1, d2);
            rgb4_to_12bytes(p[4], p[5], p[6], p[7], d3, d4, d5);
            {
                if constexpr (NTS) {
                    stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(o), v3u{d0, d1, d2});
                    stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(o + 12), v3u{d3, d4, d5});
                } else {
                    *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o) = v3u{d0, d1, d2};
                    *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o + 12) = v3u{d3, d4, d5};
                }
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)  // unrolled + predicated: a runtime-indexed p[] would live in scratch
                if (k < n) {
                    o[3 * k] = (uint8_t)sar_sat_u8x2(p[k].r, 0u, 20);
                    o[3 * k + 1] = (uint8_t)sar_sat_u8x2(p[k].g, 0u, 20);
                    o[3 * k + 2] = (uint8_t)sar_sat_u8x2(p[k].b, 0u, 20);
                }
        }
    }
};

// =============================================================================================
// FUSED_420 ("strip walk"): a workgroup owns a strip of `tx` (<= 42) MCU columns and walks the MCU rows
// [k0, k1) of it top to bottom.  Per MCU row it stages the 4*te luma blocks AND the 2*(te+2) chroma blocks under them
// (one halo block each side: the fancy upsampler reads +-1 chroma sample) in LDS, transforms all of them (one lane
// per block), and keeps the samples in LDS tiles: chroma never makes a round trip through HBM (round 1's two-pass form
// paid 0.8 of its 4.1 GB per 256 x 1080p for it).
//   * Vertical neighbours.  Output row y reads chroma rows y/2 and y/2 -+ 1 (src/upsampler.rs:200-206), so step k emits
//     output rows 16k-1 .. 16k+14: the tiles have a row 0 in front of the step's own rows — luma row 16k-1 and chroma row
//     8k-1, carried over from step k-1 through a small LDS buffer — and row 16k+15 waits for step k+1.
//   * Segment seams.  A workgroup that starts below the top of the image needs chroma row 8*k0-1, one that ends above the
//     bottom needs chroma row 8*k1 for its last output row 16*k1-1: ONE extra transform round per workgroup (the chroma
//     blocks of block rows k0-1 and k1, of which a single sample row each is kept) supplies both.  Nobody emits
//     another workgroup's rows, so there is no luma warm-up.
//   * Quantization tables sit in LDS (lanes of one wave hold blocks of different components); for the classes whose
//     products fit i16 the block is multiplied row by row while it is copied out of the staging area, so table and raw
//     coefficients never occupy registers together.
//   * Pixel phase: the 8*nch (slot, 8-pixel chunk) units of a step — a slot = two output rows that share two chroma
//     rows — are dealt to the lanes in order; waves without a unit in the last round go straight to the barrier and leave
//     their SIMD to other workgroups, so only the remainder of ONE wave idles (tx = 40: none at all).
// The sample tiles alias the coefficient staging area (consumed into registers before they are written).
// =============================================================================================
struct S420Lds {
    uint8_t *stage;   // max(6*tx + 4, 4*(tx + 2)) blocks x 128 B coefficient staging; later the tiles:
    uint8_t *ytile;   //   17 rows x ypitch : row 0 = luma row 16k-1 (carry), rows 1..16 = the step's own rows
    uint8_t *ctile;   //   2 comps x 9 rows x cpitch : row 0 = chroma row 8k-1 (carry), rows 1..8 the step's own; column lc <-> plane column 8*(x0m-1) + lc
    uint8_t *carry;   // ypitch + 2*cpitch: last luma / chroma rows of the step before (chroma part: the seam row at a segment start)
    uint8_t *bnd;     // 2*cpitch: chroma row 8*k1 (seam below the segment)
    uint8_t *qtab;    // 3 x 128 B packed quantization tables
    uint32_t ypitch, cpitch;
    // (the seam round stages 4*(tx+2) chroma blocks: more than a step's 6*tx+4 when the strip is one MCU wide)
    static __device__ __host__ __forceinline__ uint32_t stage_bytes(uint32_t tx) { return (tx < 2u ? 4u * (tx + 2u) : 6u * tx + 4u) * 128u; }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx) {
        return stage_bytes(tx) + (16u * tx + 16u * (tx + 2u)) + 16u * (tx + 2u) + 384u;
    }
    static __device__ __forceinline__ S420Lds make(uint8_t *base, uint32_t tx) {
        S420Lds l;
        l.ypitch = 16u * tx;
        l.cpitch = 8u * (tx + 2u);
        l.stage = base;
        l.ytile = base;
        l.ctile = base + 17u * l.ypitch;  // 17*16*tx + 18*8*(tx+2) = 416*tx + 288 <= (6*tx+4)*128
        l.carry = base + stage_bytes(tx);
        l.bnd = l.carry + l.ypitch + 2u * l.cpitch;
        l.qtab = l.bnd + 2u * l.cpitch;
        return l;
    }
};
constexpr uint32_t S420_TX_MAX = 42;  // 4*tx luma + 2*(tx+2) chroma blocks <= 256 lanes (<= 20 with 128-thread workgroups)

struct S420Regs {
    uint32_t cw[32];  // the lane's block between read_block and transform: dequantized (packed products) or, for the exact class, raw
    v2u carry;        // 8 bytes of the carry rows on their way into row 0 of the tiles
};

template <int ARITH, uint32_t NTHREADS = 256>
struct S420 {
    typedef S420Lds Lds;
    typedef PixelOps<ARITH> P;  // pixel helpers (upsample + colour + store)
    static constexpr uint32_t NT = NTHREADS;  // 256: strips of <= 42 MCUs; 128: <= 20
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t strip) {
        return min(g.tx, g.mcu_w - strip * g.tx);
    }

    // once per workgroup: quantization tables -> LDS (made visible by the first barrier)
    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t tid, const Lds &lds) {
        if (tid < 32u) {  // (three fixed copies: a per-lane choice of img.qt[c] would put the image struct in scratch)
            uint32_t *d = reinterpret_cast<uint32_t *>(lds.qtab);
            d[tid] = ((const JP_GLOBAL uint32_t *)img.qt[0])[tid];
            d[32u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[1])[tid];
            d[64u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[2])[tid];
        }
    }

    // Stage the coefficients of MCU row k.  Four contiguous runs of 16-B chunks: luma block rows 2k and 2k+1
    // (16*te chunks each), Cb and Cr (8*(te+2) each, one halo block either side).  Every load instruction reads
    // from one run — wave-uniform base, the lane's chunk index as 32-bit offset — so the address math stays
    // scalar; the price is 3+3+2+2 = 10 loads with idle lanes in the last load of each run.
    // Staging block index = lane that transforms it: [0,2te) luma row 0, [2te,4te) luma row 1, then Cb, Cr.
    // Two halves: stage_load issues the loads (into registers), stage_store puts them into LDS.  (Issuing the loads of
    // step k+1 before the pixel phase of step k needs 40 more live VGPRs: spills at 4 workgroups per CU, and at 3 — 166
    // VGPRs, no spills — it measured 0.8 % better, the waves of the other workgroups cover the latency as it is:
    // profiles/round2/02b_phase_clocks.md.)
    static constexpr uint32_t LY = (16u * (NT == 256 ? S420_TX_MAX : 20u) + NT - 1u) / NT;        // 3
    static constexpr uint32_t LC = (8u * ((NT == 256 ? S420_TX_MAX : 20u) + 2u) + NT - 1u) / NT;  // 2
    struct Pre {
        v4u y0[LY], y1[LY], cb[LC], cr[LC];
    };
    static __device__ __forceinline__ void stage_load(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k,
                                                      uint32_t tid, Pre &pre) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip);
        const uint32_t nl = 16u * te, ncc = 8u * (te + 2u);
        const JP_GLOBAL v4u *y0 = (const JP_GLOBAL v4u *)img.coefs[0] + ((size_t)(2u * k) * g.bw0 + 2u * x0m) * 8u;
        const JP_GLOBAL v4u *y1 = y0 + (size_t)g.bw0 * 8u;
        const JP_GLOBAL v4u *cb = (const JP_GLOBAL v4u *)img.coefs[1] + (size_t)k * g.bwc * 8u;
        const JP_GLOBAL v4u *cr = (const JP_GLOBAL v4u *)img.coefs[2] + (size_t)k * g.bwc * 8u;
        // halo blocks outside the plane (image edges) are never transformed: clamp them onto valid chunks
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
        // (byte offsets as 32-bit values next to wave-uniform bases: global_load with an SGPR base and a VGPR offset, no
        // 64-bit address arithmetic per lane)
        auto at = [](const JP_GLOBAL v4u *base, uint32_t chunk) -> v4u {
            return *reinterpret_cast<const JP_GLOBAL v4u *>(reinterpret_cast<const JP_GLOBAL uint8_t *>(base) + chunk * 16u);
        };
#pragma unroll
        for (uint32_t i = 0; i < LY; i++) {
            const uint32_t j = min(tid + NT * i, nl - 1u);  // clamped: unconditional loads
            pre.y0[i] = at(y0, j);  // (plain loads and stores in this kernel: measured 1.5 % better than the streaming hint)
            pre.y1[i] = at(y1, j);
        }
#pragma unroll
        for (uint32_t i = 0; i < LC; i++) {
            const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid + NT * i, ncc - 1u), 0), cmax);
            pre.cb[i] = at(cb, e);
            pre.cr[i] = at(cr, e);
        }
    }
    static __device__ __forceinline__ void stage_store(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, const Pre &pre) {
        const uint32_t te = txe(g, strip);
        const uint32_t nl = 16u * te, ncc = 8u * (te + 2u);
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
        const uint32_t row = tid & 7u, b = tid >> 3;
#pragma unroll
        for (uint32_t i = 0; i < LY; i++)
            if (tid + NT * i < nl) {
                dst[coef_slot(b + (NT / 8u) * i, row)] = pre.y0[i];
                dst[coef_slot(2u * te + b + (NT / 8u) * i, row)] = pre.y1[i];
            }
#pragma unroll
        for (uint32_t i = 0; i < LC; i++)
            if (tid + NT * i < ncc) {
                dst[coef_slot(4u * te + b + (NT / 8u) * i, row)] = pre.cb[i];
                dst[coef_slot(5u * te + 2u + b + (NT / 8u) * i, row)] = pre.cr[i];
            }
    }

    // lane -> block: returns false for idle lanes.  comp 0 = luma (ry, cx), 1/2 = Cb/Cr (cx = LDS block column)
    static __device__ __forceinline__ bool lane_block(const FusedGeom &g, uint32_t strip, uint32_t tid, uint32_t &comp,
                                                      uint32_t &ry, uint32_t &cx) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip);
        if (tid < 4u * te) {
            comp = 0u;
            ry = tid >= 2u * te ? 1u : 0u;
            cx = tid - ry * 2u * te;
            return true;
        }
        const uint32_t t = tid - 4u * te;
        comp = 1u;
        ry = 0u;
        cx = 0u;
        if (t >= 2u * (te + 2u)) return false;
        const uint32_t c = t >= te + 2u ? 1u : 0u;
        comp = 1u + c;
        cx = t - c * (te + 2u);
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)cx;
        return bx >= 0 && bx < (int32_t)g.bwc;
    }

    // block `lb` of the staging area -> registers, dequantized on the way for the classes whose products fit i16
    template <class L>
    static __device__ __forceinline__ void fetch_block(const L &lds, uint32_t lb, uint32_t comp, uint32_t (&cw)[32]) {
        const v4u *p = reinterpret_cast<const v4u *>(lds.stage);
        const v4u *q = reinterpret_cast<const v4u *>(lds.qtab) + comp * 8u;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const v4u v = p[coef_slot(lb, k)];
            if constexpr (ARITH == ARITH_EXACT) {
                cw[4 * k] = v.x, cw[4 * k + 1] = v.y, cw[4 * k + 2] = v.z, cw[4 * k + 3] = v.w;
            } else {
                const v4u qq = q[k];
                cw[4 * k] = pk_mul_lo_u16(v.x, qq.x), cw[4 * k + 1] = pk_mul_lo_u16(v.y, qq.y);
                cw[4 * k + 2] = pk_mul_lo_u16(v.z, qq.z), cw[4 * k + 3] = pk_mul_lo_u16(v.w, qq.w);
            }
        }
    }
    template <class L>
    static __device__ __forceinline__ void transform_block(const L &lds, uint32_t comp, const uint32_t (&cw)[32], uint32_t (&out)[16]) {
        if constexpr (ARITH == ARITH_EXACT) {
            uint32_t qw[32];
            const v4u *q = reinterpret_cast<const v4u *>(lds.qtab) + comp * 8u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v4u v = q[i];
                qw[4 * i] = v.x, qw[4 * i + 1] = v.y, qw[4 * i + 2] = v.z, qw[4 * i + 3] = v.w;
            }
            idct8x8<ARITH_EXACT>(cw, qw, out);
        } else {
            idct8x8_products<ARITH>(cw, out);
        }
    }

    // staging -> registers (a barrier follows: the tiles written by transform alias the staging area); the first lanes
    // also pick up 8 bytes each of the carry rows, which transform() puts into row 0 of the tiles
    static __device__ __forceinline__ void read_block(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds,
                                                      S420Regs &r) {
        if (tid * 8u < lds.ypitch + 2u * lds.cpitch) r.carry = *reinterpret_cast<const v2u *>(lds.carry + tid * 8u);
        uint32_t comp, ry, cx;
        if (!lane_block(g, strip, tid, comp, ry, cx)) return;
        fetch_block(lds, tid, comp, r.cw);
    }

    // transform the lane's block and write the samples to the tiles (rows 1..) and its last row to the carry buffer as well
    static __device__ __forceinline__ void transform(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, S420Regs &r) {
        if (tid * 8u < lds.ypitch + 2u * lds.cpitch) {  // carry rows -> row 0 of the tiles
            const uint32_t o = tid * 8u;
            uint8_t *dst = o < lds.ypitch ? lds.ytile + o
                                          : (o - lds.ypitch < lds.cpitch ? lds.ctile + (o - lds.ypitch) : lds.ctile + 9u * lds.cpitch + (o - lds.ypitch - lds.cpitch));
            *reinterpret_cast<v2u *>(dst) = r.carry;
        }
        uint32_t comp, ry, cx;
        if (!lane_block(g, strip, tid, comp, ry, cx)) return;
        uint32_t out[16];
        transform_block(lds, comp, r.cw, out);
        const uint32_t c = comp == 0u ? 0u : comp - 1u;
        uint8_t *base = comp == 0u ? lds.ytile + (1u + ry * 8u) * lds.ypitch + cx * 8u : lds.ctile + (c * 9u + 1u) * lds.cpitch + cx * 8u;
        const uint32_t pitch = comp == 0u ? lds.ypitch : lds.cpitch;
        const EdgeFix ef = comp == 0u ? EdgeFix{false, false} : edge_fix(g, strip * g.tx, cx, out);
#pragma unroll
        for (int row = 0; row < 8; row++) *reinterpret_cast<v2u *>(base + (uint32_t)row * pitch) = v2u{out[2 * row], out[2 * row + 1]};
        uint8_t *cy = comp == 0u ? lds.carry + cx * 8u : lds.carry + lds.ypitch + c * lds.cpitch + cx * 8u;
        if (comp != 0u || ry == 1u) *reinterpret_cast<v2u *>(cy) = v2u{out[14], out[15]};  // what the next step finds in front of its own rows
        // the sample next to the image, in the (untransformed) neighbour block's columns: one lane per chroma component at most
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

    // Chroma columns just outside the image repeat the first / last sample of their row (see PixelOps::row_pixels<.., EDGES = false>).
    // Plane column cw (= size.width: the upsampler's "last column" is pixel 2*cw-1, src/upsampler.rs:226) lies in the block
    // of column cw-1 unless that one ends the block: then, like column -1, it is a byte of the neighbouring block, which is
    // outside the plane and never transformed.  bx = plane block of the lane (x0m - 1 + cx).
    struct EdgeFix {
        bool before, after;  // write byte 0 of a row to column -1 / byte 7 to column +8 of the block
    };
    template <int ROWS = 8>  // rows of `out` in use (seam rounds: the one row that is kept)
    static __device__ __forceinline__ EdgeFix edge_fix(const FusedGeom &g, uint32_t x0m, uint32_t cx, uint32_t (&out)[16]) {
        const uint32_t bx = x0m - 1u + cx, last = g.cw - 1u;
        // (a halo block's own neighbour column is not part of the tile — nobody reads it, and the byte would land in
        // the next row or past the buffer: only the strip's own blocks, cx in [1, te], write outside themselves)
        const bool own = cx >= 1u && cx <= min(g.tx, g.mcu_w - x0m);
        EdgeFix ef{bx == 0u && own, false};
        const uint32_t r = last & 7u;  // (uniform: a property of the image)
        if (r == 7u) {
            ef.after = own && bx == (last >> 3);
        } else if (bx == (last >> 3)) {  // widths whose chroma plane ends inside a block: the copy stays in the block
#pragma unroll
            for (int row = 0; row < ROWS; row++) {
                const uint64_t v = (uint64_t)out[2 * row] | ((uint64_t)out[2 * row + 1] << 32);
                const uint64_t b = (v >> (8u * r)) & 0xffull, m = 0xffull << (8u * (r + 1u));
                const uint64_t w = (v & ~m) | (b << (8u * (r + 1u)));
                out[2 * row] = (uint32_t)w, out[2 * row + 1] = (uint32_t)(w >> 32);
            }
        }
        return ef;
    }
    static __device__ __forceinline__ void edge_bytes(const EdgeFix &ef, uint8_t *row8, uint32_t lo, uint32_t hi) {
        if (ef.before) row8[-1] = (uint8_t)lo;
        if (ef.after) row8[8] = (uint8_t)(hi >> 24);
    }

    // ---- segment seams: the chroma blocks of block rows k0-1 (their last sample row -> carry) and k1 (their first
    // sample row -> bnd).  Staging block index = lane: [0,te+2) Cb above, then Cr above, Cb below, Cr below.
    static __device__ __forceinline__ void seam_stage(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k0, uint32_t k1,
                                                      uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), ncc = 8u * (te + 2u);
        const bool above = k0 > 0u, below = k1 < g.mcu_h;
        const size_t ra = (size_t)(above ? k0 - 1u : 0u) * g.bwc * 8u, rb = (size_t)(below ? k1 : 0u) * g.bwc * 8u;
        const JP_GLOBAL v4u *run[4] = {(const JP_GLOBAL v4u *)img.coefs[1] + ra, (const JP_GLOBAL v4u *)img.coefs[2] + ra,
                                       (const JP_GLOBAL v4u *)img.coefs[1] + rb, (const JP_GLOBAL v4u *)img.coefs[2] + rb};
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
        v4u v[4][LC];
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
#pragma unroll
            for (uint32_t i = 0; i < LC; i++) {
                const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid + NT * i, ncc - 1u), 0), cmax);
                v[w][i] = run[w][e];  // (re-read by the neighbouring segment: no streaming hint)
            }
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
        const uint32_t row = tid & 7u, b = tid >> 3;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
#pragma unroll
            for (uint32_t i = 0; i < LC; i++)
                if (tid + NT * i < ncc) dst[coef_slot(w * (te + 2u) + b + (NT / 8u) * i, row)] = v[w][i];
    }
    static __device__ __forceinline__ void seam_transform(const FusedGeom &g, uint32_t strip, uint32_t k0, uint32_t k1, uint32_t tid,
                                                          const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nb = te + 2u;
        if (tid >= 4u * nb) return;
        const uint32_t which = (tid >= nb ? 1u : 0u) + (tid >= 2u * nb ? 1u : 0u) + (tid >= 3u * nb ? 1u : 0u);
        const uint32_t cx = tid - which * nb, c = which & 1u;
        const bool below = which >= 2u;
        if (below ? !(k1 < g.mcu_h) : !(k0 > 0u)) return;
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)cx;
        if (bx < 0 || bx >= (int32_t)g.bwc) return;
        uint32_t cw[32];
        fetch_block(lds, tid, 1u + c, cw);
        uint8_t *dst = below ? lds.bnd + c * lds.cpitch + cx * 8u : lds.carry + lds.ypitch + c * lds.cpitch + cx * 8u;
        uint32_t lo, hi;
        EdgeFix ef;
        if constexpr (ARITH != ARITH_EXACT) {  // (round 2 transformed the seam blocks in full: < 1 % either way, profiles/round3/04_seam_row_transform.txt)
            // only ONE sample row of a seam block is ever used: its first (block row k1, below) or its last (block row k0 - 1)
            if (below) idct8x8_products_row<ARITH, 0>(cw, lo, hi);
            else idct8x8_products_row<ARITH, 7>(cw, lo, hi);
            uint32_t row[16] = {lo, hi, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            ef = edge_fix<1>(g, x0m, cx, row);
            lo = row[0], hi = row[1];
        } else {
            uint32_t out[16];
            transform_block(lds, 1u + c, cw, out);
            ef = edge_fix(g, x0m, cx, out);
            lo = below ? out[0] : out[14], hi = below ? out[1] : out[15];
        }
        *reinterpret_cast<v2u *>(dst) = v2u{lo, hi};
        edge_bytes(ef, dst, lo, hi);
    }
    // after the last step: the carry rows and the seam row below become rows 0 / 1 of the tiles for the closing call
    static __device__ __forceinline__ void closing_tiles(uint32_t tid, const Lds &lds) {
        const uint32_t o = tid * 8u;
        if (o < lds.ypitch) *reinterpret_cast<v2u *>(lds.ytile + o) = *reinterpret_cast<const v2u *>(lds.carry + o);
        if (o < 2u * lds.cpitch) {
            const uint32_t c = o >= lds.cpitch ? 1u : 0u, x = o - c * lds.cpitch;
            *reinterpret_cast<v2u *>(lds.ctile + c * 9u * lds.cpitch + x) = *reinterpret_cast<const v2u *>(lds.carry + lds.ypitch + o);
            *reinterpret_cast<v2u *>(lds.ctile + (c * 9u + 1u) * lds.cpitch + x) = *reinterpret_cast<const v2u *>(lds.bnd + o);
        }
    }

    // SKIP_A0: slot 0 emits its lower row only
    template <bool SKIP_A0>
    static __device__ __forceinline__ void interior_loop(const FusedGeom &g, const Lds &lds, JP_GLOBAL uint8_t *base, uint32_t pitch, uint32_t x0m,
                                                         uint32_t nch, uint32_t nunits, uint32_t magic, uint32_t tid) {
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t slot = __umulhi(u, magic), chk = u - slot * nch;
            const uint8_t *pc = lds.ctile + slot * lds.cpitch + 4u * chk + 4u;  // upper chroma row of the slot, Cb; lower: + cpitch; Cr: + 9 * cpitch
            typename P::ChromaEO eu[2], el[2];
#pragma unroll
            for (uint32_t comp = 0; comp < 2; comp++) {
                eu[comp] = P::load_eo(pc + comp * 9u * lds.cpitch);
                el[comp] = P::load_eo(pc + (comp * 9u + 1u) * lds.cpitch);
            }
            const uint32_t ox0 = 16u * x0m + 8u * chk, offa = 2u * slot * pitch + ox0 * 3u;
            const uint8_t *py = lds.ytile + 2u * slot * lds.ypitch + 8u * chk;
            if (!SKIP_A0 || slot != 0u) {
                const typename P::TPrime t[2] = {P::tprime(eu[0], el[0]), P::tprime(eu[1], el[1])};
                P::template row_pixels<false, false, true, false>(g, base + offa, true, t, *reinterpret_cast<const v2u *>(py), ox0);
            }
            {
                const typename P::TPrime t[2] = {P::tprime(el[0], eu[0]), P::tprime(el[1], eu[1])};
                P::template row_pixels<false, false, true, false>(g, base + (offa + pitch), true, t, *reinterpret_cast<const v2u *>(py + lds.ypitch), ox0);
            }
        }
    }

    // output rows 16k-1 .. 16k+14 of the strip: slot p (0..7) pairs chroma tile rows (p, p+1) = plane rows 8k-1+p, 8k+p and
    // emits luma tile rows 2p (near = upper chroma row) and 2p+1 (near = lower chroma row).  Rows above `row_lo` belong
    // to the workgroup of the segment above.  closing: only slot 0 (the segment's last output row, k = k1).
    static __device__ __forceinline__ void colour(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k, uint32_t row_lo,
                                                  bool closing, uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip);
        const uint32_t nch = 2u * te, nunits = (closing ? 1u : 8u) * nch;
        const uint32_t magic = 0xffffffffu / nch + 1u;  // mul_hi(u, magic) == u / nch for u < 65536
        const uint32_t pitch = g.out_w * 3u;
        // first byte of output row 16k-1 (may lie before the image for k = 0: never dereferenced)
        JP_GLOBAL uint8_t *base = (JP_GLOBAL uint8_t *)img.out + ((ptrdiff_t)(16 * (int64_t)k - 1)) * (ptrdiff_t)pitch;
        const uint32_t base_lo = (uint32_t)((16 * (int64_t)k - 1) * (int64_t)pitch) & 3u;  // its alignment
        // Interior steps — every one of the 16 rows inside the image, no vertical clamp, the strip's chunks complete and
        // 4-byte aligned (widths that are multiples of 8) — run a loop without per-unit predicates; the first step of a
        // segment below the top one is such a step too, except that its row 16k-1 belongs to the workgroup above.
        const bool first_of_segment = 16u * k == row_lo;
        const bool interior = !closing && 16u * k + 15u <= g.out_h && k > 0u && 8u * k + 8u <= g.ch && (g.out_w & 7u) == 0u &&
                              16u * x0m + 8u * nch <= g.out_w;
        if (interior) {
            if (first_of_segment) interior_loop<true>(g, lds, base, pitch, x0m, nch, nunits, magic, tid);
            else interior_loop<false>(g, lds, base, pitch, x0m, nch, nunits, magic, tid);
            return;
        }
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t slot = __umulhi(u, magic), chk = u - slot * nch;
            const int32_t oya = 16 * (int32_t)k - 1 + 2 * (int32_t)slot;
            const uint32_t oyb = (uint32_t)(oya + 1);
            const bool va = oya >= (int32_t)row_lo && (uint32_t)oya < g.out_h, vb = !closing && oyb < g.out_h;
            const uint32_t ox0 = 16u * x0m + 8u * chk;
            if ((!va && !vb) || ox0 >= g.out_w) continue;
            const int32_t cu = 8 * (int32_t)k - 1 + (int32_t)slot;  // plane row of the slot's upper chroma row
            // row a: near U, far min(near+1, ch-1);  row b: near L, far max(near-1, 0) (src/upsampler.rs:200-206)
            const bool clamp_a = cu + 1 > (int32_t)g.ch - 1, clamp_b = cu < 0;
            const uint32_t U = clamp_b ? slot + 1u : slot, L = clamp_a ? slot : slot + 1u;
            const uint32_t coff = 4u * chk + 4u;  // tile column of plane column j0 - 4
            typename P::ChromaEO eu[2], el[2];
#pragma unroll
            for (uint32_t comp = 0; comp < 2; comp++) {
                eu[comp] = P::load_eo(lds.ctile + (comp * 9u + U) * lds.cpitch + coff);
                el[comp] = P::load_eo(lds.ctile + (comp * 9u + L) * lds.cpitch + coff);
            }
            const uint32_t offa = 2u * slot * pitch + ox0 * 3u;  // < 17 rows: fits 32 bits
            const uint8_t *py = lds.ytile + 2u * slot * lds.ypitch + 8u * chk;
            if (va) {
                const typename P::TPrime t[2] = {P::tprime(eu[0], el[0]), P::tprime(eu[1], el[1])};
                const v2u yy = *reinterpret_cast<const v2u *>(py);
                P::template row_pixels<false, false, false, false>(g, base + offa, ((base_lo + offa) & 3u) == 0, t, yy, ox0);
            }
            if (vb) {
                const typename P::TPrime t[2] = {P::tprime(el[0], eu[0]), P::tprime(el[1], eu[1])};
                const v2u yy = *reinterpret_cast<const v2u *>(py + lds.ypitch);
                P::template row_pixels<false, false, false, false>(g, base + (offa + pitch), ((base_lo + offa + pitch) & 3u) == 0, t, yy, ox0);
            }
        }
    }
};

// =============================================================================================
// FUSED_440: 4:4:0 YCbCr (H1V1 luma, chroma at half the height: UpsamplerH1V2, src/upsampler.rs:165-189) -> RGB24 in one
// launch, as a strip walk like S420: MCU = 8 x 16 pixels = two luma blocks one above the other + one Cb + one Cr block; a
// workgroup walks MCU rows [k0, k1) of a strip of tx <= 64 MCUs, four blocks per MCU = one lane each.  The vertical
// filter reads chroma rows y/2 and y/2 -+ 1 exactly as H2V2 does, so the tiles, the carry rows and the seam round are those
// of S420 (without its horizontal halo: there is no horizontal filter); a unit of the pixel phase is one 8-pixel block
// column x two output rows.
// =============================================================================================
struct S440Lds {
    uint8_t *stage;   // 4*tx blocks x 128 B coefficient staging; later the tiles:
    uint8_t *ytile;   //   17 rows x pitch : row 0 = luma row 16k-1 (carry), rows 1..16 the step's own
    uint8_t *ctile;   //   2 comps x 9 rows x pitch : row 0 = chroma row 8k-1 (carry), rows 1..8 the step's own
    uint8_t *carry;   // 3 x pitch: last luma / Cb / Cr rows of the step before (chroma: the seam row at a segment start)
    uint8_t *bnd;     // 2 x pitch: chroma row 8*k1 (seam below the segment)
    uint8_t *qtab;    // 3 x 128 B
    uint32_t pitch;   // 8*tx
    static __device__ __host__ __forceinline__ uint32_t stage_bytes(uint32_t tx) { return 4u * tx * 128u; }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx) { return stage_bytes(tx) + 5u * 8u * tx + 384u; }
    static __device__ __forceinline__ S440Lds make(uint8_t *base, uint32_t tx) {
        S440Lds l;
        l.pitch = 8u * tx;
        l.stage = base;
        l.ytile = base;
        l.ctile = base + 17u * l.pitch;  // (17 + 18) * 8 * tx = 280 * tx <= 512 * tx
        l.carry = base + stage_bytes(tx);
        l.bnd = l.carry + 3u * l.pitch;
        l.qtab = l.bnd + 2u * l.pitch;
        return l;
    }
};
constexpr uint32_t S440_TX_MAX = 64;

template <int ARITH>
struct S440 {
    typedef S440Lds Lds;
    typedef S420<ARITH, 256> W;  // block fetch / transform helpers (they only look at lds.stage and lds.qtab)
    static constexpr uint32_t NT = 256;
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t strip) { return min(g.tx, g.mcu_w - strip * g.tx); }
    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t tid, const Lds &lds) {
        if (tid < 32u) {
            uint32_t *d = reinterpret_cast<uint32_t *>(lds.qtab);
            d[tid] = ((const JP_GLOBAL uint32_t *)img.qt[0])[tid];
            d[32u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[1])[tid];
            d[64u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[2])[tid];
        }
    }
    // lane -> block: [0,te) luma row 0, [te,2te) luma row 1, [2te,3te) Cb, [3te,4te) Cr
    static __device__ __forceinline__ bool lane_block(uint32_t te, uint32_t tid, uint32_t &comp, uint32_t &ry, uint32_t &cx) {
        const uint32_t q = (tid >= te ? 1u : 0u) + (tid >= 2u * te ? 1u : 0u) + (tid >= 3u * te ? 1u : 0u);
        cx = tid - q * te;
        ry = q == 1u ? 1u : 0u;
        comp = q < 2u ? 0u : q - 1u;
        return tid < 4u * te;
    }
    // four runs of 8*te chunks (<= 512: two loads per lane and run); staging block index = lane that transforms it
    struct Pre {
        v4u v[4][2];
    };
    static __device__ __forceinline__ void stage_load(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k, uint32_t tid, Pre &pre) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nc = 8u * te;
        const JP_GLOBAL v4u *y0 = (const JP_GLOBAL v4u *)img.coefs[0] + ((size_t)(2u * k) * g.bw0 + x0m) * 8u;
        const JP_GLOBAL v4u *run[4] = {y0, y0 + (size_t)g.bw0 * 8u, (const JP_GLOBAL v4u *)img.coefs[1] + ((size_t)k * g.bwc + x0m) * 8u,
                                       (const JP_GLOBAL v4u *)img.coefs[2] + ((size_t)k * g.bwc + x0m) * 8u};
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
#pragma unroll
            for (uint32_t i = 0; i < 2; i++) pre.v[w][i] = run[w][min(tid + NT * i, nc - 1u)];
    }
    static __device__ __forceinline__ void stage_store(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, const Pre &pre) {
        const uint32_t te = txe(g, strip), nc = 8u * te;
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
#pragma unroll
            for (uint32_t i = 0; i < 2; i++)
                if (tid + NT * i < nc) dst[coef_slot(w * te + (tid >> 3) + (NT / 8u) * i, tid & 7u)] = pre.v[w][i];
    }
    static __device__ __forceinline__ void read_block(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, S420Regs &r) {
        if (tid * 8u < 3u * lds.pitch) r.carry = *reinterpret_cast<const v2u *>(lds.carry + tid * 8u);
        uint32_t comp, ry, cx;
        if (!lane_block(txe(g, strip), tid, comp, ry, cx)) return;
        W::fetch_block(lds, tid, comp, r.cw);
    }
    static __device__ __forceinline__ void transform(const FusedGeom &g, uint32_t strip, uint32_t tid, const Lds &lds, S420Regs &r) {
        if (tid * 8u < 3u * lds.pitch) {  // carry rows -> row 0 of the tiles
            const uint32_t o = tid * 8u;
            uint8_t *dst = o < lds.pitch ? lds.ytile + o : (o < 2u * lds.pitch ? lds.ctile + (o - lds.pitch) : lds.ctile + 9u * lds.pitch + (o - 2u * lds.pitch));
            *reinterpret_cast<v2u *>(dst) = r.carry;
        }
        uint32_t comp, ry, cx;
        if (!lane_block(txe(g, strip), tid, comp, ry, cx)) return;
        uint32_t out[16];
        W::transform_block(lds, comp, r.cw, out);
        uint8_t *base = comp == 0u ? lds.ytile + (1u + ry * 8u) * lds.pitch + cx * 8u : lds.ctile + ((comp - 1u) * 9u + 1u) * lds.pitch + cx * 8u;
#pragma unroll
        for (int row = 0; row < 8; row++) *reinterpret_cast<v2u *>(base + (uint32_t)row * lds.pitch) = v2u{out[2 * row], out[2 * row + 1]};
        if (comp != 0u || ry == 1u) *reinterpret_cast<v2u *>(lds.carry + comp * lds.pitch + cx * 8u) = v2u{out[14], out[15]};
    }
    // seams: chroma blocks of block rows k0-1 (last sample row -> carry) and k1 (first sample row -> bnd); staging block
    // index = lane: [0,te) Cb above, Cr above, Cb below, Cr below
    static __device__ __forceinline__ void seam_stage(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k0, uint32_t k1, uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip), nc = 8u * te;
        const size_t ra = ((size_t)(k0 > 0u ? k0 - 1u : 0u) * g.bwc + x0m) * 8u, rb = ((size_t)(k1 < g.mcu_h ? k1 : 0u) * g.bwc + x0m) * 8u;
        const JP_GLOBAL v4u *run[4] = {(const JP_GLOBAL v4u *)img.coefs[1] + ra, (const JP_GLOBAL v4u *)img.coefs[2] + ra,
                                       (const JP_GLOBAL v4u *)img.coefs[1] + rb, (const JP_GLOBAL v4u *)img.coefs[2] + rb};
        Pre pre;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
#pragma unroll
            for (uint32_t i = 0; i < 2; i++) pre.v[w][i] = run[w][min(tid + NT * i, nc - 1u)];
        stage_store(g, strip, tid, lds, pre);
    }
    static __device__ __forceinline__ void seam_transform(const FusedGeom &g, uint32_t strip, uint32_t k0, uint32_t k1, uint32_t tid, const Lds &lds) {
        const uint32_t te = txe(g, strip);
        if (tid >= 4u * te) return;
        const uint32_t which = (tid >= te ? 1u : 0u) + (tid >= 2u * te ? 1u : 0u) + (tid >= 3u * te ? 1u : 0u);
        const uint32_t cx = tid - which * te, c = which & 1u;
        const bool below = which >= 2u;
        if (below ? !(k1 < g.mcu_h) : !(k0 > 0u)) return;
        uint32_t cw[32], lo, hi;
        W::fetch_block(lds, tid, 1u + c, cw);
        if constexpr (ARITH != ARITH_EXACT) {  // the one sample row that is kept (pixel_math.hpp idct8x8_products_row)
            if (below) idct8x8_products_row<ARITH, 0>(cw, lo, hi);
            else idct8x8_products_row<ARITH, 7>(cw, lo, hi);
        } else {
            uint32_t out[16];
            W::transform_block(lds, 1u + c, cw, out);
            lo = below ? out[0] : out[14], hi = below ? out[1] : out[15];
        }
        if (below) *reinterpret_cast<v2u *>(lds.bnd + c * lds.pitch + cx * 8u) = v2u{lo, hi};
        else *reinterpret_cast<v2u *>(lds.carry + (1u + c) * lds.pitch + cx * 8u) = v2u{lo, hi};
    }
    static __device__ __forceinline__ void closing_tiles(uint32_t tid, const Lds &lds) {
        const uint32_t o = tid * 8u;
        if (o < lds.pitch) {
            *reinterpret_cast<v2u *>(lds.ytile + o) = *reinterpret_cast<const v2u *>(lds.carry + o);
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                *reinterpret_cast<v2u *>(lds.ctile + c * 9u * lds.pitch + o) = *reinterpret_cast<const v2u *>(lds.carry + (1u + c) * lds.pitch + o);
                *reinterpret_cast<v2u *>(lds.ctile + (c * 9u + 1u) * lds.pitch + o) = *reinterpret_cast<const v2u *>(lds.bnd + c * lds.pitch + o);
            }
        }
    }

    // Eight chroma samples of one output row, centred: (3*near + far + 2) >> 2 - 128 (src/upsampler.rs:186) per sample, on
    // 16-bit lane pairs — even bytes (s0,s2)(s4,s6) and odd bytes (s1,s3)(s5,s7) of the two dwords.
    struct C8 {
        uint32_t e0, o0, e1, o1;  // lanes hold c - 128 as i16
    };
    static __device__ __forceinline__ C8 vfilter(v2u n, v2u f) {
        const uint32_t m = 0x00ff00ffu, bias = 0xfe02fe02u;  // 2 - 512 per lane: ((3n + f + 2 - 512) >> 2) == ((3n + f + 2) >> 2) - 128
        C8 c;
        c.e0 = pk_sar2(pk_add(pk_mad3(n.x & m, f.x & m), bias));
        c.o0 = pk_sar2(pk_add(pk_mad3(pk_shr(n.x, 8), pk_shr(f.x, 8)), bias));
        c.e1 = pk_sar2(pk_add(pk_mad3(n.y & m, f.y & m), bias));
        c.o1 = pk_sar2(pk_add(pk_mad3(pk_shr(n.y, 8), pk_shr(f.y, 8)), bias));
        return c;
    }
    static __device__ __forceinline__ int32_t lane_lo(uint32_t x) { return (int32_t)(x << 16) >> 16; }
    static __device__ __forceinline__ int32_t lane_hi(uint32_t x) { return (int32_t)x >> 16; }
    // one 8-pixel chunk of one row; n = pixels inside the image (8 unless the image ends in the chunk), al4 = 4-byte aligned
    template <bool FULL>
    static __device__ __forceinline__ void row8(JP_GLOBAL uint8_t *o, v2u yy, const C8 &cb, const C8 &cr, uint32_t n, bool al4) {
        const w32 yb[8] = {byte_shl20<0>(yy.x), byte_shl20<1>(yy.x), byte_shl20<2>(yy.x), byte_shl20<3>(yy.x),
                           byte_shl20<0>(yy.y), byte_shl20<1>(yy.y), byte_shl20<2>(yy.y), byte_shl20<3>(yy.y)};
        const int32_t b[8] = {lane_lo(cb.e0), lane_lo(cb.o0), lane_hi(cb.e0), lane_hi(cb.o0), lane_lo(cb.e1), lane_lo(cb.o1), lane_hi(cb.e1), lane_hi(cb.o1)};
        const int32_t r[8] = {lane_lo(cr.e0), lane_lo(cr.o0), lane_hi(cr.e0), lane_hi(cr.o0), lane_lo(cr.e1), lane_lo(cr.o1), lane_hi(cr.e1), lane_hi(cr.o1)};
        RawRgb p[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) p[k] = ycbcr_raw_centred(yb[k], b[k], r[k]);
        if (FULL || (n == 8u && al4)) {
            uint32_t d0, d1, d2, d3, d4, d5;
            rgb4_to_12bytes(p[0], p[1], p[2], p[3], d0, d1, d2);
            rgb4_to_12bytes(p[4], p[5], p[6], p[7], d3, d4, d5);
            *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o) = v3u{d0, d1, d2};
            *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o + 12) = v3u{d3, d4, d5};
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
                if (k < n) {
                    o[3 * k] = (uint8_t)sar_sat_u8x2(p[k].r, 0u, 20);
                    o[3 * k + 1] = (uint8_t)sar_sat_u8x2(p[k].g, 0u, 20);
                    o[3 * k + 2] = (uint8_t)sar_sat_u8x2(p[k].b, 0u, 20);
                }
        }
    }
    // output rows 16k-1 .. 16k+14 of the strip (see S420::colour); closing: the segment's last row only
    static __device__ __forceinline__ void colour(const FusedGeom &g, const FusedImage &img, uint32_t strip, uint32_t k, uint32_t row_lo, bool closing,
                                                  uint32_t tid, const Lds &lds) {
        const uint32_t x0m = strip * g.tx, te = txe(g, strip);
        const uint32_t nunits = (closing ? 1u : 8u) * te;
        const uint32_t magic = te > 1u ? 0xffffffffu / te + 1u : 0u;  // mul_hi(u, magic) == u / te for u < 65536 (te == 1: u itself)
        const uint32_t pitch = g.out_w * 3u;
        JP_GLOBAL uint8_t *base = (JP_GLOBAL uint8_t *)img.out + ((ptrdiff_t)(16 * (int64_t)k - 1)) * (ptrdiff_t)pitch;
        const uint32_t base_lo = (uint32_t)((16 * (int64_t)k - 1) * (int64_t)pitch) & 3u;
        const bool interior = !closing && 16u * k >= row_lo + 1u && 16u * k + 15u <= g.out_h && k > 0u && 8u * k + 8u <= g.ch &&
                              (g.out_w & 7u) == 0u && 8u * (x0m + te) <= g.out_w;
        if (interior) {
#pragma unroll 1
            for (uint32_t u = tid; u < nunits; u += NT) {
                const uint32_t slot = te > 1u ? __umulhi(u, magic) : u, chk = u - slot * te;
                const uint8_t *pc = lds.ctile + slot * lds.pitch + 8u * chk, *py = lds.ytile + 2u * slot * lds.pitch + 8u * chk;
                const v2u bu = *reinterpret_cast<const v2u *>(pc), bl = *reinterpret_cast<const v2u *>(pc + lds.pitch);
                const v2u ru = *reinterpret_cast<const v2u *>(pc + 9u * lds.pitch), rl = *reinterpret_cast<const v2u *>(pc + 10u * lds.pitch);
                const uint32_t offa = 2u * slot * pitch + 8u * (x0m + chk) * 3u;
                row8<true>(base + offa, *reinterpret_cast<const v2u *>(py), vfilter(bu, bl), vfilter(ru, rl), 8u, true);
                row8<true>(base + (offa + pitch), *reinterpret_cast<const v2u *>(py + lds.pitch), vfilter(bl, bu), vfilter(rl, ru), 8u, true);
            }
            return;
        }
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t slot = te > 1u ? __umulhi(u, magic) : u, chk = u - slot * te;
            const int32_t oya = 16 * (int32_t)k - 1 + 2 * (int32_t)slot;
            const uint32_t oyb = (uint32_t)(oya + 1);
            const bool va = oya >= (int32_t)row_lo && (uint32_t)oya < g.out_h, vb = !closing && oyb < g.out_h;
            const uint32_t ox0 = 8u * (x0m + chk);
            if ((!va && !vb) || ox0 >= g.out_w) continue;
            const int32_t cu = 8 * (int32_t)k - 1 + (int32_t)slot;
            const bool clamp_a = cu + 1 > (int32_t)g.ch - 1, clamp_b = cu < 0;
            const uint32_t U = clamp_b ? slot + 1u : slot, L = clamp_a ? slot : slot + 1u;
            const uint8_t *pc = lds.ctile + 8u * chk, *py = lds.ytile + 2u * slot * lds.pitch + 8u * chk;
            const v2u bu = *reinterpret_cast<const v2u *>(pc + U * lds.pitch), bl = *reinterpret_cast<const v2u *>(pc + L * lds.pitch);
            const v2u ru = *reinterpret_cast<const v2u *>(pc + (9u + U) * lds.pitch), rl = *reinterpret_cast<const v2u *>(pc + (9u + L) * lds.pitch);
            const uint32_t offa = 2u * slot * pitch + ox0 * 3u, n = min(8u, g.out_w - ox0);
            if (va) row8<false>(base + offa, *reinterpret_cast<const v2u *>(py), vfilter(bu, bl), vfilter(ru, rl), n, ((base_lo + offa) & 3u) == 0);
            if (vb) row8<false>(base + (offa + pitch), *reinterpret_cast<const v2u *>(py + lds.pitch), vfilter(bl, bu), vfilter(rl, ru), n, ((base_lo + offa + pitch) & 3u) == 0);
        }
    }
};

// =============================================================================================
// FUSED_GEN (round 2): the layouts that fall to UpsamplerGeneric (src/upsampler.rs:230-250) — luma H x V blocks per MCU
// with H, V in {1, 2, 4} and none of the four fancy-upsampled shapes (4:1:1 = 4x1, 4:1:0 = 4x2, 1x4, 2x4, 4x4), Cb and
// Cr one block each.  Chroma is replicated: pixel (x, y) takes sample (x / H, y / V); luma is H1V1.  No neighbourhood at
// all, so a workgroup owns tx MCUs of one MCU row: (H*V + 2) * tx <= 256 blocks, one per lane, staged and transformed like
// the strip walks' (quantization tables in LDS, multiplied during the fetch), then the pixel phase walks the tile's 8V rows
// in 8-pixel chunks and reuses the 4:4:0 kernel's row arithmetic on the replicated samples.
// =============================================================================================
struct FGenLds {
    uint8_t *stage;  // tx*(H*V+2) blocks x 128 B coefficient staging; later the tiles:
    uint8_t *ytile;  //   8V rows x ypitch (8*H*tx)
    uint8_t *ctile;  //   2 components x 8 rows x cpitch (8*tx)
    uint8_t *qtab;   // 3 x 128 B
    uint32_t ypitch, cpitch;
    static __device__ __host__ __forceinline__ uint32_t stage_bytes(uint32_t tx, uint32_t hs, uint32_t vs) { return tx * ((1u << (hs + vs)) + 2u) * 128u; }
    static __device__ __host__ __forceinline__ uint32_t total_bytes(uint32_t tx, uint32_t hs, uint32_t vs) { return stage_bytes(tx, hs, vs) + 384u; }
    static __device__ __forceinline__ FGenLds make(uint8_t *base, uint32_t tx, uint32_t hs, uint32_t vs) {
        FGenLds l;
        l.ypitch = (8u << hs) * tx;
        l.cpitch = 8u * tx;
        l.stage = base;
        l.ytile = base;
        l.ctile = base + (8u << vs) * l.ypitch;  // tiles: 64*tx*(H*V + 2) bytes, half the staging area
        l.qtab = base + stage_bytes(tx, hs, vs);
        return l;
    }
};
inline __device__ __host__ uint32_t fgen_tx_max(uint32_t hs, uint32_t vs) { return 256u / ((1u << (hs + vs)) + 2u); }

template <int ARITH>
struct FGen {
    typedef FGenLds Lds;
    typedef S420<ARITH, 256> W;  // block fetch / transform helpers (they only look at lds.stage and lds.qtab)
    typedef S440<ARITH> R;       // row8: eight pixels from eight luma bytes and centred chroma lanes
    static constexpr uint32_t NT = 256;
    struct Pre {
        v4u v[8];  // (H*V + 2) * tx * 8 <= 2048 chunks: eight per lane
    };
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) { return min(g.tx, g.mcu_w - tile_x * g.tx); }
    static __device__ __forceinline__ void init(const FusedImage &img, uint32_t tid, const Lds &lds) {
        if (tid < 32u) {
            uint32_t *d = reinterpret_cast<uint32_t *>(lds.qtab);
            d[tid] = ((const JP_GLOBAL uint32_t *)img.qt[0])[tid];
            d[32u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[1])[tid];
            d[64u + tid] = ((const JP_GLOBAL uint32_t *)img.qt[2])[tid];
        }
    }
    // Chunk c of the tile (16 bytes): the V luma block rows first (8*te*H chunks each, contiguous in the plane), then Cb, Cr
    // (8*te each).  Staging block index = lane that transforms it: luma row r block j -> r*te*H + j, Cb -> te*H*V + j, Cr after.
    static __device__ __forceinline__ void stage_load(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my, uint32_t tid, Pre &pre) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nly = (8u << g.hs) * te, ncl = nly << g.vs, ncc = 8u * te, total = ncl + 2u * ncc;
        const JP_GLOBAL v4u *y = (const JP_GLOBAL v4u *)img.coefs[0] + (((size_t)my << g.vs) * g.bw0 + ((size_t)x0m << g.hs)) * 8u;
        const JP_GLOBAL v4u *cb = (const JP_GLOBAL v4u *)img.coefs[1] + ((size_t)my * g.bwc + x0m) * 8u;
        const JP_GLOBAL v4u *cr = (const JP_GLOBAL v4u *)img.coefs[2] + ((size_t)my * g.bwc + x0m) * 8u;
        const uint32_t row_chunks = g.bw0 * 8u;  // chunks between luma block rows
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t c = min(tid + NT * i, total - 1u);  // clamped: unconditional loads
            if (c < ncl) {
                const uint32_t r = (c >= nly ? 1u : 0u) + (c >= 2u * nly ? 1u : 0u) + (c >= 3u * nly ? 1u : 0u);
                pre.v[i] = y[r * row_chunks + (c - r * nly)];
            } else {
                const uint32_t d = c - ncl;
                pre.v[i] = d >= ncc ? cr[d - ncc] : cb[d];
            }
        }
    }
    static __device__ __forceinline__ void stage_store(const FusedGeom &g, uint32_t tile_x, uint32_t tid, const Lds &lds, const Pre &pre) {
        const uint32_t te = txe(g, tile_x);
        const uint32_t total = ((8u << (g.hs + g.vs)) + 16u) * te;
        v4u *dst = reinterpret_cast<v4u *>(lds.stage);
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t c = tid + NT * i;  // (runs are whole blocks: chunk c belongs to staging block c >> 3, row c & 7)
            if (c < total) dst[coef_slot(c >> 3, c & 7u)] = pre.v[i];
        }
    }
    // lane -> its block: component and position in the tiles
    static __device__ __forceinline__ bool lane_block(const FusedGeom &g, uint32_t te, uint32_t tid, uint32_t &comp, uint32_t &ry, uint32_t &cx) {
        const uint32_t nl = te << (g.hs + g.vs);
        if (tid < nl) {
            const uint32_t per_row = te << g.hs;  // luma blocks per block row of the tile
            comp = 0u;
            ry = (tid >= per_row ? 1u : 0u) + (tid >= 2u * per_row ? 1u : 0u) + (tid >= 3u * per_row ? 1u : 0u);
            cx = tid - ry * per_row;
            return true;
        }
        const uint32_t t = tid - nl;
        comp = t >= te ? 2u : 1u;
        ry = 0u;
        cx = t - (comp - 1u) * te;
        return t < 2u * te;
    }
    static __device__ __forceinline__ void read_block(const FusedGeom &g, uint32_t tile_x, uint32_t tid, const Lds &lds, S420Regs &r) {
        uint32_t comp, ry, cx;
        if (!lane_block(g, txe(g, tile_x), tid, comp, ry, cx)) return;
        W::fetch_block(lds, tid, comp, r.cw);
    }
    static __device__ __forceinline__ void transform(const FusedGeom &g, uint32_t tile_x, uint32_t tid, const Lds &lds, S420Regs &r) {
        uint32_t comp, ry, cx;
        if (!lane_block(g, txe(g, tile_x), tid, comp, ry, cx)) return;
        uint32_t out[16];
        W::transform_block(lds, comp, r.cw, out);
        const uint32_t pitch = comp == 0u ? lds.ypitch : lds.cpitch;
        uint8_t *base = comp == 0u ? lds.ytile + ry * 8u * lds.ypitch + cx * 8u : lds.ctile + (comp - 1u) * 8u * lds.cpitch + cx * 8u;
#pragma unroll
        for (int row = 0; row < 8; row++) *reinterpret_cast<v2u *>(base + (uint32_t)row * pitch) = v2u{out[2 * row], out[2 * row + 1]};
    }
    // eight replicated chroma samples of one component as centred 16-bit lanes (S440::C8: (s0,s2) (s1,s3) (s4,s6) (s5,s7))
    static __device__ __forceinline__ typename R::C8 replicate(const uint8_t *crow, uint32_t chk, uint32_t hs) {
        typename R::C8 c;
        const uint32_t m = 0x00ff00ffu, centre = 0xff80ff80u;  // - 128 per lane
        if (hs == 0u) {  // eight samples
            const v2u w = *reinterpret_cast<const v2u *>(crow + 8u * chk);
            c.e0 = w.x & m, c.o0 = (w.x >> 8) & m, c.e1 = w.y & m, c.o1 = (w.y >> 8) & m;
        } else if (hs == 1u) {  // four samples, each twice: (t0,t0,t1,t1,t2,t2,t3,t3)
            const uint32_t d = *reinterpret_cast<const uint32_t *>(crow + 4u * chk);
            c.e0 = c.o0 = (d & 0xffu) | ((d & 0xff00u) << 8);
            c.e1 = c.o1 = ((d >> 16) & 0xffu) | ((d >> 24) << 16);
        } else {  // two samples, each four times
            const uint32_t d = *reinterpret_cast<const uint16_t *>(crow + 2u * chk);
            c.e0 = c.o0 = (d & 0xffu) * 0x00010001u;
            c.e1 = c.o1 = (d >> 8) * 0x00010001u;
        }
        c.e0 = pk_add(c.e0, centre), c.o0 = pk_add(c.o0, centre), c.e1 = pk_add(c.e1, centre), c.o1 = pk_add(c.o1, centre);
        return c;
    }
    static __device__ __forceinline__ void colour(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my, uint32_t tid, const Lds &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nchk = te << g.hs, nunits = nchk << (3u + g.vs);  // rows x chunks: <= 2048
        const uint32_t magic = nchk > 1u ? 0xffffffffu / nchk + 1u : 0u;  // mul_hi(u, magic) == u / nchk for u < 65536
        const uint32_t pitch = g.out_w * 3u;
        const uint32_t oy0 = (my << g.vs) * 8u, ox_tile = (x0m << g.hs) * 8u;
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)img.out;
        const bool full = (g.out_w & 7u) == 0u && ox_tile + 8u * nchk <= g.out_w;  // (uniform) every chunk complete and 4-byte aligned
#pragma unroll 1
        for (uint32_t u = tid; u < nunits; u += NT) {
            const uint32_t row = nchk > 1u ? __umulhi(u, magic) : u, chk = u - row * nchk;
            const uint32_t oy = oy0 + row, ox0 = ox_tile + 8u * chk;
            if (oy >= g.out_h || ox0 >= g.out_w) continue;
            const v2u yy = *reinterpret_cast<const v2u *>(lds.ytile + row * lds.ypitch + 8u * chk);
            const uint32_t crow = row >> g.vs;
            const typename R::C8 cb = replicate(lds.ctile + crow * lds.cpitch, chk, g.hs);
            const typename R::C8 cr = replicate(lds.ctile + (8u + crow) * lds.cpitch, chk, g.hs);
            const size_t off = (size_t)oy * pitch + ox0 * 3u;
            if (full) R::template row8<true>(out + off, yy, cb, cr, 8u, true);
            else R::template row8<false>(out + off, yy, cb, cr, min(8u, g.out_w - ox0), (off & 3u) == 0u);
        }
    }
};

// =============================================================================================
// FUSED_422: 4:2:2 YCbCr (H2V1 / H1V1 / H1V1) -> RGB24, one launch.  MCU = 16x8 pixels = two luma blocks side by side
// + one Cb + one Cr block.  A workgroup owns TX <= 62 MCUs of one MCU row: waves 0-1 transform the 2*te luma blocks,
// wave 2 the te+2 Cb blocks and wave 3 the te+2 Cr blocks under them — one halo block either side, because
// UpsamplerH2V1 (src/upsampler.rs:134-163) reads the sample left / right of each chroma sample; there is no vertical
// neighbourhood, so nothing has to come from other MCU rows.  Every wave works on one component: its quantization
// table stays in SGPRs.  In the pixel phase wave w takes tile rows w and w + 4 and its lanes walk the row's 2*te chunks.
// =============================================================================================
template <int ARITH>
struct F422 {
    typedef PixelOps<ARITH> P;  // pixel helpers
    static constexpr uint32_t NT = 256;
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.mcu_w - tile_x * g.tx);
    }
    // lane -> block.  Staging block index = tid: [0,128) luma block column, [128,192) Cb, [192,256) Cr (cx = column in
    // the chroma tile, plane block x0m - 1 + cx).  false: idle lane.
    static __device__ __forceinline__ bool lane_block(const FusedGeom &g, uint32_t tile_x, uint32_t tid, uint32_t &comp, uint32_t &cx) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        if (tid < 128u) {
            comp = 0u;
            cx = tid;
            return tid < 2u * te;
        }
        comp = tid < 192u ? 1u : 2u;
        cx = tid & 63u;
        const int32_t bx = (int32_t)x0m - 1 + (int32_t)cx;
        return cx < te + 2u && bx >= 0 && bx < (int32_t)g.bwc;
    }
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLdsSmall &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nl = 16u * te, ncc = 8u * (te + 2u);  // chunks: luma run (<= 992), chroma run (<= 512)
        const JP_GLOBAL v4u *y = (const JP_GLOBAL v4u *)img.coefs[0] + ((size_t)my * g.bw0 + 2u * x0m) * 8u;
        const JP_GLOBAL v4u *cb = (const JP_GLOBAL v4u *)img.coefs[1] + (size_t)my * g.bwc * 8u;
        const JP_GLOBAL v4u *cr = (const JP_GLOBAL v4u *)img.coefs[2] + (size_t)my * g.bwc * 8u;
        // halo blocks outside the plane (image edges) are never transformed: clamp them onto valid chunks
        const int32_t cfirst = ((int32_t)x0m - 1) * 8, cmax = (int32_t)(g.bwc * 8u) - 1;
        v4u vy[4], vb[2], vr[2];
        load_run<NT, 4>(vy, y, nl, tid);
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t e = (uint32_t)min(max(cfirst + (int32_t)min(tid + NT * i, ncc - 1u), 0), cmax);
            vb[i] = stream_load(cb + e);
            vr[i] = stream_load(cr + e);
        }
        store_run<NT, 4>(lds.coef, vy, nl, 0u, tid);
        store_run<NT, 2>(lds.coef, vb, ncc, 128u, tid);
        store_run<NT, 2>(lds.coef, vr, ncc, 192u, tid);
    }
    // qt_of_wave: table of the wave's component (0, 0, 1, 2), fetched by the caller from the image array in memory
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const uint16_t *qt_of_wave, uint32_t tile_x, uint32_t tid,
                                                  const FusedLdsSmall &lds, FusedRegs &r) {
        uint32_t comp, cx;
        if (!lane_block(g, tile_x, tid, comp, cx)) return;
        uint32_t cw[32];
        load_block_from_lds(lds.coef, tid, cw);
        idct8x8<ARITH>(cw, as_qtab(qt_of_wave), r.out);
    }
    // sample tiles (alias the staging area): luma 8 rows x 16*tx, then [2 comps][8 rows][cpitch = 8*(tx+2)]
    static __device__ __forceinline__ uint32_t cpitch(const FusedGeom &g) { return 8u * (g.tx + 2u); }
    static __device__ __forceinline__ void phase2(const FusedGeom &g, uint32_t tile_x, uint32_t tid, FusedLdsSmall &lds,
                                                  const FusedRegs &r) {
        uint32_t comp, cx;
        if (!lane_block(g, tile_x, tid, comp, cx)) return;
        const uint32_t ypitch = 16u * g.tx, cp = cpitch(g);
        uint8_t *base = comp == 0u ? lds.coef + cx * 8u : lds.coef + 8u * ypitch + (comp - 1u) * 8u * cp + cx * 8u;
        const uint32_t pitch = comp == 0u ? ypitch : cp;
        // chroma rows repeat their first / last sample in the column outside the image, so that the pixel phase needs no
        // edge formula ((3 s + s + 2) >> 2 == s; see PixelOps::row_pixels<.., EDGES = false> and S420::edge_fix)
        uint32_t out[16];
#pragma unroll
        for (int i = 0; i < 16; i++) out[i] = r.out[i];
        typedef typename S420<ARITH, 256>::EdgeFix EdgeFix;
        const EdgeFix ef = comp == 0u ? EdgeFix{false, false} : S420<ARITH, 256>::edge_fix(g, tile_x * g.tx, cx, out);
#pragma unroll
        for (int row = 0; row < 8; row++)
            *reinterpret_cast<v2u *>(base + (uint32_t)row * pitch) = v2u{out[2 * row], out[2 * row + 1]};
        if (ef.before) {
#pragma unroll
            for (int row = 0; row < 8; row++) (base + (uint32_t)row * pitch)[-1] = (uint8_t)out[2 * row];
        }
        if (ef.after) {
#pragma unroll
            for (int row = 0; row < 8; row++) (base + (uint32_t)row * pitch)[8] = (uint8_t)(out[2 * row + 1] >> 24);
        }
    }
    static __device__ __forceinline__ void phase3(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLdsSmall &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nch = 2u * te;
        const uint32_t ypitch = 16u * g.tx, cp = cpitch(g);
        const uint8_t *ctile = lds.coef + 8u * ypitch;
        // tiles whose chunks are complete and 4-byte aligned (widths that are multiples of 8): no per-lane store predicate
        const bool full = (g.out_w & 7u) == 0u && 16u * x0m + 8u * nch <= g.out_w;
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)img.out;
        const size_t pitch = (size_t)g.out_w * 3u;
        const uint32_t wave = uniform(tid >> 6), lane = tid & 63u;
        // Wave w takes tile rows w and w + 4, its lanes walk the row's chunks (at most two each: tx <= 62): the scanline is
        // wave-uniform (scalar address math), the lane's LDS addresses are finished before the chunks (as in the walks' pixel phase).
        static_assert(F422_TX_MAX <= 64u, "a lane takes at most two chunks of a row");
#pragma unroll 1
        for (uint32_t row = wave; row < 8u; row += NT / 64u) {
            const uint32_t oy = 8u * my + row;
            if (oy >= g.out_h) continue;
            const size_t ro = (size_t)oy * pitch;
            const uint8_t *pc[2] = {opaque_lds(ctile + row * cp + 4u * lane + 4u), opaque_lds(ctile + (8u + row) * cp + 4u * lane + 4u)};
            const uint8_t *py = opaque_lds(lds.coef + row * ypitch + 8u * lane);
#pragma unroll
            for (uint32_t it = 0; it < 2u; it++) {
                const uint32_t chk = lane + 64u * it, ox0 = 16u * x0m + 8u * chk;
                if (chk >= nch || ox0 >= g.out_w) continue;
                typename P::TPrime t[2];
#pragma unroll
                for (uint32_t comp = 0; comp < 2; comp++) {
                    // tile column of plane column j0 - 4 (j0 = ox0 / 2): 4*chk + 4, as in the 4:2:0 kernels
                    const typename P::ChromaEO e = P::load_eo(pc[comp] + 256u * it);
                    t[comp].tE1 = e.E1;
                    t[comp].tO1 = e.O1;
                    t[comp].tOm = e.Om;
                    t[comp].tEp = e.Ep;
                }
                const v2u yy = *reinterpret_cast<const v2u *>(py + 512u * it);
                if (full) P::template row_pixels<true, false, true>(g, out + ro + ox0 * 3u, true, t, yy, ox0);
                else P::template row_pixels<true, false, false>(g, out + ro + ox0 * 3u, (ro & 3u) == 0, t, yy, ox0);
            }
        }
    }
};

// =============================================================================================
// FUSED_444: MCU = one 8x8 block per component; three components (YCbCr / RGB -> RGB24) or four (CMYK / YCCK -> 32-bit
// pixels, src/decoder.rs:1439-1474).  Wave c transforms component c (so the quantization table stays wave-uniform in
// SGPRs), lane = block; with three components wave 3 only helps staging and the pixel phase.
// =============================================================================================
template <int ARITH>
struct F444 {
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.mcu_w - tile_x * g.tx);
    }
    static __device__ __forceinline__ uint32_t ncomp(const FusedGeom &g) { return g.color >= FCOLOR_CMYK ? 4u : 3u; }
    // LDS block index of (component, block) = comp*64 + cx
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLdsSmall &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const size_t base = ((size_t)my * g.bwc + x0m) * 8u;  // chunk index of the tile's first block
        const JP_GLOBAL v4u *c0 = (const JP_GLOBAL v4u *)img.coefs[0] + base;
        const JP_GLOBAL v4u *c1 = (const JP_GLOBAL v4u *)img.coefs[1] + base;
        const JP_GLOBAL v4u *c2 = (const JP_GLOBAL v4u *)img.coefs[2] + base;
        const uint32_t te8 = te * 8u;  // <= 512 chunks per component: two loads each
        v4u v0[2], v1[2], v2[2], v3[2];
        load_run<FUSED_NT, 2>(v0, c0, te8, tid);
        load_run<FUSED_NT, 2>(v1, c1, te8, tid);
        load_run<FUSED_NT, 2>(v2, c2, te8, tid);
        const bool four = ncomp(g) == 4u;  // (uniform)
        if (four) load_run<FUSED_NT, 2>(v3, (const JP_GLOBAL v4u *)img.coefs[3] + base, te8, tid);
        store_run<FUSED_NT, 2>(lds.coef, v0, te8, 0u, tid);  // LDS block of (component, block) = comp*64 + cx
        store_run<FUSED_NT, 2>(lds.coef, v1, te8, 64u, tid);
        store_run<FUSED_NT, 2>(lds.coef, v2, te8, 128u, tid);
        if (four) store_run<FUSED_NT, 2>(lds.coef, v3, te8, 192u, tid);
    }
    // qt_of_wave: quantization table of component (tid >> 6), fetched by the caller from the image array in memory
    // (a runtime index — or a chain of selects, which the compiler turns into one — into the by-value image struct
    // would move the whole struct to scratch)
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const uint16_t *qt_of_wave, uint32_t tile_x, uint32_t tid,
                                                  const FusedLdsSmall &lds, FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        const uint32_t comp = uniform(tid >> 6), cx = tid & 63u;
        if (comp >= ncomp(g) || cx >= te) return;
        uint32_t cw[32];
        load_block_from_lds(lds.coef, comp * 64u + cx, cw);
        idct8x8<ARITH>(cw, as_qtab(qt_of_wave), r.out);
    }
    // sample tiles: [ncomp][8 rows][pitch 8*tx]
    static __device__ __forceinline__ void phase2(const FusedGeom &g, uint32_t tile_x, uint32_t tid, FusedLdsSmall &lds,
                                                  const FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        const uint32_t comp = tid >> 6, cx = tid & 63u;
        if (comp >= ncomp(g) || cx >= te) return;
        const uint32_t pitch = 8u * g.tx;
#pragma unroll
        for (int row = 0; row < 8; row++)
            *reinterpret_cast<v2u *>(&lds.coef[(comp * 8u + (uint32_t)row) * pitch + cx * 8u]) =
                v2u{r.out[2 * row], r.out[2 * row + 1]};
    }
    static __device__ __forceinline__ void phase3(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLdsSmall &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t pitch = 8u * g.tx;
        const uint32_t wave = uniform(tid >> 6), chk = tid & 63u;
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)img.out;
        if (chk >= te) return;
        const uint32_t ox0 = 8u * (x0m + chk);
        if (ox0 >= g.out_w) return;
        const uint32_t npx = min(8u, g.out_w - ox0);
        const bool full = (g.out_w & 7u) == 0u && 8u * (x0m + te) <= g.out_w;  // (uniform) every chunk of the tile complete and aligned
        if (ncomp(g) == 4u) {  // (uniform) 32-bit pixels: a lane's 8 pixels are 32 contiguous, 4-byte aligned bytes
            for (uint32_t row = wave; row < 8u; row += 4u) {
                const uint32_t oy = 8u * my + row;
                if (oy >= g.out_h) continue;
                v2u s[4];
#pragma unroll
                for (uint32_t comp = 0; comp < 4; comp++)
                    s[comp] = *reinterpret_cast<const v2u *>(&lds.coef[(comp * 8u + row) * pitch + chk * 8u]);
                uint32_t px[8];
                if (g.color == FCOLOR_CMYK) {  // src/decoder.rs:1456-1474: 255 - x, four times
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t v = byte_of(k < 4 ? s[0].x : s[0].y, k & 3u) | (byte_of(k < 4 ? s[1].x : s[1].y, k & 3u) << 8) |
                                           (byte_of(k < 4 ? s[2].x : s[2].y, k & 3u) << 16) | (byte_of(k < 4 ? s[3].x : s[3].y, k & 3u) << 24);
                        px[k] = ~v;
                    }
                } else {  // YCCK, src/decoder.rs:1439-1454: YCbCr -> RGB on the first three, 255 - k
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t rgb = ycbcr_to_rgb24(byte_of(k < 4 ? s[0].x : s[0].y, k & 3u), byte_of(k < 4 ? s[1].x : s[1].y, k & 3u),
                                                            byte_of(k < 4 ? s[2].x : s[2].y, k & 3u));
                        px[k] = rgb | ((255u - byte_of(k < 4 ? s[3].x : s[3].y, k & 3u)) << 24);
                    }
                }
                JP_GLOBAL uint8_t *o = out + ((size_t)oy * g.out_w + ox0) * 4u;
                if (npx == 8u) {
                    stream_store(reinterpret_cast<JP_GLOBAL v4u *>(o), v4u{px[0], px[1], px[2], px[3]});  // (an arena offset is 256-byte aligned, a pixel 4 bytes)
                    stream_store(reinterpret_cast<JP_GLOBAL v4u *>(o + 16), v4u{px[4], px[5], px[6], px[7]});
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (k < npx) reinterpret_cast<JP_GLOBAL uint32_t *>(o)[k] = px[k];
                }
            }
            return;
        }
        for (uint32_t row = wave; row < 8u; row += 4u) {
            const uint32_t oy = 8u * my + row;
            if (oy >= g.out_h) continue;
            v2u s[3];
#pragma unroll
            for (uint32_t comp = 0; comp < 3; comp++)
                s[comp] = *reinterpret_cast<const v2u *>(&lds.coef[(comp * 8u + row) * pitch + chk * 8u]);
            const size_t off = ((size_t)oy * g.out_w + ox0) * 3u;
            if (g.color == FCOLOR_RGB) {  // src/decoder.rs:1391-1404: interleave only
                uint32_t px[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                    px[k] = byte_of(k < 4 ? s[0].x : s[0].y, k & 3u) | (byte_of(k < 4 ? s[1].x : s[1].y, k & 3u) << 8) |
                            (byte_of(k < 4 ? s[2].x : s[2].y, k & 3u) << 16);
                store_rgb_run(out, off, px, npx);
            } else {
                RawRgb p[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                    p[k] = ycbcr_raw(byte_of(k < 4 ? s[0].x : s[0].y, k & 3u), byte_of(k < 4 ? s[1].x : s[1].y, k & 3u),
                                     byte_of(k < 4 ? s[2].x : s[2].y, k & 3u));
                if (full || (npx == 8u && (off & 3u) == 0)) {
                    uint32_t d0, d1, d2, d3, d4, d5;
                    rgb4_to_12bytes(p[0], p[1], p[2], p[3], d0, d1, d2);
                    rgb4_to_12bytes(p[4], p[5], p[6], p[7], d3, d4, d5);
                    stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(out + off), v3u{d0, d1, d2});
                    stream_store(reinterpret_cast<JP_GLOBAL v3u_a4 *>(out + off + 12), v3u{d3, d4, d5});
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (k < npx) {
                            out[off + 3 * k] = (uint8_t)sar_sat_u8x2(p[k].r, 0u, 20);
                            out[off + 3 * k + 1] = (uint8_t)sar_sat_u8x2(p[k].g, 0u, 20);
                            out[off + 3 * k + 2] = (uint8_t)sar_sat_u8x2(p[k].b, 0u, 20);
                        }
                }
            }
        }
    }
};

// =============================================================================================
// FUSED_GRAY: one block per lane, straight to the output rows
// =============================================================================================
template <int ARITH>
struct FGray {
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.bw0 - tile_x * g.tx);
    }
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLdsSmall &lds) {
        const uint32_t x0 = tile_x * g.tx, te = txe(g, tile_x);
        const JP_GLOBAL v4u *src = (const JP_GLOBAL v4u *)img.coefs[0] + ((size_t)my * g.bw0 + x0) * 8u;
        stage_coefficients<FUSED_NT>(lds.coef, te * 8u, tid, [&](uint32_t j) { return src + j; });
    }
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLdsSmall &lds) {
        const uint32_t x0 = tile_x * g.tx, te = txe(g, tile_x);
        if (tid >= te) return;
        uint32_t cw[32], out[16];
        load_block_from_lds(lds.coef, tid, cw);
        idct8x8<ARITH>(cw, as_qtab(img.qt[0]), out);
        const uint32_t ox = 8u * (x0 + tid);
        if (ox >= g.out_w) return;
        const uint32_t n = min(8u, g.out_w - ox);
        JP_GLOBAL uint8_t *dst = (JP_GLOBAL uint8_t *)img.out;
#pragma unroll
        for (uint32_t row = 0; row < 8; row++) {
            const uint32_t oy = 8u * my + row;
            if (oy >= g.out_h) break;
            const size_t off = (size_t)oy * g.out_w + ox;
            if (n == 8 && (off & 7u) == 0) {
                stream_store(reinterpret_cast<JP_GLOBAL v2u *>(dst + off), v2u{out[2 * row], out[2 * row + 1]});
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                    if (k < n) dst[off + k] = (uint8_t)byte_of(k < 4 ? out[2 * row] : out[2 * row + 1], k & 3u);
            }
        }
    }
};

}  // namespace jpgpu
