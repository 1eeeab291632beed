// fused_core.hpp — bodies of the fused fast-path kernels (same-geometry batches, dct_scale 8).
//
//   FUSED_420  : 4:2:0 YCbCr (H2V2 / H1V1 / H1V1) -> RGB24.  Two launches per batch:
//                  chroma pass : Cb, Cr coefficient planes -> u8 planes in a scratch arena
//                                (same body as the generic IDCT kernel);
//                  main pass   : per tile of TX MCUs x 1 MCU row: stage Y coefficients and the
//                                chroma neighbourhood (+-1 sample halo) in LDS, IDCT the 4*TX luma
//                                blocks (one lane per block), exchange through an LDS tile, then
//                                fancy-upsample (src/upsampler.rs:191-228), colour convert
//                                (src/decoder.rs:1486-1508) and store 24-B pixel runs, lanes
//                                walking consecutive 8-pixel chunks of one scanline.
//   FUSED_444  : 4:4:4 YCbCr or RGB -> RGB24, one launch, no halo.
//   FUSED_GRAY : 1 component -> L8, one launch, IDCT written straight to the output rows
//                (compute_image's stride compaction, src/decoder.rs:1310-1332, is just the
//                output pitch).
//
// Each kernel is a sequence of barrier-separated phases written as plain functions of
// (geometry, image, tile, tid, LDS, per-lane registers) so that tests/emu can run the very same
// code on the CPU (g++ -DJPGPU_HOST_EMULATION) against the oracle.  On the GPU the phases are
// called back to back from the __global__ wrappers in fused.hip with __syncthreads() between.
#pragma once
#include "pixel_math.hpp"

namespace jpgpu {

constexpr uint32_t FUSED_NT = 256;       // threads per workgroup
constexpr uint32_t F420_TX_MAX = 64;     // 4 luma blocks per MCU -> <= 256 lanes
constexpr uint32_t F444_TX_MAX = 80;     // 3 blocks per MCU -> <= 240 lanes
constexpr uint32_t FGRAY_TX_MAX = 256;   // 1 block per MCU
constexpr uint32_t F420_CPITCH = 8 * F420_TX_MAX + 16;  // chroma LDS row: 8 halo + 8*TX + 8 halo
constexpr uint32_t FUSED_COEF_LDS = 256 * 128;          // staging area (bytes), aliased by the sample tiles

enum : uint32_t { FCOLOR_YCBCR = 0, FCOLOR_RGB = 1 };
enum FusedKind : int { FUSED_NONE = 0, FUSED_420 = 1, FUSED_444 = 2, FUSED_GRAY = 3 };

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
    uint32_t chroma_plane_bytes;  // 420 scratch: bytes of one chroma plane (bwc*8 * bhc*8)
};

#ifndef JPGPU_FUSED_IMAGE_DEFINED
#define JPGPU_FUSED_IMAGE_DEFINED
struct FusedImage {
    const int16_t *coefs[4];
    const uint16_t *qt[4];
    uint8_t *out;
    uint8_t *scratch;  // 4:2:0: Cb plane followed by Cr plane
    uint32_t flags;    // bit0: every component "sane" (|c*q| < 2^15) -> 24-bit multiply path
    uint32_t _pad;
};
#endif

struct alignas(16) FusedLds {
    uint8_t coef[FUSED_COEF_LDS];           // coefficient staging, later the sample tile(s)
    uint8_t chroma[2 * 10 * F420_CPITCH];   // 4:2:0 only
};

struct FusedRegs {
    uint32_t out[16];  // one IDCT'd block: 8 rows x 2 dwords
};

// LDS slot (in 16-B units) of row k of local block lb: conflict-free for both the 8-lane
// ds_write_b128 groups (one block = 8 consecutive slots) and the 16-lane ds_read_b128 groups
// (MI355X_MICROARCH.md §LDS): lb*8 + (k ^ ((lb >> 1) & 7)).
__device__ __forceinline__ uint32_t coef_slot(uint32_t lb, uint32_t k) { return lb * 8u + (k ^ ((lb >> 1) & 7u)); }

__device__ __forceinline__ void load_block_from_lds(const FusedLds &lds, uint32_t lb, uint32_t (&cw)[32]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(lds.coef);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint4 v = p[coef_slot(lb, (uint32_t)k)];
        cw[k * 4 + 0] = v.x;
        cw[k * 4 + 1] = v.y;
        cw[k * 4 + 2] = v.z;
        cw[k * 4 + 3] = v.w;
    }
}

__device__ __forceinline__ void idct_block(bool sane, const uint32_t (&cw)[32], const uint16_t *q, uint32_t (&out)[16]) {
    if (sane) idct8x8<true>(cw, q, out);
    else idct8x8<false>(cw, q, out);
}

// store n (<= 8) RGB24 pixels held as packed 24-bit values; off = byte offset in `out`
__device__ __forceinline__ void store_rgb_run(uint8_t *out, size_t off, const uint32_t (&px)[8], uint32_t n) {
    uint8_t *o = out + off;
    if (n == 8 && (off & 3u) == 0) {
        uint32_t d0 = px[0] | (px[1] << 24);
        uint32_t d1 = (px[1] >> 8) | (px[2] << 16);
        uint32_t d2 = (px[2] >> 16) | (px[3] << 8);
        uint32_t d3 = px[4] | (px[5] << 24);
        uint32_t d4 = (px[5] >> 8) | (px[6] << 16);
        uint32_t d5 = (px[6] >> 16) | (px[7] << 8);
        if ((off & 7u) == 0) {
            uint2 *o64 = reinterpret_cast<uint2 *>(o);
            o64[0] = make_uint2(d0, d1);
            o64[1] = make_uint2(d2, d3);
            o64[2] = make_uint2(d4, d5);
        } else {
            uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
            o32[0] = d0; o32[1] = d1; o32[2] = d2; o32[3] = d3; o32[4] = d4; o32[5] = d5;
        }
    } else {
        for (uint32_t k = 0; k < n; k++) {
            o[3 * k] = (uint8_t)px[k];
            o[3 * k + 1] = (uint8_t)(px[k] >> 8);
            o[3 * k + 2] = (uint8_t)(px[k] >> 16);
        }
    }
}

__device__ __forceinline__ uint32_t byte_of(uint32_t d, uint32_t i) { return (d >> (8u * i)) & 0xffu; }

// =============================================================================================
// FUSED_420 main pass
// =============================================================================================
struct F420 {
    // effective MCUs of tile `tile_x`
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.mcu_w - tile_x * g.tx);
    }

    // phase 0: stage luma coefficients (two block rows of the MCU row) and the chroma
    // neighbourhood [8*my-1, 8*my+8] x [8*x0-8, 8*(x0+txe)+8) of both chroma planes into LDS.
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLds &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t run = 2u * te;        // luma blocks per block row of the tile
        const uint32_t nchunks = 2u * run * 8u;  // 16-B chunks to stage
        uint4 *dst = reinterpret_cast<uint4 *>(lds.coef);
        const uint4 *src = reinterpret_cast<const uint4 *>(img.coefs[0]);
        for (uint32_t j = tid; j < nchunks; j += FUSED_NT) {
            uint32_t lb = j >> 3, k = j & 7u;
            uint32_t ry = lb / run, cx = lb - ry * run;
            size_t gblock = (size_t)(2u * my + ry) * g.bw0 + (2u * x0m + cx);
            dst[coef_slot(lb, k)] = src[gblock * 8u + k];
        }
        // chroma: uint2 (8 B) granules; LDS column lc <-> plane column 8*x0m - 8 + lc
        const uint32_t stride = g.bwc * 8u;
        const uint32_t gran_per_row = te + 2u;
        const uint32_t items = 2u * 10u * gran_per_row;
        for (uint32_t it = tid; it < items; it += FUSED_NT) {
            uint32_t comp = it / (10u * gran_per_row);
            uint32_t rem = it - comp * 10u * gran_per_row;
            uint32_t rr = rem / gran_per_row, gi = rem - rr * gran_per_row;
            int32_t crow = (int32_t)(8u * my) - 1 + (int32_t)rr;
            int32_t col = (int32_t)(8u * x0m) - 8 + (int32_t)(8u * gi);
            if (crow < 0 || crow >= (int32_t)g.ch || col < 0 || col >= (int32_t)stride) continue;
            const uint8_t *plane = img.scratch + (size_t)comp * g.chroma_plane_bytes;
            uint2 v = *reinterpret_cast<const uint2 *>(plane + (size_t)crow * stride + (uint32_t)col);
            *reinterpret_cast<uint2 *>(&lds.chroma[(comp * 10u + rr) * F420_CPITCH + 8u * gi]) = v;
        }
    }

    // phase 1: one lane per luma block: LDS -> registers -> IDCT
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t tid,
                                                  const FusedLds &lds, FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        if (tid >= 4u * te) return;
        uint32_t cw[32];
        load_block_from_lds(lds, tid, cw);
        idct_block(img.flags & 1u, cw, img.qt[0], r.out);
    }

    // phase 2: luma samples into the LDS tile (16 rows x 16*te bytes, pitch 16*tx), which
    // aliases the (now consumed) coefficient staging area
    static __device__ __forceinline__ void phase2(const FusedGeom &g, uint32_t tile_x, uint32_t tid, FusedLds &lds,
                                                  const FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        if (tid >= 4u * te) return;
        const uint32_t run = 2u * te, ypitch = 16u * g.tx;
        uint32_t ry = tid / run, cx = tid - ry * run;
#pragma unroll
        for (int row = 0; row < 8; row++)
            *reinterpret_cast<uint2 *>(&lds.coef[(ry * 8u + (uint32_t)row) * ypitch + cx * 8u]) =
                make_uint2(r.out[2 * row], r.out[2 * row + 1]);
    }

    // phase 3: upsample + colour convert + store; lanes walk consecutive 8-pixel chunks
    static __device__ __forceinline__ void phase3(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLds &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nch = 2u * te, ypitch = 16u * g.tx;
        const uint32_t units = 16u * nch;
        for (uint32_t u = tid; u < units; u += FUSED_NT) {
            const uint32_t row = u / nch, chk = u - row * nch;
            const uint32_t oy = 16u * my + row, ox0 = 16u * x0m + 8u * chk;
            if (oy >= g.out_h || ox0 >= g.out_w) continue;
            const uint32_t npx = min(8u, g.out_w - ox0);
            // luma
            const uint2 yy = *reinterpret_cast<const uint2 *>(&lds.coef[row * ypitch + 8u * chk]);
            // chroma rows (src/upsampler.rs:200-206)
            const uint32_t near = oy >> 1;
            const uint32_t far = (oy & 1u) ? min(near + 1u, g.ch - 1u) : (near > 0u ? near - 1u : 0u);
            const uint32_t rn = near + 1u - 8u * my, rf = far + 1u - 8u * my;  // LDS rows (row 0 <-> 8*my-1)
            // t'[j] = 3*near[j] + far[j] + 2 for plane columns j0-4 .. j0+7 (j0 = ox0/2); LDS column of j0-4 is 4*chk+4
            uint32_t tp[2][12];
#pragma unroll
            for (uint32_t comp = 0; comp < 2; comp++) {
                const uint8_t *bn = &lds.chroma[(comp * 10u + rn) * F420_CPITCH + 4u * chk + 4u];
                const uint8_t *bf = &lds.chroma[(comp * 10u + rf) * F420_CPITCH + 4u * chk + 4u];
#pragma unroll
                for (uint32_t d = 0; d < 3; d++) {
                    uint32_t nn = *reinterpret_cast<const uint32_t *>(bn + 4u * d);
                    uint32_t ff = *reinterpret_cast<const uint32_t *>(bf + 4u * d);
#pragma unroll
                    for (uint32_t b = 0; b < 4; b++) tp[comp][4 * d + b] = 3u * byte_of(nn, b) + byte_of(ff, b) + 2u;
                }
            }
            uint32_t px[8];
            const uint32_t last_x = 2u * g.cw - 1u;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = ox0 + k;
                const uint32_t ji = 4u + (k >> 1);                 // index of t'[x>>1] in tp
                const uint32_t jo = (k & 1u) ? ji + 1u : ji - 1u;  // the "other" tap
                const bool edge = (x == 0u) || (x == last_x);      // src/upsampler.rs:213-214,226
                uint32_t c[2];
#pragma unroll
                for (uint32_t comp = 0; comp < 2; comp++)
                    c[comp] = edge ? (tp[comp][ji] >> 2) : ((3u * tp[comp][ji] + tp[comp][jo]) >> 4);
                const uint32_t y = byte_of(k < 4 ? yy.x : yy.y, k & 3u);
                px[k] = ycbcr_to_rgb24(y, c[0], c[1]);
            }
            store_rgb_run(img.out, ((size_t)oy * g.out_w + ox0) * 3u, px, npx);
        }
    }
};

// =============================================================================================
// FUSED_444: MCU = one 8x8 block per component
// =============================================================================================
struct F444 {
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.mcu_w - tile_x * g.tx);
    }
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLds &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t nchunks = 3u * te * 8u;
        uint4 *dst = reinterpret_cast<uint4 *>(lds.coef);
        for (uint32_t j = tid; j < nchunks; j += FUSED_NT) {
            uint32_t lb = j >> 3, k = j & 7u;
            uint32_t comp = lb / te, cx = lb - comp * te;
            size_t gblock = (size_t)my * g.bwc + (x0m + cx);
            dst[coef_slot(lb, k)] = reinterpret_cast<const uint4 *>(img.coefs[comp])[gblock * 8u + k];
        }
    }
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t tid,
                                                  const FusedLds &lds, FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        if (tid >= 3u * te) return;
        uint32_t cw[32];
        load_block_from_lds(lds, tid, cw);
        idct_block(img.flags & 1u, cw, img.qt[tid / te], r.out);
    }
    // sample tiles: [3 comps][8 rows][pitch 8*tx]
    static __device__ __forceinline__ void phase2(const FusedGeom &g, uint32_t tile_x, uint32_t tid, FusedLds &lds,
                                                  const FusedRegs &r) {
        const uint32_t te = txe(g, tile_x);
        if (tid >= 3u * te) return;
        const uint32_t pitch = 8u * g.tx;
        uint32_t comp = tid / te, cx = tid - comp * te;
#pragma unroll
        for (int row = 0; row < 8; row++)
            *reinterpret_cast<uint2 *>(&lds.coef[(comp * 8u + (uint32_t)row) * pitch + cx * 8u]) =
                make_uint2(r.out[2 * row], r.out[2 * row + 1]);
    }
    static __device__ __forceinline__ void phase3(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLds &lds) {
        const uint32_t x0m = tile_x * g.tx, te = txe(g, tile_x);
        const uint32_t pitch = 8u * g.tx, units = 8u * te;
        for (uint32_t u = tid; u < units; u += FUSED_NT) {
            const uint32_t row = u / te, chk = u - row * te;
            const uint32_t oy = 8u * my + row, ox0 = 8u * (x0m + chk);
            if (oy >= g.out_h || ox0 >= g.out_w) continue;
            const uint32_t npx = min(8u, g.out_w - ox0);
            uint2 s[3];
#pragma unroll
            for (uint32_t comp = 0; comp < 3; comp++)
                s[comp] = *reinterpret_cast<const uint2 *>(&lds.coef[(comp * 8u + row) * pitch + chk * 8u]);
            uint32_t px[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                uint32_t a = byte_of(k < 4 ? s[0].x : s[0].y, k & 3u);
                uint32_t b = byte_of(k < 4 ? s[1].x : s[1].y, k & 3u);
                uint32_t c = byte_of(k < 4 ? s[2].x : s[2].y, k & 3u);
                px[k] = g.color == FCOLOR_RGB ? (a | (b << 8) | (c << 16)) : ycbcr_to_rgb24(a, b, c);
            }
            store_rgb_run(img.out, ((size_t)oy * g.out_w + ox0) * 3u, px, npx);
        }
    }
};

// =============================================================================================
// FUSED_GRAY: one block per lane, straight to the output rows
// =============================================================================================
struct FGray {
    static __device__ __forceinline__ uint32_t txe(const FusedGeom &g, uint32_t tile_x) {
        return min(g.tx, g.bw0 - tile_x * g.tx);
    }
    static __device__ __forceinline__ void phase0(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, FusedLds &lds) {
        const uint32_t x0 = tile_x * g.tx, te = txe(g, tile_x);
        uint4 *dst = reinterpret_cast<uint4 *>(lds.coef);
        const uint4 *src = reinterpret_cast<const uint4 *>(img.coefs[0]) + ((size_t)my * g.bw0 + x0) * 8u;
        for (uint32_t j = tid; j < te * 8u; j += FUSED_NT) dst[coef_slot(j >> 3, j & 7u)] = src[j];
    }
    static __device__ __forceinline__ void phase1(const FusedGeom &g, const FusedImage &img, uint32_t tile_x, uint32_t my,
                                                  uint32_t tid, const FusedLds &lds) {
        const uint32_t x0 = tile_x * g.tx, te = txe(g, tile_x);
        if (tid >= te) return;
        uint32_t cw[32], out[16];
        load_block_from_lds(lds, tid, cw);
        idct_block(img.flags & 1u, cw, img.qt[0], out);
        const uint32_t ox = 8u * (x0 + tid);
        if (ox >= g.out_w) return;
        const uint32_t n = min(8u, g.out_w - ox);
#pragma unroll
        for (uint32_t row = 0; row < 8; row++) {
            const uint32_t oy = 8u * my + row;
            if (oy >= g.out_h) break;
            const size_t off = (size_t)oy * g.out_w + ox;
            if (n == 8 && (off & 7u) == 0) {
                *reinterpret_cast<uint2 *>(img.out + off) = make_uint2(out[2 * row], out[2 * row + 1]);
            } else {
                for (uint32_t k = 0; k < n; k++) img.out[off + k] = (uint8_t)byte_of(k < 4 ? out[2 * row] : out[2 * row + 1], k & 3u);
            }
        }
    }
};

}  // namespace jpgpu
