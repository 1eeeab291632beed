// fused_scaled.hpp — reduced-size decodes (Decoder::scale, src/decoder.rs:278-290: dct_scale 4 / 2 / 1, src/idct.rs:456-565) in ONE
// launch: coefficients in, interleaved pixels out, the reduced sample planes never leave LDS.  Rounds 1-3 ran such images on the
// generic pair of kernels (idct_planes_kernel<4|2|1> -> u8 planes in HBM -> upsample_color_kernel): 2.64 GB moved for 2.00 GB
// algorithmic at scale 4, 0.36-0.40 of the roofline (profiles/round3/pmc_traffic.json).
//
// A BAND kernel in address order (what the row kernel of rounds 3-4 was for four-component frames with half-size components, fused_x4.hpp): a workgroup owns `tx` MCUs of `ry`
// consecutive MCU rows of one image (ry = 1 in the first version: every chroma block row of a 4:2:0 image was then transformed by
// three workgroups — its own and, as part of their rings, the ones above and below; 10 block transforms per MCU for 6 blocks).
//   1. transform: one lane per block (up to FS_BLOCKS_PER_LANE rounds) — the tile's own blocks of every component and, for the
//      components under one of the fancy upsamplers (UpsamplerH2V1 / H1V2 / H2V2, src/upsampler.rs:134-228: they read one sample
//      beyond a block), the ring of blocks around them: at a reduced size a whole neighbour block is 16 / 4 / 1 samples and a
//      fraction of a full transform, and its coefficients are what the neighbouring workgroups read at about the same time (L2 /
//      infinity cache, not HBM).  A lane fetches the `SCALE` 16-byte pieces of its block that the reduced IDCT reads (rows
//      0 .. SCALE-1), all of its blocks' pieces before the first use, and writes SCALE x SCALE samples into the component's LDS
//      plane — the reference's plane (stride = blocks x dct_scale, src/worker/immediate.rs:39-60) cut to the tile and its ring;
//   2. pixels: the reference's row functions on those planes — every upsampler and colour function in the forms of the generic
//      path's lane body (upsample_color_body.hpp), with ABSOLUTE plane coordinates (near / far rows, first / last column) mapped
//      into the tile's copy by one offset per component.  Four pixels per unit, units dealt to the lanes in row-major order.
// Every sampling layout, colour function and component count build_image_job accepts takes this kernel when all components are at
// one reduced scale; images of different sizes and kinds share a launch (per-image geometry table).
// No HIP dependency in the planner (tests/emu runs the same phases on the CPU).
#pragma once
#include <stdint.h>

#include "fused_core.hpp"
#include "jobs.hpp"
#include "pixel_math.hpp"
#include "upsample_color_body.hpp"

namespace jpgpu {

constexpr uint32_t FS_NT = 256, FS_BLOCKS_PER_LANE = 4;

struct ScaledGeom {
    uint32_t scale, ncomp;
    uint32_t mcu_w, mcu_h;          // MCUs across / down
    uint32_t tx, tiles_x;           // MCUs per tile (a multiple of 8: tiles then begin at multiples of eight pixels at every scale)
    uint32_t ry, bands;             // MCU rows per workgroup, workgroups down the image
    uint32_t inv_full[4], inv_rest[4];  // ceil(2^20 / blocks of component c per block row of a full tile / of the last, narrower tile): b / n = b * inv >> 20 for b * n < 2^20
    uint32_t hmax, vmax;
    uint32_t h[4], v[4];            // blocks of component c per MCU
    uint32_t halo[4];               // 1: the ring of neighbour blocks is transformed as well (fancy upsamplers)
    uint32_t block_w[4], block_h[4];
    uint32_t lds_off[4], pitch[4];  // the component's LDS plane: (ry * v + 2 halo) * scale rows of `pitch` = tx * h * scale + 8 bytes — four bytes
                                    // of margin either side of the tile's own samples (the ring's columns are the inner ones of them):
                                    // sample columns that are multiples of 4 in the plane are so in LDS (aligned dword reads)
    uint32_t lds_bytes;
    uint32_t first_plane_job;       // index of component 0's PlaneJob in the launch's table (the others follow)
};

// Which images take the kernel, and their tiling.  `job` = what build_image_job made of the frame (upsampler kinds).
// tx_cap / ry_cap: widest tile in MCUs, most MCU rows per workgroup (test / tuning knobs JPGPU_SCALED_TX / JPGPU_SCALED_RY; 64 and 8 by default)
inline bool scaled_geom_from_job(const jpgpu_component *comps, uint32_t ncomp, const ImageJob &job, ScaledGeom &g, uint32_t tx_cap = 64u, uint32_t ry_cap = 8u) {  // (ry_cap up to 16)
    g = ScaledGeom{};
    if (ncomp == 0 || ncomp > 4) return false;
    const uint32_t scale = comps[0].dct_scale;
    if (scale != 4u && scale != 2u && scale != 1u) return false;
    uint32_t hmax = 0, vmax = 0;
    for (uint32_t c = 0; c < ncomp; c++) {
        if (comps[c].dct_scale != scale) return false;
        hmax = hmax > comps[c].horizontal_sampling_factor ? hmax : comps[c].horizontal_sampling_factor;
        vmax = vmax > comps[c].vertical_sampling_factor ? vmax : comps[c].vertical_sampling_factor;
    }
    if (hmax == 0 || vmax == 0 || hmax > 4 || vmax > 4) return false;
    g.scale = scale, g.ncomp = ncomp, g.hmax = hmax, g.vmax = vmax;
    g.mcu_w = comps[0].block_width / comps[0].horizontal_sampling_factor;
    g.mcu_h = comps[0].block_height / comps[0].vertical_sampling_factor;
    if (g.mcu_w == 0 || g.mcu_h == 0 || g.mcu_h > 65535u) return false;
    for (uint32_t c = 0; c < ncomp; c++) {
        g.h[c] = comps[c].horizontal_sampling_factor, g.v[c] = comps[c].vertical_sampling_factor;
        g.block_w[c] = comps[c].block_width, g.block_h[c] = comps[c].block_height;
        if (g.block_w[c] != g.mcu_w * g.h[c] || g.block_h[c] != g.mcu_h * g.v[c]) return false;  // (not a grid update_component_sizes makes)
        const uint32_t k = job.comp[c].kind;
        g.halo[c] = (job.color_fn != CC_GRAY && (k == UP_H2V1 || k == UP_H1V2 || k == UP_H2V2)) ? 1u : 0u;
    }
    const uint32_t cap = FS_NT * FS_BLOCKS_PER_LANE;
    auto blocks_of = [&](uint32_t te, uint32_t re) {  // what a workgroup of te x re MCUs transforms, rings included
        uint32_t n = 0;
        for (uint32_t c = 0; c < ncomp; c++) n += (te * g.h[c] + 2u * g.halo[c]) * (re * g.v[c] + 2u * g.halo[c]);
        return n;
    };
    auto lds_of = [&](uint32_t tx, uint32_t ry) {
        uint32_t off = 0;
        for (uint32_t c = 0; c < ncomp; c++) off = (off + (tx * g.h[c] * scale + 8u) * (ry * g.v[c] + 2u * g.halo[c]) * scale + 15u) & ~15u;
        return off;
    };
    if (blocks_of(8u, 1u) > cap) return false;
    tx_cap = (tx_cap < 8u ? 8u : (tx_cap > 64u ? 64u : tx_cap)) & ~7u;
    // Tile width and rows per workgroup: the transform phase deals blocks to 256 lanes in rounds, and what a round costs does not depend
    // on how many of its lanes hold a block — the shape whose workgroups need the fewest rounds for the whole image wins: full rounds,
    // and rings that are small next to what they surround (1080p 4:2:0 at scale 4, 256 images: 24 x 1 MCUs = 252 blocks, one full
    // round per MCU row, 0.507 ms; 64 x 1 = 652 blocks in three rounds 0.528; 40 x 1 0.552; 32 x 1 0.619; 24 x 4 = 696 blocks in three
    // rounds for FOUR MCU rows, 0.433 ms; 16 x 8 = 872 blocks in four rounds for EIGHT rows 0.424: profiles/round4/06_scaled_kernel.txt).
    // Among equally good shapes the taller one, then the one with fewer workgroups.
    uint32_t best_tx = 8u, best_ry = 1u;
    uint64_t best_slots = ~0ull, best_wgs = ~0ull;  // (rows are tried in rising order: `<=` below lets the taller band win a tie)
    static const uint32_t kRows[] = {1u, 2u, 3u, 4u, 6u, 8u, 12u, 16u};
    for (uint32_t ry : kRows) {
        if (ry > 1u && (ry > g.mcu_h || ry > ry_cap)) break;
        for (uint32_t tx = 8u; tx <= tx_cap; tx += 8u) {
            if (blocks_of(tx, ry) > cap || lds_of(tx, ry) > 24u * 1024u) break;
            const uint32_t full_x = g.mcu_w / tx, rest_x = g.mcu_w - full_x * tx, full_y = g.mcu_h / ry, rest_y = g.mcu_h - full_y * ry;
            uint64_t slots = 0;
            for (uint32_t ky = 0; ky < 2u; ky++)
                for (uint32_t kx = 0; kx < 2u; kx++) {
                    const uint32_t te = kx ? rest_x : tx, re = ky ? rest_y : ry;
                    const uint64_t n_of = (uint64_t)(kx ? (rest_x ? 1u : 0u) : full_x) * (ky ? (rest_y ? 1u : 0u) : full_y);
                    if (n_of && te && re) slots += n_of * ((blocks_of(te, re) + FS_NT - 1u) / FS_NT);
                }
            const uint64_t wgs = (uint64_t)(full_x + (rest_x ? 1u : 0u)) * (full_y + (rest_y ? 1u : 0u));
            if (slots < best_slots || (slots == best_slots && (ry > best_ry || wgs < best_wgs))) best_slots = slots, best_wgs = wgs, best_tx = tx, best_ry = ry;
            if (tx >= g.mcu_w) break;  // (wider tiles change nothing)
        }
    }
    g.tx = best_tx;
    g.ry = best_ry;
    g.tiles_x = (g.mcu_w + g.tx - 1u) / g.tx;
    g.bands = (g.mcu_h + g.ry - 1u) / g.ry;
    const uint32_t rest_x = g.mcu_w - (g.tiles_x - 1u) * g.tx;
    uint32_t off = 0;
    for (uint32_t c = 0; c < ncomp; c++) {
        g.inv_full[c] = ((1u << 20) + (g.tx * g.h[c] + 2u * g.halo[c]) - 1u) / (g.tx * g.h[c] + 2u * g.halo[c]);
        g.inv_rest[c] = ((1u << 20) + (rest_x * g.h[c] + 2u * g.halo[c]) - 1u) / (rest_x * g.h[c] + 2u * g.halo[c]);
        g.pitch[c] = g.tx * g.h[c] * scale + 8u;
        g.lds_off[c] = off;
        off += g.pitch[c] * (g.ry * g.v[c] + 2u * g.halo[c]) * scale;
        off = (off + 15u) & ~15u;
    }
    g.lds_bytes = off;
    return true;
}
// name of the path for jpgpu_batch_path: "fused420-s4", "fused444-s2", "fusedgray-s1", ... ("fusedscaled-sN" for the other layouts)
inline const char *scaled_path_name(const ScaledGeom &g) {
    static const char *names[5][3] = {{"fused420-s4", "fused420-s2", "fused420-s1"}, {"fused444-s4", "fused444-s2", "fused444-s1"},
                                      {"fusedgray-s4", "fusedgray-s2", "fusedgray-s1"}, {"fused422-s4", "fused422-s2", "fused422-s1"},
                                      {"fusedscaled-s4", "fusedscaled-s2", "fusedscaled-s1"}};
    const uint32_t si = g.scale == 4u ? 0u : (g.scale == 2u ? 1u : 2u);
    auto is = [&](uint32_t c, uint32_t h, uint32_t v) { return g.h[c] == h && g.v[c] == v; };
    uint32_t kind = 4;
    if (g.ncomp == 1) kind = 2;
    else if (g.ncomp == 3 && is(1, 1, 1) && is(2, 1, 1)) kind = is(0, 2, 2) ? 0u : (is(0, 1, 1) ? 1u : (is(0, 2, 1) ? 3u : 4u));
    return names[kind][si];
}

template <int SCALE>
struct FScaled {
    static constexpr uint32_t R = SCALE == 4 ? 4u : (SCALE == 2 ? 2u : 1u);  // 16-byte pieces of a block the reduced IDCT reads

    static __device__ __forceinline__ uint32_t txe(const ScaledGeom &g, uint32_t tile) { return min(g.tx, g.mcu_w - tile * g.tx); }
    static __device__ __forceinline__ uint32_t rye(const ScaledGeom &g, uint32_t band) { return min(g.ry, g.mcu_h - band * g.ry); }
    static __device__ __forceinline__ uint32_t tile_blocks(const ScaledGeom &g, uint32_t te, uint32_t re) {
        uint32_t n = 0;
        for (uint32_t c = 0; c < g.ncomp; c++) n += (te * g.h[c] + 2u * g.halo[c]) * (re * g.v[c] + 2u * g.halo[c]);
        return n;
    }

    // phase 1: blocks -> samples in the LDS planes
    static __device__ __forceinline__ void transform(const ScaledGeom &g, const PlaneJob *__restrict__ pj, uint32_t tile, uint32_t band, uint32_t tid,
                                                     uint8_t *lds) {
        const uint32_t te = txe(g, tile), re = rye(g, band), x0m = tile * g.tx, my = band * g.ry, total = tile_blocks(g, te, re);
        v4u pc[FS_BLOCKS_PER_LANE][R];
        uint32_t comp[FS_BLOCKS_PER_LANE], at[FS_BLOCKS_PER_LANE];  // component; LDS byte offset of the block's first sample (~0: no block)
#pragma unroll
        for (uint32_t i = 0; i < FS_BLOCKS_PER_LANE; i++) {
            uint32_t b = tid + FS_NT * i, c = 0;
            at[i] = 0xffffffffu;
            comp[i] = 0;
            if (b >= total) continue;
            uint32_t nbx = te * g.h[0] + 2u * g.halo[0], cnt = nbx * (re * g.v[0] + 2u * g.halo[0]);
            while (b >= cnt) {  // (<= 3 steps)
                b -= cnt;
                c++;
                nbx = te * g.h[c] + 2u * g.halo[c];
                cnt = nbx * (re * g.v[c] + 2u * g.halo[c]);
            }
            const uint32_t by = (b * (te == g.tx ? g.inv_full[c] : g.inv_rest[c])) >> 20, bx = b - by * nbx;  // (b / nbx: exact, b * nbx < 2^20)
            const int32_t gbx = (int32_t)(x0m * g.h[c] + bx) - (int32_t)g.halo[c], gby = (int32_t)(my * g.v[c] + by) - (int32_t)g.halo[c];
            if (gbx < 0 || gby < 0 || gbx >= (int32_t)g.block_w[c] || gby >= (int32_t)g.block_h[c]) continue;  // outside the plane: never read
            comp[i] = c;
            // (block column bx of the tile's row: its own blocks start at LDS column 4, the left ring block ends there)
            at[i] = g.lds_off[c] + by * (uint32_t)SCALE * g.pitch[c] + 4u + (bx - g.halo[c]) * (uint32_t)SCALE;
            const JP_GLOBAL v4u *src = (const JP_GLOBAL v4u *)(pj[c].coefs + ((size_t)gby * g.block_w[c] + (size_t)gbx) * 64u);
            // (components with a ring are read again by the workgroups of the neighbouring tiles and rows, at about the same
            // time: plain loads, so that those find them in the L2 / infinity cache; the others are touched once — first version:
            // streaming loads everywhere, 3.37 GB of HBM traffic for 2.00 algorithmic)
#pragma unroll
            for (uint32_t r = 0; r < R; r++) pc[i][r] = g.halo[c] ? src[r] : stream_load(src + r);
        }
#pragma unroll
        for (uint32_t i = 0; i < FS_BLOCKS_PER_LANE; i++) {
            if (at[i] == 0xffffffffu) continue;
            const uint32_t c = comp[i];
            uint32_t cw[32];
#pragma unroll
            for (uint32_t k = 0; k < 32; k++) cw[k] = 0u;
#pragma unroll
            for (uint32_t r = 0; r < R; r++) cw[4 * r] = pc[i][r].x, cw[4 * r + 1] = pc[i][r].y, cw[4 * r + 2] = pc[i][r].z, cw[4 * r + 3] = pc[i][r].w;
            const qtab_t q = as_qtab(pj[c].qt);
            uint8_t *dst = lds + at[i];
            if constexpr (SCALE == 4) {
                uint32_t out[4];
                idct4x4_exact(cw, q, out);
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) *reinterpret_cast<uint32_t *>(dst + r * g.pitch[c]) = out[r];
            } else if constexpr (SCALE == 2) {
                const uint32_t o = idct2x2_exact(cw, q);
                *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(o & 0xffffu);
                *reinterpret_cast<uint16_t *>(dst + g.pitch[c]) = (uint16_t)(o >> 16);
            } else {
                dst[0] = (uint8_t)idct1x1_exact(cw[0], q);
            }
        }
    }

    // One component's view of its LDS plane for the pixel phase: `off0` is the plane's LDS offset minus the position of the tile's
    // first sample (mod 2^32), so that lds[off0 + row * pitch + x] with ABSOLUTE plane coordinates (what the reference's row functions
    // compute with: near / far rows, first / last column) addresses the tile's copy; off0 and pitch are multiples of 4.
    struct View {
        uint32_t off0, pitch, kind, hf, vf, width, height;
    };
    static __device__ __forceinline__ View view_of(const ScaledGeom &g, const ImageJob &job, uint32_t c, uint32_t tile, uint32_t band) {
        const uint32_t my = band * g.ry;
        const uint32_t col0 = tile * g.tx * g.h[c] * (uint32_t)SCALE - 4u, row0 = (my * g.v[c] - g.halo[c]) * (uint32_t)SCALE;  // (wrapping)
        const UpComp &u = job.comp[c];
        return View{g.lds_off[c] - (row0 * g.pitch[c] + col0), g.pitch[c], u.kind, u.hf, u.vf, u.width, u.height};
    }
    static __device__ __forceinline__ uint32_t dword_at(const uint8_t *lds, uint32_t off) { return *reinterpret_cast<const uint32_t *>(lds + off); }
    // samples s[-1 .. 4] around column j0 (a multiple of 4) of the row at `base`: s[0] = column j0 - 1
    static __device__ __forceinline__ void fetch6(const uint8_t *lds, uint32_t base, uint32_t j0, uint32_t (&s)[6]) {
        const uint32_t a = dword_at(lds, base + j0 - 4u), b = dword_at(lds, base + j0), c = dword_at(lds, base + j0 + 4u);
        s[0] = a >> 24, s[1] = b & 0xffu, s[2] = (b >> 8) & 0xffu, s[3] = (b >> 16) & 0xffu, s[4] = b >> 24, s[5] = c & 0xffu;
    }
    static __device__ __forceinline__ void fetch8(const uint8_t *lds, uint32_t base, uint32_t x0, uint32_t (&s)[8]) {
        const uint32_t a = dword_at(lds, base + x0), b = dword_at(lds, base + x0 + 4u);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) s[k] = ((k < 4 ? a : b) >> (8u * (k & 3u))) & 0xffu;
    }
    // UpsamplerXxx::upsample_row (src/upsampler.rs:119-250) for the eight output samples x0 .. x0 + 7 (x0 a multiple of 8) of
    // output row `row` — up_sample8 of upsample_color_body.hpp on the LDS plane.  Samples past the image are computed from
    // whatever the margins hold and never stored.
    static __device__ __forceinline__ void sample8(const uint8_t *lds, const View &u, uint32_t x0, uint32_t row, uint32_t (&out)[8]) {
        const uint32_t W = u.width;
        if (u.kind == UP_H1V1) {  // :119-132
            fetch8(lds, u.off0 + row * u.pitch, x0, out);
        } else if (u.kind == UP_H1V2) {  // :165-189
            uint32_t near, far, n[8], f[8];
            near_far(row, u.height, near, far);
            fetch8(lds, u.off0 + near * u.pitch, x0, n);
            fetch8(lds, u.off0 + far * u.pitch, x0, f);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) out[k] = (3u * n[k] + f[k] + 2u) >> 2;
        } else if (u.kind == UP_H2V1) {  // :134-163
            uint32_t s[6];
            fetch6(lds, u.off0 + row * u.pitch, x0 >> 1, s);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = x0 + k, m = s[1u + (k >> 1)], o = (k & 1u) ? s[2u + (k >> 1)] : s[k >> 1];
                out[k] = (x == 0u || x == 2u * W - 1u) ? m : (3u * m + o + 2u) >> 2;
            }
        } else if (u.kind == UP_H2V2) {  // :191-228
            uint32_t near, far, n[6], f[6], t[6];
            near_far(row, u.height, near, far);
            fetch6(lds, u.off0 + near * u.pitch, x0 >> 1, n);
            fetch6(lds, u.off0 + far * u.pitch, x0 >> 1, f);
#pragma unroll
            for (uint32_t m = 0; m < 6; m++) t[m] = 3u * n[m] + f[m];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t x = x0 + k, tm = t[1u + (k >> 1)], to = (k & 1u) ? t[2u + (k >> 1)] : t[k >> 1];
                out[k] = (x == 0u || x == 2u * W - 1u) ? (tm + 2u) >> 2 : (3u * tm + to + 8u) >> 4;
            }
        } else {  // Generic :230-250 (sample (row / vf, x / hf); columns past the plane's last one read the last)
            const uint32_t base = u.off0 + (row / u.vf) * u.pitch;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) out[k] = lds[base + min((x0 + k) / u.hf, W - 1u)];
        }
    }

    // phase 2: the tile's output pixels, eight per unit (Upsampler::upsample_and_interleave_row + the colour functions,
    // src/upsampler.rs:47-63, src/decoder.rs:1391-1484; the 1-component copy of compute_image, :1310-1332).  First version: four per
    // unit, byte reads, one switch per sample — 600 vector instructions per unit, the launch bound by them (0.91 ms per 256 x 1080p
    // at scale 4 where the generic pair of kernels took 0.62).
    static __device__ __forceinline__ void pixels(const ScaledGeom &g, const ImageJob &job, uint32_t tile, uint32_t band, uint32_t tid, const uint8_t *lds) {
        const uint32_t te = txe(g, tile), re = rye(g, band), nc = g.ncomp, fn = job.color_fn, my = band;
        const uint32_t x0 = tile * g.tx * g.hmax * (uint32_t)SCALE, y0 = band * g.ry * g.vmax * (uint32_t)SCALE;
        const uint32_t width = te * g.hmax * (uint32_t)SCALE, rows = re * g.vmax * (uint32_t)SCALE;
        const uint32_t upr = (width + 7u) / 8u, units = upr * rows;
        const View v0 = view_of(g, job, 0u, tile, my), v1 = view_of(g, job, nc > 1u ? 1u : 0u, tile, my), v2 = view_of(g, job, nc > 2u ? 2u : 0u, tile, my),
                   v3 = view_of(g, job, nc > 3u ? 3u : 0u, tile, my);
        const uint32_t out_w = fn == CC_GRAY ? v0.width : job.out_w, out_h = fn == CC_GRAY ? v0.height : job.out_h;
        JP_GLOBAL uint8_t *out = (JP_GLOBAL uint8_t *)job.out;
        if (fn == CC_YCBCR && nc == 3u && v0.kind == UP_H1V1 && v1.kind == UP_H2V2 && v2.kind == UP_H2V2 && v1.width == v2.width && v1.height == v2.height) {
            // 4:2:0 YCbCr (uniform per image): the packed 16-bit row arithmetic of the full-size walk (PixelOps::tprime / row_pixels,
            // fused_core.hpp: H2V2 + colour conversion for 8 pixels in ~120 vector instructions, the first / last column of the
            // image fixed up inside) on the rows of the LDS planes — the general loop below spends ~290 on the same eight pixels,
            // and at 158 M vector wave-instructions per 256 x 1080p launch at scale 4 the kernel was bound by them (profiles/round4).
            typedef PixelOps<ARITH_EXACT> P;
            FusedGeom fg{};
            fg.cw = v1.width, fg.out_w = out_w;
            for (uint32_t un = tid; un < units; un += FS_NT) {
                const uint32_t r = un / upr, x = x0 + 8u * (un - r * upr), row = y0 + r;
                if (row >= out_h || x >= out_w) continue;
                uint32_t near, far;
                near_far(row, v1.height, near, far);
                const uint32_t jn = near * v1.pitch + (x >> 1) - 4u, jf = far * v1.pitch + (x >> 1) - 4u;  // (the row functions read columns j0 - 4 .. j0 + 7)
                const typename P::TPrime t[2] = {P::tprime(P::load_eo(lds + (v1.off0 + jn)), P::load_eo(lds + (v1.off0 + jf))),
                                                 P::tprime(P::load_eo(lds + (v2.off0 + jn)), P::load_eo(lds + (v2.off0 + jf)))};
                const uint32_t yo = v0.off0 + row * v0.pitch + x;
                const v2u yy = {dword_at(lds, yo), dword_at(lds, yo + 4u)};
                const size_t off = ((size_t)row * out_w + x) * 3u;
                // (plain stores: a lane's two 12-byte pieces fill half of every line an instruction touches, and with the non-temporal
                // hint such partial lines leave one by one — profiles/round2/01_store_patterns.txt)
                P::template row_pixels<false, true, false, false>(fg, out + off, (off & 3u) == 0u, t, yy, x);
            }
            return;
        }
        for (uint32_t un = tid; un < units; un += FS_NT) {
            const uint32_t r = un / upr, x = x0 + 8u * (un - r * upr), row = y0 + r;
            if (row >= out_h || x >= out_w) continue;
            const uint32_t n = min(8u, out_w - x);
            uint32_t s[4][8];
            sample8(lds, v0, x, row, s[0]);
            if (nc > 1u) sample8(lds, v1, x, row, s[1]);
            if (nc > 2u) sample8(lds, v2, x, row, s[2]);
            if (nc > 3u) sample8(lds, v3, x, row, s[3]);
            if (fn == CC_GRAY) {
                JP_GLOBAL uint8_t *o = out + (size_t)row * out_w + x;
                if (n == 8u && (((size_t)row * out_w) & 3u) == 0u) {
                    reinterpret_cast<JP_GLOBAL uint32_t *>(o)[0] = s[0][0] | (s[0][1] << 8) | (s[0][2] << 16) | (s[0][3] << 24);
                    reinterpret_cast<JP_GLOBAL uint32_t *>(o)[1] = s[0][4] | (s[0][5] << 8) | (s[0][6] << 16) | (s[0][7] << 24);
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (k < n) o[k] = (uint8_t)s[0][k];
                }
                continue;
            }
            if (fn == CC_NONE) {  // color_no_convert: planar within the row
#pragma unroll
                for (uint32_t c = 0; c < 4; c++)
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (c < nc && k < n) out[(size_t)row * out_w * nc + (size_t)c * out_w + x + k] = (uint8_t)s[c][k];
                continue;
            }
            uint32_t px[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                if (fn == CC_RGB) px[k] = s[0][k] | (s[1][k] << 8) | (s[2][k] << 16);
                else if (fn == CC_YCBCR) px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]);
                else if (fn == CC_YCCK) px[k] = ycbcr_to_rgb24(s[0][k], s[1][k], s[2][k]) | ((255u - s[3][k]) << 24);
                else px[k] = (255u - s[0][k]) | ((255u - s[1][k]) << 8) | ((255u - s[2][k]) << 16) | ((255u - s[3][k]) << 24);
            }
            const size_t off = ((size_t)row * out_w + x) * nc;
            if (nc == 4u) {
                JP_GLOBAL uint8_t *o = out + off;
                if (n == 8u) {  // (4-byte aligned whatever the image's width)
                    *reinterpret_cast<JP_GLOBAL v4u_a4 *>(o) = v4u{px[0], px[1], px[2], px[3]};
                    *reinterpret_cast<JP_GLOBAL v4u_a4 *>(o + 16) = v4u{px[4], px[5], px[6], px[7]};
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++)
                        if (k < n) reinterpret_cast<JP_GLOBAL uint32_t *>(o)[k] = px[k];
                }
            } else if (n == 8u && (off & 3u) == 0u) {  // (store_rgb_run with plain stores)
                JP_GLOBAL uint8_t *o = out + off;
                *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o) = v3u{px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8)};
                *reinterpret_cast<JP_GLOBAL v3u_a4 *>(o + 12) = v3u{px[4] | (px[5] << 24), (px[5] >> 8) | (px[6] << 16), (px[6] >> 16) | (px[7] << 8)};
            } else {
                store_rgb_run(out, off, px, n);
            }
        }
    }
};

}  // namespace jpgpu
