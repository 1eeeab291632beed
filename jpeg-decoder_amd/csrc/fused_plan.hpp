// fused_plan.hpp — host-only: which fused kernel (if any) serves a frame geometry, and its
// tiling.  No HIP dependency so that tests/emu can use the same planner.
#pragma once
#include <stdint.h>

#include "../../include/jpgpu.h"
#include "fused_core.hpp"
#include "fused_x4.hpp"

namespace jpgpu {

inline bool fused_same_component(const jpgpu_component &a, const jpgpu_component &b) {
    return a.horizontal_sampling_factor == b.horizontal_sampling_factor &&
           a.vertical_sampling_factor == b.vertical_sampling_factor && a.dct_scale == b.dct_scale &&
           a.size_width == b.size_width && a.size_height == b.size_height && a.block_width == b.block_width &&
           a.block_height == b.block_height;
}

// Returns FUSED_* and fills `g`; FUSED_NONE (with `why`) sends the batch down the generic path.
// s420_tx_max: widest strip of the 4:2:0 walk (test knob JPGPU_S420_TX: narrow strips exercise halos and seams on small images)
inline int fused_geom_from_desc(const jpgpu_image_desc &d0, FusedGeom &g, const char *&name, const char *&why, uint32_t s420_tx_max = S420_TX_MAX) {
    g = FusedGeom{};
    for (uint32_t c = 0; c < d0.ncomp; c++)
        if (d0.components[c].dct_scale != 8) {
            why = "scaled IDCT";
            return FUSED_NONE;
        }
    auto hv = [&](uint32_t c, uint32_t h, uint32_t v) {
        return d0.components[c].horizontal_sampling_factor == h && d0.components[c].vertical_sampling_factor == v;
    };
    int kind = FUSED_NONE;
    uint32_t tx_max = 0;
    g.out_w = d0.out_w;
    g.out_h = d0.out_h;
    g.bw0 = d0.components[0].block_width;
    if (d0.ncomp == 3 && hv(0, 2, 2) && hv(1, 1, 1) && hv(2, 1, 1) && d0.color_transform == JPGPU_CT_YCBCR &&
        d0.out_w > 1 && d0.out_h > 1 && fused_same_component(d0.components[1], d0.components[2])) {
        // choose_upsampler (src/upsampler.rs:80-81): an output width/height of 1 overrides H2V2,
        // such frames stay on the generic path
        kind = FUSED_420;
        name = "fused420";
        g.mcu_w = d0.components[1].block_width;
        g.mcu_h = d0.components[1].block_height;
        g.bwc = d0.components[1].block_width;
        g.cw = d0.components[1].size_width;
        g.ch = d0.components[1].size_height;
        if (d0.components[0].block_width != 2u * g.mcu_w || d0.components[0].block_height != 2u * g.mcu_h ||
            d0.out_w > 2u * g.cw || d0.out_h > 2u * g.ch) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = s420_tx_max < 1u ? 1u : (s420_tx_max > S420_TX_MAX ? S420_TX_MAX : s420_tx_max);
        g.strip = 1u;
    } else if (d0.ncomp == 3 && hv(0, 2, 1) && hv(1, 1, 1) && hv(2, 1, 1) && d0.color_transform == JPGPU_CT_YCBCR && d0.out_w > 1 &&
               fused_same_component(d0.components[1], d0.components[2])) {
        // (an output width of 1 overrides H2V1 with H1V1, src/upsampler.rs:80: generic path)
        kind = FUSED_422;
        name = "fused422";
        g.mcu_w = d0.components[1].block_width;
        g.mcu_h = d0.components[1].block_height;
        g.bwc = d0.components[1].block_width;
        g.cw = d0.components[1].size_width;
        g.ch = d0.components[1].size_height;
        if (d0.components[0].block_width != 2u * g.mcu_w || d0.components[0].block_height != g.mcu_h || d0.out_w > 2u * g.cw ||
            d0.out_h > g.ch || d0.out_h > 8u * g.mcu_h) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = F422_TX_MAX;
    } else if (d0.ncomp == 3 && hv(0, 1, 2) && hv(1, 1, 1) && hv(2, 1, 1) && d0.color_transform == JPGPU_CT_YCBCR && d0.out_h > 1 &&
               fused_same_component(d0.components[1], d0.components[2])) {
        // (an output height of 1 overrides H1V2 with H1V1, src/upsampler.rs:81: generic path)
        kind = FUSED_440;
        name = "fused440";
        g.mcu_w = d0.components[1].block_width;
        g.mcu_h = d0.components[1].block_height;
        g.bwc = d0.components[1].block_width;
        g.cw = d0.components[1].size_width;
        g.ch = d0.components[1].size_height;
        if (d0.components[0].block_width != g.mcu_w || d0.components[0].block_height != 2u * g.mcu_h || d0.out_w > g.cw ||
            d0.out_w > 8u * g.mcu_w || d0.out_h > 2u * g.ch) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        // Strips of 48 MCUs (384 px = 1152 B = nine 128-byte lines per output row) rather than the 64 the kernel can take:
        // with 60 (1080p balanced over four strips) every row of every strip starts and ends inside a line, and the walk
        // is bound by the memory system — 0.834 ms with 60, 0.809 with 64/64/64/48, 0.757 with five strips of 48 (256 x 1080p).
        tx_max = 48u;
        g.strip = 1u;
    } else if (d0.ncomp == 3 && hv(1, 1, 1) && hv(2, 1, 1) && d0.color_transform == JPGPU_CT_YCBCR && d0.out_w > 1 && d0.out_h > 1 &&
               fused_same_component(d0.components[1], d0.components[2]) &&
               (hv(0, 4, 1) || hv(0, 4, 2) || hv(0, 1, 4) || hv(0, 2, 4) || hv(0, 4, 4))) {
        // 4:1:1, 4:1:0 and their transposes: neither factor pair is one of the fancy upsamplers' (choose_upsampler,
        // src/upsampler.rs:80-105), the chroma components get UpsamplerGeneric.  (An output width / height of 1 would turn
        // the choice into H1V1: generic path.)
        kind = FUSED_GEN;
        name = "fusedgen";
        const uint32_t H = d0.components[0].horizontal_sampling_factor, V = d0.components[0].vertical_sampling_factor;
        g.hs = H == 4 ? 2u : (H == 2 ? 1u : 0u);
        g.vs = V == 4 ? 2u : (V == 2 ? 1u : 0u);
        g.mcu_w = d0.components[1].block_width;
        g.mcu_h = d0.components[1].block_height;
        g.bwc = d0.components[1].block_width;
        g.cw = d0.components[1].size_width;
        g.ch = d0.components[1].size_height;
        if (d0.components[0].block_width != H * g.mcu_w || d0.components[0].block_height != V * g.mcu_h || d0.out_w > H * g.cw ||
            d0.out_h > V * g.ch || d0.out_w > 8u * H * g.mcu_w || d0.out_h > 8u * V * g.mcu_h) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = fgen_tx_max(g.hs, g.vs);
    } else if (d0.ncomp == 3 && hv(0, 1, 1) && hv(1, 1, 1) && hv(2, 1, 1) &&
               (d0.color_transform == JPGPU_CT_YCBCR || d0.color_transform == JPGPU_CT_RGB) &&
               fused_same_component(d0.components[0], d0.components[1]) &&
               fused_same_component(d0.components[1], d0.components[2])) {
        kind = FUSED_444;
        name = "fused444";
        g.mcu_w = d0.components[0].block_width;
        g.mcu_h = d0.components[0].block_height;
        g.bwc = d0.components[0].block_width;
        g.color = d0.color_transform == JPGPU_CT_RGB ? FCOLOR_RGB : FCOLOR_YCBCR;
        if (d0.out_w > 8u * g.mcu_w || d0.out_h > 8u * g.mcu_h) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = F444_TX_MAX;
    } else if (d0.ncomp == 4 && hv(0, 1, 1) && hv(1, 1, 1) && hv(2, 1, 1) && hv(3, 1, 1) &&
               (d0.color_transform == JPGPU_CT_CMYK || d0.color_transform == JPGPU_CT_YCCK) &&
               fused_same_component(d0.components[0], d0.components[1]) && fused_same_component(d0.components[1], d0.components[2]) &&
               fused_same_component(d0.components[2], d0.components[3])) {
        kind = FUSED_444;
        name = "fused444x4";
        g.mcu_w = d0.components[0].block_width;
        g.mcu_h = d0.components[0].block_height;
        g.bwc = d0.components[0].block_width;
        g.color = d0.color_transform == JPGPU_CT_CMYK ? FCOLOR_CMYK : FCOLOR_YCCK;
        if (d0.out_w > 8u * g.mcu_w || d0.out_h > 8u * g.mcu_h) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = F444_TX_MAX;
    } else if (d0.ncomp == 4 && hv(0, 2, 2) && hv(1, 1, 1) && hv(2, 1, 1) && (hv(3, 1, 1) || hv(3, 2, 2)) &&
               (d0.color_transform == JPGPU_CT_CMYK || d0.color_transform == JPGPU_CT_YCCK) && d0.out_w > 1 && d0.out_h > 1 &&
               fused_same_component(d0.components[1], d0.components[2]) &&
               fused_same_component(d0.components[3], d0.components[hv(3, 2, 2) ? 0 : 1])) {
        // jpg-cmyk-2.jpg (22 11 11 11: M, Y, K through UpsamplerH2V2) and YCCK with K at full size (22 11 11 22): fused_x4.hpp.
        // (An output width / height of 1 overrides H2V2, src/upsampler.rs:80-81: generic path.)
        kind = FUSED_420X4;
        g.k_full = hv(3, 2, 2) ? 1u : 0u;
        name = g.k_full ? "fused420x4-2212" : "fused420x4-2211";
        g.mcu_w = d0.components[1].block_width;
        g.mcu_h = d0.components[1].block_height;
        g.bwc = d0.components[1].block_width;
        g.cw = d0.components[1].size_width;
        g.ch = d0.components[1].size_height;
        g.color = d0.color_transform == JPGPU_CT_CMYK ? FCOLOR_CMYK : FCOLOR_YCCK;
        if (d0.components[0].block_width != 2u * g.mcu_w || d0.components[0].block_height != 2u * g.mcu_h || d0.out_w > 2u * g.cw ||
            d0.out_h > 2u * g.ch) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        // (a strip walk like 4:2:0's since round 5: W4, fused_x4.hpp; the same test knob narrows its strips)
        tx_max = w4_tx_max(g.k_full != 0u);
        if (s420_tx_max < tx_max) tx_max = s420_tx_max < 1u ? 1u : s420_tx_max;
        g.strip = 1u;
    } else if (d0.ncomp == 1 && hv(0, 1, 1)) {
        kind = FUSED_GRAY;
        name = "fusedgray";
        g.mcu_w = d0.components[0].block_width;
        g.mcu_h = d0.components[0].block_height;
        g.out_w = d0.components[0].size_width;  // compute_image's 1-component size (src/decoder.rs:1314-1316)
        g.out_h = d0.components[0].size_height;
        if (g.out_w > 8u * g.mcu_w || g.out_h > 8u * g.mcu_h) {
            why = "inconsistent block grid";
            return FUSED_NONE;
        }
        tx_max = FGRAY_TX_MAX;
    } else {
        why = "no fused kernel for this sampling / colour transform";
        return FUSED_NONE;
    }
    if (g.mcu_w == 0 || g.mcu_h == 0 || g.mcu_h > 65535u) {
        why = "grid limits";
        return FUSED_NONE;
    }
    g.kind = (uint32_t)kind;
    uint32_t n_tiles = (g.mcu_w + tx_max - 1) / tx_max;
    g.tx = (g.mcu_w + n_tiles - 1) / n_tiles;
    // four components (32-bit pixels, 16 pixels per MCU = 64 bytes of an output row): an even number of MCUs per strip makes every
    // strip's piece of a row a whole number of 128-byte lines (the row kernel of rounds 3-4: eight balanced tiles of 15 MCUs, every
    // piece starting and ending inside a line, 0.97 ms against 0.80 with tiles of 16)
    if (kind == FUSED_420X4 && (g.tx & 1u) && g.tx < tx_max && g.tx < g.mcu_w) g.tx++;
    g.tiles_x = (g.mcu_w + g.tx - 1) / g.tx;
    g.seg_rows = g.mcu_h;
    g.n_seg = 1;
    return kind;
}

// S420: split the MCU rows of a strip between workgroups.  A seam costs one extra transform round (the chroma blocks
// above and below, 35 % of a step) and re-reads two chroma block rows, so segments are as long as the machine allows:
// about 2300 workgroups per launch for the 1024 the chip holds at once (measured on 256 x 1080p, 768 strips: 3 segments
// of 23 rows 0.650 ms, 4 of 17 0.667, 8 of 9 0.663, 2 of 34 0.687 — profiles/round2/02_single_launch_420.md), never
// fewer than the strips themselves; a lone image is cut down to single rows for the sake of parallelism.
inline void s420_set_segments(FusedGeom &g, uint32_t n_images, uint32_t seg_rows_override = 0) {
    uint32_t seg = seg_rows_override;
    if (seg == 0) {
        // (4:4:0: 0.766 / 0.758 / 0.754 ms with 2 / 3 / 4 segments of its five strips)
        // (round 4, with the colour phase at the lowest wave priority: 256 x 1080p — 768 strips — 2 segments of 34 rows 0.648 ms, 3 of 23
        // 0.663, 4 of 17 0.6445-0.656, 5 of 14 0.652; 64 x 2160p — 384 strips — 4 of 34 0.641, 6 of 23 0.693, 8 of 17 0.660: about 1,536
        // workgroups, one and a half times what the chip holds, instead of round 2's 2,304; profiles/round4/10_wave_priorities.txt)
        // (four components with half-size ones, 192 x 1080p, 4-5 strips: 2 segments of 34 rows 0.687 / 0.774 ms (CMYK 22 11 11 11 / YCCK
        // 22 11 11 22), 3 of 23 0.671 / 0.770, 4 of 17 0.629-0.644 / 0.709-0.731, 5 of 14 0.651 / 0.750, 8 of 9 0.642 / 0.741: profiles/round5/17_*)
        const uint32_t target = g.kind == FUSED_440 ? 4608u : (g.kind == FUSED_420X4 ? 3400u : 1536u);
        const uint32_t per_seg = g.tiles_x * (n_images ? n_images : 1u);
        uint32_t n_seg = (target + per_seg / 2u) / per_seg;
        n_seg = n_seg < 1u ? 1u : n_seg;
        seg = (g.mcu_h + n_seg - 1) / n_seg;
    }
    if (seg < 1u) seg = 1u;
    if (seg > g.mcu_h) seg = g.mcu_h;
    g.seg_rows = seg;
    g.n_seg = (g.mcu_h + seg - 1) / seg;
}

}  // namespace jpgpu
