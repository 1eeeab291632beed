// emu_fused.cpp — TEST-ONLY CPU emulation of the product's fused kernels: runs the very same
// phase functions (jpeg-decoder_amd/csrc/fused_core.hpp) for every workgroup / lane with
// sequential "barriers", so tile / halo / edge logic can be checked against the oracle without a
// GPU.  Not part of the product; the product path only ever runs on gfx950.
#include "hip_shim.hpp"
#include <algorithm>
#include <vector>
#include "../../jpeg-decoder_amd/csrc/fused_plan.hpp"

using namespace jpgpu;

// the strip-walk kernels (S420, S440): fused.hip's walk_item, phase by phase, for every work item of the launch — segments of
// `seg_rows` MCU rows, one item per workgroup
template <class K>
static void run_walk(const FusedGeom& g, const FusedImage& img) {
    constexpr uint32_t NT = K::NT;
    std::vector<uint8_t> mem(K::Lds::total_bytes(g.tx) + 64);
    std::vector<S420Regs> regs(NT);
    std::vector<FusedWork> items;
    std::vector<uint32_t> wg_first;
    for (uint32_t seg = 0; seg < g.n_seg; seg++)
        for (uint32_t strip = 0; strip < g.tiles_x; strip++) {
            wg_first.push_back((uint32_t)items.size());
            items.push_back(FusedWork{0u, strip, seg * g.seg_rows, std::min((seg + 1u) * g.seg_rows, g.mcu_h)});
        }
    wg_first.push_back((uint32_t)items.size());
#define LANES(BODY) for (uint32_t t = 0; t < NT; t++) { BODY; }
    for (size_t wg = 0; wg + 1 < wg_first.size(); wg++) {
        memset(mem.data(), 0xCD, mem.size());  // garbage, like real LDS
        for (uint32_t it = wg_first[wg]; it < wg_first[wg + 1]; it++) {
            const typename K::Lds lds = K::Lds::make(mem.data(), g.tx);
            const uint32_t strip = items[it].a, k0 = items[it].b, k1 = items[it].c;
            LANES(K::init(img, t, lds))
            if (k0 > 0 || k1 < g.mcu_h) {
                LANES(K::seam_stage(g, img, strip, k0, k1, t, lds))
                LANES(K::seam_transform(g, strip, k0, k1, t, lds))
            }
            std::vector<typename K::Pre> pre(NT);
            LANES(K::stage_load(g, img, strip, k0, t, pre[t]))
            LANES(K::stage_store(g, strip, t, lds, pre[t]))
            for (uint32_t k = k0; k < k1; k++) {
                LANES(K::read_block(g, strip, t, lds, regs[t]))
                LANES(K::transform(g, strip, t, lds, regs[t]))
                if (k + 1 < k1) LANES(K::stage_load(g, img, strip, k + 1, t, pre[t]))
                LANES(K::colour(g, img, strip, k, 16u * k0, false, t, lds))
                if (k + 1 < k1) LANES(K::stage_store(g, strip, t, lds, pre[t]))
            }
            if (16u * k1 - 1u < g.out_h) {
                LANES(K::closing_tiles(t, lds))
                LANES(K::colour(g, img, strip, k1, 16u * k0, true, t, lds))
            }
        }
    }
#undef LANES
}

extern "C" {

// returns the fused kind the planner picked (0 = none -> generic path on the GPU)
int emu_fused_decode(const jpgpu_image_desc* desc, const int16_t* const* coefs, int sane, uint8_t* out,
                     uint32_t* tx_out, uint32_t seg_rows, uint32_t s420_tx_max) {
    FusedGeom g;
    const char *name = "", *why = "";
    int kind = fused_geom_from_desc(*desc, g, name, why, s420_tx_max ? s420_tx_max : S420_TX_MAX);
    if (kind == FUSED_NONE) return 0;
    if ((kind == FUSED_420 || kind == FUSED_440) && g.strip) {
        s420_set_segments(g, 1, seg_rows);
        FusedImage im{};
        for (uint32_t c = 0; c < desc->ncomp; c++) {
            im.coefs[c] = coefs[c];
            im.qt[c] = desc->quantization_tables[c];
        }
        im.out = out;
        if (tx_out) *tx_out = g.tx;
        im.flags = sane == 2 ? 3u : (sane ? 1u : 0u);
        if (kind == FUSED_440) {
            if (sane == 2) run_walk<S440<ARITH_TIGHT>>(g, im);
            else if (sane) run_walk<S440<ARITH_SANE>>(g, im);
            else run_walk<S440<ARITH_EXACT>>(g, im);
        } else if (sane == 2) run_walk<S420<ARITH_TIGHT, 256>>(g, im);
        else if (sane) run_walk<S420<ARITH_SANE, 256>>(g, im);
        else run_walk<S420<ARITH_EXACT, 256>>(g, im);
        return kind;
    }
    if (kind == FUSED_420X4) {  // w4_kernel (fused_x4.hpp): a strip walk like the two above
        s420_set_segments(g, 1, seg_rows);
        FusedImage im{};
        for (uint32_t c = 0; c < desc->ncomp; c++) {
            im.coefs[c] = coefs[c];
            im.qt[c] = desc->quantization_tables[c];
        }
        im.out = out;
        if (tx_out) *tx_out = g.tx;
        im.flags = sane == 2 ? 3u : (sane ? 1u : 0u);
        if (g.k_full) {
            if (sane == 2) run_walk<W4<ARITH_TIGHT, true>>(g, im);
            else if (sane) run_walk<W4<ARITH_SANE, true>>(g, im);
            else run_walk<W4<ARITH_EXACT, true>>(g, im);
        } else {
            if (sane == 2) run_walk<W4<ARITH_TIGHT, false>>(g, im);
            else if (sane) run_walk<W4<ARITH_SANE, false>>(g, im);
            else run_walk<W4<ARITH_EXACT, false>>(g, im);
        }
        return kind;
    }
    if (kind == FUSED_GEN) {  // fgen_kernel, phase by phase
        FusedImage im{};
        for (uint32_t c = 0; c < desc->ncomp; c++) {
            im.coefs[c] = coefs[c];
            im.qt[c] = desc->quantization_tables[c];
        }
        im.out = out;
        if (tx_out) *tx_out = g.tx;
        std::vector<uint8_t> mem(FGenLds::total_bytes(g.tx, g.hs, g.vs) + 64);
        std::vector<S420Regs> regs(256);
        auto run = [&](auto K) {
            typedef decltype(K) KK;
            std::vector<typename KK::Pre> pre(256);
            for (uint32_t my = 0; my < g.mcu_h; my++)
                for (uint32_t tile = 0; tile < g.tiles_x; tile++) {
                    memset(mem.data(), 0xCD, mem.size());
                    const FGenLds lds = FGenLds::make(mem.data(), g.tx, g.hs, g.vs);
                    for (uint32_t t = 0; t < 256; t++) KK::init(im, t, lds);
                    for (uint32_t t = 0; t < 256; t++) KK::stage_load(g, im, tile, my, t, pre[t]);
                    for (uint32_t t = 0; t < 256; t++) KK::stage_store(g, tile, t, lds, pre[t]);
                    for (uint32_t t = 0; t < 256; t++) KK::read_block(g, tile, t, lds, regs[t]);
                    for (uint32_t t = 0; t < 256; t++) KK::transform(g, tile, t, lds, regs[t]);
                    for (uint32_t t = 0; t < 256; t++) KK::colour(g, im, tile, my, t, lds);
                }
        };
        if (sane == 2) run(FGen<ARITH_TIGHT>{});
        else if (sane) run(FGen<ARITH_SANE>{});
        else run(FGen<ARITH_EXACT>{});
        return kind;
    }
    if (tx_out) *tx_out = g.tx;
    FusedImage img{};
    for (uint32_t c = 0; c < desc->ncomp; c++) {
        img.coefs[c] = coefs[c];
        img.qt[c] = desc->quantization_tables[c];
    }
    img.out = out;
    img.flags = sane == 2 ? 3u : (sane ? 1u : 0u);
    FusedLdsSmall* lds_s = new FusedLdsSmall;
    std::vector<FusedRegs> regs(FUSED_NT);
#define RUN(NTH, BODY) for (uint32_t t = 0; t < NTH; t++) { BODY; }
    for (uint32_t my = 0; my < g.mcu_h; my++)
        for (uint32_t tile = 0; tile < g.tiles_x; tile++) {
            memset(lds_s, 0xCD, sizeof(FusedLdsSmall));
            if (kind == FUSED_422) {
#define RUN422(S) RUN(256, F422<S>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, F422<S>::phase1(g, img.qt[(t >> 6) < 2 ? 0 : (t >> 6) - 1], tile, t, *lds_s, regs[t])) \
    RUN(256, F422<S>::phase2(g, tile, t, *lds_s, regs[t])) RUN(256, F422<S>::phase3(g, img, tile, my, t, *lds_s))
                if (sane == 2) { RUN422(ARITH_TIGHT) } else if (sane) { RUN422(ARITH_SANE) } else { RUN422(ARITH_EXACT) }
#undef RUN422
            } else if (kind == FUSED_444 && sane == 2) {
                RUN(256, F444<ARITH_TIGHT>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, F444<ARITH_TIGHT>::phase1(g, img.qt[std::min(t >> 6, desc->ncomp - 1u)], tile, t, *lds_s, regs[t]))
                RUN(256, F444<ARITH_TIGHT>::phase2(g, tile, t, *lds_s, regs[t])) RUN(256, F444<ARITH_TIGHT>::phase3(g, img, tile, my, t, *lds_s))
            } else if (kind == FUSED_444 && sane) {
                RUN(256, F444<ARITH_SANE>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, F444<ARITH_SANE>::phase1(g, img.qt[std::min(t >> 6, desc->ncomp - 1u)], tile, t, *lds_s, regs[t]))
                RUN(256, F444<ARITH_SANE>::phase2(g, tile, t, *lds_s, regs[t])) RUN(256, F444<ARITH_SANE>::phase3(g, img, tile, my, t, *lds_s))
            } else if (kind == FUSED_444) {
                RUN(256, F444<ARITH_EXACT>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, F444<ARITH_EXACT>::phase1(g, img.qt[std::min(t >> 6, desc->ncomp - 1u)], tile, t, *lds_s, regs[t]))
                RUN(256, F444<ARITH_EXACT>::phase2(g, tile, t, *lds_s, regs[t])) RUN(256, F444<ARITH_EXACT>::phase3(g, img, tile, my, t, *lds_s))
            } else if (sane == 2) {
                RUN(256, FGray<ARITH_TIGHT>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, FGray<ARITH_TIGHT>::phase1(g, img, tile, my, t, *lds_s))
            } else if (sane) {
                RUN(256, FGray<ARITH_SANE>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, FGray<ARITH_SANE>::phase1(g, img, tile, my, t, *lds_s))
            } else {
                RUN(256, FGray<ARITH_EXACT>::phase0(g, img, tile, my, t, *lds_s)) RUN(256, FGray<ARITH_EXACT>::phase1(g, img, tile, my, t, *lds_s))
            }
        }
#undef RUN
    delete lds_s;
    return kind;
}
}
