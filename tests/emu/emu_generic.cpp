// emu_generic.cpp — TEST-ONLY CPU emulation of the generic upsample + colour kernel: the product's planner
// (csrc/image_job.cpp) and lane body (csrc/upsample_color_body.hpp) run for every (lane, row) of the launch grid.
#include "hip_shim.hpp"
#include <string>
#include <vector>
#include "../../jpeg-decoder_amd/csrc/host_common.hpp"
#include "../../jpeg-decoder_amd/csrc/upsample_color_body.hpp"

using namespace jpgpu;

extern "C" {
// planes[c]: the component's u8 plane (any alignment: copied to 8-byte aligned storage like device planes are).
// Returns the status build_image_job returns; on success `out` holds the image (out_len bytes written to *len).
int emu_compute_image(const jpgpu_component* comps, uint32_t ncomp, const uint8_t* const* planes, uint16_t out_w,
                      uint16_t out_h, int color_transform, uint8_t* out, size_t* len, int force_slow, int* used_fast) {
    std::vector<std::vector<uint64_t>> store(ncomp);
    uint8_t* d_planes[4] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t c = 0; c < ncomp; c++) {
        const size_t n = plane_bytes(comps[c]);
        store[c].assign(n / 8 + 2, 0xCDCDCDCDCDCDCDCDull);
        memcpy(store[c].data(), planes[c], n);
        d_planes[c] = reinterpret_cast<uint8_t*>(store[c].data());
    }
    ImageJob job;
    size_t out_len = 0;
    std::string err;
    int rc = build_image_job(comps, ncomp, d_planes, out_w, out_h, color_transform, out, job, out_len, err);
    if (rc) return rc;
    if (len) *len = out_len;
    if (force_slow) job.fast8 = 0;
    if (used_fast) *used_fast = (int)job.fast8;
    const uint32_t w = job.color_fn == CC_GRAY ? job.comp[0].width : job.out_w;
    const uint32_t h = job.color_fn == CC_GRAY ? job.comp[0].height : job.out_h;
    const uint32_t lanes = ((w + 7u) / 8u + 255u) / 256u * 256u;  // the launch grid, idle lanes included
    for (uint32_t row = 0; row < h; row++)
        for (uint32_t t = 0; t < lanes; t++) upsample_color_lane(job, t * 8u, row);
    return 0;
}
}

// ---- reduced-size decodes in one launch (csrc/fused_scaled.hpp): every workgroup of the launch grid, phase by phase ----------
#include "../../jpeg-decoder_amd/csrc/fused_scaled.hpp"
extern "C" {
// Returns -1 if the planner keeps the image on the generic pair of kernels, else the status of build_image_job (0: `out` holds the image).
int emu_scaled_fused(const jpgpu_image_desc* desc, const int16_t* const* coefs, uint8_t* out, size_t* len, uint32_t* tx_out, uint32_t* tiles_out) {
    ImageJob job;
    size_t out_len = 0;
    std::string err;
    uint8_t* no_planes[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = build_image_job(desc->components, desc->ncomp, no_planes, desc->out_w, desc->out_h, desc->color_transform, out, job, out_len, err);
    if (rc) return rc;
    ScaledGeom g;
    if (!scaled_geom_from_job(desc->components, desc->ncomp, job, g)) return -1;
    if (len) *len = out_len;
    if (tx_out) *tx_out = g.tx;
    if (tiles_out) *tiles_out = g.tiles_x;
    PlaneJob pj[4];
    memset(pj, 0, sizeof(pj));
    for (uint32_t c = 0; c < desc->ncomp; c++) {
        pj[c].coefs = coefs[c];
        pj[c].qt = desc->quantization_tables[c];
        pj[c].block_w = desc->components[c].block_width;
        pj[c].n_blocks = (uint32_t)desc->components[c].block_width * desc->components[c].block_height;
        pj[c].scale = desc->components[c].dct_scale;
    }
    std::vector<uint8_t> lds(g.lds_bytes + 64);
    for (uint32_t my = 0; my < g.bands; my++)  // (a workgroup: a tile of a band of g.ry MCU rows)
        for (uint32_t tile = 0; tile < g.tiles_x; tile++) {
            memset(lds.data(), 0xCD, lds.size());  // garbage, like real LDS
#define RUNS(S)                                                                                            \
    {                                                                                                      \
        for (uint32_t t = 0; t < FS_NT; t++) FScaled<S>::transform(g, pj, tile, my, t, lds.data());         \
        for (uint32_t t = 0; t < FS_NT; t++) FScaled<S>::pixels(g, job, tile, my, t, lds.data());           \
    }
            if (g.scale == 4) RUNS(4) else if (g.scale == 2) RUNS(2) else RUNS(1)
#undef RUNS
        }
    return 0;
}
}
