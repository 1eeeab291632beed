// emu_prog.cpp — TEST-ONLY CPU run of the device decoder for progressive frames (csrc/huff_prog_wave.hpp) on the plan the host
// front-end makes (Frontend::plan_progressive_scans): the tracks of a frame one lane at a time, in the order given (any order must
// give the same planes: tracks share no coefficient).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "hip_shim.hpp"
#include <vector>
#include "../../jpeg-decoder_amd/csrc/host/frontend.hpp"
#include "../../jpeg-decoder_amd/csrc/huff_core.hpp"
#include "../../jpeg-decoder_amd/csrc/huff_prog_wave.hpp"

using namespace jpgpu;
using jpgpu::host::Frontend;
using jpgpu::host::ProgPlan;
using jpgpu::host::ProgPlannedScan;

extern "C" {

// -> 0 and the frame's descriptor, scans and tracks if the planner takes the stream; 1: it stays with the host
int emu_prog_plan(const uint8_t *data, size_t len, jpgpu_image_desc *desc, uint32_t *n_scans, uint32_t *n_tracks) {
    try {
        Frontend fe(data, len, Frontend::Borrowed{});
        fe.read_info();
        ProgPlan plan;
        if (!fe.plan_progressive_scans(plan)) return 1;
        memset(desc, 0, sizeof(*desc));
        desc->ncomp = fe.ncomp();
        for (uint32_t c = 0; c < desc->ncomp; c++) {
            desc->components[c] = fe.components()[c];
            memcpy(desc->quantization_tables[c], fe.qtable_of_component(c), 128);
        }
        desc->out_w = fe.output_width();
        desc->out_h = fe.output_height();
        desc->color_transform = fe.color_transform();
        *n_scans = (uint32_t)plan.scans.size();
        *n_tracks = plan.n_tracks;
        return 0;
    } catch (...) {
        return 1;
    }
}

// The dependencies of scan j as the launch sees them (host::prog_plan_dependencies): deps[3] (-1: none), rank, whole[3] (1: the producer
// must have ended before the scan starts: another walk order).  Returns the number of scans, -1: not eligible, -2: walked as tracks.
int emu_prog_dependencies(const uint8_t *data, size_t len, int32_t *deps, uint32_t *rank, uint32_t *whole, uint32_t cap) {
    Frontend fe(data, len, Frontend::Borrowed{});
    fe.read_info();
    ProgPlan plan;
    if (!fe.plan_progressive_scans(plan)) return -1;
    const host::ProgDependencies pd = host::prog_plan_dependencies(plan);
    if (!pd.ok) return -2;
    for (size_t j = 0; j < plan.scans.size() && j < cap; j++) {
        rank[j] = pd.rank[j];
        for (int w = 0; w < 3; w++) {
            deps[3 * j + w] = pd.deps[j][w];
            whole[3 * j + w] = pd.deps[j][w] >= 0 && !host::prog_same_walk(plan.scans[j], plan.scans[(size_t)pd.deps[j][w]]) ? 1u : 0u;
        }
    }
    return (int)plan.scans.size();
}

// Decode into planes[c] (zero-filled by the caller).  order: 0 = tracks in plan order, 1 = reversed, 2 = interleaved scan by scan
// across tracks (round robin: what concurrent lanes may do to one another's dwords).  Returns the status word (0: decoded).
int emu_prog_decode(const uint8_t *data, size_t len, int16_t *const planes[4], int order) {
    Frontend fe(data, len, Frontend::Borrowed{});
    fe.read_info();
    ProgPlan plan;
    if (!fe.plan_progressive_scans(plan)) return -1;
    const uint32_t nc = fe.ncomp();
    std::vector<std::vector<uint64_t>> masks(nc);
    for (uint32_t c = 0; c < nc; c++) masks[c].assign((size_t)fe.components()[c].block_width * fe.components()[c].block_height * 2u, 0);
    // staging: every scan unstuffed into its own slot (huff_stage_segment), as batch.cpp does
    std::vector<std::vector<uint8_t>> slots(plan.scans.size());
    std::vector<ProgScan> scans(plan.scans.size());
    uint32_t status = 0;
    for (size_t i = 0; i < plan.scans.size(); i++) {
        const ProgPlannedScan &ps = plan.scans[i];
        slots[i].assign(huff_slot_bytes(ps.stuffed_bytes) + 16, 0xEE);
        bool clean = true;
        const uint32_t n = huff_stage_segment(slots[i].data(), data + ps.data_off, ps.stuffed_bytes, &clean);
        if (!clean) status |= PROG_ST_STAGING | PROG_ST_HOST;
        ProgScan &s = scans[i];
        memset(&s, 0, sizeof(s));
        s.data = slots[i].data();
        s.n_bytes = n;
        s.ss = ps.ss, s.se = ps.se, s.ah = ps.ah, s.al = ps.al;
        s.ncomp = ps.ncomp, s.cols = ps.cols, s.rows = ps.rows;
        for (uint32_t c = 0; c < ps.ncomp; c++) {
            s.comp[c].coefs = planes[ps.comp[c].frame_index];
            s.comp[c].masks = masks[ps.comp[c].frame_index].data();
            s.comp[c].block_w = ps.comp[c].block_w;
            s.comp[c].h = ps.comp[c].h;
            s.comp[c].v = ps.comp[c].v;
            s.comp[c].table = ps.comp[c].table;
        }
        for (int t = 0; t < 4; t++) s.table[t] = ps.table[t].get();
    }
    if (status) return (int)status;
    // order 0 / 1: a wave per track, tracks in plan order / reversed; 2: scan by scan, round robin over the tracks (what concurrent waves
    // may do to one another's dwords); 3 / 4: a wave per SCAN with its waits, producers first (stream order); 5: a wave per track
    const host::ProgDependencies pd = host::prog_plan_dependencies(plan);
    std::vector<uint32_t> progress(plan.scans.size(), 0u);
    if (order == 3 || order == 4) {
        if (!pd.ok) return -2;
        for (size_t i = 0; i < plan.scans.size(); i++) {
            scans[i].progress = &progress[i];
            for (int w = 0; w < 3; w++)
                if (pd.deps[i][w] >= 0) {
                    scans[i].wait[w] = &progress[(size_t)pd.deps[i][w]];
                    if (!host::prog_same_walk(plan.scans[i], plan.scans[(size_t)pd.deps[i][w]])) scans[i].wait_whole |= 1u << w;
                }
        }
        for (size_t i = 0; i < plan.scans.size() && !(status & 1u); i++) {
            ProgTrack tr{&scans[i], 1u, &status};
            progw_run_track(tr);
            if (progress[i] != PROG_DONE) status |= 0x8000u;  // a scan must announce its end, whatever happened
        }
    } else if (order == 2) {
        std::vector<std::vector<ProgScan>> pt(plan.n_tracks);
        for (size_t i = 0; i < plan.scans.size(); i++) pt[plan.scans[i].track].push_back(scans[i]);
        size_t longest = 0;
        for (auto &t : pt) longest = std::max(longest, t.size());
        for (size_t step = 0; step < longest; step++)
            for (uint32_t t = 0; t < plan.n_tracks; t++)
                if (step < pt[t].size() && !(status & 1u)) {
                    ProgTrack tr{&pt[t][step], 1u, &status};
                    progw_run_track(tr);
                }
    } else {
        std::vector<std::vector<ProgScan>> pt(plan.n_tracks);
        for (size_t i = 0; i < plan.scans.size(); i++) pt[plan.scans[i].track].push_back(scans[i]);
        for (uint32_t k = 0; k < plan.n_tracks; k++) {
            const uint32_t t = order == 1 ? plan.n_tracks - 1u - k : k;
            ProgTrack tr{pt[t].data(), (uint32_t)pt[t].size(), &status};
            progw_run_track(tr);
        }
    }
    return (int)status;
}

// the C++ twin of the hand-scheduled refinement loop on given states (tests/test_gpu_progw_asm.py compares the device with this)
void emu_progw_refine_fast(void *cases, uint32_t n) {
    PwFastCase *c = static_cast<PwFastCase *>(cases);
    for (uint32_t i = 0; i < n; i++) {
        c[i].table = c[i].lut8;
        pw_refine_fast_case(c[i]);
    }
}
void emu_progw_first_fast(void *cases, uint32_t n) {
    PwFirstCase *c = static_cast<PwFirstCase *>(cases);
    for (uint32_t i = 0; i < n; i++) {
        c[i].table = c[i].lut8;
        pw_first_fast_case(c[i]);
    }
}
void emu_progw_dc_fast(void *cases, uint32_t n) {
    PwDcCase *c = static_cast<PwDcCase *>(cases);
    for (uint32_t i = 0; i < n; i++) pw_dc_fast_case(c[i]);
}
}
