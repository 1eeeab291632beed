// frontend.hpp — host front-end of the decoder: marker loop, frame/scan/table parsers and the
// Huffman / progressive entropy decoder, restated in C++ from the reference's Rust
// (src/decoder.rs:297-1298, src/parser.rs, src/huffman.rs, src/marker.rs).  Its only output is
// what the crate hands to the Worker boundary: per component, `start(RowData)`, one
// `append_row` per MCU row of natural-order i16 coefficients, and `get_result`.
// Entropy decoding is inherently serial and stays on the host; all pixel work is on the GPU.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <array>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/jpgpu.h"
#include "../../../include/jpgpu_decoder.h"
#include "../huff_job.hpp"
#include "../huff_prog_job.hpp"

namespace jpgpu {
namespace host {

struct DecodeError {
    int code;  // JPGPU_ERR_*
    std::string message;
};

// The Worker boundary as the front-end sees it (trait Worker, src/worker/mod.rs:24-35).
// `index` is scan-local in decode_scan and frame-local in decode_planes, exactly as in the
// reference; finish() carries the frame-level slot the plane belongs to.
void trim_coefficient_pool();  // frees the idle accumulation planes of progressive frames (see frontend.cpp, CoefPool)

class RowSink {
public:
    virtual ~RowSink() {}
    // Progressive frames only: called after every scan, once per component of the scan (a clock for tools/host_bench.cpp; round 2-3's
    // per-scan change lists — SURVEY §8f n3, coefficients accumulated on the device — were measured 2.5 x slower than the compact
    // planes and deleted in round 4: profiles/round4/04_progressive_sizing.txt).
    virtual void scan_finished(uint32_t /*frame_slot*/) {}
    // called right before start(index, ...): the frame component this worker index will deliver (finish()'s frame_slot),
    // for sinks that place rows at their final address instead of collecting them
    virtual void frame_slot_hint(uint32_t /*index*/, uint32_t /*frame_slot*/) {}
    virtual void start(uint32_t index, const jpgpu_component &component, const uint16_t qt[64]) = 0;
    virtual void append_row(uint32_t index, const int16_t *coefficients, size_t len) = 0;
    virtual void finish(uint32_t index, uint32_t frame_slot) = 0;  // get_result + keep for compute_image
};

// What the device entropy decoders (csrc/huff_sync_core.hpp) need for one sequential Huffman scan: its restart segments, or
// (no restart interval in force) the whole scan as one segment for the chunk decoder.
struct PlannedScan {
    size_t data_off = 0;            // offset of the scan's entropy-coded bytes in the stream given to the Frontend
    std::vector<uint32_t> seg_off;  // 2 * n_seg offsets relative to data_off: segment s = [seg_off[2s], seg_off[2s+1]), no markers
    uint32_t ri = 0, cols = 0, n_mcu = 0, ncomp = 0;  // ri == 0: no restart interval (one segment)
    bool check_at_staging = false;  // ri == 0: the segment was taken to run up to the last EOI of the stream without looking inside:
                                    // huff_stage_segment must find nothing but 0xFF00 pairs in it, else the image is the host's
    struct Comp {
        uint32_t frame_index, block_w, h, v, dc, ac;
    } comp[4];
    // dc 0..3, ac 0..3 as they stand at this scan, in device form (slot of a table: huff_table_slot).  Shared and immutable: files of one encoder repeat their tables,
    // the planner hands every scan that uses the set it built last the same object (equal pointers = equal tables; 27 kB not
    // copied per file), and the staging code uploads a set once per run of scans that share it.
    struct TableSet {
        DevHuffTable t[8];
    };
    std::shared_ptr<const TableSet> tables;
};

// What the device decoder for PROGRESSIVE frames (csrc/huff_prog_wave.hpp) needs for one scan of such a frame, and which scans of
// the frame depend on one another: scans that (transitively) share a coefficient of a component form a TRACK, decoded by one lane in
// stream order; different tracks touch disjoint coefficients and run side by side.
struct ProgPlannedScan {
    size_t data_off = 0;          // offset of the scan's entropy-coded bytes in the stream
    uint32_t stuffed_bytes = 0;   // up to the marker that ends the scan: nothing but 0xFF00 pairs inside (checked here)
    uint8_t ss = 0, se = 0, ah = 0, al = 0;  // se inclusive
    uint32_t ncomp = 0, cols = 0, rows = 0;
    struct Comp {
        uint32_t frame_index, block_w, h, v, table;
    } comp[4];
    std::shared_ptr<const ProgHuffTable> table[4];  // DC first scans: indexed by DC table id; AC scans: [0]
    uint32_t track = 0;
};
struct ProgPlan {
    std::vector<ProgPlannedScan> scans;  // stream order
    uint32_t n_tracks = 0;
};

// Which scans does a scan of a progressive frame depend on (huff_prog_job.hpp: scans pipelined over lanes)?  For every coefficient it
// covers, the LAST earlier scan that covered it — that one stayed behind ITS predecessors block by block, so staying behind it is
// staying behind them all.  deps[j]: up to three scan numbers (-1: none); rank[j]: how deep the dependencies go; ok = false: some scan
// depends on more than three others or a chain is 64 scans deep — such a frame is walked one lane per track.
struct ProgDependencies {
    std::vector<std::array<int32_t, 3>> deps;
    std::vector<uint32_t> rank;
    bool ok = true;
};
inline ProgDependencies prog_plan_dependencies(const ProgPlan &pl) {
    ProgDependencies out;
    const uint32_t ns = (uint32_t)pl.scans.size();
    out.deps.assign(ns, std::array<int32_t, 3>{-1, -1, -1});
    out.rank.assign(ns, 0u);
    int16_t last[4][64];
    for (auto &row : last)
        for (auto &v : row) v = -1;
    for (uint32_t j = 0; j < ns && out.ok; j++) {
        const ProgPlannedScan &ps = pl.scans[j];
        uint32_t nd = 0;
        for (uint32_t c = 0; c < ps.ncomp && out.ok; c++)
            for (uint32_t k = ps.ss; k <= ps.se; k++) {
                const int32_t w = last[ps.comp[c].frame_index & 3u][k];
                last[ps.comp[c].frame_index & 3u][k] = (int16_t)j;
                if (w < 0 || (nd > 0 && out.deps[j][0] == w) || (nd > 1 && out.deps[j][1] == w) || (nd > 2 && out.deps[j][2] == w)) continue;
                if (nd == 3) {
                    out.ok = false;
                    break;
                }
                out.deps[j][nd++] = w;
                out.rank[j] = std::max(out.rank[j], out.rank[(uint32_t)w] + 1u);
            }
        if (out.rank[j] >= 64u) out.ok = false;
    }
    return out;
}
// do two scans walk the same blocks in the same order (then one can stay behind the other block for block)?
inline bool prog_same_walk(const ProgPlannedScan &a, const ProgPlannedScan &b) {
    bool same = a.ncomp == b.ncomp && a.cols == b.cols && a.rows == b.rows;
    for (uint32_t c = 0; same && c < a.ncomp; c++) same = a.comp[c].frame_index == b.comp[c].frame_index && a.comp[c].h == b.comp[c].h && a.comp[c].v == b.comp[c].v;
    return same;
}

struct IccChunk {
    uint8_t num_markers, seq_no;
    std::vector<uint8_t> data;
};

class Frontend {
public:
    Frontend(const uint8_t *data, size_t len);  // copies the stream (Decoder::new(reader) reads it in, src/decoder.rs:134-154)
    struct Borrowed {};                          // ... or works on the caller's bytes, which must outlive every decoding call
    Frontend(const uint8_t *data, size_t len, Borrowed);
    ~Frontend();

    // Decoder::read_info / scale / decode_internal(false) — throw DecodeError
    void read_info();
    void scale(uint16_t req_w, uint16_t req_h, uint16_t &out_w, uint16_t &out_h);
    // Runs the marker loop to EOI feeding `sink`; afterwards planes_present()[i] tells which
    // frame components produced a plane (compute_image fails if any is missing).
    void decode_to(RowSink &sink);
    // Instead of decoding: walk the markers to EOI and describe every scan for the device entropy decoder.  Returns
    // false — the object is then spent, decode with a fresh Frontend — unless the stream is plainly eligible: 8-bit
    // sequential Huffman, one scan carrying all components (at most 16 blocks per MCU), RST markers — if a restart
    // interval is in force — exactly where and as numbered as the interval says, nothing else inside or after the
    // entropy data.  Anything doubtful (including every error the marker loop would raise) is "not eligible": the host
    // decoder is the one whose behaviour on odd streams is pinned.  On success the tables handed to Worker::start
    // (qtable_of_component) and planes_present() are set as decode_to would have set them.
    bool plan_device_scans(std::vector<PlannedScan> &scans);
    // The same for a PROGRESSIVE frame.  Eligible: 8-bit, Huffman, no restart interval in force at any scan, every scan's
    // data free of markers and fill bytes, every coefficient of every component refined one bit at a time from a first scan on (the
    // order the standard prescribes: the device's refinement scans rely on it), every component with a quantization table at the end
    // of the stream (so that every plane exists, finished or not: src/decoder.rs:643-684), DC tables whose symbols are categories.
    // As with plan_device_scans the object is spent afterwards; qtable_of_component() / planes_present() are set on success.
    bool plan_progressive_scans(ProgPlan &plan);

    const uint8_t *stream_bytes(size_t *len) const;  // the copy of the stream this object works on
    bool has_frame() const;
    jpgpu_image_info info() const;
    int color_transform() const;  // determine_color_transform(), src/decoder.rs:698-764
    uint32_t ncomp() const;
    const jpgpu_component *components() const;
    uint16_t output_width() const;
    uint16_t output_height() const;
    const bool *planes_present() const;
    // table handed to Worker::start for frame component c (valid once its plane is present)
    const uint16_t *qtable_of_component(uint32_t c) const;

    void set_color_transform(int ct);
    void set_max_decoding_buffer_size(size_t n);
    size_t max_decoding_buffer_size() const;

    const std::vector<uint8_t> *exif() const;
    const std::vector<uint8_t> *xmp() const;
    bool icc_profile(std::vector<uint8_t> &out) const;  // src/decoder.rs:213-243

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace host
}  // namespace jpgpu
