// frontend.hpp — host front-end of the decoder: marker loop, frame/scan/table parsers and the
// Huffman / progressive entropy decoder, restated in C++ from the reference's Rust
// (src/decoder.rs:297-1298, src/parser.rs, src/huffman.rs, src/marker.rs).  Its only output is
// what the crate hands to the Worker boundary: per component, `start(RowData)`, one
// `append_row` per MCU row of natural-order i16 coefficients, and `get_result`.
// Entropy decoding is inherently serial and stays on the host; all pixel work is on the GPU.
#pragma once
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include "../../../include/jpgpu.h"
#include "../../../include/jpgpu_decoder.h"

namespace jpgpu {
namespace host {

struct DecodeError {
    int code;  // JPGPU_ERR_*
    std::string message;
};

// The Worker boundary as the front-end sees it (trait Worker, src/worker/mod.rs:24-35).
// `index` is scan-local in decode_scan and frame-local in decode_planes, exactly as in the
// reference; finish() carries the frame-level slot the plane belongs to.
class RowSink {
public:
    virtual ~RowSink() {}
    // called right before start(index, ...): the frame component this worker index will deliver (finish()'s frame_slot),
    // for sinks that place rows at their final address instead of collecting them
    virtual void frame_slot_hint(uint32_t /*index*/, uint32_t /*frame_slot*/) {}
    virtual void start(uint32_t index, const jpgpu_component &component, const uint16_t qt[64]) = 0;
    virtual void append_row(uint32_t index, const int16_t *coefficients, size_t len) = 0;
    virtual void finish(uint32_t index, uint32_t frame_slot) = 0;  // get_result + keep for compute_image
};

struct IccChunk {
    uint8_t num_markers, seq_no;
    std::vector<uint8_t> data;
};

class Frontend {
public:
    Frontend(const uint8_t *data, size_t len);
    ~Frontend();

    // Decoder::read_info / scale / decode_internal(false) — throw DecodeError
    void read_info();
    void scale(uint16_t req_w, uint16_t req_h, uint16_t &out_w, uint16_t &out_h);
    // Runs the marker loop to EOI feeding `sink`; afterwards planes_present()[i] tells which
    // frame components produced a plane (compute_image fails if any is missing).
    void decode_to(RowSink &sink);

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
