// decoder_api.cpp — C ABI of include/jpgpu_decoder.h: the crate's `Decoder` surface
// (src/decoder.rs:101-295) = host front-end (frontend.cpp) + MI355X pixel backend (jpgpu.h).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../host_common.hpp"
#include "frontend.hpp"

using jpgpu::host::DecodeError;
using jpgpu::host::Frontend;
using jpgpu::host::RowSink;

namespace {

// Worker backed by the GPU: rows go straight to jpgpu_worker_append_row, planes stay in HBM.
class GpuSink : public RowSink {
public:
    explicit GpuSink(jpgpu_worker *w) : w_(w) {}
    void check(int rc) {
        if (rc) throw DecodeError{rc, jpgpu_worker_last_error(w_)};
    }
    void start(uint32_t index, const jpgpu_component &c, const uint16_t qt[64]) override {
        const auto t0 = std::chrono::steady_clock::now();
        check(jpgpu_worker_start(w_, index, &c, qt));
        other_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void append_row(uint32_t index, const int16_t *co, size_t len) override {
        const auto t0 = std::chrono::steady_clock::now();
        check(jpgpu_worker_append_row(w_, index, co, len));
        append_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        appends++;
    }
    double append_ms = 0;  // (JPGPU_DECODER_TRACE)
    size_t appends = 0;
    void finish(uint32_t index, uint32_t slot) override {
        const auto t0 = std::chrono::steady_clock::now();
        check(jpgpu_worker_finish_plane(w_, index, slot));
        other_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    double other_ms = 0;  // start + finish_plane (JPGPU_DECODER_TRACE)

private:
    jpgpu_worker *w_;
};

// Collects what crosses the boundary (for the batch driver and for GPU-less tests).
class CoefSink : public RowSink {
public:
    std::vector<int16_t> work[JPGPU_MAX_COMPONENTS];   // by worker index
    std::vector<int16_t> frame[JPGPU_MAX_COMPONENTS];  // by frame component
    void start(uint32_t index, const jpgpu_component &, const uint16_t *) override { work[index].clear(); }
    void append_row(uint32_t index, const int16_t *co, size_t len) override { work[index].insert(work[index].end(), co, co + len); }
    void finish(uint32_t index, uint32_t slot) override {
        std::vector<int16_t> taken;
        taken.swap(work[index]);  // (index may equal slot)
        frame[slot].swap(taken);
    }
};

}  // namespace

// How many decodes may hold a device context (a worker, or a one-image pipeline: streams, pinned staging, device buffers) at the
// same time.  The reference's rayon-2 test spawns 1,024 decoding threads (tests/rayon-2.rs:14-20); its workers are CPU threads
// of one global pool, so 1,024 decodes queue up behind the machine's cores.  Here the resource is the device: decode() calls
// beyond the cap wait for a context to come back instead of creating a thousand streams and a thousand sets of buffers.
// JPGPU_MAX_CONCURRENT_DECODES overrides the default of 64.  A context is held only inside jpgpu_decoder_decode (it goes back
// on the failure paths as well), so waiting cannot deadlock.
namespace {
struct DeviceSlots {
    std::mutex m;
    std::condition_variable cv;
    int in_use = 0, cap = 64;
    DeviceSlots() {
        if (const char *e = getenv("JPGPU_MAX_CONCURRENT_DECODES")) cap = std::max(1, atoi(e));
    }
    void acquire() {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return in_use < cap; });
        in_use++;
    }
    void release() {
        {
            std::lock_guard<std::mutex> g(m);
            in_use--;
        }
        cv.notify_one();
    }
};
DeviceSlots &device_slots() {
    static DeviceSlots *s = new DeviceSlots;
    return *s;
}
struct SlotHold {  // scope guard
    bool held = false;
    void take() {
        if (!held) device_slots().acquire(), held = true;
    }
    void drop() {
        if (held) device_slots().release(), held = false;
    }
    ~SlotHold() { drop(); }
};
}  // namespace

// Idle workers per device: a Decoder borrows one for its decode() and hands it back with its streams, pinned staging and
// device buffers intact — creating those for every image was 4-5 ms, more than decoding a 512x512 image takes.
namespace {
struct WorkerPool {
    std::mutex m;
    std::vector<std::pair<int, jpgpu_worker *>> idle;
    jpgpu_worker *take(int device) {
        std::lock_guard<std::mutex> g(m);
        for (size_t k = 0; k < idle.size(); k++)
            if (idle[k].first == device) {
                jpgpu_worker *w = idle[k].second;
                idle.erase(idle.begin() + (long)k);
                return w;
            }
        return nullptr;
    }
    void give(int device, jpgpu_worker *w) {
        {
            std::lock_guard<std::mutex> g(m);
            if (idle.size() < 16) {
                idle.emplace_back(device, w);
                return;
            }
        }
        jpgpu_worker_destroy(w);
    }
};
WorkerPool &worker_pool() {
    static WorkerPool *p = new WorkerPool;  // (never destroyed: the HIP runtime may be gone before static destructors run)
    return *p;
}
}  // namespace

// Large sequential images: entropy decoding on the device as well (a one-image jpgpu_pipeline_decode with
// JPGPU_PIPELINE_DEVICE_ENTROPY) — 2.2 instead of 12 ms for a 2268x1512 photo.  Idle pipelines are kept like idle workers.
namespace {
struct PipelinePool {
    std::mutex m;
    std::vector<std::pair<int, jpgpu_pipeline *>> idle;
    jpgpu_pipeline *take(int device) {
        std::lock_guard<std::mutex> g(m);
        for (size_t k = 0; k < idle.size(); k++)
            if (idle[k].first == device) {
                jpgpu_pipeline *p = idle[k].second;
                idle.erase(idle.begin() + (long)k);
                return p;
            }
        return nullptr;
    }
    void give(int device, jpgpu_pipeline *p) {
        {
            std::lock_guard<std::mutex> g(m);
            if (idle.size() < 4) {
                idle.emplace_back(device, p);
                return;
            }
        }
        jpgpu_pipeline_destroy(p);
    }
};
PipelinePool &pipeline_pool() {
    static PipelinePool *p = new PipelinePool;
    return *p;
}
constexpr uint64_t kDeviceEntropyMinPixels = 900000;  // below, the Worker path is as fast (round 2, q85 4:2:0: 1024x768 1.32-1.48 vs 1.53 ms, 1280x720 1.49 vs 1.33, 1600x900 2.25 vs 1.72)
}  // namespace

struct jpgpu_decoder {
    std::unique_ptr<Frontend> fe;
    int device = 0;
    jpgpu_worker *worker = nullptr;
    std::string err;
    bool decoded = false;
    bool fe_spent = false;    // decode() went through the device entropy decoder: `fe` holds the metadata, it cannot decode any more
    bool customized = false;  // set_color_transform / scale / set_max_decoding_buffer_size were used: the plain Worker path only
    std::vector<uint8_t> pixels;
    std::vector<uint8_t> icc;
    CoefSink coefs;
    bool coefs_done = false;
};

static int fail(jpgpu_decoder *d, const DecodeError &e) {
    d->err = e.message;
    return e.code;
}

extern "C" {

int jpgpu_decoder_create(const uint8_t *data, size_t len, int device, jpgpu_decoder **out) {
    if (!out || (!data && len)) return JPGPU_ERR_FORMAT;
    jpgpu_decoder *d = new jpgpu_decoder();
    d->fe.reset(new Frontend(data, len));
    d->device = device;
    *out = d;
    return JPGPU_OK;
}

void jpgpu_decoder_destroy(jpgpu_decoder *d) {
    if (!d) return;
    if (d->worker) jpgpu_worker_destroy(d->worker);
    delete d;
}

void jpgpu_trim_caches(void) {
    for (;;) {
        jpgpu_worker *w = nullptr;
        {
            std::lock_guard<std::mutex> g(worker_pool().m);
            if (worker_pool().idle.empty()) break;
            w = worker_pool().idle.back().second;
            worker_pool().idle.pop_back();
        }
        jpgpu_worker_destroy(w);
    }
    for (;;) {
        jpgpu_pipeline *p = nullptr;
        {
            std::lock_guard<std::mutex> g(pipeline_pool().m);
            if (pipeline_pool().idle.empty()) break;
            p = pipeline_pool().idle.back().second;
            pipeline_pool().idle.pop_back();
        }
        jpgpu_pipeline_destroy(p);
    }
    jpgpu::host::trim_coefficient_pool();
}

const char *jpgpu_decoder_last_error(const jpgpu_decoder *d) { return d ? d->err.c_str() : ""; }

int jpgpu_decoder_set_color_transform(jpgpu_decoder *d, int ct) {
    if (!d || ct < 0 || ct > JPGPU_CT_JCS_BG_RGB) return JPGPU_ERR_FORMAT;
    d->fe->set_color_transform(ct);
    d->customized = true;
    return JPGPU_OK;
}

int jpgpu_decoder_set_max_decoding_buffer_size(jpgpu_decoder *d, size_t max_bytes) {
    if (!d) return JPGPU_ERR_FORMAT;
    d->fe->set_max_decoding_buffer_size(max_bytes);
    d->customized = true;
    return JPGPU_OK;
}

int jpgpu_decoder_read_info(jpgpu_decoder *d) {
    if (!d) return JPGPU_ERR_FORMAT;
    try {
        d->fe->read_info();
    } catch (const DecodeError &e) {
        return fail(d, e);
    }
    return JPGPU_OK;
}

int jpgpu_decoder_info(const jpgpu_decoder *d, jpgpu_image_info *info) {
    if (!d || !info || !d->fe->has_frame()) return JPGPU_ERR_FORMAT;
    *info = d->fe->info();
    return JPGPU_OK;
}

int jpgpu_decoder_scale(jpgpu_decoder *d, uint16_t rw, uint16_t rh, uint16_t *ow, uint16_t *oh) {
    if (!d) return JPGPU_ERR_FORMAT;
    try {
        uint16_t w = 0, h = 0;
        d->fe->scale(rw, rh, w, h);
        d->customized = true;
        if (ow) *ow = w;
        if (oh) *oh = h;
    } catch (const DecodeError &e) {
        return fail(d, e);
    }
    return JPGPU_OK;
}

size_t jpgpu_decoder_output_bytes(const jpgpu_decoder *d) {
    if (!d || !d->fe->has_frame()) return 0;
    const jpgpu_image_info i = d->fe->info();
    const size_t bpp = i.pixel_format == JPGPU_PIXEL_L8 ? 1 : i.pixel_format == JPGPU_PIXEL_L16 ? 2 : i.pixel_format == JPGPU_PIXEL_RGB24 ? 3 : 4;
    return (size_t)i.width * i.height * bpp;
}

int jpgpu_decoder_decode(jpgpu_decoder *d, uint8_t *dst, size_t cap, size_t *len) {
    if (!d) return JPGPU_ERR_FORMAT;
    if (!d->decoded && d->fe_spent) {  // (an earlier call delivered straight into the caller's buffer and kept nothing)
        size_t n = 0;
        const uint8_t *bytes = d->fe->stream_bytes(&n);
        std::unique_ptr<Frontend> nf(new Frontend(bytes, n));
        d->fe = std::move(nf);
        d->fe_spent = false;
    }
    if (!d->decoded && d->device >= 0 && !d->customized && !getenv("JPGPU_DECODER_NO_DEVICE_ENTROPY")) {
        // same pixels, other route: see kDeviceEntropyMinPixels.  Anything the planner or the device decoder does not
        // like — and every error — goes through the ordinary path below, whose behaviour is the pinned one.
        try {
            d->fe->read_info();
            const jpgpu_image_info inf = d->fe->info();
            if (inf.coding_process == JPGPU_CODING_DCT_SEQUENTIAL && (uint64_t)inf.width * inf.height >= kDeviceEntropyMinPixels) {
                std::vector<jpgpu::host::PlannedScan> plans;
                const bool eligible = d->fe->plan_device_scans(plans);  // (walks every marker: the metadata accessors are complete afterwards)
                size_t n = 0;
                const uint8_t *bytes = d->fe->stream_bytes(&n);
                bool done = false;
                if (eligible) {
                    SlotHold slot;
                    slot.take();
                    jpgpu_pipeline *p = pipeline_pool().take(d->device);
                    if (!p && jpgpu_pipeline_create(d->device, 2, &p) != JPGPU_OK) p = nullptr;
                    if (p) {
                        if (jpgpu_pipeline_decode(p, &bytes, &n, 1, JPGPU_PIPELINE_DOWNLOAD | JPGPU_PIPELINE_DEVICE_ENTROPY) == JPGPU_OK &&
                            jpgpu_pipeline_image_status(p, 0) == JPGPU_OK) {
                            const size_t nb = jpgpu_pipeline_pixel_bytes(p, 0);
                            const uint8_t *px = jpgpu_pipeline_pixels_host(p, 0);
                            if (px || nb == 0) {
                                done = true;
                                d->fe_spent = true;
                                if (dst && cap >= nb) {  // pinned memory -> the caller's buffer, no copy kept (25 MB for a 4K image)
                                    if (nb) memcpy(dst, px, nb);
                                    if (len) *len = nb;
                                    pipeline_pool().give(d->device, p);
                                    return JPGPU_OK;
                                }
                                d->pixels.assign(px, px + nb);
                                d->decoded = true;
                            }
                        }
                        pipeline_pool().give(d->device, p);
                    }
                }
                if (!done) {  // the planning pass spent the front-end: a fresh one for the ordinary path
                    std::unique_ptr<Frontend> nf(new Frontend(bytes, n));
                    d->fe = std::move(nf);
                }
            }
        } catch (const DecodeError &) {
            size_t n = 0;
            const uint8_t *bytes = d->fe->stream_bytes(&n);
            std::unique_ptr<Frontend> nf(new Frontend(bytes, n));  // (the ordinary path reports the error its own way)
            d->fe = std::move(nf);
        }
    }
    if (!d->decoded) {
        SlotHold slot;
        try {
            if (d->device < 0) throw DecodeError{JPGPU_ERR_NO_DEVICE, "decoder was created without a device (host-only)"};
            slot.take();
            if (!d->worker) d->worker = worker_pool().take(d->device);
            if (!d->worker) {
                int rc = jpgpu_worker_create(d->device, &d->worker);
                if (rc) throw DecodeError{rc, "no usable MI355X device: the pixel pipeline has no CPU fallback"};
            }
            {
                // A header that announces far more blocks than the stream can hold (two bits per block at the very least):
                // let the entropy decoder fail into a sink that keeps nothing before planes are allocated for it (a 200-byte
                // file claiming 65535 x 65535 pixels cost 2.5 s of allocations otherwise).  A stream that decodes goes on.
                d->fe->read_info();
                uint64_t blocks = 0;
                for (uint32_t c = 0; c < d->fe->ncomp(); c++) blocks += (uint64_t)d->fe->components()[c].block_width * d->fe->components()[c].block_height;
                size_t n = 0;
                const uint8_t *bytes = d->fe->stream_bytes(&n);
                if ((uint64_t)n * 8u < blocks * 2u) {
                    struct Nothing : RowSink {
                        void start(uint32_t, const jpgpu_component &, const uint16_t *) override {}
                        void append_row(uint32_t, const int16_t *, size_t) override {}
                        void finish(uint32_t, uint32_t) override {}
                    } nothing;
                    Frontend probe(bytes, n, Frontend::Borrowed{});
                    probe.set_max_decoding_buffer_size(d->fe->max_decoding_buffer_size());
                    probe.read_info();
                    probe.decode_to(nothing);
                }
            }
            const bool trace = getenv("JPGPU_DECODER_TRACE") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
            GpuSink sink(d->worker);
            d->fe->decode_to(sink);
            const double t_decode = ms_since(t0);
            const uint32_t n = d->fe->ncomp();
            const jpgpu_component *comps = d->fe->components();
            const uint16_t w = d->fe->output_width(), h = d->fe->output_height();
            const size_t out_len = n == 1 ? (size_t)comps[0].size_width * comps[0].size_height : (size_t)w * h * n;
            const auto t1 = std::chrono::steady_clock::now();
            d->pixels.resize(out_len ? out_len : 1);
            const double t_alloc = ms_since(t1);
            const auto t2 = std::chrono::steady_clock::now();
            size_t got = 0;
            sink.check(jpgpu_compute_image(d->worker, comps, n, nullptr, w, h, d->fe->color_transform(), d->pixels.data(),
                                           d->pixels.size(), &got));
            d->pixels.resize(got);
            if (trace) fprintf(stderr, "decoder trace: entropy decoding + row uploads %.2f ms (of which %zu append_row calls %.2f ms, start + finish_plane %.2f ms), result buffer %.2f ms, compute_image + download %.2f ms\n", t_decode, sink.appends, sink.append_ms, sink.other_ms, t_alloc, ms_since(t2));
            d->decoded = true;
            jpgpu::worker_recycle(d->worker);
            worker_pool().give(d->device, d->worker);  // the pixels are on the host: the next Decoder may have the worker
            d->worker = nullptr;
        } catch (const DecodeError &e) {
            if (d->worker) {  // the context goes back with the failed decode (a decoder decodes once): nothing is held between calls
                jpgpu::worker_recycle(d->worker);
                worker_pool().give(d->device, d->worker);
                d->worker = nullptr;
            }
            return fail(d, e);
        }
    }
    if (len) *len = d->pixels.size();
    if (!dst || cap < d->pixels.size()) {
        d->err = "decode: destination too small";
        return JPGPU_ERR_FORMAT;
    }
    if (!d->pixels.empty()) memcpy(dst, d->pixels.data(), d->pixels.size());
    return JPGPU_OK;
}

const uint8_t *jpgpu_decoder_exif_data(const jpgpu_decoder *d, size_t *len) {
    const std::vector<uint8_t> *v = d ? d->fe->exif() : nullptr;
    if (len) *len = v ? v->size() : 0;
    return v ? v->data() : nullptr;
}
const uint8_t *jpgpu_decoder_xmp_data(const jpgpu_decoder *d, size_t *len) {
    const std::vector<uint8_t> *v = d ? d->fe->xmp() : nullptr;
    if (len) *len = v ? v->size() : 0;
    return v ? v->data() : nullptr;
}
const uint8_t *jpgpu_decoder_icc_profile(jpgpu_decoder *d, size_t *len) {
    if (len) *len = 0;
    if (!d || !d->fe->icc_profile(d->icc)) return nullptr;
    if (len) *len = d->icc.size();
    return d->icc.data();
}

int jpgpu_decoder_decode_coefficients(jpgpu_decoder *d, jpgpu_image_desc *desc, const int16_t **coefs, size_t *n_coefs) {
    if (!d || !desc || !coefs || !n_coefs) return JPGPU_ERR_FORMAT;
    if (d->fe_spent) {
        size_t n = 0;
        const uint8_t *bytes = d->fe->stream_bytes(&n);
        std::unique_ptr<Frontend> nf(new Frontend(bytes, n));
        d->fe = std::move(nf);
        d->fe_spent = false;
    }
    if (!d->coefs_done) {
        try {
            d->fe->decode_to(d->coefs);
            d->coefs_done = true;
        } catch (const DecodeError &e) {
            return fail(d, e);
        }
    }
    memset(desc, 0, sizeof(*desc));
    desc->ncomp = d->fe->ncomp();
    for (uint32_t c = 0; c < desc->ncomp; c++) {
        desc->components[c] = d->fe->components()[c];
        memcpy(desc->quantization_tables[c], d->fe->qtable_of_component(c), 128);
        coefs[c] = d->coefs.frame[c].data();
        n_coefs[c] = d->coefs.frame[c].size();
    }
    desc->out_w = d->fe->output_width();
    desc->out_h = d->fe->output_height();
    desc->color_transform = d->fe->color_transform();
    return JPGPU_OK;
}

}  // extern "C"
