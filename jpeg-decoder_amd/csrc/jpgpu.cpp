// jpgpu.cpp — C ABI (include/jpgpu.h) over the gfx950 kernels: Worker + compute_image
// (the crate's drop-in boundary) and the batch driver.  Host-side logic mirrors
// src/worker/rayon.rs / src/worker/mod.rs / src/decoder.rs:1300-1389 of the reference; the
// arithmetic lives in the kernels.  There is NO CPU fallback: without a usable HIP device
// every compute entry point fails with JPGPU_ERR_NO_DEVICE.
#include "../../include/jpgpu.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fused.hpp"
#include "host_common.hpp"
#include "huff.hpp"
#include "kernels.hpp"
#include "range_stats.hpp"

using namespace jpgpu;

extern "C" {

const char *jpgpu_version(void) { return "jpgpu 0.2 (gfx950)"; }

int jpgpu_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? JPGPU_OK : JPGPU_ERR_NO_DEVICE;
}

// Opt-in process set-up (include/jpgpu.h).  jpgpu_pipeline_decode keeps several sub-batches in flight on 16 compute + 4 copy
// streams; the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a
// queue serialise (4,096 1080p files per call: 91 ms with 4 queues, 64 ms with 16, 58 with 24).  The runtime reads the variable
// when it initialises — at the first HIP call of the process — so a host calls this before it touches HIP, or exports the
// variable itself.  (Rounds 1-3 did this from a load-time constructor: a silent side effect on every HIP user of the process.)
int jpgpu_process_init(void) {
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;
    return setenv("GPU_MAX_HW_QUEUES", "24", 0) == 0 ? 1 : 0;
}

const char *jpgpu_status_string(int status) {
    switch (status) {
    case JPGPU_OK: return "Ok";
    case JPGPU_ERR_FORMAT: return "Format";
    case JPGPU_ERR_UNSUPPORTED: return "Unsupported";
    case JPGPU_ERR_IO: return "Io";
    case JPGPU_ERR_INTERNAL: return "Internal";
    case JPGPU_ERR_NO_DEVICE: return "NoDevice";
    default: return "Unknown";
    }
}

}  // extern "C"

namespace jpgpu {

const RoctxApi &roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        if (getenv("JPGPU_NO_ROCTX")) return a;
        void *h = nullptr;
        for (const char *name : {"libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!h) return a;
        a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
        return a;
    }();
    return api;
}

int use_device(int device, std::string &err) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(err, JPGPU_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return set_err(err, JPGPU_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return set_err(err, JPGPU_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return JPGPU_OK;
}

}  // namespace jpgpu

// ------------------------------------------------------------------------------------------
// Worker
// ------------------------------------------------------------------------------------------
struct jpgpu_worker {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    struct Slot {
        bool started = false;
        jpgpu_component c{};
        uint16_t *d_qt = nullptr;
        int16_t *d_coefs = nullptr;
        size_t coef_cap = 0;
        int16_t *h_pinned = nullptr;
        size_t pinned_cap = 0;
        uint8_t *d_plane = nullptr;
        size_t plane_cap = 0;
        uint16_t *h_qt = nullptr;  // pinned copies of the callers' tables (a ring of QT_RING): the upload does not have to finish before
                                   // start() returns, and start() does not have to wait for the stream (a wait on an idle GPU costs ~8 ms)
        uint32_t qt_next = 0, qt_unsynced = 0;
        hipEvent_t uploaded = nullptr;  // recorded behind the last row upload of the slot's previous use (non-interleaved and
        bool upload_pending = false;    // progressive scans reuse worker index 0 for one component after the other)
        size_t rows = 0;      // MCU rows appended since start()
        size_t sent_rows = 0; // MCU rows whose upload has been enqueued
        size_t idct_rows = 0; // MCU rows already transformed
    } slot[JPGPU_MAX_COMPONENTS];
    struct Frame {
        uint8_t *d_plane = nullptr;
        size_t len = 0, cap = 0;
        // deferred: the plane has not been transformed — its coefficients (a complete plane at dct_scale 8) wait in d_coefs,
        // and compute_image runs the fused kernel of the frame's kind on them (coefficients -> pixels in one launch, as in a
        // batch) if every component came this way; anything else that needs the samples transforms them first (materialise)
        bool deferred = false;
        int16_t *d_coefs = nullptr;
        size_t coef_cap = 0;
        jpgpu_component c{};
    } frame[JPGPU_MAX_COMPONENTS];
    std::vector<std::pair<int16_t *, size_t>> spare_coefs;  // coefficient buffers between uses, like spare_planes
    uint16_t *d_frame_qt = nullptr;                          // 4 x 64: the tables of the deferred planes, by frame slot
    jpgpu::FusedPlan fplan;                                  // one-image plan of the last geometry that took the fused route
    bool fplan_valid = false;
    uint64_t fplan_bound[6] = {0, 0, 0, 0, 0, 0};            // what the plan's tables were last bound to (coefficients x 4, pixels, tables)
    uint32_t *d_cls = nullptr;                               // RS_WORDS statistics words + 4 class-source bytes (all CLS_FROM_DEVICE)
    jpgpu_image_desc fplan_desc{};
    std::string last_path = "generic";
    // planes that went out of use (a frame slot was overwritten): start() takes them back instead of allocating —
    // a worker that decodes image after image reaches a state without any hipMalloc / hipFree (each a device-wide sync)
    std::vector<std::pair<uint8_t *, size_t>> spare_planes;
    uint8_t *d_out = nullptr;
    size_t out_cap = 0;
    uint8_t *h_out = nullptr;  // pinned bounce buffer for results going to pageable memory (see worker_download)
    size_t h_out_cap = 0;
    uint8_t *d_tmp[JPGPU_MAX_COMPONENTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t tmp_cap[JPGPU_MAX_COMPONENTS] = {0, 0, 0, 0};
};

// JPGPU_WORKER_TRACE=1: report runtime calls of the Worker that take longer than 0.2 ms (diagnostics)
static const bool g_worker_trace = getenv("JPGPU_WORKER_TRACE") != nullptr;
#define W_HIP(call)                                                                                     \
    do {                                                                                                \
        const auto _t0 = std::chrono::steady_clock::now();                                              \
        hipError_t _e = (call);                                                                         \
        if (g_worker_trace) {                                                                           \
            const double _ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - _t0).count(); \
            if (_ms > 0.2) fprintf(stderr, "worker trace: %.2f ms  %s (%s)\n", _ms, #call, __func__);      \
        }                                                                                               \
        if (_e != hipSuccess) return set_err(w->err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

// rows staged in pinned memory but not yet on their way: one copy for all of them (a copy per MCU row was 200 enqueues per
// 1080p image)
static int worker_send_rows(jpgpu_worker *w, uint32_t index) {
    auto &s = w->slot[index];
    if (s.sent_rows == s.rows) return JPGPU_OK;
    const size_t per_row = (size_t)s.c.block_width * s.c.vertical_sampling_factor * 64;
    W_HIP(hipMemcpyAsync(s.d_coefs + s.sent_rows * per_row, s.h_pinned + s.sent_rows * per_row, (s.rows - s.sent_rows) * per_row * sizeof(int16_t),
                         hipMemcpyHostToDevice, w->stream));
    s.sent_rows = s.rows;
    // every copy out of the pinned staging memory leaves its mark: a scan that is abandoned half-way never reaches
    // finish_plane, and the next start() of the slot must not overwrite (or free) memory a copy is still reading
    if (!s.uploaded) W_HIP(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
    W_HIP(hipEventRecord(s.uploaded, w->stream));
    s.upload_pending = true;
    return JPGPU_OK;
}

// Device -> caller's buffer, then wait.  A copy straight into pageable memory ran at 0.5 GB/s (10 MB: 20 ms — the runtime pins
// the destination page by page); through a pinned buffer of the worker's and a memcpy it is PCIe speed plus ~10 GB/s.
static int worker_download(jpgpu_worker *w, uint8_t *dst, const uint8_t *d_src, size_t n) {
    if (n == 0) {
        W_HIP(hipStreamSynchronize(w->stream));
        return JPGPU_OK;
    }
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();  // (an ordinary pointer is "invalid value" to the query: not an error here)
    if (pinned || n < (64u << 10)) {
        W_HIP(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, w->stream));
        W_HIP(hipStreamSynchronize(w->stream));
        return JPGPU_OK;
    }
    if (w->h_out_cap < n) {
        if (w->h_out) W_HIP(hipHostFree(w->h_out));
        w->h_out = nullptr;
        w->h_out_cap = 0;
        W_HIP(hipHostMalloc((void **)&w->h_out, n + n / 4, hipHostMallocDefault));
        w->h_out_cap = n + n / 4;
    }
    W_HIP(hipMemcpyAsync(w->h_out, d_src, n, hipMemcpyDeviceToHost, w->stream));
    W_HIP(hipStreamSynchronize(w->stream));
    memcpy(dst, w->h_out, n);
    return JPGPU_OK;
}

static int worker_run_idct(jpgpu_worker *w, uint32_t index) {
    auto &s = w->slot[index];
    if (s.rows == s.idct_rows) return JPGPU_OK;
    int rc0 = worker_send_rows(w, index);
    if (rc0) return rc0;
    const size_t blocks_per_row = (size_t)s.c.block_width * s.c.vertical_sampling_factor;
    const size_t row_bytes = blocks_per_row * s.c.dct_scale * s.c.dct_scale;
    PlaneJob job{};
    job.coefs = s.d_coefs + s.idct_rows * blocks_per_row * 64;
    job.plane = s.d_plane + s.idct_rows * row_bytes;
    job.qt = s.d_qt;
    job.block_w = s.c.block_width;
    job.n_blocks = (uint32_t)((s.rows - s.idct_rows) * blocks_per_row);
    job.scale = s.c.dct_scale;
    job.flags = 0;
    W_HIP(launch_idct_plane_one(job, w->stream));
    s.idct_rows = s.rows;
    return JPGPU_OK;
}

// Forget the planes of the image just finished (a recycled worker must not offer them to the next image's compute_image:
// "not all components have data" has to stay what it is); their memory stays with the worker.
void jpgpu::worker_recycle(jpgpu_worker *w) {
    if (!w) return;
    for (auto &f : w->frame) {
        if (f.d_plane) w->spare_planes.emplace_back(f.d_plane, f.cap);
        if (f.d_coefs) w->spare_coefs.emplace_back(f.d_coefs, f.coef_cap);
        f.d_plane = nullptr;
        f.d_coefs = nullptr;
        f.deferred = false;
        f.len = f.cap = f.coef_cap = 0;
    }
    for (auto &s : w->slot) s.started = false;
    w->err.clear();
}

extern "C" {

int jpgpu_worker_create(int device, jpgpu_worker **out) {
    if (!out) return JPGPU_ERR_FORMAT;
    *out = nullptr;
    std::string err;
    int rc = use_device(device, err);
    if (rc) return rc;
    jpgpu_worker *w = new jpgpu_worker();
    w->device = device;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
        delete w;
        return JPGPU_ERR_IO;
    }
    *out = w;
    return JPGPU_OK;
}

void jpgpu_worker_destroy(jpgpu_worker *w) {
    if (!w) return;
    std::string err;
    if (use_device(w->device, err) == JPGPU_OK) {
        hipStreamSynchronize(w->stream);
        for (auto &s : w->slot) {
            if (s.d_qt) hipFree(s.d_qt);
            if (s.d_coefs) hipFree(s.d_coefs);
            if (s.h_pinned) hipHostFree(s.h_pinned);
            if (s.h_qt) hipHostFree(s.h_qt);
            if (s.uploaded) hipEventDestroy(s.uploaded);
            if (s.d_plane) hipFree(s.d_plane);
        }
        for (auto &sp : w->spare_planes) hipFree(sp.first);
        for (auto &f : w->frame) {
            if (f.d_plane) hipFree(f.d_plane);
            if (f.d_coefs) hipFree(f.d_coefs);
        }
        for (auto &sp : w->spare_coefs) hipFree(sp.first);
        if (w->d_frame_qt) hipFree(w->d_frame_qt);
        if (w->d_cls) hipFree(w->d_cls);
        if (w->fplan_valid) jpgpu::fused_free(w->fplan);
        for (auto &t : w->d_tmp)
            if (t) hipFree(t);
        if (w->d_out) hipFree(w->d_out);
        if (w->h_out) hipHostFree(w->h_out);
        hipStreamDestroy(w->stream);
    }
    delete w;
}

const char *jpgpu_worker_last_error(const jpgpu_worker *w) { return w ? w->err.c_str() : ""; }
const char *jpgpu_worker_last_path(const jpgpu_worker *w) { return w ? w->last_path.c_str() : ""; }
int jpgpu_worker_last_class(jpgpu_worker *w) {
    if (!w || !w->fplan_valid || w->last_path == "generic") return -1;
    if (use_device(w->device, w->err) != JPGPU_OK) return -1;
    std::vector<uint8_t> bits;
    if (fused_read_classes(w->fplan, bits, w->err) != JPGPU_OK || bits.empty()) return -1;
    return (bits[0] & 2u) ? 3 : ((bits[0] & 1u) ? 1 : 0);
}

int jpgpu_worker_start(jpgpu_worker *w, uint32_t index, const jpgpu_component *component,
                       const uint16_t quantization_table[64]) {
    if (!w) return JPGPU_ERR_FORMAT;
    if (index >= JPGPU_MAX_COMPONENTS || !component || !quantization_table)
        return set_err(w->err, JPGPU_ERR_FORMAT, "start: bad arguments");
    const jpgpu_component &c = *component;
    if (!(c.dct_scale == 8 || c.dct_scale == 4 || c.dct_scale == 2 || c.dct_scale == 1))
        return set_err(w->err, JPGPU_ERR_INTERNAL, "Unsupported IDCT scale %u/8", c.dct_scale);  // src/idct.rs:237
    if (c.vertical_sampling_factor == 0 || c.horizontal_sampling_factor == 0)
        return set_err(w->err, JPGPU_ERR_FORMAT, "start: zero sampling factor");
    int rc = use_device(w->device, w->err);
    if (rc) return rc;
    auto &s = w->slot[index];
    s.c = c;
    s.rows = s.idct_rows = 0;
    const size_t pbytes = plane_bytes(c);
    const size_t cbytes = (size_t)c.block_width * c.block_height * 64 * sizeof(int16_t);
    s.sent_rows = 0;
    if (s.upload_pending) {  // the staging memory below is about to be overwritten: its last upload must have left it
        W_HIP(hipEventSynchronize(s.uploaded));
        s.upload_pending = false;
    }
    if (!s.d_qt) W_HIP(hipMalloc((void **)&s.d_qt, 128));
    constexpr uint32_t QT_RING = 16;
    if (!s.h_qt) W_HIP(hipHostMalloc((void **)&s.h_qt, 128 * QT_RING, hipHostMallocDefault));
    if (s.qt_unsynced >= QT_RING - 1) {  // sixteen start() calls on this slot without a get_result / compute_image in between
        W_HIP(hipStreamSynchronize(w->stream));
        s.qt_unsynced = 0;
    }
    // the table is caller-owned (Arc<[u16;64]>): keep a copy the upload can read after start() has returned
    uint16_t *hq = s.h_qt + 64 * (s.qt_next++ % QT_RING);
    s.qt_unsynced++;
    memcpy(hq, quantization_table, 128);
    W_HIP(hipMemcpyAsync(s.d_qt, hq, 128, hipMemcpyHostToDevice, w->stream));
    if (s.coef_cap < cbytes) {
        if (s.d_coefs) w->spare_coefs.emplace_back(s.d_coefs, s.coef_cap);
        s.d_coefs = nullptr;
        s.coef_cap = 0;
        for (size_t k = 0; k < w->spare_coefs.size(); k++)
            if (w->spare_coefs[k].second >= cbytes && w->spare_coefs[k].second <= 2 * std::max<size_t>(cbytes, 256)) {
                s.d_coefs = w->spare_coefs[k].first;
                s.coef_cap = w->spare_coefs[k].second;
                w->spare_coefs.erase(w->spare_coefs.begin() + (long)k);
                break;
            }
        if (!s.d_coefs) {
            while (w->spare_coefs.size() > 8) {
                W_HIP(hipFree(w->spare_coefs.front().first));
                w->spare_coefs.erase(w->spare_coefs.begin());
            }
            W_HIP(hipMalloc((void **)&s.d_coefs, std::max<size_t>(cbytes, 256)));
            s.coef_cap = std::max<size_t>(cbytes, 256);
        }
    }
    if (s.pinned_cap < cbytes) {
        if (s.h_pinned) W_HIP(hipHostFree(s.h_pinned));
        s.h_pinned = nullptr;
        s.pinned_cap = 0;
        W_HIP(hipHostMalloc((void **)&s.h_pinned, std::max<size_t>(cbytes, 256), hipHostMallocDefault));
        s.pinned_cap = std::max<size_t>(cbytes, 256);
    }
    if (!s.d_plane || s.plane_cap < pbytes) {
        if (s.d_plane) w->spare_planes.emplace_back(s.d_plane, s.plane_cap);
        s.d_plane = nullptr;
        s.plane_cap = 0;
        for (size_t k = 0; k < w->spare_planes.size(); k++)
            if (w->spare_planes[k].second >= pbytes && w->spare_planes[k].second <= 2 * std::max<size_t>(pbytes, 256)) {
                s.d_plane = w->spare_planes[k].first;
                s.plane_cap = w->spare_planes[k].second;
                w->spare_planes.erase(w->spare_planes.begin() + (long)k);
                break;
            }
        if (!s.d_plane) {
            while (w->spare_planes.size() > 8) {  // sizes nobody asks for any more
                W_HIP(hipFree(w->spare_planes.front().first));
                w->spare_planes.erase(w->spare_planes.begin());
            }
            W_HIP(hipMalloc((void **)&s.d_plane, std::max<size_t>(pbytes, 256)));
            s.plane_cap = std::max<size_t>(pbytes, 256);
        }
    }
    // results[index].resize(elements, 0u8) — rows never appended stay 0 (src/worker/rayon.rs:40-49): the transform writes
    // every appended row completely, finish_plane clears what is left
    s.started = true;
    return JPGPU_OK;
}

int jpgpu_worker_append_rows(jpgpu_worker *w, uint32_t index, const int16_t *coefficients, size_t n_rows) {
    if (!w) return JPGPU_ERR_FORMAT;
    if (index >= JPGPU_MAX_COMPONENTS || !w->slot[index].started)
        return set_err(w->err, JPGPU_ERR_INTERNAL, "append_row on a component that was not started");
    if (n_rows == 0) return JPGPU_OK;
    if (!coefficients) return set_err(w->err, JPGPU_ERR_FORMAT, "append_row: null coefficients");
    int rc = use_device(w->device, w->err);
    if (rc) return rc;
    auto &s = w->slot[index];
    const size_t per_row = (size_t)s.c.block_width * s.c.vertical_sampling_factor * 64;
    const size_t total_rows = s.c.block_height / s.c.vertical_sampling_factor;
    if (s.rows + n_rows > total_rows)
        return set_err(w->err, JPGPU_ERR_INTERNAL, "reference would panic: append_row beyond the plane");
    memcpy(s.h_pinned + s.rows * per_row, coefficients, n_rows * per_row * sizeof(int16_t));
    s.rows += n_rows;
    // upload in pieces of a megabyte or more (the rest at finish_plane): overlaps the host's entropy decoding of the rows to come
    if ((s.rows - s.sent_rows) * per_row * sizeof(int16_t) >= (1u << 20)) return worker_send_rows(w, index);
    return JPGPU_OK;
}

int jpgpu_worker_append_row(jpgpu_worker *w, uint32_t index, const int16_t *coefficients, size_t len) {
    if (!w) return JPGPU_ERR_FORMAT;
    if (index >= JPGPU_MAX_COMPONENTS || !w->slot[index].started)
        return set_err(w->err, JPGPU_ERR_INTERNAL, "append_row on a component that was not started");
    const auto &c = w->slot[index].c;
    // assert_eq!(data.len(), block_count * 64), src/worker/rayon.rs:85
    if (len != (size_t)c.block_width * c.vertical_sampling_factor * 64)
        return set_err(w->err, JPGPU_ERR_INTERNAL, "reference would panic: append_row length %zu != %zu", len,
                       (size_t)c.block_width * c.vertical_sampling_factor * 64);
    return jpgpu_worker_append_rows(w, index, coefficients, 1);
}

int jpgpu_worker_finish_plane(jpgpu_worker *w, uint32_t index, uint32_t plane_slot) {
    if (!w) return JPGPU_ERR_FORMAT;
    if (index >= JPGPU_MAX_COMPONENTS || plane_slot >= JPGPU_MAX_COMPONENTS || !w->slot[index].started)
        return set_err(w->err, JPGPU_ERR_INTERNAL, "get_result on a component that was not started");
    int rc = use_device(w->device, w->err);
    if (rc) return rc;
    auto &s = w->slot[index];
    auto &f = w->frame[plane_slot];
    // what the frame slot held goes back to the spare lists; the slot forgets it right away — a failing call below must not
    // leave a pointer owned twice (a later start() would hand the buffer out while the frame still names it)
    if (f.d_plane) w->spare_planes.emplace_back(f.d_plane, f.cap);
    if (f.d_coefs) w->spare_coefs.emplace_back(f.d_coefs, f.coef_cap);
    f.d_plane = nullptr;
    f.len = f.cap = 0;
    f.d_coefs = nullptr;
    f.coef_cap = 0;
    f.deferred = false;
    static const bool defer_ok = getenv("JPGPU_WORKER_NO_DEFER") == nullptr;  // (A/B and test knob)
    if (defer_ok && s.c.dct_scale == 8 && s.rows * s.c.vertical_sampling_factor == s.c.block_height && s.rows > 0 && s.idct_rows == 0) {
        // a complete plane: keep the coefficients, transform later (or never: compute_image's fused route)
        rc = worker_send_rows(w, index);
        if (rc) return rc;
        if (!w->d_frame_qt) W_HIP(hipMalloc((void **)&w->d_frame_qt, JPGPU_MAX_COMPONENTS * 128));
        W_HIP(hipMemcpyAsync(w->d_frame_qt + plane_slot * 64, s.d_qt, 128, hipMemcpyDeviceToDevice, w->stream));
        f.deferred = true;
        f.d_coefs = s.d_coefs;
        f.coef_cap = s.coef_cap;
        f.c = s.c;
        s.d_coefs = nullptr;
        s.coef_cap = 0;
    } else {
        rc = worker_run_idct(w, index);
        if (rc) return rc;
        if (s.rows) {
            if (!s.uploaded) W_HIP(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
            W_HIP(hipEventRecord(s.uploaded, w->stream));
            s.upload_pending = true;
        }
        const size_t row_bytes = (size_t)s.c.block_width * s.c.vertical_sampling_factor * s.c.dct_scale * s.c.dct_scale, done = s.rows * row_bytes;
        const size_t pbytes = plane_bytes(s.c);
        if (done < pbytes) W_HIP(hipMemsetAsync(s.d_plane + done, 0, pbytes - done, w->stream));
    }
    f.d_plane = s.d_plane;  // mem::take
    f.len = plane_bytes(s.c);
    f.cap = s.plane_cap;
    s.d_plane = nullptr;
    s.plane_cap = 0;
    s.started = false;
    return JPGPU_OK;
}

// A deferred plane after all: transform its coefficients into the plane the frame slot holds.
static int worker_materialise(jpgpu_worker *w, uint32_t plane_slot) {
    auto &f = w->frame[plane_slot];
    if (!f.deferred) return JPGPU_OK;
    PlaneJob job{};
    job.coefs = f.d_coefs;
    job.plane = f.d_plane;
    job.qt = w->d_frame_qt + plane_slot * 64;
    job.block_w = f.c.block_width;
    job.n_blocks = (uint32_t)f.c.block_width * f.c.block_height;
    job.scale = 8;
    job.flags = 0;
    W_HIP(launch_idct_plane_one(job, w->stream));
    f.deferred = false;
    return JPGPU_OK;
}

int jpgpu_worker_get_result(jpgpu_worker *w, uint32_t index, uint8_t *dst, size_t cap, size_t *len) {
    if (!w) return JPGPU_ERR_FORMAT;
    if (index >= JPGPU_MAX_COMPONENTS) return set_err(w->err, JPGPU_ERR_FORMAT, "get_result: bad index");
    if (!w->slot[index].started) {  // mem::take of an empty Vec: empty result
        if (len) *len = 0;
        return JPGPU_OK;
    }
    const size_t n = plane_bytes(w->slot[index].c);
    if (len) *len = n;
    if (!dst || cap < n) return set_err(w->err, JPGPU_ERR_FORMAT, "get_result: destination too small (%zu < %zu)", cap, n);
    int rc = jpgpu_worker_finish_plane(w, index, index);
    if (rc) return rc;
    rc = worker_materialise(w, index);
    if (rc) return rc;
    rc = worker_download(w, dst, w->frame[index].d_plane, n);
    if (rc) return rc;
    for (auto &sl : w->slot) sl.qt_unsynced = 0, sl.upload_pending = false;
    return JPGPU_OK;
}

int jpgpu_compute_image(jpgpu_worker *w, const jpgpu_component *components, uint32_t ncomp,
                        const uint8_t *const *host_planes, uint16_t out_w, uint16_t out_h, int color_transform,
                        uint8_t *dst, size_t cap, size_t *len) {
    if (!w) return JPGPU_ERR_FORMAT;
    jpgpu::TraceRange roctx_range("jpgpu_compute_image");
    if (!components || ncomp == 0 || ncomp > JPGPU_MAX_COMPONENTS)
        return set_err(w->err, JPGPU_ERR_FORMAT, "not all components have data");  // src/decoder.rs:1306-1308
    int rc = use_device(w->device, w->err);
    if (rc) return rc;
    uint8_t *d_planes[JPGPU_MAX_COMPONENTS] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t i = 0; i < ncomp; i++) {
        const size_t n = plane_bytes(components[i]);
        if (host_planes) {
            if (!host_planes[i] || n == 0) return set_err(w->err, JPGPU_ERR_FORMAT, "not all components have data");
            if (w->tmp_cap[i] < n) {
                if (w->d_tmp[i]) W_HIP(hipFree(w->d_tmp[i]));
                w->d_tmp[i] = nullptr;
                w->tmp_cap[i] = 0;
                W_HIP(hipMalloc((void **)&w->d_tmp[i], n));
                w->tmp_cap[i] = n;
            }
            W_HIP(hipMemcpyAsync(w->d_tmp[i], host_planes[i], n, hipMemcpyHostToDevice, w->stream));
            d_planes[i] = w->d_tmp[i];
        } else {
            if (!w->frame[i].d_plane || w->frame[i].len == 0)
                return set_err(w->err, JPGPU_ERR_FORMAT, "not all components have data");
            if (w->frame[i].len != n)
                return set_err(w->err, JPGPU_ERR_INTERNAL, "plane %u has %zu bytes, component geometry needs %zu", i,
                               w->frame[i].len, n);
            d_planes[i] = w->frame[i].d_plane;
        }
    }
    ImageJob job;
    size_t out_len = 0;
    rc = build_image_job(components, ncomp, d_planes, out_w, out_h, color_transform, nullptr, job, out_len, w->err);
    if (rc) return rc;
    if (len) *len = out_len;
    if (out_len == 0) return JPGPU_OK;
    if (!dst || cap < out_len) return set_err(w->err, JPGPU_ERR_FORMAT, "compute_image: destination too small (%zu < %zu)", cap, out_len);
    if (w->out_cap < out_len) {
        if (w->d_out) W_HIP(hipFree(w->d_out));
        w->d_out = nullptr;
        w->out_cap = 0;
        W_HIP(hipMalloc((void **)&w->d_out, out_len));
        w->out_cap = out_len;
    }
    // Every component still as coefficients and the frame of a kind the batch path has a fused kernel for: that kernel, on
    // this one image (coefficients -> pixels in one launch; wrap-exact arithmetic: nobody classified these coefficients).
    bool fused = !host_planes;
    for (uint32_t i = 0; fused && i < ncomp; i++)
        fused = w->frame[i].deferred && memcmp(&w->frame[i].c, &components[i], sizeof(jpgpu_component)) == 0;
    if (fused) {
        jpgpu_image_desc d{};
        d.ncomp = ncomp;
        for (uint32_t i = 0; i < ncomp; i++) d.components[i] = components[i];
        d.out_w = out_w;
        d.out_h = out_h;
        d.color_transform = color_transform;
        fused = fused_kind_key(d) != 0;
        if (fused && !(w->fplan_valid && memcmp(&w->fplan_desc, &d, sizeof(d)) == 0)) {
            if (w->fplan_valid) fused_free(w->fplan);
            w->fplan_valid = false;
            std::string why;
            if (!fused_plan(std::vector<jpgpu_image_desc>{d}, std::vector<uint32_t>{0u}, w->fplan, why)) fused = false;
            else {
                rc = fused_alloc(w->fplan, w->err);
                if (rc) return rc;
                w->fplan_desc = d;
                w->fplan_valid = true;
                memset(w->fplan_bound, 0, sizeof(w->fplan_bound));
            }
        }
        if (fused) {
            // the plan's tables are rewritten (a blocking copy) only when the buffers behind them changed; a worker that decodes
            // frame after frame of one geometry keeps its buffers
            uint64_t key[6] = {0, 0, 0, 0, (uint64_t)(uintptr_t)w->d_out, (uint64_t)(uintptr_t)w->d_frame_qt};
            for (uint32_t i = 0; i < ncomp; i++) key[i] = (uint64_t)(uintptr_t)w->frame[i].d_coefs;
            if (memcmp(key, w->fplan_bound, sizeof(key)) != 0) {
                std::vector<size_t> coef_off(4, 0), out_off(1, 0);
                for (uint32_t i = 0; i < ncomp; i++) coef_off[i] = (size_t)key[i];  // absolute: the base is null
                rc = fused_bind(w->fplan, nullptr, w->d_out, w->d_frame_qt, coef_off, out_off, std::vector<uint8_t>(4, 0), w->err);
                if (rc) return rc;
                memcpy(w->fplan_bound, key, sizeof(key));
            }
            // Nobody classified these coefficients on the way (that would be a pass over them on the host, 0.2-0.5 ms per 1080p
            // frame): the device ranges them where they lie — a scan at HBM speed, microseconds for one frame — and the `_dyn`
            // kernel takes the class from those statistics (range_stats.hpp).
            if (!w->d_cls) {
                W_HIP(hipMalloc((void **)&w->d_cls, (RS_WORDS + 1) * sizeof(uint32_t)));
                // (on the worker's own stream: it is non-blocking, i.e. NOT ordered behind the null stream a plain hipMemset uses,
                // and the finalize kernel below is the first reader — ADVICE r3)
                W_HIP(hipMemsetAsync(w->d_cls + RS_WORDS, 0xff, sizeof(uint32_t), w->stream));
            }
            W_HIP(hipMemsetAsync(w->d_cls, 0, RS_WORDS * sizeof(uint32_t), w->stream));
            for (uint32_t i = 0; i < ncomp; i++)
                W_HIP(launch_range_scan_one(w->frame[i].d_coefs, (uint32_t)components[i].block_width * components[i].block_height,
                                            w->d_frame_qt + i * 64, w->d_cls, w->stream));
            W_HIP(fused_launch(w->fplan, w->stream, w->d_cls, reinterpret_cast<const uint8_t *>(w->d_cls + RS_WORDS)));
            w->last_path = w->fplan.name;
        }
    }
    if (!fused) {
        w->last_path = "generic";
        if (!host_planes)
            for (uint32_t i = 0; i < ncomp; i++) {
                rc = worker_materialise(w, i);
                if (rc) return rc;
            }
        job.out = w->d_out;
        W_HIP(launch_upsample_color_one(job, w->stream));
    }
    rc = worker_download(w, dst, w->d_out, out_len);
    if (rc) return rc;
    for (auto &sl : w->slot) sl.qt_unsynced = 0, sl.upload_pending = false;
    return JPGPU_OK;
}

}  // extern "C"
