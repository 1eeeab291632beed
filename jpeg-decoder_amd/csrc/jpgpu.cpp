// jpgpu.cpp — C ABI (include/jpgpu.h) over the gfx950 kernels: Worker + compute_image
// (the crate's drop-in boundary) and the batch driver.  Host-side logic mirrors
// src/worker/rayon.rs / src/worker/mod.rs / src/decoder.rs:1300-1389 of the reference; the
// arithmetic lives in the kernels.  There is NO CPU fallback: without a usable HIP device
// every compute entry point fails with JPGPU_ERR_NO_DEVICE.
#include "../../include/jpgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "kernels.hpp"

using namespace jpgpu;

extern "C" {

const char *jpgpu_version(void) { return "jpgpu 0.1 (gfx950)"; }

int jpgpu_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? JPGPU_OK : JPGPU_ERR_NO_DEVICE;
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
        size_t rows = 0;      // MCU rows appended since start()
        size_t idct_rows = 0; // MCU rows already transformed
    } slot[JPGPU_MAX_COMPONENTS];
    struct Frame {
        uint8_t *d_plane = nullptr;
        size_t len = 0;
    } frame[JPGPU_MAX_COMPONENTS];
    uint8_t *d_out = nullptr;
    size_t out_cap = 0;
    uint8_t *d_tmp[JPGPU_MAX_COMPONENTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t tmp_cap[JPGPU_MAX_COMPONENTS] = {0, 0, 0, 0};
};

#define W_HIP(call)                                                                                     \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) return set_err(w->err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

static int worker_run_idct(jpgpu_worker *w, uint32_t index) {
    auto &s = w->slot[index];
    if (s.rows == s.idct_rows) return JPGPU_OK;
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
            if (s.d_plane) hipFree(s.d_plane);
        }
        for (auto &f : w->frame)
            if (f.d_plane) hipFree(f.d_plane);
        for (auto &t : w->d_tmp)
            if (t) hipFree(t);
        if (w->d_out) hipFree(w->d_out);
        hipStreamDestroy(w->stream);
    }
    delete w;
}

const char *jpgpu_worker_last_error(const jpgpu_worker *w) { return w ? w->err.c_str() : ""; }

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
    if (!s.d_qt) W_HIP(hipMalloc((void **)&s.d_qt, 128));
    W_HIP(hipMemcpyAsync(s.d_qt, quantization_table, 128, hipMemcpyHostToDevice, w->stream));
    // the table is caller-owned (Arc<[u16;64]>): finish reading it before returning
    W_HIP(hipStreamSynchronize(w->stream));
    if (s.coef_cap < cbytes) {
        if (s.d_coefs) W_HIP(hipFree(s.d_coefs));
        s.d_coefs = nullptr;
        s.coef_cap = 0;
        W_HIP(hipMalloc((void **)&s.d_coefs, std::max<size_t>(cbytes, 256)));
        s.coef_cap = std::max<size_t>(cbytes, 256);
    }
    if (s.pinned_cap < cbytes) {
        if (s.h_pinned) W_HIP(hipHostFree(s.h_pinned));
        s.h_pinned = nullptr;
        s.pinned_cap = 0;
        W_HIP(hipHostMalloc((void **)&s.h_pinned, std::max<size_t>(cbytes, 256), hipHostMallocDefault));
        s.pinned_cap = std::max<size_t>(cbytes, 256);
    }
    if (!s.d_plane || s.plane_cap < pbytes) {
        if (s.d_plane) W_HIP(hipFree(s.d_plane));
        s.d_plane = nullptr;
        s.plane_cap = 0;
        W_HIP(hipMalloc((void **)&s.d_plane, std::max<size_t>(pbytes, 256)));
        s.plane_cap = std::max<size_t>(pbytes, 256);
    }
    // results[index].resize(elements, 0u8) — rows never appended stay 0 (src/worker/rayon.rs:40-49)
    W_HIP(hipMemsetAsync(s.d_plane, 0, std::max<size_t>(pbytes, 1), w->stream));
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
    int16_t *stage = s.h_pinned + s.rows * per_row;
    memcpy(stage, coefficients, n_rows * per_row * sizeof(int16_t));
    W_HIP(hipMemcpyAsync(s.d_coefs + s.rows * per_row, stage, n_rows * per_row * sizeof(int16_t), hipMemcpyHostToDevice,
                         w->stream));
    s.rows += n_rows;
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
    rc = worker_run_idct(w, index);
    if (rc) return rc;
    auto &s = w->slot[index];
    auto &f = w->frame[plane_slot];
    if (f.d_plane) W_HIP(hipFree(f.d_plane));
    f.d_plane = s.d_plane;  // mem::take
    f.len = plane_bytes(s.c);
    s.d_plane = nullptr;
    s.plane_cap = 0;
    s.started = false;
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
    if (n) W_HIP(hipMemcpyAsync(dst, w->frame[index].d_plane, n, hipMemcpyDeviceToHost, w->stream));
    W_HIP(hipStreamSynchronize(w->stream));
    return JPGPU_OK;
}

int jpgpu_compute_image(jpgpu_worker *w, const jpgpu_component *components, uint32_t ncomp,
                        const uint8_t *const *host_planes, uint16_t out_w, uint16_t out_h, int color_transform,
                        uint8_t *dst, size_t cap, size_t *len) {
    if (!w) return JPGPU_ERR_FORMAT;
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
    job.out = w->d_out;
    W_HIP(launch_upsample_color_one(job, w->stream));
    W_HIP(hipMemcpyAsync(dst, w->d_out, out_len, hipMemcpyDeviceToHost, w->stream));
    W_HIP(hipStreamSynchronize(w->stream));
    return JPGPU_OK;
}

}  // extern "C"
