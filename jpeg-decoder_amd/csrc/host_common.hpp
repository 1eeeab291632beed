// host_common.hpp — helpers shared by the C-ABI translation units (internal).
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/jpgpu.h"
#include "jobs.hpp"

namespace jpgpu {

int set_err(std::string &dst, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int use_device(int device, std::string &err);
size_t plane_bytes(const jpgpu_component &c);
int choose_color_fn(uint32_t ncomp, int color_transform, uint32_t &fn, std::string &err);
int build_image_job(const jpgpu_component *comps, uint32_t ncomp, uint8_t *const *d_planes, uint16_t out_w,
                    uint16_t out_h, int color_transform, uint8_t *d_out, ImageJob &job, size_t &out_len,
                    std::string &err);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// jpgpu_batch_upload_compact for buffers written by CompactWriter inside this library (no consistency pass)
int batch_upload_compact(jpgpu_batch *b, uint32_t image, uint32_t comp, const void *compact, size_t bytes, int range_class,
                         void *hip_stream, bool trusted);

}  // namespace jpgpu
