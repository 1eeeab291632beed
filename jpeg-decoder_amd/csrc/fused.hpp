// fused.hpp — interface of the fused fast-path kernels (fused.hip) used by batch.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/jpgpu.h"
#include "fused_core.hpp"

namespace jpgpu {

struct FusedPlan {
    std::string name;
    int kind = FUSED_NONE;
    uint32_t n_images = 0, ncomp = 0;
    std::vector<uint32_t> ids;       // batch-level index of each image of this plan (a batch may hold several plans)
    bool strip = false;              // a strip walk (4:2:0, 4:4:0): work items are {image, strip, MCU rows [k0, k1)}
    bool uniform = false;            // every image has the same geometry: 3-D grid, no work table
    uint32_t nt = 256;               // threads per workgroup of the main launch
    size_t lds_bytes = 0;            // dynamic LDS of the main launch (largest tile of the batch)
    std::vector<FusedGeom> geoms;    // per image
    std::vector<FusedWork> work_main;  // one entry per workgroup
    FusedImage *d_images = nullptr;
    FusedGeom *d_geoms = nullptr;
    FusedWork *d_work_main = nullptr;
    uint32_t *d_ids = nullptr;       // `ids` on the device (class_finalize_fused_kernel)
    hipEvent_t launched = nullptr;   // recorded behind every launch of the plan: fused_bind waits for it before it rewrites the
    bool launch_pending = false;     // tables a launch in flight may still be reading
    std::vector<FusedImage> images;
    int arith = 0;  // ARITH_* variant every image of the plan qualifies for (when they all agree)
    // Images of different arithmetic classes in one plan: one launch per class present, each over its own work table
    // (an image that needs the wrap-exact kernels costs only itself).  Built by fused_bind.
    bool by_class = false;
    uint32_t class_images[3] = {0, 0, 0};      // images per ARITH_* class (statistics: jpgpu_batch_class_counts)
    FusedWork *d_work_cls = nullptr;           // work tables of the three classes, back to back
    size_t work_cls_cap = 0;
    uint32_t n_main_cls[3] = {0, 0, 0};
};

// descs: the images of ONE kind (fused_kind_key); ids: their indices in the batch (for fused_bind's offset tables)
bool fused_plan(const std::vector<jpgpu_image_desc> &descs, const std::vector<uint32_t> &ids, FusedPlan &plan, std::string &why);
// 0 if the image cannot take a fused kernel, else a key shared by the images that can share a launch
uint32_t fused_kind_key(const jpgpu_image_desc &d);
int fused_alloc(FusedPlan &plan, std::string &err);
int fused_bind(FusedPlan &plan, uint8_t *d_coef, uint8_t *d_out, uint16_t *d_qt, const std::vector<size_t> &coef_off,
               const std::vector<size_t> &out_off, const std::vector<uint8_t> &sane, std::string &err);
// d_stats (RS_WORDS per batch image, range_stats.hpp) and d_host_cls (per batch image * 4 + component: 0 / 1 / 3 or
// CLS_FROM_DEVICE) given: the classes are taken from the device's own statistics and the `_dyn` kernels run; else the
// classes fused_bind was given (one launch per class present).
hipError_t fused_launch(FusedPlan &plan, hipStream_t stream, const uint32_t *d_stats = nullptr, const uint8_t *d_host_cls = nullptr);
// fused_entries.hpp: the plan's images whose coefficients are the device entropy decoder's entry lists (d_srcs: per BATCH image)
struct EntrySrc;
hipError_t fused_launch_entries(FusedPlan &plan, hipStream_t stream, const EntrySrc *d_srcs);
// the finalize step alone (jpgpu_batch_class_counts)
hipError_t fused_finalize_classes(FusedPlan &plan, hipStream_t stream, const uint32_t *d_stats, const uint8_t *d_host_cls);
// class bits (bit 0 sane, bit 1 tight) of the plan's images as the last launch saw them (blocking read-back; diagnostics)
int fused_read_classes(FusedPlan &plan, std::vector<uint8_t> &bits, std::string &err);
void fused_free(FusedPlan &plan);

}  // namespace jpgpu
