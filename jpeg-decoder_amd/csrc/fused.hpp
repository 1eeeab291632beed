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
    uint32_t n_images = 0;
    jpgpu_image_desc desc{};  // the shared geometry
    FusedGeom geom{};
    size_t scratch_per_image = 0;
    uint32_t chunk = 1;  // 4:2:0: images per (chroma pass, main pass) pair
    uint32_t scratch_slots = 1;
    uint32_t n_streams = 1;
    hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t *d_scratch = nullptr;
    FusedImage *d_images = nullptr;
    std::vector<FusedImage> images;
    int arith = 0;  // ARITH_* variant every image of the batch qualifies for
};

bool fused_plan(const std::vector<jpgpu_image_desc> &descs, FusedPlan &plan, std::string &why);
int fused_alloc(FusedPlan &plan, std::string &err);
int fused_bind(FusedPlan &plan, uint8_t *d_coef, uint8_t *d_out, uint16_t *d_qt, const std::vector<size_t> &coef_off,
               const std::vector<size_t> &out_off, const std::vector<uint8_t> &sane, std::string &err);
hipError_t fused_launch(FusedPlan &plan, hipStream_t stream);
void fused_free(FusedPlan &plan);

}  // namespace jpgpu
