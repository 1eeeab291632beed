// fused.hpp — interface of the fused fast-path kernels (fused.hip) used by batch.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/jpgpu.h"

namespace jpgpu {

enum FusedKind : int { FUSED_NONE = 0, FUSED_420 = 1, FUSED_444 = 2, FUSED_GRAY = 3 };

// Per-image pointers of a same-geometry batch.
struct FusedImage {
    const int16_t *coefs[4];
    const uint16_t *qt[4];
    uint8_t *out;
    uint8_t *scratch;   // 4:2:0: Cb plane followed by Cr plane
    uint32_t flags;     // bit0: all components "sane" -> 24-bit multiply path
    uint32_t _pad;
};

struct FusedPlan {
    std::string name;
    int kind = FUSED_NONE;
    uint32_t n_images = 0;
    jpgpu_image_desc desc{};  // the shared geometry
    uint32_t mcu_w = 0, mcu_h = 0;
    size_t scratch_per_image = 0;
    uint8_t *d_scratch = nullptr;
    FusedImage *d_images = nullptr;
    std::vector<FusedImage> images;
    bool all_sane = false;
};

bool fused_plan(const std::vector<jpgpu_image_desc> &descs, FusedPlan &plan, std::string &why);
int fused_alloc(FusedPlan &plan, std::string &err);
int fused_bind(FusedPlan &plan, uint8_t *d_coef, uint8_t *d_out, uint16_t *d_qt, const std::vector<size_t> &coef_off,
               const std::vector<size_t> &out_off, const std::vector<uint8_t> &sane, std::string &err);
hipError_t fused_launch(FusedPlan &plan, hipStream_t stream);
void fused_free(FusedPlan &plan);

}  // namespace jpgpu
