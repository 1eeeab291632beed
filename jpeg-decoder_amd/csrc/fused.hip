// fused.hip — placeholder until the fused kernels land (next commit): no plan resolves.
#include "fused.hpp"
#include "host_common.hpp"
namespace jpgpu {
bool fused_plan(const std::vector<jpgpu_image_desc> &, FusedPlan &, std::string &why) { why = "not built"; return false; }
int fused_alloc(FusedPlan &, std::string &) { return JPGPU_OK; }
int fused_bind(FusedPlan &, uint8_t *, uint8_t *, uint16_t *, const std::vector<size_t> &, const std::vector<size_t> &,
               const std::vector<uint8_t> &, std::string &) { return JPGPU_OK; }
hipError_t fused_launch(FusedPlan &, hipStream_t) { return hipSuccess; }
void fused_free(FusedPlan &) {}
}  // namespace jpgpu
