// kernels.hpp — host-visible launch interface of the gfx950 kernels (internal; the public
// boundary is include/jpgpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jpgpu.h"
#include "jobs.hpp"

namespace jpgpu {

struct ExpandJob;
hipError_t launch_expand_compact(const ExpandJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, hipStream_t stream);
hipError_t launch_idct_planes(const PlaneJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t scale,
                              hipStream_t stream);
hipError_t launch_idct_plane_one(const PlaneJob &job, hipStream_t stream);
// device-side classes (range_stats.hpp): d_jobs[j].flags from the statistics of image d_slot[j] / 4 and the host's class table
hipError_t launch_class_finalize_planes(PlaneJob *d_jobs, const uint32_t *d_slot, uint32_t n_jobs, const uint32_t *d_stats, const uint8_t *d_host_cls,
                                        hipStream_t stream);
hipError_t launch_upsample_color(const ImageJob *d_jobs, uint32_t n_jobs, uint32_t max_w, uint32_t max_h,
                                 hipStream_t stream);
hipError_t launch_upsample_color_one(const ImageJob &job, hipStream_t stream);
// reduced-size decodes in one launch (fused_scaled.hpp): n_images geometries / jobs, PlaneJobs indexed by ScaledGeom::first_plane_job;
// scales[s]: some image of the launch decodes at dct_scale s (one launch per scale present)
struct ScaledGeom;
hipError_t launch_scaled_fused(const ScaledGeom *d_geoms, const ImageJob *d_jobs, const PlaneJob *d_planes, uint32_t n_images, uint32_t max_tiles_x,
                               uint32_t max_bands, uint32_t lds_bytes, const bool (&scales)[9], hipStream_t stream);

}  // namespace jpgpu
