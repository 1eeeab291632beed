// idct_plane_body.hpp — workgroup body of the plane IDCT (shared by kernels.hip and fused.hip).
#pragma once
#include "kernels.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

// One lane per 8x8 block, 256 blocks per workgroup.  Coefficients are fetched with fully
// coalesced 16-B loads (lane j of the workgroup reads chunk j of the workgroup's contiguous
// 32 KiB) and staged in LDS so that each lane can then pull its own 128-B block with 8
// ds_read_b128.  LDS slot of (block b, row k): b*8 + (k ^ ((b >> 1) & 7)), conflict-free for both
// the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups (MI355X_MICROARCH.md §LDS).
template <int SCALE>
__device__ __forceinline__ void idct_planes_body(const PlaneJob &job, uint32_t wg, v4u *lds) {
    const uint32_t tid = threadIdx.x;
    const uint32_t first = wg * 256u;
    if (first >= job.n_blocks) return;  // whole workgroup out of range (uniform)
    const uint32_t nb = min(256u, job.n_blocks - first);
    const JP_GLOBAL v4u *src = (const JP_GLOBAL v4u *)(job.coefs + (size_t)first * 64);
    uint32_t cw[32];
    if constexpr (SCALE == 8) {
    // All eight 16-B loads are issued before the first LDS store (one exposed memory latency, not
    // eight).  Named scalars + clamped indices on purpose: a predicated `v4u v[8]` array is
    // kept in scratch memory by hipcc (ROCm 7.2) instead of VGPRs.
    const uint32_t lastc = nb * 8u - 1u;
#define JP_LD(i) const v4u v##i = src[min((i) * 256u + tid, lastc)];
#define JP_ST(i)                                                      \
    {                                                                 \
        const uint32_t j = (i) * 256u + tid;                          \
        if (j <= lastc) lds[(j >> 3) * 8u + ((j & 7u) ^ ((j >> 4) & 7u))] = v##i; \
    }
    JP_LD(0) JP_LD(1) JP_LD(2) JP_LD(3) JP_LD(4) JP_LD(5) JP_LD(6) JP_LD(7)
    JP_ST(0) JP_ST(1) JP_ST(2) JP_ST(3) JP_ST(4) JP_ST(5) JP_ST(6) JP_ST(7)
#undef JP_LD
#undef JP_ST
    __syncthreads();
    if (tid >= nb) return;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v4u v = lds[tid * 8u + ((uint32_t)k ^ ((tid >> 1) & 7u))];
        cw[k * 4 + 0] = v.x;
        cw[k * 4 + 1] = v.y;
        cw[k * 4 + 2] = v.z;
        cw[k * 4 + 3] = v.w;
    }
    } else {
        // Reduced IDCTs use the top-left SCALE x SCALE coefficients only (src/idct.rs:456-565): rows 0 .. SCALE-1 of a block,
        // i.e. its first SCALE 16-byte pieces — half, a quarter, an eighth of the arena's bytes (round 3: the kernel fetched
        // all eight).  Piece j of the workgroup = row j % R of block j / R, R = SCALE (scale 1: the one piece with the DC).
        constexpr uint32_t R = SCALE == 4 ? 4u : (SCALE == 2 ? 2u : 1u);
        const uint32_t lastp = nb * R - 1u;
        v4u v[R];
#pragma unroll
        for (uint32_t i = 0; i < R; i++) {
            const uint32_t j = min(i * 256u + tid, lastp);
            v[i] = src[(j / R) * 8u + (j % R)];
        }
#pragma unroll
        for (uint32_t i = 0; i < R; i++) {
            const uint32_t j = i * 256u + tid;
            if (j <= lastp) lds[(j / R) * 8u + ((j % R) ^ (((j / R) >> 1) & 7u))] = v[i];
        }
        __syncthreads();
        if (tid >= nb) return;
#pragma unroll
        for (int k = 0; k < 32; k++) cw[k] = 0u;
#pragma unroll
        for (uint32_t k = 0; k < R; k++) {
            const v4u w = lds[tid * 8u + (k ^ ((tid >> 1) & 7u))];
            cw[k * 4 + 0] = w.x;
            cw[k * 4 + 1] = w.y;
            cw[k * 4 + 2] = w.z;
            cw[k * 4 + 3] = w.w;
        }
    }
    const uint32_t b = first + tid;
    const uint32_t bx = b % job.block_w, by = b / job.block_w;
    const size_t stride = (size_t)job.block_w * SCALE;
    JP_GLOBAL uint8_t *dst = (JP_GLOBAL uint8_t *)job.plane + (size_t)by * SCALE * stride + (size_t)bx * SCALE;
    if constexpr (SCALE == 8) {
        uint32_t out[16];
        if (job.flags & 2u) idct8x8<ARITH_TIGHT>(cw, as_qtab(job.qt), out);  // uniform: one job per workgroup
        else if (job.flags & 1u) idct8x8<ARITH_SANE>(cw, as_qtab(job.qt), out);
        else idct8x8<ARITH_EXACT>(cw, as_qtab(job.qt), out);
#pragma unroll
        for (int r = 0; r < 8; r++)
            *reinterpret_cast<JP_GLOBAL v2u *>(dst + (size_t)r * stride) = v2u{out[2 * r], out[2 * r + 1]};
    } else if constexpr (SCALE == 4) {
        uint32_t out[4];
        idct4x4_exact(cw, as_qtab(job.qt), out);
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<JP_GLOBAL uint32_t *>(dst + (size_t)r * stride) = out[r];
    } else if constexpr (SCALE == 2) {
        uint32_t o = idct2x2_exact(cw, as_qtab(job.qt));
        *reinterpret_cast<JP_GLOBAL uint16_t *>(dst) = (uint16_t)(o & 0xffffu);
        *reinterpret_cast<JP_GLOBAL uint16_t *>(dst + stride) = (uint16_t)(o >> 16);
    } else {
        dst[0] = (uint8_t)idct1x1_exact(cw[0], as_qtab(job.qt));
    }
}


}  // namespace jpgpu
