// fused.hip — gfx950 wrappers + host plan of the fused fast-path kernels (fused_core.hpp).
//
// Launch geometry: grid = (tiles per MCU row, MCU rows, images), 256 threads.  One workgroup
// covers TX consecutive MCUs of one MCU row, so its pixel stores are 16 (4:2:0) or 8 scanline
// runs of TX*48 / TX*24 contiguous bytes, and its coefficient loads are two (4:2:0 luma) or
// three (4:4:4) contiguous runs of TX*256 / TX*128 bytes.
#include "fused.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fused_plan.hpp"
#include "host_common.hpp"
#include "idct_plane_body.hpp"

namespace jpgpu {

__global__ __launch_bounds__(256) void f420_chroma_kernel(FusedGeom g, const FusedImage *__restrict__ imgs, uint32_t n_blocks) {
    __shared__ v4u lds[256 * 8];
    const FusedImage &img = imgs[blockIdx.z];
    PlaneJob job;
    job.coefs = img.coefs[1 + blockIdx.y];
    job.plane = img.scratch + (size_t)blockIdx.y * g.chroma_plane_bytes;
    job.qt = img.qt[1 + blockIdx.y];
    job.block_w = g.bwc;
    job.n_blocks = n_blocks;
    job.scale = 8;
    job.flags = img.flags;
    idct_planes_body<8>(job, blockIdx.x, lds);
}

template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, NT == 128 ? 4 : 3) void f420_main_kernel(FusedGeom g, const FusedImage *__restrict__ imgs) {
    typedef F420<ARITH, NT> K;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const F420Lds lds = F420Lds::make(lds_raw, g.tx);
    const FusedImage img = imgs[blockIdx.z];
    FusedRegs r;
    K::phase0(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
    __syncthreads();
    K::phase1(g, img, blockIdx.x, threadIdx.x, lds, r);
    __syncthreads();
    K::phase2(g, blockIdx.x, threadIdx.x, lds, r);
    __syncthreads();
    K::phase3(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
}

// 4:2:0 in one launch: grid = (strips, row segments, images); see S420 in fused_core.hpp
template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, 4) void s420_kernel(FusedGeom g, const FusedImage *__restrict__ imgs) {
    typedef S420<ARITH, NT> K;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const S420Lds lds = S420Lds::make(lds_raw, g.tx);
    const FusedImage img = imgs[blockIdx.z];
    const uint32_t strip = blockIdx.x, tid = threadIdx.x;
    const uint32_t k0 = blockIdx.y * g.seg_rows, k1 = min(k0 + g.seg_rows, g.mcu_h);
    S420Regs r;
    K::init(img, tid, lds);
    if (k0 > 0) {  // carry rows of the MCU row above this segment
        K::stage(g, img, strip, k0 - 1, tid, lds);
        __syncthreads();
        K::read_block(g, strip, tid, lds, r);
        K::transform(g, strip, k0 - 1, tid, lds, r, true);
        __syncthreads();
    }
    for (uint32_t k = k0; k < k1; k++) {
        // A fresh, opaque copy of the lane id per phase: otherwise every per-lane address of every phase is hoisted
        // out of the loop and kept live across it (~100 VGPRs), which spills the coefficient block at 4 waves/SIMD.
        uint32_t t0 = tid, t1 = tid, t2 = tid;
        asm volatile("" : "+v"(t0));
        K::stage(g, img, strip, k, t0, lds);
        __syncthreads();
        asm volatile("" : "+v"(t1));
        K::read_block(g, strip, t1, lds, r);
        __syncthreads();  // the tiles alias the staging area
        K::transform(g, strip, k, t1, lds, r, false);
        __syncthreads();
        asm volatile("" : "+v"(t2));
        K::colour(g, img, strip, k, t2, lds);
        __syncthreads();
    }
    if (k1 == g.mcu_h) K::colour(g, img, strip, g.mcu_h, tid, lds);  // the image's last row (far chroma row clamped onto the near one)
}

template <int ARITH>
__global__ __launch_bounds__(256) void f444_kernel(FusedGeom g, const FusedImage *__restrict__ imgs) {
    __shared__ FusedLdsSmall lds;
    const FusedImage img = imgs[blockIdx.z];
    FusedRegs r;
    F444<ARITH>::phase0(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
    __syncthreads();
    const uint32_t wcomp = min((uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), 2u);
    F444<ARITH>::phase1(g, imgs[blockIdx.z].qt[wcomp], blockIdx.x, threadIdx.x, lds, r);
    __syncthreads();
    F444<ARITH>::phase2(g, blockIdx.x, threadIdx.x, lds, r);
    __syncthreads();
    F444<ARITH>::phase3(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
}

template <int ARITH>
__global__ __launch_bounds__(256) void fgray_kernel(FusedGeom g, const FusedImage *__restrict__ imgs) {
    __shared__ FusedLdsSmall lds;
    const FusedImage img = imgs[blockIdx.z];
    FGray<ARITH>::phase0(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
    __syncthreads();
    FGray<ARITH>::phase1(g, img, blockIdx.x, blockIdx.y, threadIdx.x, lds);
}

// ---- host side ------------------------------------------------------------------------------
bool fused_plan(const std::vector<jpgpu_image_desc> &descs, FusedPlan &plan, std::string &why) {
    plan.kind = FUSED_NONE;
    if (descs.empty() || descs.size() > 65535u) return false;
    const jpgpu_image_desc &d0 = descs[0];
    for (const auto &d : descs) {
        if (d.ncomp != d0.ncomp || d.out_w != d0.out_w || d.out_h != d0.out_h || d.color_transform != d0.color_transform) {
            why = "mixed geometry";
            return false;
        }
        for (uint32_t c = 0; c < d.ncomp; c++)
            if (!fused_same_component(d.components[c], d0.components[c])) {
                why = "mixed geometry";
                return false;
            }
    }
    FusedGeom g{};
    const char *name = "", *w = "";
    // JPGPU_F420_TX=32 selects the 128-thread / 32-MCU tiling of the 4:2:0 main pass (tuning knob)
    const char *txenv = getenv("JPGPU_F420_TX");
    // JPGPU_420_STRIP=1 selects the single-launch strip walk (S420) for 4:2:0 instead of chroma pass + main pass;
    // JPGPU_S420_TX / JPGPU_S420_SEG set its strip width and MCU rows per workgroup.  Off by default: on MI355X it
    // moves 15 % fewer bytes but runs 3 % slower (0.88 vs 0.85 ms per 256 x 1080p) — DESIGN.md §5.
    const char *tp = getenv("JPGPU_420_STRIP");
    const bool strip420 = tp && atoi(tp) != 0;
    const char *stx = getenv("JPGPU_S420_TX");
    int kind = fused_geom_from_desc(d0, g, name, w, txenv ? (uint32_t)atoi(txenv) : 64u, strip420, stx ? (uint32_t)atoi(stx) : S420_TX_MAX);
    if (kind == FUSED_NONE) {
        why = w;
        return false;
    }
    if (kind == FUSED_420 && g.strip) {
        const char *sr = getenv("JPGPU_S420_SEG");
        s420_set_segments(g, (uint32_t)descs.size(), sr ? (uint32_t)atoi(sr) : 0u);
    }
    plan.kind = kind;
    plan.name = name;
    plan.geom = g;
    plan.desc = d0;
    plan.n_images = (uint32_t)descs.size();
    plan.scratch_per_image = plan.kind == FUSED_420 && !g.strip ? align_up(2 * (size_t)g.chroma_plane_bytes, 256) : 0;
    // images per chunk of the 4:2:0 path: chroma planes of a chunk <= 64 MiB (JPGPU_CHUNK overrides)
    {
        const char *ce = getenv("JPGPU_CHUNK");
        uint32_t chunk = ce ? (uint32_t)atoi(ce) : 0u;
        if (chunk == 0u) chunk = plan.n_images;  // measured on MI355X: chunking (16..128 images) is slower than one pass over the batch
        plan.chunk = std::max(1u, std::min(chunk, plan.n_images));
        const char *se = getenv("JPGPU_STREAMS");
        plan.n_streams = se ? (uint32_t)std::max(1, std::min(4, atoi(se))) : 1u;
        if (plan.kind != FUSED_420 || g.strip) plan.n_streams = 1;
        // scratch slots are shared modulo `chunk`: chunks in flight on different streams need their own
        if (plan.n_streams > 1 && !ce) plan.chunk = std::max(1u, (plan.n_images + 2u * plan.n_streams - 1u) / (2u * plan.n_streams));
    }
    plan.images.assign(plan.n_images, FusedImage{});
    return true;
}

int fused_alloc(FusedPlan &plan, std::string &err) {
    if (plan.kind == FUSED_NONE) return JPGPU_OK;
    hipError_t e;
    if (plan.scratch_per_image) {
        plan.scratch_slots = std::min(plan.n_images, plan.chunk * plan.n_streams);
        e = hipMalloc((void **)&plan.d_scratch, plan.scratch_per_image * plan.scratch_slots);
        if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMalloc(scratch): %s", hipGetErrorString(e));
    }
    e = hipMalloc((void **)&plan.d_images, sizeof(FusedImage) * plan.n_images);
    if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMalloc(images): %s", hipGetErrorString(e));
    if (plan.n_streams > 1) {
        e = hipEventCreateWithFlags(&plan.ev_fork, hipEventDisableTiming);
        for (uint32_t k = 0; k < plan.n_streams && e == hipSuccess; k++) {
            e = hipStreamCreateWithFlags(&plan.streams[k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&plan.ev_join[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "stream setup: %s", hipGetErrorString(e));
    }
    return JPGPU_OK;
}

int fused_bind(FusedPlan &plan, uint8_t *d_coef, uint8_t *d_out, uint16_t *d_qt, const std::vector<size_t> &coef_off,
               const std::vector<size_t> &out_off, const std::vector<uint8_t> &sane, std::string &err) {
    if (plan.kind == FUSED_NONE) return JPGPU_OK;
    uint32_t common = 3u;  // AND of the per-image flags: one hostile image sends the whole batch down the wrap-exact kernels
    for (uint32_t i = 0; i < plan.n_images; i++) {
        FusedImage &im = plan.images[i];
        uint32_t fl = 3u;
        for (uint32_t c = 0; c < plan.desc.ncomp; c++) {
            im.coefs[c] = reinterpret_cast<const int16_t *>(d_coef + coef_off[i * 4 + c]);
            im.qt[c] = d_qt + ((size_t)i * 4 + c) * 64;
            fl &= sane[i * 4 + c];
        }
        im.out = d_out + out_off[i];
        im.scratch = plan.d_scratch ? plan.d_scratch + (size_t)(i % plan.scratch_slots) * plan.scratch_per_image : nullptr;
        if (!(fl & 1u)) fl = 0u;  // tight implies sane
        im.flags = fl;
        common &= fl;
    }
    plan.arith = (common & 2u) ? ARITH_TIGHT : ((common & 1u) ? ARITH_SANE : ARITH_EXACT);
    if (const char *ae = getenv("JPGPU_ARITH")) plan.arith = std::min(plan.arith, atoi(ae));  // tuning/testing knob: cap the variant
    hipError_t e = hipMemcpy(plan.d_images, plan.images.data(), sizeof(FusedImage) * plan.n_images, hipMemcpyHostToDevice);
    if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMemcpy(images): %s", hipGetErrorString(e));
    return JPGPU_OK;
}

hipError_t fused_launch(FusedPlan &plan, hipStream_t stream) {
    const FusedGeom &g = plan.geom;
    dim3 block(FUSED_NT);
    dim3 grid(g.tiles_x, g.mcu_h, plan.n_images);
    switch (plan.kind) {
    case FUSED_420: {
        if (g.strip) {
            const size_t shm = S420Lds::total_bytes(g.tx);
            dim3 sgrid(g.tiles_x, g.n_seg, plan.n_images);
            if (g.tx <= 20u) {  // 128-thread workgroups
                if (plan.arith == ARITH_TIGHT) s420_kernel<ARITH_TIGHT, 128><<<sgrid, dim3(128), shm, stream>>>(g, plan.d_images);
                else if (plan.arith == ARITH_SANE) s420_kernel<ARITH_SANE, 128><<<sgrid, dim3(128), shm, stream>>>(g, plan.d_images);
                else s420_kernel<ARITH_EXACT, 128><<<sgrid, dim3(128), shm, stream>>>(g, plan.d_images);
            } else {
                if (plan.arith == ARITH_TIGHT) s420_kernel<ARITH_TIGHT, 256><<<sgrid, block, shm, stream>>>(g, plan.d_images);
                else if (plan.arith == ARITH_SANE) s420_kernel<ARITH_SANE, 256><<<sgrid, block, shm, stream>>>(g, plan.d_images);
                else s420_kernel<ARITH_EXACT, 256><<<sgrid, block, shm, stream>>>(g, plan.d_images);
            }
            break;
        }
        // The batch may be walked in chunks of `chunk` images (chroma pass, then main pass) sharing one
        // scratch area (JPGPU_CHUNK).  Default: one chunk — keeping a chunk's chroma planes within the
        // 256 MiB Infinity Cache did not pay on MI355X (profiles/round1: 16/32/64/128-image chunks
        // were 23/9/4/1 % slower than the whole 256-image batch).
        const uint32_t nblk = g.bwc * g.mcu_h;  // chroma blocks per component
        size_t shm = F420Lds::total_bytes(g.tx);
        if (const char *pe = getenv("JPGPU_F420_LDS")) shm = std::max(shm, (size_t)atoi(pe));  // experiment knob: pad the workgroup's LDS claim
        // The chroma pass is HBM-bound and the main pass VALU-bound (profiles/round1), so with
        // JPGPU_STREAMS=2 the chunks alternate between two internal streams: the chroma pass of one
        // chunk can share the machine with the main pass of another.  Forked from / joined to the
        // caller's stream with events, so ordering on that stream is unchanged.
        const uint32_t ns = plan.n_streams;
        if (ns > 1) {
            hipError_t e = hipEventRecord(plan.ev_fork, stream);
            if (e != hipSuccess) return e;
            for (uint32_t k = 0; k < ns; k++)
                if ((e = hipStreamWaitEvent(plan.streams[k], plan.ev_fork, 0)) != hipSuccess) return e;
        }
        uint32_t ci = 0;
        for (uint32_t first = 0; first < plan.n_images; first += plan.chunk, ci++) {
            const uint32_t n = std::min(plan.chunk, plan.n_images - first);
            const FusedImage *imgs = plan.d_images + first;
            hipStream_t st = ns > 1 ? plan.streams[ci % ns] : stream;
            dim3 cgrid((nblk + 255u) / 256u, 2, n), mgrid(g.tiles_x, g.mcu_h, n);
            f420_chroma_kernel<<<cgrid, block, 0, st>>>(g, imgs, nblk);
            const int ar = plan.arith;
            if (g.tx <= 32u) {  // 128-thread workgroups, 32 MCUs per tile
                if (ar == ARITH_TIGHT) f420_main_kernel<ARITH_TIGHT, 128><<<mgrid, dim3(128), shm, st>>>(g, imgs);
                else if (ar == ARITH_SANE) f420_main_kernel<ARITH_SANE, 128><<<mgrid, dim3(128), shm, st>>>(g, imgs);
                else f420_main_kernel<ARITH_EXACT, 128><<<mgrid, dim3(128), shm, st>>>(g, imgs);
            } else {
                if (ar == ARITH_TIGHT) f420_main_kernel<ARITH_TIGHT, 256><<<mgrid, block, shm, st>>>(g, imgs);
                else if (ar == ARITH_SANE) f420_main_kernel<ARITH_SANE, 256><<<mgrid, block, shm, st>>>(g, imgs);
                else f420_main_kernel<ARITH_EXACT, 256><<<mgrid, block, shm, st>>>(g, imgs);
            }
        }
        if (ns > 1)
            for (uint32_t k = 0; k < ns; k++) {
                hipError_t e = hipEventRecord(plan.ev_join[k], plan.streams[k]);
                if (e == hipSuccess) e = hipStreamWaitEvent(stream, plan.ev_join[k], 0);
                if (e != hipSuccess) return e;
            }
        break;
    }
    case FUSED_444:
        if (plan.arith == ARITH_TIGHT) f444_kernel<ARITH_TIGHT><<<grid, block, 0, stream>>>(g, plan.d_images);
        else if (plan.arith == ARITH_SANE) f444_kernel<ARITH_SANE><<<grid, block, 0, stream>>>(g, plan.d_images);
        else f444_kernel<ARITH_EXACT><<<grid, block, 0, stream>>>(g, plan.d_images);
        break;
    case FUSED_GRAY:
        if (plan.arith == ARITH_TIGHT) fgray_kernel<ARITH_TIGHT><<<grid, block, 0, stream>>>(g, plan.d_images);
        else if (plan.arith == ARITH_SANE) fgray_kernel<ARITH_SANE><<<grid, block, 0, stream>>>(g, plan.d_images);
        else fgray_kernel<ARITH_EXACT><<<grid, block, 0, stream>>>(g, plan.d_images);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

void fused_free(FusedPlan &plan) {
    if (plan.d_scratch) (void)hipFree(plan.d_scratch);
    if (plan.d_images) (void)hipFree(plan.d_images);
    for (uint32_t k = 0; k < 4; k++) {
        if (plan.streams[k]) (void)hipStreamDestroy(plan.streams[k]);
        if (plan.ev_join[k]) (void)hipEventDestroy(plan.ev_join[k]);
        plan.streams[k] = nullptr;
        plan.ev_join[k] = nullptr;
    }
    if (plan.ev_fork) (void)hipEventDestroy(plan.ev_fork);
    plan.ev_fork = nullptr;
    plan.d_scratch = nullptr;
    plan.d_images = nullptr;
}

}  // namespace jpgpu
