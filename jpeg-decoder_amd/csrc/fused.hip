// fused.hip — gfx950 wrappers + host plan of the fused fast-path kernels (fused_core.hpp).
//
// A batch takes the fused path when every image is of ONE fusable kind (4:2:0 YCbCr, 4:4:4 YCbCr/RGB, or gray) at
// dct_scale 8; the images may differ in size.  The launch is a flat list of workgroups: a work table (built once
// per batch on the host) tells workgroup i which image and which tile it owns, and a per-image geometry table gives
// the tiling, so mixed-size batches run the same kernels as uniform ones (two scalar loads per workgroup).
// One workgroup covers TX consecutive MCUs of one MCU row, so its pixel stores are 16 (4:2:0) or 8 scanline runs of
// TX*48 / TX*24 contiguous bytes, and its coefficient loads are two (4:2:0 luma) or three (4:4:4) contiguous runs.
// Order of the table = x fastest, then MCU row, then image: neighbouring workgroups touch neighbouring memory.
#include "fused.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "fused_plan.hpp"
#include "host_common.hpp"
#include "range_stats.hpp"

#include "fused_entries.hpp"

namespace jpgpu {

// Which image / tile a workgroup owns: from the work table (mixed-size batches, 1-D grid) or, when the batch is uniform
// and no table is passed, straight from the 3-D grid (x, y, image) — that saves the dependent scalar load at the
// start of every workgroup (measured: 2 % on the 4:2:0 bench).
__device__ __forceinline__ FusedWork locate(const FusedWork *__restrict__ work) {
    if (work) return work[blockIdx.x];
    // (renumbering the workgroups so that each XCD walks one contiguous eighth of the (tile, MCU row, image) space — neighbours in
    // x and y sharing an L2 — was measured in round 1 and changed nothing: the kernels read and write every byte once)
    return FusedWork{blockIdx.z, blockIdx.x, blockIdx.y, 0u};
}

// Every pixel kernel exists in four forms: one per arithmetic class (the host knows the classes of a launch's images and has
// split the work tables accordingly: fused_bind) and `_dyn`, which reads the class of its workgroup's image from the image
// table ON THE DEVICE and branches to the body of that class (workgroup-uniform) — for batches whose classes come from
// statistics the device gathered itself (range_stats.hpp, class_finalize_fused_kernel): no host in between.  The three
// bodies share registers and LDS (allocation = the largest, which the wrap-exact body already set under the same launch bounds).
#define JP_DYN_DISPATCH(FLAGS, CALL)                \
    do {                                            \
        const uint32_t _fl = (FLAGS);               \
        if (_fl & 2u) { CALL(ARITH_TIGHT); }        \
        else if (_fl & 1u) { CALL(ARITH_SANE); }    \
        else { CALL(ARITH_EXACT); }                 \
    } while (0)
__device__ __forceinline__ uint32_t image_flags(const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)imgs[locate(work).image].flags);
}

// Strip walks (4:2:0: S420, 4:4:0: S440 in fused_core.hpp).  One work item = MCU rows [k0, k1) of one strip of one image, one
// item per workgroup: work[blockIdx.x], or without a table (uniform batch) strip blockIdx.x, segment blockIdx.y of image blockIdx.z.
// (Round 3 also had "balanced shares" — the launch's steps dealt to exactly as many workgroups as the device holds, several items
// per workgroup — which measured 5-8 % slower: workgroups that start together stay in step, and it is the spread of phases across
// the workgroups of a CU that overlaps one's memory phases with another's arithmetic; profiles/round3/03_balanced_walk.txt.)
__device__ __forceinline__ FusedWork walk_item_at(const FusedGeom *__restrict__ geoms, const FusedWork *__restrict__ work, uint32_t it) {
    if (work) return work[it];
    FusedWork w = locate(nullptr);  // (strip, segment, image) from the grid
    const uint32_t seg = geoms[w.image].seg_rows, k0 = w.b * seg;
    w.b = k0;
    w.c = min(k0 + seg, geoms[w.image].mcu_h);
    return w;
}

template <class K>
__device__ __forceinline__ void walk_item(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork w, uint8_t *lds_raw) {
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const typename K::Lds lds = K::Lds::make(lds_raw, g.tx);
    const uint32_t strip = w.a, tid = threadIdx.x;
    const uint32_t k0 = w.b, k1 = w.c;
    S420Regs r;
    // Wave priorities (round 4): the colour phase — upsampling, conversion, the store burst — runs BELOW everything else of the walk.
    // A CU holds four workgroups in different phases; with equal priorities the waves that convert and store (long runs of
    // independent vector instructions) crowd out the ones that issue the next row's loads or transform, whose results other phases
    // wait for: 0.6655 -> 0.6495 ms per 256 x 1080p (same box, interleaved; any pair of levels with colour the lower one measures the
    // same; raising the colour phase instead: 0.662-0.681).  Box to box the gain is 0.2-2.4 %, never a loss; the ROW kernels (one pass per
    // workgroup, nothing to prefetch for) lose 2-6 % with the same split and keep equal priorities (profiles/round4/10_wave_priorities.txt).
    __builtin_amdgcn_s_setprio(1);
    K::init(img, tid, lds);
    if (k0 > 0 || k1 < g.mcu_h) {  // seam rows of the segments above / below
        K::seam_stage(g, img, strip, k0, k1, tid, lds);
        __syncthreads();
        K::seam_transform(g, strip, k0, k1, tid, lds);
        __syncthreads();
    }
    {
        typename K::Pre pre;
        K::stage_load(g, img, strip, k0, tid, pre);
        K::stage_store(g, strip, tid, lds, pre);
    }
    __syncthreads();
    for (uint32_t k = k0; k < k1; k++) {
        typename K::Pre pre;  // (declared per iteration: not live around the loop)
        // A fresh, opaque copy of the lane id per phase: otherwise every per-lane address of every phase is hoisted
        // out of the loop and kept live across it (~100 VGPRs), which spills the coefficient block at 4 waves/SIMD.
        uint32_t t0 = tid, t1 = tid, t2 = tid;
        asm volatile("" : "+v"(t1));
        K::read_block(g, strip, t1, lds, r);
        __syncthreads();  // the tiles alias the staging area
        K::transform(g, strip, t1, lds, r);
        __syncthreads();
        const bool more = k + 1u < k1;
        asm volatile("" : "+v"(t2));
        __builtin_amdgcn_s_setprio(0);
        K::colour(g, img, strip, k, 16u * k0, false, t2, lds);
        __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        if (more) {
            asm volatile("" : "+v"(t0));
            K::stage_load(g, img, strip, k + 1u, t0, pre);
            K::stage_store(g, strip, t0, lds, pre);
            __syncthreads();
        }
    }
    if (16u * k1 - 1u < g.out_h) {  // the segment's last output row: its far chroma row is the seam row below (or itself at the image's end)
        K::closing_tiles(tid, lds);
        __syncthreads();
        K::colour(g, img, strip, k1, 16u * k0, true, tid, lds);
    }
}

// Workgroups per CU the walks are compiled for: four — 128 registers per lane — except the wrap-exact bodies (class 0: hostile
// coefficients, exact 32-bit arithmetic), which need 140 and get three (170 registers).  At four they spilled 58 registers into a
// scratch segment: 1.19 ms per 256 x 1080p = 0.335 of the roofline; at three (or two: the same) 0.815 ms = 0.49 (round 5).
constexpr uint32_t walk_wgs(bool exact) { return exact ? 3u : 4u; }
template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, walk_wgs(ARITH == ARITH_EXACT)) void s420_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                     const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    walk_item<S420<ARITH, NT>>(geoms, imgs, walk_item_at(geoms, work, blockIdx.x), lds_raw);
}
// Classes from the device (one item per workgroup), as TWO launches: the first runs the images whose coefficients are in range
// (tight / sane bodies), the second the others (wrap-exact body); a workgroup whose image belongs to the other launch leaves at
// once.  One kernel with all three bodies inherits the scratch memory of the wrap-exact one (it spills at 128 VGPRs), and a
// kernel with scratch is dispatched more slowly even where no wave touches it: measured 0.737 ms against 0.676 for the tight
// kernel on the same box (profiles/round3/10_kernel_trace_stats_and_pmc.json).
template <uint32_t NT, bool EXACT_PASS>
__global__ __launch_bounds__(NT, walk_wgs(EXACT_PASS)) void s420_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                         const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const FusedWork w = walk_item_at(geoms, work, blockIdx.x);
    const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)imgs[w.image].flags);
    if (fl & 8u) return;  // the image's pixels come from the entry lists (s420_entries_kernel, fused_entries.hpp)
    if constexpr (EXACT_PASS) {
        if (!(fl & 1u)) walk_item<S420<ARITH_EXACT, NT>>(geoms, imgs, w, lds_raw);
    } else {
        if (fl & 2u) walk_item<S420<ARITH_TIGHT, NT>>(geoms, imgs, w, lds_raw);
        else if (fl & 1u) walk_item<S420<ARITH_SANE, NT>>(geoms, imgs, w, lds_raw);
    }
}

// The 4:2:0 walk fed by the device entropy decoder's entry lists (fused_entries.hpp): the staging step of walk_item replaced by
// "clear the staging area, scatter the run's entries into it"; everything else is S420<ARITH_SANE>'s.
__global__ __launch_bounds__(256, 4) void s420_entries_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                              const FusedWork *__restrict__ work, const uint32_t *__restrict__ ids,
                                                              const EntrySrc *__restrict__ srcs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    typedef S420E::K K;
    const FusedWork w = walk_item_at(geoms, work, blockIdx.x);
    const EntrySrc src = srcs[ids[w.image]];
    if (!src.job) return;  // (a dense image of a mixed launch group)
    const HuffSyncJob *__restrict__ job = src.job;
    if (__hip_atomic_load(job->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // refused by the decoder: the host decodes the image
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const S420Lds lds = S420Lds::make(lds_raw, g.tx);
    const S420ELds el = S420ELds::make(lds_raw + S420Lds::total_bytes(g.tx), g.tx);
    const uint32_t strip = w.a, tid = threadIdx.x, k0 = w.b, k1 = w.c, te = K::txe(g, strip);
    S420Regs r;
    // {chunk, entry} of MCU row k of this strip — written by huff_strip_index_kernel before this launch: constant here, read with scalar loads
    // (a scalar load waits where its result is used, so the next row's pair, asked for a step ahead, costs no register a lane owns)
    const JP_CONST uint32_t *tab = (const JP_CONST uint32_t *)src.tab + 2u * strip;
    auto at = [&](uint32_t k, uint32_t i) { return tab[2u * k * g.tiles_x + i]; };
    __builtin_amdgcn_s_setprio(1);
    K::init(img, tid, lds);
    S420E::init(img, te, tid, el);
    if (k0 > 0 || k1 < g.mcu_h) {  // seam rows of the segments above / below: the chroma blocks of MCU rows k0 - 1 and k1
        S420E::clear_stage(lds, 4u * (te + 2u), tid);
        __syncthreads();
        if (k0 > 0) S420E::scatter_row<false>(job, at(k0 - 1u, 0u), at(k0 - 1u, 1u), g, strip, k0 - 1u, tid, lds, el.blk[1], el, S420E::Meta{});
        if (k1 < g.mcu_h) S420E::scatter_row<false>(job, at(k1, 0u), at(k1, 1u), g, strip, k1, tid, lds, el.blk[2], el, S420E::Meta{});
        __syncthreads();
        K::seam_transform(g, strip, k0, k1, tid, lds);
        __syncthreads();
    }
    S420E::clear_stage(lds, 6u * te + 4u, tid);
    __syncthreads();
    S420E::scatter_row<false>(job, at(k0, 0u), at(k0, 1u), g, strip, k0, tid, lds, el.blk[0], el, S420E::Meta{});
    uint32_t nc0 = k0 + 1u < k1 ? at(k0 + 1u, 0u) : 0u, ne0 = k0 + 1u < k1 ? at(k0 + 1u, 1u) : 0u;  // (the next row's place in the lists: asked for a step ahead)
    __syncthreads();
    for (uint32_t k = k0; k < k1; k++) {
        uint32_t t0 = tid, t1 = tid, t2 = tid;  // (opaque copies of the lane id per phase: walk_item)
        asm volatile("" : "+v"(t1));
        K::read_block(g, strip, t1, lds, r);
        __syncthreads();  // the tiles alias the staging area
        K::transform(g, strip, t1, lds, r);
        __syncthreads();
        const bool more = k + 1u < k1;
        asm volatile("" : "+v"(t2));
        S420E::Meta pre{};
        if (more) pre = S420E::request_meta(job, nc0, t2);  // (the next row's chunks: on their way during the colour phase)
        __builtin_amdgcn_s_setprio(0);
        K::colour(g, img, strip, k, 16u * k0, false, t2, lds);
        __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        if (more) {
            asm volatile("" : "+v"(t0));
            S420E::clear_stage(lds, 6u * te + 4u, t0);
            __syncthreads();
            S420E::scatter_row<true>(job, nc0, ne0, g, strip, k + 1u, t0, lds, el.blk[0], el, pre);
            if (k + 2u < k1) nc0 = at(k + 2u, 0u), ne0 = at(k + 2u, 1u);
            __syncthreads();
        }
    }
    if (16u * k1 - 1u < g.out_h) {
        K::closing_tiles(tid, lds);
        __syncthreads();
        K::colour(g, img, strip, k1, 16u * k0, true, tid, lds);
    }
    // the sane body was run on trust: an image with a coefficient outside its range goes back to the host
    __syncthreads();
    uint32_t t3 = tid;
    asm volatile("" : "+v"(t3));  // (or the address of el.rg[tid], computed in init, is kept — spilled — across the whole walk)
    if (t3 < 4u && el.rg[t3] >= (1u << 15)) atomicOr(job->status, 1u | ENTRY_ST_RANGE);
}

template <int ARITH>
__global__ __launch_bounds__(256, walk_wgs(ARITH == ARITH_EXACT)) void s440_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                      const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    walk_item<S440<ARITH>>(geoms, imgs, walk_item_at(geoms, work, blockIdx.x), lds_raw);
}
template <bool EXACT_PASS>
__global__ __launch_bounds__(256, walk_wgs(EXACT_PASS)) void s440_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                          const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const FusedWork w = walk_item_at(geoms, work, blockIdx.x);
    const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)imgs[w.image].flags);
    if constexpr (EXACT_PASS) {
        if (!(fl & 1u)) walk_item<S440<ARITH_EXACT>>(geoms, imgs, w, lds_raw);
    } else {
        if (fl & 2u) walk_item<S440<ARITH_TIGHT>>(geoms, imgs, w, lds_raw);
        else if (fl & 1u) walk_item<S440<ARITH_SANE>>(geoms, imgs, w, lds_raw);
    }
}

// a = tile, b = MCU row
template <int ARITH>
__device__ __forceinline__ void fgen_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work,
                                          uint8_t *lds_raw) {
    typedef FGen<ARITH> K;
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const FGenLds lds = FGenLds::make(lds_raw, g.tx, g.hs, g.vs);
    const uint32_t tid = threadIdx.x;
    S420Regs r;
    K::init(img, tid, lds);
    {
        typename K::Pre pre;
        K::stage_load(g, img, w.a, w.b, tid, pre);
        K::stage_store(g, w.a, tid, lds, pre);
    }
    __syncthreads();
    K::read_block(g, w.a, tid, lds, r);
    __syncthreads();  // the tiles alias the staging area
    K::transform(g, w.a, tid, lds, r);
    __syncthreads();
    K::colour(g, img, w.a, w.b, tid, lds);
}
template <int ARITH>
__global__ __launch_bounds__(256, 4) void fgen_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                      const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    fgen_body<ARITH>(geoms, imgs, work, lds_raw);
}
__global__ __launch_bounds__(256, 4) void fgen_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                          const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
#define JP_CALL(A) fgen_body<A>(geoms, imgs, work, lds_raw)
    JP_DYN_DISPATCH(image_flags(imgs, work), JP_CALL);
#undef JP_CALL
}

// four components, some at half size: a strip walk (W4, fused_x4.hpp): a = strip, MCU rows [b, c) of it
template <int ARITH, bool K_FULL>
__global__ __launch_bounds__(256, walk_wgs(ARITH == ARITH_EXACT)) void w4_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                    const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    walk_item<W4<ARITH, K_FULL>>(geoms, imgs, walk_item_at(geoms, work, blockIdx.x), lds_raw);
}
template <bool K_FULL, bool EXACT_PASS>  // (two launches, like the other walks: the wrap-exact body wants more registers)
__global__ __launch_bounds__(256, walk_wgs(EXACT_PASS)) void w4_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                        const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const FusedWork w = walk_item_at(geoms, work, blockIdx.x);
    const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)imgs[w.image].flags);
    if constexpr (EXACT_PASS) {
        if (!(fl & 1u)) walk_item<W4<ARITH_EXACT, K_FULL>>(geoms, imgs, w, lds_raw);
    } else {
        if (fl & 2u) walk_item<W4<ARITH_TIGHT, K_FULL>>(geoms, imgs, w, lds_raw);
        else if (fl & 1u) walk_item<W4<ARITH_SANE, K_FULL>>(geoms, imgs, w, lds_raw);
    }
}

template <int ARITH>
__device__ __forceinline__ void f444_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work,
                                          FusedLdsSmall &lds) {
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    FusedRegs r;
    F444<ARITH>::phase0(g, img, w.a, w.b, threadIdx.x, lds);
    __syncthreads();
    const uint32_t wcomp = min((uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), F444<ARITH>::ncomp(g) - 1u);
    F444<ARITH>::phase1(g, imgs[w.image].qt[wcomp], w.a, threadIdx.x, lds, r);
    __syncthreads();
    F444<ARITH>::phase2(g, w.a, threadIdx.x, lds, r);
    __syncthreads();
    F444<ARITH>::phase3(g, img, w.a, w.b, threadIdx.x, lds);
}
template <int ARITH>
__global__ __launch_bounds__(256) void f444_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                   const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
    f444_body<ARITH>(geoms, imgs, work, lds);
}
__global__ __launch_bounds__(256) void f444_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                       const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
#define JP_CALL(A) f444_body<A>(geoms, imgs, work, lds)
    JP_DYN_DISPATCH(image_flags(imgs, work), JP_CALL);
#undef JP_CALL
}

template <int ARITH>
__device__ __forceinline__ void f422_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work,
                                          FusedLdsSmall &lds) {
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    FusedRegs r;
    F422<ARITH>::phase0(g, img, w.a, w.b, threadIdx.x, lds);
    __syncthreads();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    F422<ARITH>::phase1(g, imgs[w.image].qt[wave < 2u ? 0u : wave - 1u], w.a, threadIdx.x, lds, r);
    __syncthreads();
    F422<ARITH>::phase2(g, w.a, threadIdx.x, lds, r);
    __syncthreads();
    F422<ARITH>::phase3(g, img, w.a, w.b, threadIdx.x, lds);
}
template <int ARITH>
__global__ __launch_bounds__(256) void f422_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                   const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
    f422_body<ARITH>(geoms, imgs, work, lds);
}
__global__ __launch_bounds__(256) void f422_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                       const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
#define JP_CALL(A) f422_body<A>(geoms, imgs, work, lds)
    JP_DYN_DISPATCH(image_flags(imgs, work), JP_CALL);
#undef JP_CALL
}

template <int ARITH>
__device__ __forceinline__ void fgray_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work,
                                           FusedLdsSmall &lds) {
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    FGray<ARITH>::phase0(g, img, w.a, w.b, threadIdx.x, lds);
    __syncthreads();
    FGray<ARITH>::phase1(g, img, w.a, w.b, threadIdx.x, lds);
}
template <int ARITH>
__global__ __launch_bounds__(256) void fgray_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                    const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
    fgray_body<ARITH>(geoms, imgs, work, lds);
}
__global__ __launch_bounds__(256) void fgray_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                        const FusedWork *__restrict__ work) {
    __shared__ FusedLdsSmall lds;
#define JP_CALL(A) fgray_body<A>(geoms, imgs, work, lds)
    JP_DYN_DISPATCH(image_flags(imgs, work), JP_CALL);
#undef JP_CALL
}

// Statistics -> class bits of every image of a plan, in its image table, right in front of the `_dyn` launch (range_stats.hpp).
// host_cls[image * 4 + comp]: the class the host knows (0, 1, 3) or CLS_FROM_DEVICE; stats: RS_WORDS per batch image.
__global__ __launch_bounds__(256) void class_finalize_fused_kernel(FusedImage *__restrict__ imgs, const uint32_t *__restrict__ ids, uint32_t n,
                                                                   uint32_t ncomp, const uint32_t *__restrict__ stats,
                                                                   const uint8_t *__restrict__ host_cls, uint32_t cap_bits) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t gi = ids[i];
    const uint32_t *st = stats + (size_t)gi * RS_WORDS;
    const uint32_t dev = range_class_from_stats(st[RS_MAX_DC], st[RS_MAX_AC], st[RS_MAX_COL], st[RS_COL_EXACT]);
    uint32_t fl = 3u;
    bool skip = false;  // CLS_SKIP: the entry-list walk makes this image's pixels; the dense kernels leave it alone (flag bit 3)
    for (uint32_t c = 0; c < ncomp; c++) {
        const uint32_t h = host_cls[gi * 4u + c];
        skip = skip || h == CLS_SKIP;
        fl &= h == CLS_FROM_DEVICE ? dev : h;
    }
    if (!(fl & 1u)) fl = 0u;  // tight implies sane
    imgs[i].flags = skip ? (8u | 1u) : (fl & cap_bits);
}

// ---- host side ------------------------------------------------------------------------------
// JPGPU_ARITH (tuning / testing knob): cap the arithmetic variant — 0 wrap-exact, 1 sane, 2 (default) tight
static int arith_cap() {
    int cap = (int)ARITH_TIGHT;
    if (const char *ae = getenv("JPGPU_ARITH")) cap = std::max(0, std::min(cap, atoi(ae)));
    return cap;
}
uint32_t fused_kind_key(const jpgpu_image_desc &d) {
    FusedGeom g;
    const char *nm = "", *w = "";
    const int kind = fused_geom_from_desc(d, g, nm, w);
    return kind == FUSED_NONE ? 0u : (uint32_t)kind * 8u + g.color + 4u * g.k_full;  // (color < 4; k_full only with CMYK / YCCK frames)
}

bool fused_plan(const std::vector<jpgpu_image_desc> &descs, const std::vector<uint32_t> &ids, FusedPlan &plan, std::string &why) {
    plan = FusedPlan();
    plan.uniform = false;
    if (descs.empty() || descs.size() > 65535u || ids.size() != descs.size()) return false;
    plan.ids = ids;
    // JPGPU_S420_TX / JPGPU_S420_SEG (test knobs): strip width and MCU rows per workgroup of the 4:2:0 walk — narrow strips and short
    // segments exercise halos and seams on small images (tests/test_gpu_parity.py)
    const char *stx = getenv("JPGPU_S420_TX");
    const uint32_t n = (uint32_t)descs.size();
    bool uniform = true;
    for (uint32_t i = 1; i < n && uniform; i++) {
        const jpgpu_image_desc &d = descs[i], &d0 = descs[0];
        uniform = d.ncomp == d0.ncomp && d.out_w == d0.out_w && d.out_h == d0.out_h && d.color_transform == d0.color_transform;
        for (uint32_t c = 0; uniform && c < d.ncomp; c++) uniform = fused_same_component(d.components[c], d0.components[c]);
    }
    plan.uniform = uniform;
    plan.geoms.resize(n);
    const char *name = "";
    for (uint32_t i = 0; i < n; i++) {
        const char *nm = "", *w = "";
        int kind = fused_geom_from_desc(descs[i], plan.geoms[i], nm, w, stx ? (uint32_t)atoi(stx) : S420_TX_MAX);
        if (kind == FUSED_NONE) {
            why = w;
            return false;
        }
        if (const char *t = getenv("JPGPU_TX"); t && kind != FUSED_420) {  // experiment: tile / strip width of the other kinds (<= the kind's maximum)
            FusedGeom &g = plan.geoms[i];
            g.tx = std::min<uint32_t>(std::max(1, atoi(t)), g.tx * g.tiles_x);
            g.tiles_x = (g.mcu_w + g.tx - 1) / g.tx;
        }
        if (i == 0) {
            plan.kind = kind;
            name = nm;
        } else if (kind != plan.kind || plan.geoms[i].color != plan.geoms[0].color || plan.geoms[i].k_full != plan.geoms[0].k_full) {
            why = "images of different kinds";  // e.g. gray next to 4:2:0: the generic path takes the batch
            plan.kind = FUSED_NONE;
            return false;
        }
    }
    plan.name = name;
    plan.n_images = n;
    plan.ncomp = descs[0].ncomp;
    plan.strip = (plan.kind == FUSED_420 || plan.kind == FUSED_440 || plan.kind == FUSED_420X4) && plan.geoms[0].strip != 0;
    if (plan.strip) {
        // segments per strip by the heuristic of s420_set_segments, one (strip, segment) per workgroup; JPGPU_S420_SEG = n: fixed
        // segments of n MCU rows (test knob)
        const char *sr = getenv("JPGPU_S420_SEG");
        for (auto &g : plan.geoms) s420_set_segments(g, n, sr ? (uint32_t)atoi(sr) : 0u);
    }
    // workgroup size, LDS claim, work tables
    uint32_t tx_max = 0;
    for (const auto &g : plan.geoms) tx_max = std::max(tx_max, g.tx);
    plan.nt = 256;
    plan.lds_bytes = plan.kind == FUSED_440 ? S440Lds::total_bytes(tx_max) : (plan.kind == FUSED_420 ? S420Lds::total_bytes(tx_max) : 0);
    if (plan.kind == FUSED_420X4) plan.lds_bytes = plan.geoms[0].k_full ? W4Lds<2, 2>::total_bytes(tx_max) : W4Lds<1, 3>::total_bytes(tx_max);
    if (plan.kind == FUSED_GEN) {  // (images of one launch group may differ in H x V: the largest claim)
        plan.lds_bytes = 0;
        for (const auto &g : plan.geoms) plan.lds_bytes = std::max<size_t>(plan.lds_bytes, FGenLds::total_bytes(g.tx, g.hs, g.vs));
    }
    if (const char *pad = getenv("JPGPU_LDS_PAD")) plan.lds_bytes += (size_t)atoi(pad);  // occupancy experiments: claim more LDS than needed
    // Unequal segments, longest first (round 6).  When the equal split comes out at two or a few segments per strip the launch
    // is between one and two rounds of the device's resident workgroups (256 x 1080p: 1,536 workgroups of 34 MCU rows on 1,024 slots) — the
    // second round runs on half the machine.  Cut every strip at 65 / 82 / 94 % of its rows instead and put ALL first segments in front
    // of all second ones, and so on (work table in segment order): the 1,024 slots take the 768 long segments and a third of the short
    // ones, the short ones' slots work through the rest, and everybody ends together — 256 x 1080p 0.633-0.668 -> 0.608-0.657 ms on two
    // boxes (+1.7 % / +4 %; cuts 44,56,64 of 68 rows best of 24 variants: profiles/round6/04_launch_shape.txt).  JPGPU_S420_CUTS="r1,r2,..."
    // sets the cuts by hand (experiments), JPGPU_S420_EQUAL=1 keeps the equal split (A/B).
    std::vector<uint32_t> cuts;
    if (const char *cs = plan.strip ? getenv("JPGPU_S420_CUTS") : nullptr) {
        for (const char *q = cs; *q;) {
            cuts.push_back((uint32_t)strtoul(q, const_cast<char **>(&q), 10));
            if (*q == ',') q++;
        }
    } else if (plan.strip && uniform && plan.geoms[0].n_seg >= 2u && plan.geoms[0].mcu_h >= 16u && (uint64_t)plan.geoms[0].tiles_x * n >= 512u &&
               !getenv("JPGPU_S420_SEG") && !getenv("JPGPU_S420_EQUAL")) {
        // (every strip walk — 4:2:0, 4:4:0, the four-component ones — whose launch has at least half as many strips as the device has
        // slots: 256 x 1080p 4:4:0 0.6244 -> 0.6355 of the roofline, CMYK 22 11 11 11 0.631 -> 0.664, YCCK 22 11 11 22 0.628 -> 0.647;
        // launches of fewer strips keep many short equal segments: they need the workgroups)
        const uint32_t h = plan.geoms[0].mcu_h;
        for (uint32_t pct : {65u, 82u, 94u}) cuts.push_back((h * pct + 50u) / 100u);
    }
    if (!cuts.empty()) {
        cuts.push_back(0xffffffffu);
        uint32_t lo = 0;
        for (uint32_t hi : cuts) {
            for (uint32_t i = 0; i < n; i++) {
                const FusedGeom &g = plan.geoms[i];
                const uint32_t a = std::min(lo, g.mcu_h), b = std::min(hi, g.mcu_h);
                for (uint32_t x = 0; a < b && x < g.tiles_x; x++) plan.work_main.push_back(FusedWork{i, x, a, b});
            }
            lo = hi;
        }
        plan.uniform = false;
    } else
    for (uint32_t i = 0; i < n; i++) {
        const FusedGeom &g = plan.geoms[i];
        const uint32_t ny = plan.strip ? g.n_seg : g.mcu_h;
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < g.tiles_x; x++)
                plan.work_main.push_back(plan.strip ? FusedWork{i, x, y * g.seg_rows, std::min((y + 1u) * g.seg_rows, g.mcu_h)} : FusedWork{i, x, y, 0u});
    }
    // JPGPU_FUSED_TABLE=1 forces the table form (test knob)
    if (const char *ft = getenv("JPGPU_FUSED_TABLE")) if (atoi(ft) != 0) plan.uniform = false;
    plan.images.assign(n, FusedImage{});
    return true;
}

int fused_alloc(FusedPlan &plan, std::string &err) {
    if (plan.kind == FUSED_NONE) return JPGPU_OK;
#define F_HIP(call)                                                                                            \
    do {                                                                                                       \
        hipError_t _e = (call);                                                                                \
        if (_e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(_e));        \
    } while (0)
    F_HIP(hipMalloc((void **)&plan.d_images, sizeof(FusedImage) * plan.n_images));
    F_HIP(hipMalloc((void **)&plan.d_geoms, sizeof(FusedGeom) * plan.n_images));
    F_HIP(hipMemcpy(plan.d_geoms, plan.geoms.data(), sizeof(FusedGeom) * plan.n_images, hipMemcpyHostToDevice));
    F_HIP(hipMalloc((void **)&plan.d_ids, sizeof(uint32_t) * plan.n_images));
    F_HIP(hipMemcpy(plan.d_ids, plan.ids.data(), sizeof(uint32_t) * plan.n_images, hipMemcpyHostToDevice));
    F_HIP(hipEventCreateWithFlags(&plan.launched, hipEventDisableTiming));
    F_HIP(hipMalloc((void **)&plan.d_work_main, sizeof(FusedWork) * std::max<size_t>(plan.work_main.size(), 1)));
    F_HIP(hipMemcpy(plan.d_work_main, plan.work_main.data(), sizeof(FusedWork) * plan.work_main.size(), hipMemcpyHostToDevice));
#undef F_HIP
    return JPGPU_OK;
}

int fused_bind(FusedPlan &plan, uint8_t *d_coef, uint8_t *d_out, uint16_t *d_qt, const std::vector<size_t> &coef_off,
               const std::vector<size_t> &out_off, const std::vector<uint8_t> &sane, std::string &err) {
    if (plan.kind == FUSED_NONE) return JPGPU_OK;
    std::vector<uint8_t> cls(plan.n_images, 0);
    const int cap = arith_cap();
    // The previous launch of this plan may still be reading the tables rewritten below (decodes are asynchronous on the
    // caller's stream, possibly a non-blocking one that the copies on the null stream do not order against)
    if (plan.launched && plan.launch_pending) {
        (void)hipEventSynchronize(plan.launched);
        plan.launch_pending = false;
    }
    plan.class_images[0] = plan.class_images[1] = plan.class_images[2] = 0;
    for (uint32_t i = 0; i < plan.n_images; i++) {
        FusedImage &im = plan.images[i];
        uint32_t fl = 3u;
        const size_t gi = plan.ids[i];  // index in the batch
        for (uint32_t c = 0; c < plan.ncomp; c++) {
            im.coefs[c] = reinterpret_cast<const int16_t *>(d_coef + coef_off[gi * 4 + c]);
            im.qt[c] = d_qt + (gi * 4 + c) * 64;
            fl &= sane[gi * 4 + c];
        }
        im.out = d_out + out_off[gi];
        if (!(fl & 1u)) fl = 0u;  // tight implies sane
        im.flags = fl;
        cls[i] = (uint8_t)std::min(cap, (fl & 2u) ? (int)ARITH_TIGHT : ((fl & 1u) ? (int)ARITH_SANE : (int)ARITH_EXACT));
        plan.class_images[cls[i]]++;
    }
    const int present = (plan.class_images[0] != 0) + (plan.class_images[1] != 0) + (plan.class_images[2] != 0);
    plan.arith = plan.class_images[2] ? ARITH_TIGHT : (plan.class_images[1] ? ARITH_SANE : ARITH_EXACT);
    plan.by_class = present > 1;
    hipError_t e = hipMemcpy(plan.d_images, plan.images.data(), sizeof(FusedImage) * plan.n_images, hipMemcpyHostToDevice);
    if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMemcpy(images): %s", hipGetErrorString(e));
    if (plan.by_class) {
        // the plan's work tables, split by the class of the image each entry belongs to
        std::vector<FusedWork> all;
        for (int c = 0; c < 3; c++) {
            plan.n_main_cls[c] = 0;
            for (const FusedWork &w : plan.work_main)
                if (cls[w.image] == c) all.push_back(w), plan.n_main_cls[c]++;
        }
        if (all.size() > plan.work_cls_cap) {
            if (plan.d_work_cls) (void)hipFree(plan.d_work_cls);
            plan.d_work_cls = nullptr;
            plan.work_cls_cap = 0;
            e = hipMalloc((void **)&plan.d_work_cls, sizeof(FusedWork) * all.size());
            if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMalloc(class work tables): %s", hipGetErrorString(e));
            plan.work_cls_cap = all.size();
        }
        e = hipMemcpy(plan.d_work_cls, all.data(), sizeof(FusedWork) * all.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMemcpy(class work tables): %s", hipGetErrorString(e));
    }
    return JPGPU_OK;
}

// One launch group: `n_main` workgroups over the work table W, or the 3-D grid of a uniform batch (W null).
static hipError_t fused_launch_one(FusedPlan &plan, hipStream_t stream, int ar, const FusedWork *W, uint32_t n_main) {
    const FusedGeom *G = plan.d_geoms;
    const FusedImage *I = plan.d_images;
    const FusedGeom &g0 = plan.geoms[0];
    const dim3 grid = W ? dim3(n_main) : dim3(g0.tiles_x, plan.strip ? g0.n_seg : g0.mcu_h, plan.n_images);
    const dim3 block(plan.nt);
    const size_t shm = plan.lds_bytes;
    // ar < 0: the `_dyn` form (class per image from the image table on the device)
#define ARITH_SWITCH(KERNEL, DYN, ...)                                                                 \
    do {                                                                                               \
        if (ar < 0) DYN<<<grid, block, shm, stream>>>(G, I, W);                                        \
        else if (ar == ARITH_TIGHT) KERNEL<ARITH_TIGHT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);  \
        else if (ar == ARITH_SANE) KERNEL<ARITH_SANE, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W); \
        else KERNEL<ARITH_EXACT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);                  \
    } while (0)
    // strip walks; ar < 0: the two `_dyn` passes
#define WALK_SWITCH(KERNEL, DYN_FAST, DYN_EXACT, ...)                                                               \
    do {                                                                                                            \
        if (ar < 0) {                                                                                               \
            DYN_FAST<<<grid, block, shm, stream>>>(G, I, W);                                                        \
            DYN_EXACT<<<grid, block, shm, stream>>>(G, I, W);                                                       \
        } else if (ar == ARITH_TIGHT) KERNEL<ARITH_TIGHT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);    \
        else if (ar == ARITH_SANE) KERNEL<ARITH_SANE, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);        \
        else KERNEL<ARITH_EXACT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);                             \
    } while (0)
    switch (plan.kind) {
    case FUSED_420: WALK_SWITCH(s420_kernel, (s420_kernel_dyn<256, false>), (s420_kernel_dyn<256, true>), 256); break;
    case FUSED_440: WALK_SWITCH(s440_kernel, s440_kernel_dyn<false>, s440_kernel_dyn<true>); break;
    case FUSED_GEN: ARITH_SWITCH(fgen_kernel, fgen_kernel_dyn); break;
    case FUSED_420X4:
        if (ar < 0) {
            if (g0.k_full) {
                w4_kernel_dyn<true, false><<<grid, block, shm, stream>>>(G, I, W);
                w4_kernel_dyn<true, true><<<grid, block, shm, stream>>>(G, I, W);
            } else {
                w4_kernel_dyn<false, false><<<grid, block, shm, stream>>>(G, I, W);
                w4_kernel_dyn<false, true><<<grid, block, shm, stream>>>(G, I, W);
            }
        } else if (g0.k_full) {
            if (ar == ARITH_TIGHT) w4_kernel<ARITH_TIGHT, true><<<grid, block, shm, stream>>>(G, I, W);
            else if (ar == ARITH_SANE) w4_kernel<ARITH_SANE, true><<<grid, block, shm, stream>>>(G, I, W);
            else w4_kernel<ARITH_EXACT, true><<<grid, block, shm, stream>>>(G, I, W);
        } else {
            if (ar == ARITH_TIGHT) w4_kernel<ARITH_TIGHT, false><<<grid, block, shm, stream>>>(G, I, W);
            else if (ar == ARITH_SANE) w4_kernel<ARITH_SANE, false><<<grid, block, shm, stream>>>(G, I, W);
            else w4_kernel<ARITH_EXACT, false><<<grid, block, shm, stream>>>(G, I, W);
        }
        break;
    case FUSED_444: ARITH_SWITCH(f444_kernel, f444_kernel_dyn); break;
    case FUSED_422: ARITH_SWITCH(f422_kernel, f422_kernel_dyn); break;
    case FUSED_GRAY: ARITH_SWITCH(fgray_kernel, fgray_kernel_dyn); break;
    default: return hipErrorInvalidValue;
    }
#undef ARITH_SWITCH
#undef WALK_SWITCH
    return hipGetLastError();
}

// One batch = one launch — per arithmetic class when the images disagree.
// d_stats / d_host_cls != null: the classes are decided on the device — class_finalize_fused_kernel writes them into the image
// table from the statistics there (and the classes the host does know), then ONE `_dyn` launch over the whole work table.
hipError_t fused_finalize_classes(FusedPlan &plan, hipStream_t stream, const uint32_t *d_stats, const uint8_t *d_host_cls) {
    if (plan.kind == FUSED_NONE || !d_stats || !d_host_cls) return hipSuccess;
    const int cap = arith_cap();
    class_finalize_fused_kernel<<<dim3((plan.n_images + 255u) / 256u), dim3(256), 0, stream>>>(
        plan.d_images, plan.d_ids, plan.n_images, plan.ncomp, d_stats, d_host_cls, cap >= (int)ARITH_TIGHT ? 3u : (cap == (int)ARITH_SANE ? 1u : 0u));
    return hipGetLastError();
}

hipError_t fused_launch(FusedPlan &plan, hipStream_t stream, const uint32_t *d_stats, const uint8_t *d_host_cls) {
    hipError_t e = hipSuccess;
    const bool table = !plan.uniform;
    if (d_stats && d_host_cls) {
        e = fused_finalize_classes(plan, stream, d_stats, d_host_cls);
        if (e == hipSuccess) e = fused_launch_one(plan, stream, -1, table ? plan.d_work_main : nullptr, (uint32_t)plan.work_main.size());
    } else if (!plan.by_class) {
        e = fused_launch_one(plan, stream, plan.arith, table ? plan.d_work_main : nullptr, (uint32_t)plan.work_main.size());
    } else {
        const FusedWork *w = plan.d_work_cls;
        for (int c = 0; c < 3 && e == hipSuccess; c++) {
            if (plan.n_main_cls[c]) e = fused_launch_one(plan, stream, c, w, plan.n_main_cls[c]);
            w += plan.n_main_cls[c];
        }
    }
    if (e == hipSuccess && plan.launched && hipEventRecord(plan.launched, stream) == hipSuccess) plan.launch_pending = true;
    return e;
}

// The images of a 4:2:0 plan whose coefficients are entry lists (srcs[batch image].job set): one launch over the plan's work table.
hipError_t fused_launch_entries(FusedPlan &plan, hipStream_t stream, const EntrySrc *d_srcs) {
    if (plan.kind != FUSED_420 || !plan.strip || !d_srcs || plan.work_main.empty()) return hipErrorInvalidValue;
    uint32_t tx_max = 0;
    for (const auto &g : plan.geoms) tx_max = std::max(tx_max, g.tx);
    const size_t shm = S420Lds::total_bytes(tx_max) + S420ELds::total_bytes(tx_max);
    s420_entries_kernel<<<dim3((uint32_t)plan.work_main.size()), dim3(256), shm, stream>>>(plan.d_geoms, plan.d_images, plan.d_work_main, plan.d_ids, d_srcs);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && plan.launched && hipEventRecord(plan.launched, stream) == hipSuccess) plan.launch_pending = true;
    return e;
}

int fused_read_classes(FusedPlan &plan, std::vector<uint8_t> &bits, std::string &err) {
    bits.assign(plan.n_images, 0);
    if (plan.kind == FUSED_NONE || !plan.d_images) return JPGPU_OK;
    if (plan.launched && plan.launch_pending) {
        (void)hipEventSynchronize(plan.launched);
        plan.launch_pending = false;
    }
    std::vector<FusedImage> imgs(plan.n_images);
    hipError_t e = hipMemcpy(imgs.data(), plan.d_images, sizeof(FusedImage) * plan.n_images, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return set_err(err, JPGPU_ERR_IO, "hipMemcpy(image table): %s", hipGetErrorString(e));
    for (uint32_t i = 0; i < plan.n_images; i++) bits[i] = (uint8_t)(imgs[i].flags & 3u);
    return JPGPU_OK;
}

void fused_free(FusedPlan &plan) {
    if (plan.d_images) (void)hipFree(plan.d_images);
    if (plan.d_geoms) (void)hipFree(plan.d_geoms);
    if (plan.d_work_main) (void)hipFree(plan.d_work_main);
    if (plan.d_work_cls) (void)hipFree(plan.d_work_cls);
    if (plan.d_ids) (void)hipFree(plan.d_ids);
    if (plan.launched) (void)hipEventDestroy(plan.launched);
    plan.d_ids = nullptr;
    plan.launched = nullptr;
    plan.launch_pending = false;
    plan.d_work_cls = nullptr;
    plan.work_cls_cap = 0;
    plan.d_images = nullptr;
    plan.d_geoms = nullptr;
    plan.d_work_main = nullptr;
}

}  // namespace jpgpu
