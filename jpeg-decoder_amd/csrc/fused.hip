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
#include "idct_plane_body.hpp"
#include "range_stats.hpp"

namespace jpgpu {

// Which image / tile a workgroup owns: from the work table (mixed-size batches, 1-D grid) or, when the batch is uniform
// and no table is passed, straight from the 3-D grid (x, y, image) — that saves the dependent scalar load at the
// start of every workgroup (measured: 2 % on the 4:2:0 bench).
__device__ __forceinline__ FusedWork locate(const FusedWork *__restrict__ work) {
    if (work) return work[blockIdx.x];
#ifdef JPGPU_XCD_SWIZZLE
    // Experiment: workgroups are handed to the 8 XCDs round-robin in launch order; renumber them so that each
    // XCD walks one contiguous eighth of the (tile, MCU row, image) space — neighbours in x and y then share an L2.
    const uint32_t gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    uint32_t l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if ((n & 7u) == 0u) l = (l & 7u) * (n >> 3) + (l >> 3);
    const uint32_t t = l / gx;
    return FusedWork{t / gy, l - t * gx, t - (t / gy) * gy, 0u};
#else
    return FusedWork{blockIdx.z, blockIdx.x, blockIdx.y, 0u};
#endif
}

// work item of the chroma pass: a = component (0 Cb, 1 Cr), b = 256-block group within the plane
__global__ __launch_bounds__(256) void f420_chroma_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                          const FusedWork *__restrict__ work) {
    __shared__ v4u lds[256 * 8];
    const FusedWork w = locate(work);
    const FusedGeom &g = geoms[w.image];
    const FusedImage *img = imgs + w.image;  // (indexed through memory: a runtime index into a by-value copy would go to scratch)
    PlaneJob job;
    job.coefs = img->coefs[1u + w.a];
    job.plane = img->scratch + (size_t)w.a * g.chroma_plane_bytes;
    job.qt = img->qt[1u + w.a];
    job.block_w = g.bwc;
    job.n_blocks = g.bwc * g.mcu_h;
    job.scale = 8;
    job.flags = img->flags;
    idct_planes_body<8>(job, w.b, lds);
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

// main pass / 4:4:4 / gray: a = tile within the MCU row, b = MCU row
template <int ARITH, uint32_t NT>
__device__ __forceinline__ void f420_main_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                               const FusedWork *__restrict__ work, uint8_t *lds_raw) {
    typedef F420<ARITH, NT> K;
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const F420Lds lds = F420Lds::make(lds_raw, g.tx);
    FusedRegs r;
    K::phase0(g, img, w.a, w.b, threadIdx.x, lds);
    __syncthreads();
    K::phase1(g, img, w.a, threadIdx.x, lds, r);
    __syncthreads();
    K::phase2(g, w.a, threadIdx.x, lds, r);
    __syncthreads();
    K::phase3(g, img, w.a, w.b, threadIdx.x, lds);
}
template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, NT == 128 ? 4 : 3) void f420_main_kernel(const FusedGeom *__restrict__ geoms,
                                                                           const FusedImage *__restrict__ imgs,
                                                                           const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    f420_main_body<ARITH, NT>(geoms, imgs, work, lds_raw);
}
template <uint32_t NT>
__global__ __launch_bounds__(NT, NT == 128 ? 4 : 3) void f420_main_kernel_dyn(const FusedGeom *__restrict__ geoms,
                                                                               const FusedImage *__restrict__ imgs,
                                                                               const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
#define JP_CALL(A) f420_main_body<A, NT>(geoms, imgs, work, lds_raw)
    JP_DYN_DISPATCH(image_flags(imgs, work), JP_CALL);
#undef JP_CALL
}

// Phase clocks (a diagnostic build: -DJPGPU_PHASE_CLOCKS, tools/gpu_phase_clocks.sh): every wave sums, per phase of the
// walk, the shader-clock time from the barrier that opened the phase to the end of its own work and from there to the
// release of the barrier that closes it; jpgpu_debug_phase_clocks() reads the totals.
#ifdef JPGPU_PHASE_CLOCKS
__device__ unsigned long long g_phase_clocks[16];
struct PhaseClock {
    unsigned long long last, acc[10];
    __device__ __forceinline__ PhaseClock() : last(__builtin_readcyclecounter()) {
        for (auto &a : acc) a = 0;
    }
    __device__ __forceinline__ void mark(int slot) {
        const unsigned long long t = __builtin_readcyclecounter();
        acc[slot] += t - last;
        last = t;
    }
    __device__ __forceinline__ void flush() {
        if ((threadIdx.x & 63u) == 0u) {
            for (int i = 0; i < 10; i++) atomicAdd(&g_phase_clocks[i], acc[i]);
            atomicAdd(&g_phase_clocks[15], 1ull);
        }
    }
};
#define PHASE_MARK(slot) pc.mark(slot)
#else
#define PHASE_MARK(slot) (void)0
#endif

// Strip walks (4:2:0: S420, 4:4:0: S440 in fused_core.hpp).  One work item = MCU rows [k0, k1) of one strip of one image.
// Which items a workgroup owns (walk_items_of): with `wg_first` (balanced launches: walk_balanced_items, fused_plan.hpp) items
// wg_first[blockIdx.x] .. wg_first[blockIdx.x + 1] of the table, a contiguous share of the launch's steps; with a table alone the
// one item work[blockIdx.x]; without a table (uniform batch, fixed segments) strip blockIdx.x, segment blockIdx.y of image
// blockIdx.z.
struct WalkItems {
    uint32_t first, end;  // table indices (table forms); first = 0, end = 1 for the 3-D grid form
};
__device__ __forceinline__ WalkItems walk_items_of(const FusedWork *__restrict__ work, const uint32_t *__restrict__ wg_first) {
    if (wg_first) return WalkItems{wg_first[blockIdx.x], wg_first[blockIdx.x + 1u]};
    if (work) return WalkItems{blockIdx.x, blockIdx.x + 1u};
    return WalkItems{0u, 1u};
}
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
#ifdef JPGPU_PHASE_CLOCKS
    PhaseClock pc;
#endif
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const typename K::Lds lds = K::Lds::make(lds_raw, g.tx);
    const uint32_t strip = w.a, tid = threadIdx.x;
    const uint32_t k0 = w.b, k1 = w.c;
    S420Regs r;
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
    PHASE_MARK(8);  // prologue: set-up, seam round, first stage
    for (uint32_t k = k0; k < k1; k++) {
        typename K::Pre pre;  // (declared per iteration: not live around the loop)
        // A fresh, opaque copy of the lane id per phase: otherwise every per-lane address of every phase is hoisted
        // out of the loop and kept live across it (~100 VGPRs), which spills the coefficient block at 4 waves/SIMD.
        uint32_t t0 = tid, t1 = tid, t2 = tid;
        asm volatile("" : "+v"(t1));
        K::read_block(g, strip, t1, lds, r);
        PHASE_MARK(0);
        __syncthreads();  // the tiles alias the staging area
        PHASE_MARK(1);
        K::transform(g, strip, t1, lds, r);
        PHASE_MARK(2);
        __syncthreads();
        PHASE_MARK(3);
        const bool more = k + 1u < k1;
        asm volatile("" : "+v"(t2));
        K::colour(g, img, strip, k, 16u * k0, false, t2, lds);
        PHASE_MARK(4);
        __syncthreads();
        PHASE_MARK(5);
        if (more) {
            asm volatile("" : "+v"(t0));
            K::stage_load(g, img, strip, k + 1u, t0, pre);
            K::stage_store(g, strip, t0, lds, pre);
            PHASE_MARK(6);
            __syncthreads();
            PHASE_MARK(7);
        }
    }
    if (16u * k1 - 1u < g.out_h) {  // the segment's last output row: its far chroma row is the seam row below (or itself at the image's end)
        K::closing_tiles(tid, lds);
        __syncthreads();
        K::colour(g, img, strip, k1, 16u * k0, true, tid, lds);
    }
    PHASE_MARK(9);  // epilogue
#ifdef JPGPU_PHASE_CLOCKS
    pc.flush();
#endif
}

// The default form: ONE work item per workgroup (work[blockIdx.x], or strip / segment / image from the grid).  Kept free of the
// item loop below: with it the tight 4:2:0 kernel needed 128 VGPRs, 32 spilled SGPRs and scratch memory where it takes 119
// VGPRs and none on its own.
template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, 4) void s420_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                     const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    walk_item<S420<ARITH, NT>>(geoms, imgs, walk_item_at(geoms, work, blockIdx.x), lds_raw);
}
// Balanced shares (JPGPU_WALK_BALANCE=1, the A/B partner): workgroup w runs items wg_first[w] .. wg_first[w + 1], with a barrier
// between them (the closing phase of one reads the tiles the next one's staging overwrites)
template <int ARITH, uint32_t NT>
__global__ __launch_bounds__(NT, 4) void s420_kernel_items(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                           const FusedWork *__restrict__ work, const uint32_t *__restrict__ wg_first) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const WalkItems its = walk_items_of(work, wg_first);
    for (uint32_t it = its.first; it < its.end; it++) {
        walk_item<S420<ARITH, NT>>(geoms, imgs, walk_item_at(geoms, work, it), lds_raw);
        if (it + 1u < its.end) __syncthreads();
    }
}
// Classes from the device (one item per workgroup), as TWO launches: the first runs the images whose coefficients are in range
// (tight / sane bodies), the second the others (wrap-exact body); a workgroup whose image belongs to the other launch leaves at
// once.  One kernel with all three bodies inherits the scratch memory of the wrap-exact one (it spills at 128 VGPRs), and a
// kernel with scratch is dispatched more slowly even where no wave touches it: measured 0.737 ms against 0.676 for the tight
// kernel on the same box (profiles/round3/10_kernel_trace_stats_and_pmc.json).
template <uint32_t NT, bool EXACT_PASS>
__global__ __launch_bounds__(NT, 4) void s420_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                         const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const FusedWork w = walk_item_at(geoms, work, blockIdx.x);
    const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)imgs[w.image].flags);
    if constexpr (EXACT_PASS) {
        if (!(fl & 1u)) walk_item<S420<ARITH_EXACT, NT>>(geoms, imgs, w, lds_raw);
    } else {
        if (fl & 2u) walk_item<S420<ARITH_TIGHT, NT>>(geoms, imgs, w, lds_raw);
        else if (fl & 1u) walk_item<S420<ARITH_SANE, NT>>(geoms, imgs, w, lds_raw);
    }
}

template <int ARITH>
__global__ __launch_bounds__(256, 4) void s440_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                      const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    walk_item<S440<ARITH>>(geoms, imgs, walk_item_at(geoms, work, blockIdx.x), lds_raw);
}
template <int ARITH>
__global__ __launch_bounds__(256, 4) void s440_kernel_items(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                            const FusedWork *__restrict__ work, const uint32_t *__restrict__ wg_first) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const WalkItems its = walk_items_of(work, wg_first);
    for (uint32_t it = its.first; it < its.end; it++) {
        walk_item<S440<ARITH>>(geoms, imgs, walk_item_at(geoms, work, it), lds_raw);
        if (it + 1u < its.end) __syncthreads();
    }
}
template <bool EXACT_PASS>
__global__ __launch_bounds__(256, 4) void s440_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
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

// four components, some at half size: a = tile, b = MCU row (fused_x4.hpp)
template <int ARITH, bool K_FULL>
__device__ __forceinline__ void r4_body(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs, const FusedWork *__restrict__ work,
                                        uint8_t *lds_raw) {
    typedef R4<ARITH, K_FULL> K;
    const FusedWork w = locate(work);
    const FusedGeom g = geoms[w.image];
    const FusedImage img = imgs[w.image];
    const R4Lds lds = R4Lds::make(lds_raw, g.tx, K::NL, K::NH);
    const uint32_t tid = threadIdx.x;
    S420Regs r;
    K::init(img, tid, lds);
    {
        typename K::Pre pre;
        K::stage_load(g, img, w.a, w.b, tid, pre);
        K::stage_store(g, w.a, tid, lds, pre);
    }
    __syncthreads();
    // Which wave transforms which blocks rotates from workgroup to workgroup: the first waves hold the blocks that are
    // transformed in full (660 instructions), the last ones only one-row transforms (150), and wave i of every workgroup runs
    // on SIMD i of its CU — unrotated, one SIMD did most of every workgroup's arithmetic while two idled.
    const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t rot = (lin ^ (lin >> 2) ^ (lin >> 5) ^ (lin >> 8) ^ (lin >> 11)) & 3u;
    const uint32_t role = ((((tid >> 6) + rot) & 3u) << 6) | (tid & 63u);  // the lane whose block this lane takes
    K::read_block(g, w.a, w.b, role, lds, r);
    __syncthreads();  // the tiles alias the staging area
    K::transform(g, w.a, w.b, role, lds, r);
    __syncthreads();
    K::colour(g, img, w.a, w.b, tid, lds);
}
template <int ARITH, bool K_FULL>
__global__ __launch_bounds__(256, 4) void r4_kernel(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                    const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    r4_body<ARITH, K_FULL>(geoms, imgs, work, lds_raw);
}
template <bool K_FULL, bool EXACT_PASS>  // (two launches, like the walks: the wrap-exact body spills)
__global__ __launch_bounds__(256, 4) void r4_kernel_dyn(const FusedGeom *__restrict__ geoms, const FusedImage *__restrict__ imgs,
                                                        const FusedWork *__restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t fl = image_flags(imgs, work);
    if constexpr (EXACT_PASS) {
        if (!(fl & 1u)) r4_body<ARITH_EXACT, K_FULL>(geoms, imgs, work, lds_raw);
    } else {
        if (fl & 2u) r4_body<ARITH_TIGHT, K_FULL>(geoms, imgs, work, lds_raw);
        else if (fl & 1u) r4_body<ARITH_SANE, K_FULL>(geoms, imgs, work, lds_raw);
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
    for (uint32_t c = 0; c < ncomp; c++) {
        const uint32_t h = host_cls[gi * 4u + c];
        fl &= h == CLS_FROM_DEVICE ? dev : h;
    }
    if (!(fl & 1u)) fl = 0u;  // tight implies sane
    imgs[i].flags = fl & cap_bits;
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
    // Knobs (A/B experiments, profiles/round2/02_single_launch_420.md):
    //   JPGPU_420_STRIP  0 = 4:2:0 as chroma pass + main pass (F420, round 1's default); default 1 = the single-launch
    //                    strip walk (S420): chroma never goes through HBM (3.3 instead of 4.1 GB per 256 x 1080p)
    //   JPGPU_F420_TX    MCUs per tile of the two-pass main kernel (<= 32 selects 128-thread workgroups)
    //   JPGPU_S420_TX / JPGPU_S420_SEG  strip width and MCU rows per workgroup of the strip walk
    const char *txenv = getenv("JPGPU_F420_TX"), *tp = getenv("JPGPU_420_STRIP"), *stx = getenv("JPGPU_S420_TX");
    const uint32_t f420_tx = txenv ? (uint32_t)atoi(txenv) : 64u;
    const bool strip420 = !(tp && atoi(tp) == 0);
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
        int kind = fused_geom_from_desc(descs[i], plan.geoms[i], nm, w, f420_tx, strip420, stx ? (uint32_t)atoi(stx) : S420_TX_MAX);
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
    plan.strip = (plan.kind == FUSED_420 || plan.kind == FUSED_440) && plan.geoms[0].strip != 0;
    if (plan.strip) {
        // Default: segments per strip by the heuristic of s420_set_segments, one (strip, segment) per workgroup.
        // JPGPU_S420_SEG = n: fixed segments of n MCU rows (A/B and test knob).
        // JPGPU_WALK_BALANCE=1 (round 3 experiment, kept as the A/B partner): balanced shares of the launch's steps, one share
        // per workgroup the device holds at once (walk_balanced_items), JPGPU_WALK_ROUNDS = r: r times as many, shorter shares.
        // Measured SLOWER (256 x 1080p, same box, profiles/round3/03_balanced_walk.txt): segments 0.654 ms; one share per
        // resident workgroup 0.685-0.708; two 0.687-0.702; three 0.649-0.671 — workgroups that all start together stay in step
        // (all loading, then all computing), and it is the spread of phases across the workgroups of a CU that overlaps the
        // memory phases of one with the arithmetic of another.
        const char *sr = getenv("JPGPU_S420_SEG"), *wb = getenv("JPGPU_WALK_BALANCE"), *wr = getenv("JPGPU_WALK_ROUNDS");
        for (auto &g : plan.geoms) s420_set_segments(g, n, sr ? (uint32_t)atoi(sr) : 0u);
        plan.balanced = !sr && wb && atoi(wb) != 0;
        if (plan.balanced) {
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
            else
                (void)hipGetLastError();
            const uint32_t per_cu = 4u;  // workgroups per CU: 33.6 KB of LDS and <= 128 VGPRs (s420_kernel / s440_kernel launch bounds)
            plan.walk_wgs = (uint32_t)cus * per_cu * (wr ? (uint32_t)std::max(1, atoi(wr)) : 1u);
        }
    }
    // workgroup size, LDS claim, scratch layout, work tables
    uint32_t tx_max = 0;
    for (const auto &g : plan.geoms) tx_max = std::max(tx_max, g.tx);
    plan.nt = 256;
    if (plan.kind == FUSED_420) plan.nt = plan.strip ? 256u : (tx_max <= 32u ? 128u : 256u);
    plan.lds_bytes = plan.kind == FUSED_440 ? S440Lds::total_bytes(tx_max)
                     : plan.kind != FUSED_420 ? 0 : (plan.strip ? S420Lds::total_bytes(tx_max) : F420Lds::total_bytes(tx_max));
    if (plan.kind == FUSED_420X4) plan.lds_bytes = R4Lds::total_bytes(tx_max, plan.geoms[0].k_full ? 2u : 1u, plan.geoms[0].k_full ? 2u : 3u);
    if (plan.kind == FUSED_GEN) {  // (images of one launch group may differ in H x V: the largest claim)
        plan.lds_bytes = 0;
        for (const auto &g : plan.geoms) plan.lds_bytes = std::max<size_t>(plan.lds_bytes, FGenLds::total_bytes(g.tx, g.hs, g.vs));
    }
    if (const char *pad = getenv("JPGPU_LDS_PAD")) plan.lds_bytes += (size_t)atoi(pad);  // occupancy experiments: claim more LDS than needed
    plan.scratch_off.assign(n, 0);
    size_t so = 0;
    for (uint32_t i = 0; i < n; i++) {
        const FusedGeom &g = plan.geoms[i];
        plan.scratch_off[i] = so;
        if (plan.kind == FUSED_420 && !plan.strip) {
            so += align_up(2 * (size_t)g.chroma_plane_bytes, 256);
            const uint32_t groups = (g.bwc * g.mcu_h + 255u) / 256u;
            for (uint32_t comp = 0; comp < 2; comp++)
                for (uint32_t wg = 0; wg < groups; wg++) plan.work_pre.push_back(FusedWork{i, comp, wg, 0u});
        }
        if (plan.strip && plan.balanced) continue;  // (the items of a balanced walk: below, over all images)
        const uint32_t ny = plan.strip ? g.n_seg : g.mcu_h;
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < g.tiles_x; x++)
                plan.work_main.push_back(plan.strip ? FusedWork{i, x, y * g.seg_rows, std::min((y + 1u) * g.seg_rows, g.mcu_h)} : FusedWork{i, x, y, 0u});
    }
    if (plan.strip && plan.balanced) walk_balanced_items(plan.geoms.data(), nullptr, n, plan.walk_wgs, plan.work_main, plan.wg_first);
    plan.scratch_bytes = so;
    // the 3-D grid needs its y extent within 65535; JPGPU_FUSED_TABLE=1 forces the table form (test knob)
    if (plan.kind == FUSED_420 && !plan.strip && (plan.geoms[0].bwc * plan.geoms[0].mcu_h + 255u) / 256u > 65535u) plan.uniform = false;
    if (const char *ft = getenv("JPGPU_FUSED_TABLE")) if (atoi(ft) != 0) plan.uniform = false;
    if (plan.strip && plan.balanced) plan.uniform = false;  // (always through the item table)
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
    if (plan.scratch_bytes) F_HIP(hipMalloc((void **)&plan.d_scratch, plan.scratch_bytes));
    F_HIP(hipMalloc((void **)&plan.d_images, sizeof(FusedImage) * plan.n_images));
    F_HIP(hipMalloc((void **)&plan.d_geoms, sizeof(FusedGeom) * plan.n_images));
    F_HIP(hipMemcpy(plan.d_geoms, plan.geoms.data(), sizeof(FusedGeom) * plan.n_images, hipMemcpyHostToDevice));
    F_HIP(hipMalloc((void **)&plan.d_ids, sizeof(uint32_t) * plan.n_images));
    F_HIP(hipMemcpy(plan.d_ids, plan.ids.data(), sizeof(uint32_t) * plan.n_images, hipMemcpyHostToDevice));
    F_HIP(hipEventCreateWithFlags(&plan.launched, hipEventDisableTiming));
    F_HIP(hipMalloc((void **)&plan.d_work_main, sizeof(FusedWork) * std::max<size_t>(plan.work_main.size(), 1)));
    F_HIP(hipMemcpy(plan.d_work_main, plan.work_main.data(), sizeof(FusedWork) * plan.work_main.size(), hipMemcpyHostToDevice));
    if (!plan.wg_first.empty()) {
        F_HIP(hipMalloc((void **)&plan.d_wg_first, sizeof(uint32_t) * plan.wg_first.size()));
        F_HIP(hipMemcpy(plan.d_wg_first, plan.wg_first.data(), sizeof(uint32_t) * plan.wg_first.size(), hipMemcpyHostToDevice));
    }
    if (!plan.work_pre.empty()) {
        F_HIP(hipMalloc((void **)&plan.d_work_pre, sizeof(FusedWork) * plan.work_pre.size()));
        F_HIP(hipMemcpy(plan.d_work_pre, plan.work_pre.data(), sizeof(FusedWork) * plan.work_pre.size(), hipMemcpyHostToDevice));
    }
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
        im.scratch = plan.d_scratch ? plan.d_scratch + plan.scratch_off[i] : nullptr;
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
        for (int c = 0; c < 3; c++) {
            plan.n_pre_cls[c] = 0;
            for (const FusedWork &w : plan.work_pre)
                if (cls[w.image] == c) all.push_back(w), plan.n_pre_cls[c]++;
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

// One batch = the chroma pass (two-pass 4:2:0 only) and the main launch.  (Walking the batch in chunks so that a
// chunk's chroma planes stay in the 256 MiB Infinity Cache, and alternating chunks between two streams so that the
// HBM-bound chroma pass overlaps the VALU-bound main pass, were both measured on MI355X and did not pay: chunks of
// 16/32/64/128 images were 23/9/4/1 % slower, two streams 3 % slower — profiles/round1.)
// wg_first (strip walks): the workgroups' shares of the item table W (balanced launches); null: one item per workgroup
static hipError_t fused_launch_one(FusedPlan &plan, hipStream_t stream, int ar, const FusedWork *W, uint32_t n_main, const FusedWork *Wpre,
                                   uint32_t n_pre, const uint32_t *wg_first = nullptr, uint32_t n_wg = 0) {
    const FusedGeom *G = plan.d_geoms;
    const FusedImage *I = plan.d_images;
    const FusedGeom &g0 = plan.geoms[0];
    const dim3 grid = wg_first ? dim3(n_wg) : (W ? dim3(n_main) : dim3(g0.tiles_x, plan.strip ? g0.n_seg : g0.mcu_h, plan.n_images));
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
    // strip walks: one item per workgroup (KERNEL) or balanced shares (ITEMS, with wg_first); ar < 0: the two `_dyn` passes
#define WALK_SWITCH(KERNEL, ITEMS, DYN_FAST, DYN_EXACT, ...)                                                        \
    do {                                                                                                            \
        if (ar < 0) {                                                                                               \
            DYN_FAST<<<grid, block, shm, stream>>>(G, I, W);                                                        \
            DYN_EXACT<<<grid, block, shm, stream>>>(G, I, W);                                                       \
        } else if (wg_first) {                                                                                      \
            if (ar == ARITH_TIGHT) ITEMS<ARITH_TIGHT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W, wg_first);  \
            else if (ar == ARITH_SANE) ITEMS<ARITH_SANE, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W, wg_first); \
            else ITEMS<ARITH_EXACT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W, wg_first);                \
        } else if (ar == ARITH_TIGHT) KERNEL<ARITH_TIGHT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);    \
        else if (ar == ARITH_SANE) KERNEL<ARITH_SANE, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);        \
        else KERNEL<ARITH_EXACT, ##__VA_ARGS__><<<grid, block, shm, stream>>>(G, I, W);                             \
    } while (0)
    switch (plan.kind) {
    case FUSED_420:
        if (plan.strip) {
            WALK_SWITCH(s420_kernel, s420_kernel_items, (s420_kernel_dyn<256, false>), (s420_kernel_dyn<256, true>), 256);
            break;
        }
        if (!Wpre)  // (component, 256-block group, image)
            f420_chroma_kernel<<<dim3(2, (g0.bwc * g0.mcu_h + 255u) / 256u, plan.n_images), dim3(256), 0, stream>>>(G, I, nullptr);
        else
            f420_chroma_kernel<<<dim3(n_pre), dim3(256), 0, stream>>>(G, I, Wpre);
        if (plan.nt == 128) ARITH_SWITCH(f420_main_kernel, f420_main_kernel_dyn<128>, 128);
        else ARITH_SWITCH(f420_main_kernel, f420_main_kernel_dyn<256>, 256);
        break;
    case FUSED_440: WALK_SWITCH(s440_kernel, s440_kernel_items, s440_kernel_dyn<false>, s440_kernel_dyn<true>); break;
    case FUSED_GEN: ARITH_SWITCH(fgen_kernel, fgen_kernel_dyn); break;
    case FUSED_420X4:
        if (ar < 0) {
            if (g0.k_full) {
                r4_kernel_dyn<true, false><<<grid, block, shm, stream>>>(G, I, W);
                r4_kernel_dyn<true, true><<<grid, block, shm, stream>>>(G, I, W);
            } else {
                r4_kernel_dyn<false, false><<<grid, block, shm, stream>>>(G, I, W);
                r4_kernel_dyn<false, true><<<grid, block, shm, stream>>>(G, I, W);
            }
        } else if (g0.k_full) {
            if (ar == ARITH_TIGHT) r4_kernel<ARITH_TIGHT, true><<<grid, block, shm, stream>>>(G, I, W);
            else if (ar == ARITH_SANE) r4_kernel<ARITH_SANE, true><<<grid, block, shm, stream>>>(G, I, W);
            else r4_kernel<ARITH_EXACT, true><<<grid, block, shm, stream>>>(G, I, W);
        } else {
            if (ar == ARITH_TIGHT) r4_kernel<ARITH_TIGHT, false><<<grid, block, shm, stream>>>(G, I, W);
            else if (ar == ARITH_SANE) r4_kernel<ARITH_SANE, false><<<grid, block, shm, stream>>>(G, I, W);
            else r4_kernel<ARITH_EXACT, false><<<grid, block, shm, stream>>>(G, I, W);
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

// One batch = the chroma pass (two-pass 4:2:0 only) and the main launch — per arithmetic class when the images disagree.
// (Walking the batch in chunks so that a chunk's chroma planes stay in the 256 MiB Infinity Cache, and alternating chunks
// between two streams so that the HBM-bound chroma pass overlaps the VALU-bound main pass, were both measured on MI355X and
// did not pay: chunks of 16/32/64/128 images were 23/9/4/1 % slower, two streams 3 % slower — profiles/round1.)
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
        if (e == hipSuccess)  // (a balanced plan: one item per workgroup here — the `_dyn` kernels have no item loop)
            e = fused_launch_one(plan, stream, -1, table ? plan.d_work_main : nullptr, (uint32_t)plan.work_main.size(),
                                 table && !plan.work_pre.empty() ? plan.d_work_pre : nullptr, (uint32_t)plan.work_pre.size());
    } else if (!plan.by_class) {
        e = fused_launch_one(plan, stream, plan.arith, table ? plan.d_work_main : nullptr, (uint32_t)plan.work_main.size(),
                             table && !plan.work_pre.empty() ? plan.d_work_pre : nullptr, (uint32_t)plan.work_pre.size(), plan.d_wg_first,
                             plan.wg_first.empty() ? 0u : (uint32_t)plan.wg_first.size() - 1u);
    } else {  // (per class: one item per workgroup — the shares of a balanced walk were cut over all images)
        const FusedWork *w = plan.d_work_cls, *wp = plan.d_work_cls + plan.n_main_cls[0] + plan.n_main_cls[1] + plan.n_main_cls[2];
        for (int c = 0; c < 3 && e == hipSuccess; c++) {
            if (plan.n_main_cls[c]) e = fused_launch_one(plan, stream, c, w, plan.n_main_cls[c], plan.n_pre_cls[c] ? wp : nullptr, plan.n_pre_cls[c]);
            w += plan.n_main_cls[c];
            wp += plan.n_pre_cls[c];
        }
    }
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
    if (plan.d_scratch) (void)hipFree(plan.d_scratch);
    if (plan.d_images) (void)hipFree(plan.d_images);
    if (plan.d_geoms) (void)hipFree(plan.d_geoms);
    if (plan.d_work_main) (void)hipFree(plan.d_work_main);
    if (plan.d_work_pre) (void)hipFree(plan.d_work_pre);
    if (plan.d_work_cls) (void)hipFree(plan.d_work_cls);
    if (plan.d_wg_first) (void)hipFree(plan.d_wg_first);
    plan.d_wg_first = nullptr;
    if (plan.d_ids) (void)hipFree(plan.d_ids);
    if (plan.launched) (void)hipEventDestroy(plan.launched);
    plan.d_ids = nullptr;
    plan.launched = nullptr;
    plan.launch_pending = false;
    plan.d_work_cls = nullptr;
    plan.work_cls_cap = 0;
    plan.d_scratch = nullptr;
    plan.d_images = nullptr;
    plan.d_geoms = nullptr;
    plan.d_work_main = plan.d_work_pre = nullptr;
}

}  // namespace jpgpu

#ifdef JPGPU_PHASE_CLOCKS
extern "C" int jpgpu_debug_phase_clocks(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(jpgpu::g_phase_clocks), sizeof(unsigned long long) * 16) != hipSuccess) return 4;
    if (reset) {
        unsigned long long zero[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(jpgpu::g_phase_clocks), zero, sizeof(zero)) != hipSuccess) return 4;
    }
    return 0;
}
#endif
