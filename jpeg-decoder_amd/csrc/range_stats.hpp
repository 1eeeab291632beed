// range_stats.hpp — the range statistics behind the arithmetic classes of include/jpgpu.h, as they are kept
// ON THE DEVICE: four words per image, raised with atomicMax by whoever writes or reads the coefficients there —
//   * the device entropy decoder's expansion (huff_expand_kernel) and huff_dc_prefix_kernel,
//   * expand_compact_kernel (compact transport with an unknown class),
//   * range_scan_kernel (jpgpu_batch_classify_on_device: coefficients a caller's own kernels put into a bound arena),
// and turned into class bits by class_finalize_* right in front of the pixel kernels, without the host looking at them
// (round 2 read them back: a host synchronisation between entropy decoding and the pixel kernels, and a second pass over
// the arena).  No HIP dependency.
//
// Class rule (s = coefficient * quantization value, per image over all its components):
//   class >= 1 ("sane"):  every |s| < 2^15;
//   class 3 ("tight"):    additionally every block column's sum of |s| <= 5900.
// Writers that see one coefficient at a time (the entropy decoder's lanes walk chunks of the bit stream: a block may straddle
// lanes) do not build column sums; they keep the largest |s| among the DC coefficients and among the AC coefficients.  Column 0
// of a block holds the DC coefficient and seven AC coefficients, every other column eight AC coefficients, so
//   max column sum <= max(max_dc + 7 * max_ac, 8 * max_ac)
// is a sound bound (legal 8-bit data: |DC * q| <= 1024, so class 3 is granted up to max_ac = 696; images with harder edges
// than that run class 1, measured 1.5 % slower).  range_scan_kernel has whole blocks in front of it and stores the exact
// column maximum instead (RS_COL_EXACT set); every other writer clears that word again (stat_mark_inexact), and the bound
// then also covers what the scan had seen: it leaves max |s| of ALL coefficients in RS_MAX_AC, and 8 * max_ac bounds any column.
#pragma once
#include <stdint.h>

namespace jpgpu {

enum : uint32_t { RS_MAX_DC = 0, RS_MAX_AC = 1, RS_MAX_COL = 2, RS_COL_EXACT = 3, RS_WORDS = 4 };
constexpr uint8_t CLS_FROM_DEVICE = 0xffu;  // host-side class table entry: "look at the device statistics"
constexpr uint8_t CLS_SKIP = 0xfeu;         // ...: "the entry-list walk makes this image's pixels, leave it alone" (fused_entries.hpp)

inline
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t
    range_class_from_stats(uint32_t max_dc, uint32_t max_ac, uint32_t max_col, uint32_t col_exact) {
    const uint32_t max_abs = max_dc > max_ac ? max_dc : max_ac;
    if (max_abs >= (1u << 15)) return 0u;
    uint32_t bound = max_col;
    if (!col_exact) {
        const uint32_t a = max_dc + 7u * max_ac, b = 8u * max_ac;  // (< 2^19: no overflow)
        bound = a > b ? a : b;
    }
    return bound <= 5900u ? 3u : 1u;
}

#if defined(__HIPCC__) && !defined(JPGPU_HOST_EMULATION)
// Raise a statistics word to `v`: most workgroups of a launch bring nothing new, and a plain look first keeps them from
// queueing up at one L2 line with their atomics.
__device__ __forceinline__ void stat_raise(uint32_t *p, uint32_t v) {
    if (v && __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(p, v);
}
// A writer that does not keep block-column sums has touched the image's coefficients: whatever exact column maximum an earlier
// range scan stored (RS_COL_EXACT) no longer describes them — a value added to a column raises its sum without raising any
// per-coefficient maximum — so the class falls back to the bound from max_dc / max_ac until the next scan.
// A look first: most waves of a launch find the word cleared already.
__device__ __forceinline__ void stat_mark_inexact(uint32_t *stats) {
    if (__hip_atomic_load(stats + RS_COL_EXACT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        __hip_atomic_store(stats + RS_COL_EXACT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// per-lane maxima -> the image's statistics: one pair of atomics per wave at most.  Every lane of the wave calls it.
// (the by-product writers' publisher: column sums are unknown to them, stat_mark_inexact)
__device__ __forceinline__ void stat_publish_wave(uint32_t *stats, uint32_t max_dc, uint32_t max_ac) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_dc = max(max_dc, (uint32_t)__shfl_xor((int)max_dc, off));
        max_ac = max(max_ac, (uint32_t)__shfl_xor((int)max_ac, off));
    }
    if ((threadIdx.x & 63u) == 0u && stats) {
        stat_mark_inexact(stats);
        stat_raise(stats + RS_MAX_DC, max_dc);
        stat_raise(stats + RS_MAX_AC, max_ac);
    }
}
#endif

}  // namespace jpgpu
