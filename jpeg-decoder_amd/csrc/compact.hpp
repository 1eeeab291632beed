// compact.hpp — compact coefficient transport: what crosses PCIe for one component is
//     [ n_blocks x u64 bitmap | n_blocks x u32 first-value index | nnz x i16 values ]
// instead of n_blocks x 64 x i16.  Bit k of a block's bitmap = natural-order coefficient k is non-zero; its values
// follow in ascending k; the index is the block's position in the value array.  Photographic content at usual
// qualities has 5-15 non-zero coefficients per block: 22-42 B per block instead of 128.
// No HIP dependency (host encoder + layout); the expand kernel is in kernels.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace jpgpu {

struct ExpandJob {
    const uint8_t *compact;  // device: layout above, 8-B aligned
    int16_t *dense;          // device: n_blocks * 64 i16 (the coefficient arena)
    uint32_t n_blocks;
    uint32_t _pad;
    const uint16_t *qt;      // device: the component's quantization table, natural order (only read when `stats` is set)
    uint32_t *stats;         // device, or null: the image's RangeStats words (range_stats.hpp) — raised on the way when the
                             // sender did not classify the coefficients itself (jpgpu_batch_upload_compact, range_class < 0)
};

// Incremental host encoder: blocks arrive in order (MCU row by MCU row); finish() pads the blocks never delivered
// with empty ones (a plane keeps zeros where no row was appended) and returns the bytes to send.  Also gathers the
// range class of include/jpgpu.h (|c*q| maxima and block-column sums) on the way: only non-zero coefficients matter.
struct CompactWriter {
    uint64_t *bitmaps;
    uint32_t *first;
    int16_t *values;
    size_t n_blocks, done = 0;
    uint32_t n_values = 0;
    int32_t qq[64], max_abs = 0, max_col = 0;
    bool have_q, simd_total = false;
#if defined(__SSE2__)
    __m128i qv[8];
#endif
    CompactWriter(void *dst, size_t n_blocks_, const uint16_t *q) : n_blocks(n_blocks_), have_q(q != nullptr) {
        uint8_t *out = static_cast<uint8_t *>(dst);
        bitmaps = reinterpret_cast<uint64_t *>(out);
        first = reinterpret_cast<uint32_t *>(out + n_blocks * 8u);
        values = reinterpret_cast<int16_t *>(out + n_blocks * 12u);
        for (int k = 0; k < 64; k++) qq[k] = q ? q[k] : 0;
#if defined(__SSE2__)
        if (q) {
            simd_total = true;
            // 8-bit tables only: then 32 pmaddwd pair sums (each <= 2 * 32767 * 255) cannot overflow the 32-bit total, so
            // hostile coefficients can never wrap it into the accepted range; 16-bit tables take the exact path
            for (int k = 0; k < 64; k++) simd_total = simd_total && q[k] <= 255;
            for (int g = 0; g < 8; g++) qv[g] = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + 8 * g));
        }
#endif
    }
    void add_blocks(const int16_t *coefficients, size_t count) {
        if (count > n_blocks - done) count = n_blocks - done;  // rows past the plane are dropped
        for (size_t b = 0; b < count; b++) {
            const int16_t *p = coefficients + b * 64;
            uint64_t bm = 0;
            bool detail = true;  // per-coefficient range bookkeeping needed for this block
#if defined(__SSE2__)
            // bitmap of non-zeros, 16 coefficients per step; and sum of |c|*q over the block: if even that total is
            // within the column-sum limit of class 3 (8-bit tables only, see the constructor), every
            // column sum and every single product is, and the block needs no further range work
            const __m128i zero = _mm_setzero_si128();
            __m128i total = zero;
            for (int g = 0; g < 4; g++) {
                const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16 * g));
                const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16 * g + 8));
                const uint32_t z = (uint32_t)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(a, zero), _mm_cmpeq_epi16(c, zero)));
                bm |= (uint64_t)(~z & 0xffffu) << (16 * g);
                if (simd_total) {
                    // |x| as max(x, -x) with a saturating negate: |-32768| reads 32767, and the exact path below
                    // takes over long before that matters (32767 * 1 already exceeds the limit)
                    const __m128i aa = _mm_max_epi16(a, _mm_subs_epi16(zero, a)), ca = _mm_max_epi16(c, _mm_subs_epi16(zero, c));
                    total = _mm_add_epi32(total, _mm_madd_epi16(aa, qv[2 * g]));
                    total = _mm_add_epi32(total, _mm_madd_epi16(ca, qv[2 * g + 1]));
                }
            }
            if (simd_total) {
                total = _mm_add_epi32(total, _mm_shuffle_epi32(total, 0x4e));
                total = _mm_add_epi32(total, _mm_shuffle_epi32(total, 0xb1));
                const int32_t t = _mm_cvtsi128_si32(total);
                if (t >= 0 && t <= 5900) {
                    detail = false;
                    max_abs = t > max_abs ? t : max_abs;  // upper bounds are enough: both stay inside class 3
                    max_col = t > max_col ? t : max_col;
                }
            }
#else
            for (int k = 0; k < 64; k++) bm |= (uint64_t)(p[k] != 0) << k;
#endif
            bitmaps[done] = bm;
            first[done] = n_values;
            done++;
            if (!detail) {
                for (uint64_t m = bm; m; m &= m - 1) values[n_values++] = p[__builtin_ctzll(m)];
                continue;
            }
            int64_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (64-bit: eight products of up to 2^31 each)
            for (uint64_t m = bm; m; m &= m - 1) {
                const int k = __builtin_ctzll(m);
                values[n_values++] = p[k];
                int32_t v = (int32_t)p[k] * qq[k];
                v = v < 0 ? -v : v;
                max_abs = v > max_abs ? v : max_abs;
                col[k & 7] += v;
            }
            for (int i = 0; i < 8; i++) max_col = col[i] > max_col ? (int32_t)(col[i] > 0x7fffffff ? 0x7fffffff : col[i]) : max_col;
        }
    }
    size_t finish(int *range_class) {
        for (; done < n_blocks; done++) {
            bitmaps[done] = 0;
            first[done] = n_values;
        }
        if (range_class) *range_class = (have_q && max_abs < (1 << 15)) ? ((max_col <= 5900) ? 3 : 1) : 0;
        return n_blocks * 12u + (size_t)n_values * 2u;
    }
};

inline size_t compact_fixed_bytes(size_t n_blocks) { return n_blocks * 12u; }
inline size_t compact_max_bytes(size_t n_blocks) { return n_blocks * (12u + 128u); }

}  // namespace jpgpu
