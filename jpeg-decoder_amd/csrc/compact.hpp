// compact.hpp — compact coefficient transport: what crosses PCIe for one component is
//     [ n_blocks x u64 bitmap | n_blocks x u32 first-value index | nnz x i16 values ]
// instead of n_blocks x 64 x i16.  Bit k of a block's bitmap = natural-order coefficient k is non-zero; its values
// follow in ascending k; the index is the block's position in the value array.  Photographic content at usual
// qualities has 5-15 non-zero coefficients per block: 22-42 B per block instead of 128.
// No HIP dependency (host encoder + layout); the expand kernel is in kernels.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace jpgpu {

struct ExpandJob {
    const uint8_t *compact;  // device: layout above, 8-B aligned
    int16_t *dense;          // device: n_blocks * 64 i16 (the coefficient arena)
    uint32_t n_blocks;
    uint32_t _pad;
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
    bool have_q;
    CompactWriter(void *dst, size_t n_blocks_, const uint16_t *q) : n_blocks(n_blocks_), have_q(q != nullptr) {
        uint8_t *out = static_cast<uint8_t *>(dst);
        bitmaps = reinterpret_cast<uint64_t *>(out);
        first = reinterpret_cast<uint32_t *>(out + n_blocks * 8u);
        values = reinterpret_cast<int16_t *>(out + n_blocks * 12u);
        for (int k = 0; k < 64; k++) qq[k] = q ? q[k] : 0;
    }
    void add_blocks(const int16_t *coefficients, size_t count) {
        if (count > n_blocks - done) count = n_blocks - done;  // rows past the plane are dropped
        for (size_t b = 0; b < count; b++) {
            const int16_t *p = coefficients + b * 64;
            uint64_t bm = 0;
            for (int k = 0; k < 64; k++) bm |= (uint64_t)(p[k] != 0) << k;  // (vectorises)
            bitmaps[done] = bm;
            first[done] = n_values;
            done++;
            int32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint64_t m = bm; m; m &= m - 1) {
                const int k = __builtin_ctzll(m);
                values[n_values++] = p[k];
                int32_t v = (int32_t)p[k] * qq[k];
                v = v < 0 ? -v : v;
                max_abs = v > max_abs ? v : max_abs;
                col[k & 7] += v;
            }
            for (int i = 0; i < 8; i++) max_col = col[i] > max_col ? col[i] : max_col;
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
