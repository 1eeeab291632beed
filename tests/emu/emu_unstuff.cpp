// emu_unstuff.cpp — TEST-ONLY CPU run of the device's staging pass ("host light", csrc/huff_unstuff_core.hpp): the per-16-bytes rule of
// huff_unstuff_count_kernel / huff_unstuff_compact_kernel applied piece by piece in order, against the host's huff_stage_segment.
#include <cstring>
#include <vector>
#include "hip_shim.hpp"
#include "../../jpeg-decoder_amd/csrc/huff_unstuff_core.hpp"

using namespace jpgpu;

extern "C" {
// raw: the scan's bytes as they stand in `buf` at offset `lead` (any alignment relative to the 16-byte pieces of buf); dst: room for n
// bytes.  Returns the unstuffed length, or -1 if the rule refuses the scan (a 0xFF not followed by its stuffing zero).
int emu_unstuff(const uint8_t *buf, uint32_t lead, uint32_t n, uint8_t *dst) {
    const uint32_t total = lead + n;
    uint32_t o = 0;
    bool bad = false;
    for (uint32_t lo = 0; lo < total; lo += 16u) {
        uint32_t w[4];
        memcpy(w, buf + lo, 16);  // (the caller pads buf to whole pieces)
        uint32_t prev = 0, next = 0;
        if (lo > lead) prev = buf[lo - 1u];
        const bool has_next = lo + 16u < total;
        if (has_next) next = buf[lo + 16u];
        const uint32_t first = lo >= lead ? 0u : std::min(16u, lead - lo), last = std::min(16u, total - lo);
        uint32_t keep = unstuff_piece_flags(w, prev, next, has_next, first, last, bad);
        while (keep) {
            const uint32_t j = (uint32_t)__builtin_ctz(keep);
            keep &= keep - 1u;
            dst[o++] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
        }
    }
    return bad ? -1 : (int)o;
}
}
