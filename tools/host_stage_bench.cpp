// Host staging of entropy-coded data (huff_stage_segment: 0xFF00 -> 0xFF, copy into the slot), portable form against the AVX2 form:
//   g++ -O3 -std=c++17 -o /tmp/host_stage_bench tools/host_stage_bench.cpp && /tmp/host_stage_bench
// Input: pseudo-random bytes with the 0xFF density of entropy-coded data (1 in 256, each followed by its stuffing zero), 391 kB per "file"
// (the bench's 1080p input), many files so that nothing stays in the caches.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jpeg-decoder_amd/csrc/huff_job.hpp"

using namespace jpgpu;

int main() {
    const uint32_t n = 391 * 1024, files = 512;
    std::vector<uint8_t> src((size_t)n * files), dst((size_t)huff_slot_bytes(n) * files);
    uint32_t x = 12345;
    for (size_t i = 0; i < src.size(); i++) {
        x = x * 1664525u + 1013904223u;
        src[i] = (uint8_t)(x >> 24);
        if (i && src[i - 1] == 0xFF) src[i] = 0;
    }
    for (int form = 0; form < 2; form++) {
        double best = 1e9;
        uint64_t sum = 0;
        for (int rep = 0; rep < 5; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t f = 0; f < files; f++) {
                uint8_t *d = dst.data() + (size_t)f * huff_slot_bytes(n);
                const uint8_t *s = src.data() + (size_t)f * n;
                uint32_t o;
                if (form == 0) {
                    o = huff_unstuff_portable(d, s, n, 0, 0, nullptr);
                    memset(d + o, 0, huff_slot_bytes(n) - o);
                } else {
                    o = huff_stage_segment(d, s, n);
                }
                sum += o;
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        printf("%s: %.1f us per 391 kB file, %.2f GB/s per thread (checksum %llu)\n", form ? "huff_stage_segment (AVX2 where the host has it)" : "portable (memchr + memcpy runs)",
               best * 1e3 / files, (double)n * files / best / 1e6, (unsigned long long)sum);
    }
    return 0;
}
