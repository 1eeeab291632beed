// jobs.hpp — device job descriptors of the generic path (no HIP dependency: tests/emu compiles it with g++).
#pragma once
#include <stdint.h>

namespace jpgpu {

// One component plane to (de)quantize + IDCT: row a3 of SURVEY §8a
// (src/worker/rayon.rs:71-112 for every block of the appended MCU rows).
struct PlaneJob {
    const int16_t *coefs;  // block-raster, 64 natural-order i16 per block, 16-B aligned
    uint8_t *plane;        // stride = block_w * scale
    const uint16_t *qt;    // 64 u16, natural order (device memory)
    uint32_t block_w;
    uint32_t n_blocks;     // blocks to transform (appended MCU rows * block_w * v)
    uint32_t scale;        // dct_scale: 8, 4, 2, 1
    uint32_t flags;        // bit0: coefficients proven "sane" (|c*q| < 2^15) -> 24-bit multiply path allowed
};

enum UpKind : uint32_t { UP_H1V1 = 0, UP_H2V1 = 1, UP_H1V2 = 2, UP_H2V2 = 3, UP_GENERIC = 4 };
enum ColorFn : uint32_t { CC_NONE = 0, CC_RGB = 1, CC_YCBCR = 2, CC_CMYK = 3, CC_YCCK = 4, CC_GRAY = 5 };

struct UpComp {
    const uint8_t *plane;
    uint32_t kind;    // UpKind, src/upsampler.rs:76-105
    uint32_t hf, vf;  // Generic scaling factors
    uint32_t width, height;  // component.size
    uint32_t stride;         // block_w * dct_scale
};

// One image to upsample + colour-convert: rows a8-a15 of SURVEY §8a.
struct ImageJob {
    UpComp comp[4];
    uint8_t *out;
    uint32_t ncomp;
    uint32_t out_w, out_h;
    uint32_t color_fn;  // ColorFn
    uint32_t fast8;     // every component at dct_scale 8 with one of the four fixed upsamplers and an 8-B aligned plane:
                        // the kernel may work on aligned dwords, 8 pixels per lane (set by build_image_job)
};

// Fused 4:2:0 / 4:4:4 / gray fast paths (same-geometry batches); see fused.hip.
struct FusedJob {
    const int16_t *coefs[4];
    const uint16_t *qt[4];
    uint8_t *chroma[2];  // 4:2:0 two-stage path: Cb / Cr planes (scratch)
    uint8_t *out;
};

}  // namespace jpgpu
