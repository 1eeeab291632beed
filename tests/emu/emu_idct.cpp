// Host build of the product's device math (TEST ONLY; see hip_shim.hpp).
#include "hip_shim.hpp"
#include "../../jpeg-decoder_amd/csrc/pixel_math.hpp"
extern "C" {
void emu_idct8x8(int sane, const int16_t* c, const uint16_t* q, uint8_t* out, int nblocks) {
    for (int b = 0; b < nblocks; b++) {
        uint32_t cw[32]; memcpy(cw, c + b * 64, 128);
        uint32_t o[16];
        if (sane == 2) jpgpu::idct8x8<jpgpu::ARITH_TIGHT>(cw, jpgpu::as_qtab(q), o); else if (sane) jpgpu::idct8x8<jpgpu::ARITH_SANE>(cw, jpgpu::as_qtab(q), o); else jpgpu::idct8x8<jpgpu::ARITH_EXACT>(cw, jpgpu::as_qtab(q), o);
        memcpy(out + b * 64, o, 64);
    }
}
void emu_idct_small(int scale, const int16_t* c, const uint16_t* q, uint8_t* out, int nblocks) {
    for (int b = 0; b < nblocks; b++) {
        uint32_t cw[32]; memcpy(cw, c + b * 64, 128);
        if (scale == 4) { uint32_t o[4]; jpgpu::idct4x4_exact(cw, jpgpu::as_qtab(q), o); memcpy(out + b * 16, o, 16); }
        else if (scale == 2) { uint32_t o = jpgpu::idct2x2_exact(cw, jpgpu::as_qtab(q)); memcpy(out + b * 4, &o, 4); }
        else { out[b] = (uint8_t)jpgpu::idct1x1_exact(cw[0], jpgpu::as_qtab(q)); }
    }
}
uint32_t emu_ycbcr(uint32_t y, uint32_t cb, uint32_t cr) { return jpgpu::ycbcr_to_rgb24(y, cb, cr); }
// every (y, cb, cr): the form the fused 4:2:0 / 4:2:2 passes use (chroma with its 128 already taken off, one rounding term)
// against src/decoder.rs:1486-1508 written out; returns the number of inputs that differ
uint32_t emu_ycbcr_centred_mismatches(void) {
    uint32_t bad = 0;
    for (int y = 0; y < 256; y++)
        for (int cb = 0; cb < 256; cb++)
            for (int cr = 0; cr < 256; cr++) {
                const jpgpu::RawRgb p = jpgpu::ycbcr_raw_centred((uint32_t)y << 20, cb - 128, cr - 128);
                const uint32_t got = jpgpu::sar_sat_u8x2(p.r, p.g, 20) | (jpgpu::sar_sat_u8x2(p.b, 0u, 20) << 16);
                const int64_t Y = (int64_t)y * (1 << 20) + (1 << 19);
                int64_t r = (Y + 1470104ll * (cr - 128)) >> 20, g = (Y - 360857ll * (cb - 128) - 748830ll * (cr - 128)) >> 20, b = (Y + 1858077ll * (cb - 128)) >> 20;
                r = r < 0 ? 0 : r > 255 ? 255 : r; g = g < 0 ? 0 : g > 255 ? 255 : g; b = b < 0 ? 0 : b > 255 ? 255 : b;
                bad += got != (uint32_t)(r | (g << 8) | (b << 16));
            }
    return bad;
}
}
