// clock_probe.hip — the shader clock while the 4:2:0 kernel runs: a one-wave kernel on a second stream reads s_memtime
// (shader clock) and s_memrealtime (100 MHz) side by side while jpgpu_batch_decode launches run back to back on the first.
// For comparison: the same while the device is otherwise idle and while the pure-VALU loop of tools/ubench_idct.hip runs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/clock_probe.bin tools/clock_probe.hip -L jpeg-decoder_amd -ljpgpu -Wl,-rpath,'$ORIGIN/../jpeg-decoder_amd'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "jpgpu.h"

__global__ void probe(uint64_t *out, uint64_t ticks_100mhz) {
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    uint64_t r = r0;
    while (r - r0 < ticks_100mhz) {
        __builtin_amdgcn_s_sleep(64);
        r = wall_clock64();
    }
    const uint64_t c1 = clock64();
    out[0] = r - r0, out[1] = c1 - c0;
}

__global__ __launch_bounds__(256) void valu_burn(uint32_t *out, uint32_t iters) {
    uint32_t a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 977u + i;
    for (uint32_t it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] * 5u + a[(i + 1) & 15];
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double run_probe(hipStream_t s2, uint64_t *d_out, double seconds) {
    probe<<<1, 64, 0, s2>>>(d_out, (uint64_t)(seconds * 1e8));
    (void)hipStreamSynchronize(s2);
    uint64_t h[2];
    (void)hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    return (double)h[1] / (double)h[0] * 100.0;  // MHz
}

int main() {
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1);
    (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1);
    uint64_t *d_out;
    (void)hipMalloc(&d_out, 16);
    printf("idle device:            s_memtime runs at %.0f MHz\n", run_probe(s2, d_out, 0.05));

    // 256 x 1920x1080 4:2:0, all-zero coefficients (class 3: the arithmetic does not depend on the data)
    const uint32_t n = 256;
    jpgpu_image_desc d;
    memset(&d, 0, sizeof(d));
    d.ncomp = 3, d.out_w = 1920, d.out_h = 1080, d.color_transform = JPGPU_CT_YCBCR;
    const uint32_t mcu_w = (1920 + 15) / 16, mcu_h = (1080 + 15) / 16;
    for (int c = 0; c < 3; c++) {
        jpgpu_component &k = d.components[c];
        k.identifier = (uint8_t)(c + 1), k.horizontal_sampling_factor = k.vertical_sampling_factor = c ? 1 : 2, k.dct_scale = 8;
        k.block_width = (uint16_t)(mcu_w * k.horizontal_sampling_factor), k.block_height = (uint16_t)(mcu_h * k.vertical_sampling_factor);
        k.size_width = (uint16_t)(c ? 960 : 1920), k.size_height = (uint16_t)(c ? 540 : 1080);
        for (int i = 0; i < 64; i++) d.quantization_tables[c][i] = 2;
    }
    std::vector<jpgpu_image_desc> descs(n, d);
    jpgpu_batch *b = nullptr;
    if (jpgpu_batch_create(0, descs.data(), n, JPGPU_BATCH_DEFAULT, &b)) { printf("batch_create failed\n"); return 1; }
    (void)hipMemset(jpgpu_batch_coef_arena(b), 0, jpgpu_batch_coef_arena_bytes(b));
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < 3; c++) jpgpu_batch_set_range_class(b, i, c, 3);
    float ms = 0;
    jpgpu_batch_time(b, s1, 50, &ms);
    printf("path %s, %.4f ms per launch (50 launches, nothing else running)\n", jpgpu_batch_path(b), ms);
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 1200; i++) jpgpu_batch_decode(b, s1);  // ~0.8 s of back-to-back launches
        const double mhz = run_probe(s2, d_out, 0.3);
        (void)hipStreamSynchronize(s1);
        printf("under the 4:2:0 kernel: s_memtime runs at %.0f MHz\n", mhz);
    }
    uint32_t *d_burn;
    (void)hipMalloc(&d_burn, 2048 * 256 * 4);
    for (int rep = 0; rep < 2; rep++) {
        for (int i = 0; i < 40; i++) valu_burn<<<2048, 256, 0, s1>>>(d_burn, 200000);
        const double mhz = run_probe(s2, d_out, 0.3);
        (void)hipStreamSynchronize(s1);
        printf("under a pure VALU loop:  s_memtime runs at %.0f MHz\n", mhz);
    }
    jpgpu_batch_destroy(b);
    return 0;
}
