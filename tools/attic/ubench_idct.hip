// Micro-benchmark: the 8x8 IDCT (class 3) and the 4:2:0 pixel arithmetic as pure register loops — no LDS, no global
// memory in the loop — at 1..8 waves per SIMD: cycles per block / per 16 pixels at the issue limit, to be compared with the
// sums of tools/ubench_valu2.hip's per-instruction costs and with what the kernels achieve.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I jpeg-decoder_amd/csrc -o tools/ubench_idct.bin tools/ubench_idct.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fused_core.hpp"
using namespace jpgpu;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t iters, uint32_t seed) {
    uint32_t d[32], o[16];
    const uint32_t t = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 32; i++) d[i] = ((t >> (i & 15)) & 0x003f003fu) + (uint32_t)i;  // small products: class 3
    uint32_t acc = 0;
    if (MODE == 0) {
        for (uint32_t it = 0; it < iters; it++) {
            idct8x8_products<ARITH_TIGHT>(d, o);
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= o[i];
#pragma unroll
            for (int i = 0; i < 32; i++) d[i] = (d[i] + (acc & 0x00010001u)) & 0x007f007fu;  // next block depends on this one (no hoisting)
        }
    } else {
        typedef F420<ARITH_TIGHT, 256> P;
        v2u yy = {t, t ^ 0x5a5a5a5au};
        uint32_t c0 = t & 0x00ff00ffu, c1 = (t >> 3) & 0x00ff00ffu;
        for (uint32_t it = 0; it < iters; it++) {
            // one unit of the pixel phase: two chroma rows x two components unpacked, two rows of 8 pixels produced
            typename P::ChromaEO eu[2], el[2];
#pragma unroll
            for (uint32_t comp = 0; comp < 2; comp++) {
                eu[comp].E1 = c0 + comp, eu[comp].O1 = c1 + comp, eu[comp].Om = c0 ^ 0x00110011u, eu[comp].Ep = c1 ^ 0x00220022u;
                el[comp].E1 = c1 + comp, el[comp].O1 = c0 + comp, el[comp].Om = c1 ^ 0x00110011u, el[comp].Ep = c0 ^ 0x00220022u;
            }
            uint32_t pk[2][2][6];
#pragma unroll
            for (int row = 0; row < 2; row++) {
                const typename P::TPrime tp[2] = {row ? P::tprime(el[0], eu[0]) : P::tprime(eu[0], el[0]), row ? P::tprime(el[1], eu[1]) : P::tprime(eu[1], el[1])};
                // row_pixels' arithmetic without its store
                uint32_t m[2][4];
#pragma unroll
                for (uint32_t comp = 0; comp < 2; comp++) {
                    const typename P::TPrime &q = tp[comp];
                    m[comp][0] = pk_mad3(q.tE1, q.tOm), m[comp][1] = pk_mad3(q.tE1, q.tO1), m[comp][2] = pk_mad3(q.tO1, q.tE1), m[comp][3] = pk_mad3(q.tO1, q.tEp);
                }
                RawRgb p[8];
                const w32 yb[8] = {byte_shl20<0>(yy.x), byte_shl20<1>(yy.x), byte_shl20<2>(yy.x), byte_shl20<3>(yy.x),
                                   byte_shl20<0>(yy.y), byte_shl20<1>(yy.y), byte_shl20<2>(yy.y), byte_shl20<3>(yy.y)};
#pragma unroll
                for (uint32_t kk = 0; kk < 8; kk++) {
                    const int32_t cb = (kk < 4) ? ((int32_t)(m[0][kk & 3u] << 16) >> 20) : ((int32_t)m[0][kk & 3u] >> 20);
                    const int32_t cr = (kk < 4) ? ((int32_t)(m[1][kk & 3u] << 16) >> 20) : ((int32_t)m[1][kk & 3u] >> 20);
                    p[kk] = ycbcr_raw_centred(yb[kk], cb, cr);
                }
                rgb4_to_12bytes(p[0], p[1], p[2], p[3], pk[row][0][0], pk[row][0][1], pk[row][0][2]);
                rgb4_to_12bytes(p[4], p[5], p[6], p[7], pk[row][0][3], pk[row][0][4], pk[row][0][5]);
#pragma unroll
                for (int i = 0; i < 6; i++) acc ^= pk[row][0][i];
            }
            c0 = (c0 + (acc & 0x00010001u)) & 0x00ff00ffu;
            c1 = (c1 ^ (acc & 0x00030003u)) & 0x00ff00ffu;
            yy.x += acc & 1u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char *name, uint32_t *d, int wg_per_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const uint32_t iters = MODE == 0 ? 400 : 800;
    const int blocks = 256 * wg_per_cu;
    k<MODE><<<blocks, 256>>>(d, iters, 1);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, iters, 2 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-units per SIMD: blocks * 4 waves * iters / (256 CUs * 4 SIMDs)
    const double units_per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("%-34s %d waves/SIMD  %8.3f ms  %8.1f ns per %s per SIMD (= %.0f cycles @2.3 GHz)\n", name, wg_per_cu, best,
           best * 1e6 / units_per_simd, MODE == 0 ? "block" : "16 px", best * 1e6 / units_per_simd * 2.3);
}

int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 8}) run<0>("IDCT class 3 (products -> samples)", d, w);
    for (int w : {1, 2, 3, 4, 8}) run<1>("4:2:0 pixel unit (2 rows x 8 px)", d, w);
    return 0;
}
