// huff_core.hpp — bit reader and Huffman symbol lookup of the device entropy decoder (huff_sync_core.hpp: the self-synchronising
// chunk decoder; restart segments — SURVEY §8f n1 "DRI segments are independently decodable", src/decoder.rs:920-956 — are scans
// in miniature with chunk slots of their own).  The decoding procedure is the reference's (src/huffman.rs:31-96) on the same wide tables
// the host front-end uses (csrc/host/frontend.cpp: an exact cache of the 8-bit LUT + maxcode walk).  Compiled by hipcc for
// the kernels (huff.hip) and by g++ for tests/emu.
//
// Shape of the code.  The lanes of a wave walk unrelated bit streams: whatever any lane does, the wave executes, so the
// run time is the number of instructions on the UNION of the lanes' paths (and with one or two waves per SIMD every
// dependent instruction costs most of its latency).  Hence:
//   * the host removes the 0xFF00 stuffing and aligns the data (huff_stage_segment): the bit reader appends one aligned
//     dword when it holds <= 32 bits — no byte loop, no marker logic on the device;
//   * decode_block is unrolled into ONE step per Huffman symbol (k == 0: the DC symbol of the next block; k >= 1: an AC
//     symbol); what a symbol means comes from a small LDS table (huff_sym_info), only the table miss, the refill and
//     the end of a block are branches;
//   * per-step memory operations are LDS reads through address-space-3 pointers (generic pointers made them flat_*
//     operations at several hundred cycles each) and one 2-byte store per non-zero coefficient.
// History of the restart-segment decoder (MI355X, 68 one-MCU-row segments per 1080p image): block-structured decoder with
// a byte-wise reader 22 ms for 64 images — the same 22 ms for 256 (latency-bound); select-based symbol step 13.4 ms for
// 1,024 images; the table-driven step shared with the chunk decoder: see DESIGN.md §5.
#pragma once
#include "huff_job.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

// How the stream is fetched: one aligned dword per refill — from the lane's column of the weave (huff_job.hpp: the 64 chunks of a
// wave side by side), requested one refill AHEAD into the register the previous one just left (the empty asm keeps the load behind
// the last use of the old value, so no copy and no wait until the next refill).
// Rounds 1-3 also carried a reader of 16-byte pieces (sync passes of 256 1080p images 2.53 ms against 2.14 with dwords) and an LDS
// ring for the kernels that stored coefficients themselves (the write pass and the one-lane-per-restart-segment decoder, both
// replaced by speculative emission + huff_expand_kernel and deleted in round 4: profiles/round3/14_emission_path.txt).
struct DevBits {
    uint64_t bits;   // unread bits, left-aligned
    uint32_t nbits;
    uint32_t wpos;   // dwords taken from the slot so far
    const JP_GLOBAL uint32_t *g;  // dword w of the scan is g[w * HUFF_WEAVE_LANES]: the lane's column of the weave (huff_job.hpp), moved back
                                  // by the chunk's first dword.  An address-space-1 pointer: through a generic one the fetches are
                                  // flat_load instructions, which count as LDS operations too — every wait for an LDS read behind one
                                  // (and the loop is full of them) then waits for the stream fetch as well
    uint32_t ahead;  // dword wpos
    bool bad;
};

// at most once per step: afterwards more than 32 bits are available (a step reads <= 16 + 15)
__device__ __forceinline__ void huff_refill(DevBits &b) {
    if (b.nbits <= 32u) {
        b.bits |= (uint64_t)__builtin_bswap32(b.ahead) << (32u - b.nbits);
        b.nbits += 32u;
        b.wpos++;
#ifndef JPGPU_HOST_EMULATION
        asm volatile("" : "+v"(b.bits) : : "memory");  // the old `ahead` is dead from here on
#endif
        b.ahead = b.g[(size_t)b.wpos * HUFF_WEAVE_LANES];
    }
}
__device__ __forceinline__ uint32_t huff_peek(const DevBits &b, uint32_t n) { return n ? (uint32_t)(b.bits >> (64u - n)) : 0u; }
__device__ __forceinline__ void huff_consume(DevBits &b, uint32_t n) {
    b.bits <<= n;
    b.nbits -= n;
}
__device__ __forceinline__ int32_t huff_extend(uint32_t v, uint32_t n) {  // src/huffman.rs:98-101 (n == 0 -> 0)
    const int32_t vt = n ? 1 << (n - 1u) : 0;
    return (int32_t)v < vt ? (int32_t)v + (int32_t)(0xffffffffu << n) + 1 : (int32_t)v;
}

__device__ __forceinline__ void atomicOr_status(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    atomicOr(p, v);
#endif
}
// state words other lanes (possibly of other workgroups, in the same launch) read while we run: straight to/from L2
__device__ __forceinline__ uint32_t huff_load_shared(const uint32_t *p) {
#ifdef JPGPU_HOST_EMULATION
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void huff_store_shared(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// The slow tail of src/huffman.rs:31-58 for prefixes the wide table does not resolve: the maxcode walk from length 9 —
// first length whose code does not exceed that length's largest code.  All eight comparisons at once (two 16-byte LDS
// reads) and a find-first-set instead of a loop: with 64 lanes per wave some lane nearly always needs the tail, and as a
// loop it cost every lane up to eight dependent LDS round trips per symbol.
__device__ __forceinline__ uint32_t huff_walk(DevBits &b, const JP_LDS DevHuffTable &t) {
    const uint32_t b16 = huff_peek(b, 16);
    const JP_LDS v4u *mc = (const JP_LDS v4u *)&t.maxcode[8];
    const v4u m0 = mc[0], m1 = mc[1];
    const int32_t m[8] = {(int32_t)m0.x, (int32_t)m0.y, (int32_t)m0.z, (int32_t)m0.w, (int32_t)m1.x, (int32_t)m1.y, (int32_t)m1.z, (int32_t)m1.w};
    uint32_t hits = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) hits |= ((int32_t)(b16 >> (7 - j)) <= m[j] ? 1u : 0u) << j;
    if (hits == 0u) {
        b.bad = true;
        return 0u;
    }
    const uint32_t i = 8u + (uint32_t)__builtin_ctz(hits);
    const int32_t code = (int32_t)(b16 >> (15u - i));
    huff_consume(b, i + 1u);
    const int32_t index = code + t.delta[i];
    if (index < 0 || index >= t.nvalues) {
        b.bad = true;
        return 0u;
    }
    return t.values[index];
}

// zig-zag -> natural order (src/decoder.rs:27-36), written to LDS once per workgroup
__device__ __forceinline__ void huff_fill_unzigzag(JP_LDS uint8_t *dst, uint32_t lane) {
    static const uint8_t unzig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                      15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    if (lane < 64u) dst[lane] = unzig[lane];
}

}  // namespace jpgpu
