// hip_shim.hpp — TEST-ONLY host shim so that the product's device headers (pixel_math.hpp,
// fused_core.hpp) can be compiled by g++ and exercised on the CPU by tests/ (kernel-logic
// emulation against the oracle before spending GPU minutes).  Never part of the product.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
#define __restrict__
#define JPGPU_HOST_EMULATION 1
using std::max;
using std::min;
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
