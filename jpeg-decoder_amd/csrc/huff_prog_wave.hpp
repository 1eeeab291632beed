// huff_prog_wave.hpp — the device decoder for progressive frames, round 6: one WAVE per scan (round 5, huff_prog_core.hpp: one LANE per
// scan).  What the scans of a progressive frame do is restated from src/decoder.rs:1086-1298 (decode_block,
// decode_block_successive_approximation, refine_non_zeroes); the host twin this file is checked against is csrc/host/frontend.cpp.
// Compiled by hipcc for huff_progw_kernel (huff.hip) and by g++ for tests/emu (a wave = arrays of 64).
//
// Why a wave per scan.  A scan is one long dependent walk: the bits of a symbol mean what they mean only once the symbol in front of
// them is decoded, and a refinement scan's bits only given which coefficients of the block are non-zero already.  Round 5 gave every scan a
// lane; a lane's step was ~50 instructions at ~24 cycles each (LDS round trips for the table and the stream window, 64-bit shifts on
// the vector unit, loops over single bits for "skip zrl zeros" and for the correction bits, divergence between the 64 different streams
// of a wave) = 0.5 us per step, 31 ms for the longest scan of benches/tower_progressive.jpg whatever the number of frames.  Here the
// walk is SCALAR code — the wave's program counter, bit position, masks and run lengths live in SGPRs, there is no divergence because
// there is one stream per wave — and the vector registers are its memory and its parallel arm:
//   * the stream: lane i of two registers holds dwords i and i + 1 of a 256-byte window; a look at any bit position is two
//     v_readlane_b32 and one 64-bit scalar shift (no refill state, no LDS, no wait for memory: the next window is requested a window ahead);
//   * the Huffman table: the 8-bit lookup of the scan's table (256 x u16) is 128 dwords = two registers; a symbol is a v_readlane_b32;
//   * lane k is zig-zag position k of the current block: "the (zrl + 1)-th zero coefficient from k on" is one rank computation
//     (v_mbcnt) and a compare whose result mask is an SGPR pair (the reference loops over coefficients: src/decoder.rs:1260-1298); up
//     to 32 correction bits are handed to their coefficients by one more; the block's new values and corrections leave in one store
//     and one atomic instruction per block;
//   * DC refinement scans are a bit per block and nothing else: 64 blocks per step.
// The chain that is left per symbol: window (2 readlanes + shift) -> table (readlane) -> fields -> rank/compare -> popcount -> next.
//
// Anything the reference would answer with an error, and the few places where an INVALID stream makes it do something that depends
// on its table layout (a run that leaves the band, src/decoder.rs:1138-1146 — see decode_block in frontend.cpp), raise the image's
// status word instead: the host decoder, whose behaviour on odd streams is pinned, then decodes that image.
#pragma once
#include "huff_prog_job.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

// ---- a wave's vector registers: per-lane values on the device, arrays of 64 on the CPU (tests/emu) --------------------------------
#ifdef JPGPU_HOST_EMULATION
struct WV32 {
    uint32_t l[64];
};
#define WV_EACH for (uint32_t lane = 0; lane < 64u; lane++)
#define WV(x) ((x).l[lane])
static inline uint32_t wv_readlane(const WV32 &v, uint32_t i) { return v.l[i & 63u]; }
static inline void wv_writelane(WV32 &v, uint32_t i, uint32_t x) {
    if (i < 64u) v.l[i] = x;  // (a lane that does not exist: nobody writes)
}
#define WV_BALLOT(out, cond)                               \
    do {                                                   \
        uint64_t m_ = 0;                                   \
        for (uint32_t lane = 0; lane < 64u; lane++)        \
            if (cond) m_ |= 1ull << lane;                  \
        (out) = m_;                                        \
    } while (0)
static inline uint32_t wv_rank(uint64_t mask, uint32_t lane) { return (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull)); }  // set bits below `lane`
static inline uint32_t wv_uniform(uint32_t x) { return x; }
static inline void wv_sleep() {}
#else
typedef uint32_t WV32;
#define WV_EACH for (uint32_t lane __attribute__((unused)) = threadIdx.x, once_ = 1u; once_; once_ = 0u)
#define WV(x) (x)
__device__ __forceinline__ uint32_t wv_readlane(const WV32 &v, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i); }
__device__ __forceinline__ void wv_writelane(WV32 &v, uint32_t i, uint32_t x) { v = threadIdx.x == i ? x : v; }  // (v_cmp + v_cndmask; this clang has no writelane builtin)
#define WV_BALLOT(out, cond)                          \
    do {                                              \
        const uint32_t lane = threadIdx.x;            \
        (void)lane;                                   \
        (out) = __builtin_amdgcn_ballot_w64(cond);    \
    } while (0)
__device__ __forceinline__ uint32_t wv_rank(uint64_t mask, uint32_t) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wv_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ void wv_sleep() { __builtin_amdgcn_s_sleep(32); }  // ~2k cycles
#endif

__device__ __forceinline__ uint64_t wv_uniform64(uint64_t x) { return ((uint64_t)wv_uniform((uint32_t)(x >> 32)) << 32) | wv_uniform((uint32_t)x); }

// ---- memory operations other waves must see: agent scope (past this XCD's L2 where another XCD may look) ----------------------------
// EVERY access to memory that two waves share — coefficients, masks, progress words, status — is an agent-scope atomic (sc1: stores and
// read-modify-writes go through to the point where the agent's XCDs agree, loads come from there); nothing shared is ever read or
// written by a plain access.  That is why publishing progress needs no cache maintenance: a release FENCE (buffer_wbl2: write this
// XCD's L2 back) and an acquire (buffer_inv: drop its clean lines) exist for plain accesses — with 2,560 waves fencing every 32
// blocks they cost 13 % of the walk at 256 frames and 30 % at 4,096 (profiles/round6); `s_waitcnt vmcnt(0)` in front of the progress
// store (everything this wave has issued is COMPLETE at agent scope) and the consumer's control dependency on the progress word it
// loaded are the order that is needed.  -DPROGW_FENCES: the fences anyway (A/B).
#ifndef PW_SCOPE
#define PW_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ void pw_or32(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    (void)__hip_atomic_fetch_or((JP_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, PW_SCOPE);  // (result unused: the no-return form)
#endif
}
__device__ __forceinline__ void pw_add32(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p += v;
#else
    (void)__hip_atomic_fetch_add((JP_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}
__device__ __forceinline__ void pw_or64(uint64_t *p, uint64_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    (void)__hip_atomic_fetch_or((JP_GLOBAL uint64_t *)p, v, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}
__device__ __forceinline__ void pw_and64(uint64_t *p, uint64_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p &= v;
#else
    (void)__hip_atomic_fetch_and((JP_GLOBAL uint64_t *)p, v, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}
__device__ __forceinline__ uint32_t pw_load32(const uint32_t *p) {
#ifdef JPGPU_HOST_EMULATION
    return *p;
#else
    return __hip_atomic_load((const JP_GLOBAL uint32_t *)p, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}
__device__ __forceinline__ void pw_store16(int16_t *p, int16_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p = v;
#else
    __hip_atomic_store((JP_GLOBAL int16_t *)p, v, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}
__device__ __forceinline__ void pw_flag(uint32_t *status, uint32_t bits) {
    WV_EACH {
        if (lane == 0u) pw_or32(status, bits | PROG_ST_HOST);
    }
}

#ifdef JPGPU_PROG_COUNTERS  // (tests/emu: how many steps of each kind a scan takes)
struct ProgwCounters {
    unsigned long long symbols, slow_symbols, corrections, blocks, windows;
};
extern ProgwCounters g_progw_counters;
extern unsigned long long g_len_hist[17];
#define PROGW_COUNT(what, n) (g_progw_counters.what += (n))
#define PROGW_LEN(len) (g_len_hist[(len) & 15u]++)
#else
#define PROGW_COUNT(what, n) ((void)0)
#define PROGW_LEN(len) ((void)0)
#endif

__device__ __forceinline__ uint32_t pw_unzig(uint32_t k) {
    // natural index of zig-zag position k (src/parser.rs UNZIGZAG)
    constexpr uint8_t unzig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return unzig[k & 63u];
}

// ---- the stream -------------------------------------------------------------------------------------------------------------------
// The unread bits live in an SGPR pair, left-aligned (`win`, `pos` of them valid); they are topped up 32 at a time from `nx`, a dword
// that has been taken out of the register window `w` (lane i: dword base + i of the scan's data, most significant byte first) one
// refill AHEAD — v_readlane's result takes ~20 cycles to reach the scalar unit (tools/ubench_scalar_chain.hip), a refill never waits
// for it.  A window is 2,048 bits; the next one is loaded when this one is used up, and the wave waits for it there (~2 us, 65 times
// in the longest scan of benches/tower_progressive.jpg).  NOT requested a window ahead, as first built: a register with a load
// pending cannot be copied, and the compiler keeps loop-carried values in copies — it put `s_waitcnt vmcnt(0)` on the block loop's
// back edge, which on this architecture is a wait for the atomics of the block before: 1 us per block, 4 of a scan's 10 ms.
// Zeros behind the scan's data (the reference feeds zeros once it has met the marker that ends the scan: src/huffman.rs:123-160).
struct PwBits {
    WV32 w;
    uint64_t win;
    uint32_t pos;
    uint32_t nx;        // the next dword of the stream
    uint32_t dp;        // lane of `w` that holds the dword after nx (64: the next window's first)
    uint32_t base;      // dword index of lane 0 of w
    const uint32_t *src;
    uint32_t n_dwords;  // dwords that hold data or the slot's zero padding
};
__device__ __forceinline__ void pw_window_enter(PwBits &b) {  // the 64 dwords from `base` on
    PROGW_COUNT(windows, 1);
    WV_EACH {
        const uint32_t i = b.base + lane;
        WV(b.w) = i < b.n_dwords ? __builtin_bswap32(((const JP_GLOBAL uint32_t *)b.src)[i]) : 0u;
    }
}
// the refill proper; the caller has checked pos < 32.  Afterwards 32 <= pos < 64.
__device__ __forceinline__ void pw_refill(PwBits &b) {
    b.win |= (uint64_t)b.nx << (32u - b.pos);
    b.pos += 32u;
    if (__builtin_expect(b.dp == 64u, 0)) {
        b.base += 64u;
        pw_window_enter(b);
        b.dp = 0u;
    }
    b.nx = wv_readlane(b.w, b.dp);
    b.dp++;
}
__device__ __forceinline__ void pw_bits_open(PwBits &b, const uint8_t *data, uint32_t n_bytes) {
    b.src = (const uint32_t *)data;                 // (16-byte aligned slots, zero-filled behind the data: huff_stage_segment)
    b.n_dwords = ((n_bytes + 15u) / 16u) * 4u;
    b.base = 0u;
    pw_window_enter(b);
    b.nx = wv_readlane(b.w, 0u);
    b.dp = 1u;
    b.win = 0ull;
    b.pos = 0u;
    pw_refill(b);
}
// at least 32 valid bits from here on (a symbol with its extra bits is at most 16 + 15)
#define PW_NEED32(b)                                      \
    do {                                                  \
        if (__builtin_expect((b).pos < 32u, 0)) pw_refill(b); \
    } while (0)
__device__ __forceinline__ uint32_t pw_look(const PwBits &b) { return (uint32_t)(b.win >> 32); }  // the next 32 bits, the first in bit 31
__device__ __forceinline__ void pw_consume(PwBits &b, uint32_t n) {  // n <= pos, n < 64
    b.win <<= n;
    b.pos -= n;
}
// n (0..31) bits of a look, after `skip` bits of it (skip + n <= 32); 0 for n == 0
__device__ __forceinline__ uint32_t pw_field(uint32_t look, uint32_t skip, uint32_t n) { return ((look << skip) >> 1) >> (31u - n); }
__device__ __forceinline__ int32_t pw_extend(uint32_t v, uint32_t n) {  // src/huffman.rs:165-173 (n >= 1)
    const int32_t vt = 1 << (n - 1u);
    return (int32_t)v < vt ? (int32_t)v + (int32_t)(0xffffffffu << n) + 1 : (int32_t)v;
}

// ---- the table ----------------------------------------------------------------------------------------------------------------------
// A scan's Huffman table as the walk wants it: ONE register, lane i = what the six bits i at the head of the stream mean — a whole
// 32-bit entry with every field the step needs, so that a symbol costs one v_readlane and no extraction (96 % of the symbols of
// benches/tower_progressive.jpg's longest scan have codes of up to six bits).  Longer codes: the 8-bit lookup in memory through the
// scalar cache (the reference's own first step, src/huffman.rs:31-58), then its walk over maxcode (:44-58).
//   entry: bits 0-4 code length (0: not a code of up to six bits) | 5-9 extra bits that follow the code | 10-16 run | 17-18 kind | 19-22 size
constexpr uint32_t PW_KIND_COEF = 0u, PW_KIND_EOB = 1u, PW_KIND_ZRL = 2u, PW_KIND_BAD = 3u;
__device__ __forceinline__ uint32_t pw_entry(uint32_t len, uint32_t extra, uint32_t run, uint32_t kind, uint32_t size) {
    return len | (extra << 5) | (run << 10) | (kind << 17) | (size << 19);
}
__device__ __forceinline__ uint32_t pw_e_len(uint32_t e) { return e & 31u; }
__device__ __forceinline__ uint32_t pw_e_extra(uint32_t e) { return (e >> 5) & 31u; }
__device__ __forceinline__ uint32_t pw_e_run(uint32_t e) { return (e >> 10) & 127u; }
__device__ __forceinline__ uint32_t pw_e_kind(uint32_t e) { return (e >> 17) & 3u; }
__device__ __forceinline__ uint32_t pw_e_size(uint32_t e) { return (e >> 19) & 15u; }
// what symbol `sym` of length `len` means in an AC first scan (mode 0), an AC refinement scan (1), a DC first scan (2)
__device__ __forceinline__ uint32_t pw_entry_of(uint32_t sym, uint32_t len, uint32_t mode) {
    const uint32_t r = sym >> 4, sz = sym & 15u;
    if (mode == 2u) return pw_entry(len, sym > 11u ? 0u : sym, 0u, sym > 11u ? PW_KIND_BAD : PW_KIND_COEF, sym & 15u);  // category = extra bits
    if (sz == 0u) return r == 15u ? pw_entry(len, 0u, mode ? 15u : 16u, PW_KIND_ZRL, 0u) : pw_entry(len, r, mode ? 64u : 0u, PW_KIND_EOB, r);
    if (mode == 1u) return sz == 1u ? pw_entry(len, 1u, r, PW_KIND_COEF, 1u) : pw_entry(len, 0u, 0u, PW_KIND_BAD, sz);  // "unexpected huffman code"
    return pw_entry(len, sz, r, PW_KIND_COEF, sz);
}
struct PwTable {
    WV32 lut6;
    const JP_CONST ProgHuffTable *g;
    uint32_t mode;
};
__device__ __forceinline__ void pw_table_load(PwTable &t, const ProgHuffTable *src, uint32_t mode) {
    t.g = (const JP_CONST ProgHuffTable *)src;
    t.mode = mode;
    WV_EACH {
        uint32_t e = 0u;
        if (src) {
            const uint32_t l = ((const JP_GLOBAL uint16_t *)src)[4u * lane];  // (a code of up to six bits fills all four entries under its prefix)
            if ((l >> 8) && (l >> 8) <= 6u) e = pw_entry_of(l & 0xffu, l >> 8, mode);
        }
        WV(t.lut6) = e;
    }
}
// codes of seven bits and more; -> an entry (kind PW_KIND_BAD with length 0: no such code — "failed to decode huffman code")
__device__ __forceinline__ uint32_t pw_symbol_slow(const PwTable &t, uint32_t look) {
    PROGW_COUNT(slow_symbols, 1);
    const uint32_t idx = look >> 24, e8 = (reinterpret_cast<const JP_CONST uint32_t *>(t.g->lut)[idx >> 1] >> (16u * (idx & 1u))) & 0xffffu;
    if (e8 >> 8) return pw_entry_of(e8 & 0xffu, e8 >> 8, t.mode);
    const uint32_t b16 = look >> 16;  // the walk (src/huffman.rs:44-58)
    for (uint32_t i = 8; i < 16u; i++) {
        const int32_t code = (int32_t)(b16 >> (15u - i));
        if (code <= t.g->maxcode[i]) {
            const int32_t index = code + t.g->delta[i];
            if (index < 0 || index >= t.g->nvalues) break;  // ("reference would panic": the host reports it)
            // (a dword through the scalar cache: a byte load is a vector load, and waiting for a vector load is waiting for every
            // store and atomic the wave has in flight)
            const uint32_t sym = (reinterpret_cast<const JP_CONST uint32_t *>(t.g->values)[index >> 2] >> (8u * ((uint32_t)index & 3u))) & 0xffu;
            return pw_entry_of(sym, i + 1u, t.mode);
        }
    }
    return pw_entry(0u, 0u, 0u, PW_KIND_BAD, 0u);
}
__device__ __forceinline__ uint32_t pw_symbol(const PwTable &t, uint32_t look) {
    PROGW_COUNT(symbols, 1);
    uint32_t e = wv_readlane(t.lut6, look >> 26);
    if (__builtin_expect(pw_e_len(e) == 0u, 0)) e = pw_symbol_slow(t, look);
    return e;
}

// ---- staying behind the scans this one depends on, telling the ones that depend on this one -------------------------------------------
// Progress is counted in blocks of the scan's walk order and published per CHUNK (the unit a wave works in: PROGW_CHUNK blocks of an
// AC scan, 64 of a DC scan); a consumer starts a chunk when its producers have completed the same blocks (or, for a producer that
// walks its blocks in another order, when that one has ended).
constexpr uint32_t PROGW_CHUNK = 32u;
struct PwSync {
    uint32_t *progress;
    const uint32_t *wait[3];
    uint32_t seen[3];
    uint32_t whole;
    uint32_t *status;
};
__device__ __forceinline__ void pw_sync_open(PwSync &y, const JP_GLOBAL ProgScan &s, uint32_t *status) {
    y.progress = s.progress;
    y.whole = s.wait_whole;
    y.status = status;
    for (int i = 0; i < 3; i++) {
        y.wait[i] = s.wait[i];
        y.seen[i] = 0u;
    }
}
// true once `done` blocks of every producer are complete; false: gave up (a producer that never moves — cannot happen while
// workgroups are dispatched in launch order, the launch puts producers in front — must not hang the device: the image goes to the host)
__device__ __forceinline__ bool pw_wait_for(PwSync &y, uint32_t done) {
    for (int i = 0; i < 3; i++) {
        if (y.wait[i] == nullptr) continue;
        const uint32_t need = ((y.whole >> i) & 1u) ? PROG_DONE : done;
        if (y.seen[i] >= need) continue;
#ifdef JPGPU_HOST_EMULATION
        y.seen[i] = *y.wait[i];  // (tests/emu runs the waves one after the other, producers first)
        if (y.seen[i] < need) {
            pw_flag(y.status, PROG_ST_WAIT);
            return false;
        }
#else
        uint32_t spins = 0;
        for (;;) {
#ifndef PROGW_FENCES
            y.seen[i] = wv_uniform(__hip_atomic_load((const JP_GLOBAL uint32_t *)y.wait[i], __ATOMIC_RELAXED, PW_SCOPE));
#else
            y.seen[i] = wv_uniform(__hip_atomic_load((const JP_GLOBAL uint32_t *)y.wait[i], __ATOMIC_ACQUIRE, PW_SCOPE));
#endif
            if (y.seen[i] >= need) break;
            if (++spins > (1u << 20)) {  // (~2 s)
                pw_flag(y.status, PROG_ST_WAIT);
                return false;
            }
            wv_sleep();
        }
#endif
    }
    return true;
}
// `done` blocks are complete: everything this wave has stored is made visible first (release)
__device__ __forceinline__ void pw_publish(PwSync &y, uint32_t done) {
    if (y.progress == nullptr) return;
#ifdef JPGPU_HOST_EMULATION
    *y.progress = done;
#else
#ifndef PROGW_FENCES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (the whole wave: every lane's stores and atomics, then lane 0's word)
#endif
    if (threadIdx.x == 0u) __hip_atomic_store((JP_GLOBAL uint32_t *)y.progress, done, __ATOMIC_RELAXED, PW_SCOPE);
#endif
}

// ---- where the blocks of a scan lie ------------------------------------------------------------------------------------------------------
// what the walk needs of the descriptor, read once
struct PwGeom {
    uint32_t ncomp, cols, rows, bpm;  // bpm: blocks per MCU
    uint32_t h[4], v[4], block_w[4];  // (indexed by unrolled loops only: registers, never memory)
    int16_t *coefs[4];
    uint32_t nblk;                    // h * v of component c in byte c (<= 16): what the scalar walk indexes
};
__device__ __forceinline__ void pw_geom(PwGeom &g, const JP_GLOBAL ProgScan &s) {
    g.ncomp = s.ncomp, g.cols = s.cols, g.rows = s.rows, g.bpm = 0u, g.nblk = 0u;
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        const bool on = c < g.ncomp;
        g.h[c] = on ? s.comp[c].h : 1u, g.v[c] = on ? s.comp[c].v : 1u, g.block_w[c] = on ? s.comp[c].block_w : 0u;
        g.coefs[c] = on ? s.comp[c].coefs : nullptr;
        if (on) g.bpm += g.h[c] * g.v[c];
        g.nblk |= (on ? (g.h[c] * g.v[c]) & 0xffu : 0u) << (8u * c);
    }
}
// block `idx` of the walk (decode_scan's loops, src/decoder.rs:871-1000: MCU rows, MCUs, components, the component's blocks of the
// MCU): -> its first coefficient.  Per lane.
__device__ __forceinline__ int16_t *pw_block_of(const PwGeom &g, uint32_t idx) {
    const uint32_t m = idx / g.bpm;
    uint32_t r = idx - m * g.bpm;
    const uint32_t my = m / g.cols, mx = m - my * g.cols;
    uint32_t h = g.h[0], v = g.v[0], bw = g.block_w[0];
    int16_t *base = g.coefs[0];
    bool found = g.ncomp == 1u || r < g.h[0] * g.v[0];
#pragma unroll
    for (uint32_t c = 1; c < 4u; c++) {  // (selects, not an index: the component differs from lane to lane)
        if (!found) {
            r -= g.h[c - 1u] * g.v[c - 1u];
            h = g.h[c], v = g.v[c], bw = g.block_w[c], base = g.coefs[c];
            found = c + 1u >= g.ncomp || r < g.h[c] * g.v[c];
        }
    }
    const uint32_t vp = r / h, hp = r - vp * h;
    return base + ((size_t)(my * v + vp) * bw + (mx * h + hp)) * 64u;
}

// ---- DC scans (ss == se == 0; one to four components, src/decoder.rs:1100-1126 and :1181-1190), 64 blocks per chunk ----------------------
// A DC first scan's chunk: which component block i of the chunk belongs to, and which of the scan's tables decodes it, as bit masks
// over the chunk's 64 blocks (bit i of cm0 / cm1: the component's index bits, of ts0 / ts1: its table's) — the lanes work them out once
// per chunk, the walk shifts them.
// the component of block `idx` of a scan's walk (decode_scan's loops: components in order inside an MCU, h * v blocks each).  Per lane.
__device__ __forceinline__ uint32_t pw_dc_comp_of(const PwGeom &g, uint32_t idx) {
    uint32_t r = idx % g.bpm, c = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 3u; k++) {
        const uint32_t nb = (g.nblk >> (8u * k)) & 0xffu;
        if (c == k && k + 1u < g.ncomp && r >= nb) r -= nb, c = k + 1u;
    }
    return c;
}
struct PwDcChunk {
    uint64_t cm0, cm1, ts0, ts1;
    uint32_t i, n, err;  // the next block, how many the chunk has
    uint64_t pred;       // four 16-bit predictors (wrapping_add on i16 in the reference)
};
// ONE block of a DC first scan, in portable C++
__device__ __forceinline__ void pw_dc_symbol(PwBits &b, const PwTable &tab0, const PwTable &tab1, const PwTable &tab2, const PwTable &tab3, WV32 &val, PwDcChunk &D, uint32_t al) {
    PW_NEED32(b);
    const uint32_t i = D.i, look = pw_look(b);
    const uint32_t c = (uint32_t)((D.cm0 >> i) & 1ull) | ((uint32_t)((D.cm1 >> i) & 1ull) << 1), t = (uint32_t)((D.ts0 >> i) & 1ull) | ((uint32_t)((D.ts1 >> i) & 1ull) << 1);
    // (the tables live in registers: the choice between them is a scalar branch, not an index)
    const uint32_t e = t == 0u ? pw_symbol(tab0, look) : t == 1u ? pw_symbol(tab1, look) : t == 2u ? pw_symbol(tab2, look) : pw_symbol(tab3, look);
    if (__builtin_expect(pw_e_kind(e) == PW_KIND_BAD, 0)) {  // no such code / "invalid DC difference magnitude category"
        D.err = pw_e_len(e) ? PROG_ST_BAD_DC : PROG_ST_BAD_CODE;
        D.i = D.n;
        return;
    }
    const uint32_t len = pw_e_len(e), cat = pw_e_extra(e);
    uint32_t diff = 0;
    if (cat) diff = (uint32_t)pw_extend(pw_field(look, len, cat), cat);  // len + cat <= 16 + 11
    pw_consume(b, len + cat);
    const uint32_t pv = (uint32_t)((D.pred >> (16u * c)) + diff) & 0xffffu;
    D.pred = (D.pred & ~(0xffffull << (16u * c))) | ((uint64_t)pv << (16u * c));
    wv_writelane(val, i, (pv << al) & 0xffffu);
    D.i = i + 1u;
}
#if defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// tests/emu: the hand-scheduled loop below in C++ (-> 0: the chunk is through, 1: the next block is the portable path's, 2: the window is
// used up).  It knows tables 0 and 1 (what encoders use); a block of tables 2 / 3 and a code of seven bits and more are handed back.
static inline uint32_t pw_dc_fast(PwBits &b, const PwTable &tab0, const PwTable &tab1, WV32 &val, PwDcChunk &D, uint32_t al) {
    for (;;) {
        if (b.pos < 32u) {
            if (b.dp == 64u) return 2u;
            b.win |= (uint64_t)b.nx << (32u - b.pos);
            b.pos += 32u;
            b.nx = wv_readlane(b.w, b.dp);
            b.dp++;
        }
        const uint32_t i = D.i, hi = (uint32_t)(b.win >> 32);
        if ((D.ts1 >> i) & 1ull) return 1u;
        const uint32_t e = ((D.ts0 >> i) & 1ull) ? wv_readlane(tab1.lut6, hi >> 26) : wv_readlane(tab0.lut6, hi >> 26), len = e & 31u;
        if (len == 0u || ((e >> 17) & 3u) == 3u) return 1u;
        const uint32_t cat = (e >> 5) & 31u, bits = ((hi << len) >> 1) >> (31u - cat), cons = len + cat;
        b.win <<= cons, b.pos -= cons;
        PROGW_COUNT(symbols, 1);
        const uint32_t diff = bits < (1u << ((cat - 1u) & 31u)) ? bits + (0xffffffffu << cat) + 1u : bits;  // (category 0: 0)
        const uint32_t sh = 16u * ((uint32_t)((D.cm0 >> i) & 1ull) | ((uint32_t)((D.cm1 >> i) & 1ull) << 1));
        const uint32_t pv = (uint32_t)((D.pred >> sh) + diff) & 0xffffu, v = (pv << al) & 0xffffu;
        D.pred = (D.pred & ~(0xffffull << sh)) | ((uint64_t)pv << sh);
        WV_EACH {
            if (lane == i) WV(val) = v;
        }
        D.i = i + 1u;
        if (!(D.i < D.n)) return 0u;
    }
}
#endif
#if !defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// The blocks of a DC first scan's chunk, hand-scheduled (gfx950) like pw_refine_fast below.  s[40:41] window, s42 valid bits, s43 next
// dword, s44 its successor's lane, s45 the block, s46 the chunk's count, s[48:49] the predictors, s[50:51] / s[52:53] component index
// bits, s[72:73] / s[74:75] table index bits, s55 al, s58 -> 0 / 1 / 2 as in the twin above.
__device__ __forceinline__ uint32_t pw_dc_fast(PwBits &b, const PwTable &tab0, const PwTable &tab1, WV32 &val, PwDcChunk &D, uint32_t al) {
    uint32_t code, t0;
    const uint32_t al_s = wv_uniform(al), n_s = wv_uniform(D.n);
    const uint64_t cm0 = wv_uniform64(D.cm0), cm1 = wv_uniform64(D.cm1), ts0 = wv_uniform64(D.ts0), ts1 = wv_uniform64(D.ts1);
    uint64_t win = wv_uniform64(b.win), pred = wv_uniform64(D.pred);
    uint32_t pos = wv_uniform(b.pos), nx = wv_uniform(b.nx), dp = wv_uniform(b.dp), i = wv_uniform(D.i);
    asm volatile(
        "s_mov_b64 s[40:41], %[win]\n s_mov_b32 s42, %[pos]\n s_mov_b32 s43, %[nx]\n s_mov_b32 s44, %[dp]\n s_mov_b32 s45, %[i]\n s_mov_b32 s46, %[n]\n"
        "s_mov_b64 s[48:49], %[pred]\n s_mov_b64 s[50:51], %[cm0]\n s_mov_b64 s[52:53], %[cm1]\n s_mov_b64 s[72:73], %[ts0]\n s_mov_b64 s[74:75], %[ts1]\n s_mov_b32 s55, %[al]\n"
        "Ltop%=:\n"
        "s_cmp_lt_u32 s42, 32\n"
        "s_cbranch_scc1 Lrefill%=\n"
        "Lsym%=:\n"
        "s_lshr_b64 s[60:61], s[74:75], s45\n"
        "s_bitcmp1_b32 s60, 0\n"                    // a block of table 2 / 3: the portable path
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_lshr_b32 s62, s41, 26\n"
        "v_readlane_b32 s63, %[lut0], s62\n"
        "v_readlane_b32 s64, %[lut1], s62\n"
        "s_lshr_b64 s[60:61], s[72:73], s45\n"
        "s_bitcmp1_b32 s60, 0\n"
        "s_cselect_b32 s63, s64, s63\n"             // the entry of this block's table
        "s_and_b32 s62, s63, 31\n"                  // code length; SCC = (length != 0)
        "s_cbranch_scc0 Lgeneric%=\n"
        "s_bfe_u32 s64, s63, 0x20011\n"
        "s_cmp_eq_u32 s64, 3\n"
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_bfe_u32 s65, s63, 0x50005\n"             // category = extra bits
        "s_lshl_b32 s66, s41, s62\n"
        "s_lshr_b32 s66, s66, 1\n"
        "s_sub_u32 s67, 31, s65\n"
        "s_lshr_b32 s66, s66, s67\n"                // their value
        "s_add_u32 s67, s62, s65\n"
        "s_lshl_b64 s[40:41], s[40:41], s67\n"
        "s_sub_u32 s42, s42, s67\n"
        "s_sub_u32 s67, s65, 1\n"
        "s_lshl_b32 s67, 1, s67\n"                  // 1 << (category - 1); category 0: 1 << 31, and the difference comes out 0
        "s_lshl_b32 s68, -1, s65\n"
        "s_add_u32 s68, s68, 1\n"
        "s_add_u32 s68, s66, s68\n"
        "s_cmp_lt_u32 s66, s67\n"
        "s_cselect_b32 s68, s68, s66\n"             // the difference (src/huffman.rs:165-173)
        "s_lshr_b64 s[60:61], s[50:51], s45\n"
        "s_and_b32 s69, s60, 1\n"
        "s_lshr_b64 s[60:61], s[52:53], s45\n"
        "s_and_b32 s60, s60, 1\n"
        "s_lshl_b32 s60, s60, 1\n"
        "s_or_b32 s69, s69, s60\n"
        "s_lshl_b32 s69, s69, 4\n"                  // 16 x the block's component
        "s_lshr_b64 s[60:61], s[48:49], s69\n"
        "s_add_u32 s68, s60, s68\n"
        "s_and_b32 s68, s68, 0xffff\n"              // the predictor + the difference, 16 bits
        "s_mov_b32 s60, 0xffff\n"
        "s_mov_b32 s61, 0\n"
        "s_lshl_b64 s[60:61], s[60:61], s69\n"
        "s_andn2_b64 s[48:49], s[48:49], s[60:61]\n"
        "s_mov_b32 s60, s68\n"
        "s_mov_b32 s61, 0\n"
        "s_lshl_b64 s[60:61], s[60:61], s69\n"
        "s_or_b64 s[48:49], s[48:49], s[60:61]\n"
        "s_lshl_b32 s68, s68, s55\n"
        "s_and_b32 s68, s68, 0xffff\n"
        "v_mov_b32_e32 %[t0], s68\n"
        "s_lshl_b64 s[60:61], 1, s45\n"
        "s_add_u32 s45, s45, 1\n"
        "v_cndmask_b32_e64 %[val], %[val], %[t0], s[60:61]\n"
        "s_cmp_lt_u32 s45, s46\n"
        "s_cbranch_scc1 Ltop%=\n"
        "s_mov_b32 s58, 0\n"
        "s_branch Lend%=\n"
        "Lrefill%=:\n"
        "s_cmp_eq_u32 s44, 64\n"
        "s_cbranch_scc1 Lwindow%=\n"
        "s_sub_u32 s60, 32, s42\n"
        "s_mov_b32 s62, s43\n"
        "s_mov_b32 s63, 0\n"
        "s_lshl_b64 s[62:63], s[62:63], s60\n"
        "s_or_b64 s[40:41], s[40:41], s[62:63]\n"
        "s_add_u32 s42, s42, 32\n"
        "v_readlane_b32 s43, %[w], s44\n"
        "s_add_u32 s44, s44, 1\n"
        "s_branch Lsym%=\n"
        "Lgeneric%=:\n"
        "s_mov_b32 s58, 1\n"
        "s_branch Lend%=\n"
        "Lwindow%=:\n"
        "s_mov_b32 s58, 2\n"
        "Lend%=:\n"
        "s_mov_b64 %[win], s[40:41]\n s_mov_b32 %[pos], s42\n s_mov_b32 %[nx], s43\n s_mov_b32 %[dp], s44\n s_mov_b32 %[i], s45\n s_mov_b64 %[pred], s[48:49]\n s_mov_b32 %[code], s58\n"
        : [win] "+s"(win), [pos] "+s"(pos), [nx] "+s"(nx), [dp] "+s"(dp), [i] "+s"(i), [pred] "+s"(pred), [code] "=s"(code), [val] "+v"(val), [t0] "=&v"(t0)
        : [n] "s"(n_s), [al] "s"(al_s), [cm0] "s"(cm0), [cm1] "s"(cm1), [ts0] "s"(ts0), [ts1] "s"(ts1), [lut0] "v"(tab0.lut6), [lut1] "v"(tab1.lut6), [w] "v"(b.w)
        : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s48", "s49", "s50", "s51", "s52", "s53", "s55", "s58", "s60", "s61", "s62", "s63", "s64", "s65",
          "s66", "s67", "s68", "s69", "s72", "s73", "s74", "s75");
    b.win = win, b.pos = pos, b.nx = nx, b.dp = dp, D.i = i, D.pred = pred;
    return code;
}
#endif
#if !defined(PROGW_PORTABLE)
// One call of pw_dc_fast on a given state (tests: the device against the C++ twin)
struct PwDcCase {
    uint64_t win, pred, cm0, cm1, ts0, ts1;
    uint32_t pos, nx, dp, i, n, al, code, pad_;
    uint32_t lut0[64], lut1[64], w[64], val[64];
};
__device__ inline void pw_dc_fast_case(PwDcCase &c) {
    PwBits b;
    PwTable tab0, tab1;
    PwDcChunk D{c.cm0, c.cm1, c.ts0, c.ts1, c.i, c.n, 0u, c.pred};
    WV32 val;
    b.win = c.win, b.pos = c.pos, b.nx = c.nx, b.dp = c.dp, b.base = 0u, b.src = nullptr, b.n_dwords = 0u;
    tab0.g = tab1.g = nullptr, tab0.mode = tab1.mode = 2u;
    WV_EACH { WV(b.w) = c.w[lane], WV(tab0.lut6) = c.lut0[lane], WV(tab1.lut6) = c.lut1[lane], WV(val) = c.val[lane]; }
    const uint32_t code = pw_dc_fast(b, tab0, tab1, val, D, c.al);
    WV_EACH { c.val[lane] = WV(val); }
    c.win = b.win, c.pos = b.pos, c.nx = b.nx, c.dp = b.dp, c.i = D.i, c.pred = D.pred, c.code = code;
}
#endif

__device__ inline bool pw_scan_dc(const JP_GLOBAL ProgScan &s, uint32_t *status, PwSync &y) {
    PwBits b;
    PwGeom g;
    pw_bits_open(b, s.data, s.n_bytes);
    pw_geom(g, s);
    const bool first = s.ah == 0;
    const uint32_t al = wv_uniform(s.al), total = g.rows * g.cols * g.bpm;
    PwTable tab0, tab1, tab2, tab3;
    uint32_t table_of = 0u;  // two bits per component
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) table_of |= (c < g.ncomp ? s.comp[c].table & 3u : 0u) << (2u * c);
    if (first) pw_table_load(tab0, s.table[0], 2u), pw_table_load(tab1, s.table[1], 2u), pw_table_load(tab2, s.table[2], 2u), pw_table_load(tab3, s.table[3], 2u);
    uint64_t pred = 0;  // four 16-bit predictors (wrapping_add on i16 in the reference)
    for (uint32_t cb = 0; cb < total; cb += 64u) {
        const uint32_t n = total - cb < 64u ? total - cb : 64u;
        if (!pw_wait_for(y, cb + n)) return false;
        if (first) {
            WV32 val;
            PwDcChunk D{0ull, 0ull, 0ull, 0ull, 0u, n, 0u, pred};
            // which component — and with it which table — every block of the chunk belongs to, as masks over the chunk
            WV_EACH { WV(val) = 0u; }
            WV_BALLOT(D.cm0, lane < n && (pw_dc_comp_of(g, cb + lane) & 1u) != 0u);
            WV_BALLOT(D.cm1, lane < n && (pw_dc_comp_of(g, cb + lane) & 2u) != 0u);
            WV_BALLOT(D.ts0, lane < n && ((table_of >> (2u * pw_dc_comp_of(g, cb + lane))) & 1u) != 0u);
            WV_BALLOT(D.ts1, lane < n && ((table_of >> (2u * pw_dc_comp_of(g, cb + lane))) & 2u) != 0u);
#if !defined(PROGW_PORTABLE) && !defined(PROGW_NO_DC_FAST)
            for (;;) {
                const uint32_t code = pw_dc_fast(b, tab0, tab1, val, D, al);
                if (code == 0u) break;
                if (code == 2u) {
                    pw_refill(b);  // (into the next window)
                    continue;
                }
                pw_dc_symbol(b, tab0, tab1, tab2, tab3, val, D, al);
                if (D.i >= n) break;
            }
#else
            do pw_dc_symbol(b, tab0, tab1, tab2, tab3, val, D, al);
            while (D.i < n);
#endif
            if (__builtin_expect(D.err != 0u, 0)) {
                pw_flag(status, D.err);
                return false;
            }
            pred = D.pred;
            WV_EACH {
                if (lane < n) pw_store16(pw_block_of(g, cb + lane), (int16_t)(uint16_t)WV(val));
            }
        } else {  // a bit per block
            PW_NEED32(b);
            const uint32_t hi = pw_look(b);
            pw_consume(b, n < 32u ? n : 32u);
            uint32_t lo = 0;
            if (n > 32u) {
                PW_NEED32(b);
                lo = pw_look(b);
                pw_consume(b, n - 32u);
            }
            WV_EACH {
                if (lane < n && (((lane < 32u ? hi : lo) >> (31u - (lane & 31u))) & 1u))
                    pw_or32(reinterpret_cast<uint32_t *>(pw_block_of(g, cb + lane)), 1u << al);  // co[0] |= bit (the low half of the block's first dword)
            }
        }
        pw_publish(y, cb + n);
    }
    return true;
}

// ---- AC scans: one component, h = v = 1, band [ss, se] ----------------------------------------------------------------------------------
// block `bi` of the walk -> index in the plane (the walk stops at the image's edge: cols <= block_w)
__device__ __forceinline__ size_t pw_ac_block(uint32_t bi, uint32_t cols, uint32_t block_w) {
    const uint32_t my = bi / cols;
    return (size_t)my * block_w + (bi - my * cols);
}

// ... kept as the walk goes (the division above costs ~40 instructions: once per chunk and lane for the masks, never per block)
struct PwWalk {
    uint32_t mx, row, cols, block_w;
};
__device__ __forceinline__ void pw_walk_open(PwWalk &w, uint32_t cols, uint32_t block_w) { w.mx = 0u, w.row = 0u, w.cols = cols, w.block_w = block_w; }
__device__ __forceinline__ size_t pw_walk_block(const PwWalk &w) { return (size_t)w.row * w.block_w + w.mx; }
__device__ __forceinline__ void pw_walk_advance(PwWalk &w, uint32_t n) {
    w.mx += n;
    while (w.mx >= w.cols) w.mx -= w.cols, w.row++;
}

// the state of an AC first scan inside a block
struct PwFirstBlock {
    uint32_t k, eob_run, err;
    uint64_t nz, neg;
};
// ONE symbol of an AC first scan's block, in portable C++ (the hand-scheduled loop below hands back what it does not take itself);
// the block is through when B.k > se
__device__ __forceinline__ void pw_first_symbol(PwBits &b, const PwTable &tab, WV32 &cf, PwFirstBlock &B, uint32_t se, uint32_t al) {
    PW_NEED32(b);
    const uint32_t look = pw_look(b), e = pw_symbol(tab, look);
    const uint32_t len = pw_e_len(e), nb = pw_e_extra(e), kind = pw_e_kind(e);
    const uint32_t bits = pw_field(look, len, nb);  // the magnitude bits of a coefficient / the low bits of an end-of-band run
    pw_consume(b, len + nb);
    const uint32_t k = B.k + pw_e_run(e);  // (ZRL: 16; an end-of-band symbol: 0)
    if (__builtin_expect(kind != PW_KIND_COEF, 0)) {
        if (kind == PW_KIND_ZRL) {
            B.k = k;
            return;
        }
        if (kind == PW_KIND_EOB) B.eob_run = (1u << nb) - 1u + bits;
        else B.err = PROG_ST_BAD_CODE;
        B.k = se + 1u;
        return;
    }
    // a run that leaves the band: what the reference then does with the magnitude bits depends on its table layout
    // (frontend.cpp, decode_block) — the host's business; so is a magnitude that could make a later correction carry
    if (__builtin_expect(k > se || nb + al > 14u, 0)) {
        B.err = k > se ? PROG_ST_BAND : PROG_ST_RANGE;
        B.k = se + 1u;
        return;
    }
    const int32_t v = pw_extend(bits, nb);
    wv_writelane(cf, k, ((uint32_t)v << al) & 0xffffu);
    B.nz |= 1ull << k;
    B.neg |= (uint64_t)((uint32_t)v >> 31) << k;
    B.k = k + 1u;
}
#if defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// tests/emu: the hand-scheduled loop below in C++ (-> 0: the block is through, 1: the next symbol is the portable path's, 2: the window is used up)
static inline uint32_t pw_first_fast(PwBits &b, const PwTable &tab, WV32 &cf, PwFirstBlock &B, uint32_t se, uint32_t al) {
    for (;;) {
        if (b.pos < 32u) {
            if (b.dp == 64u) return 2u;
            b.win |= (uint64_t)b.nx << (32u - b.pos);
            b.pos += 32u;
            b.nx = wv_readlane(b.w, b.dp);
            b.dp++;
        }
        const uint32_t hi = (uint32_t)(b.win >> 32), e = wv_readlane(tab.lut6, hi >> 26);
        uint32_t len = e & 31u, nb = (e >> 5) & 31u, kind = (e >> 17) & 3u, run = (e >> 10) & 127u;
        if (len == 0u) {  // codes of seven and eight bits, from the 8-bit lookup in memory
            const uint32_t idx8 = hi >> 24, e8 = (reinterpret_cast<const JP_CONST uint32_t *>(tab.g->lut)[idx8 >> 1] >> (16u * (idx8 & 1u))) & 0xffffu;
            len = e8 >> 8;
            if (len == 0u) return 1u;
            const uint32_t sz = e8 & 15u, r = (e8 >> 4) & 15u;
            if (sz) nb = sz, kind = 0u, run = r;
            else if (r == 15u) nb = 0u, kind = 2u, run = 16u;
            else nb = r, kind = 1u, run = 0u;
        }
        const uint32_t bits = ((hi << len) >> 1) >> (31u - nb), cons = len + nb, k2 = B.k + run;
        if (kind == 0u) {
            if (k2 > se || nb + al > 14u) return 1u;
            b.win <<= cons, b.pos -= cons;
            PROGW_COUNT(symbols, 1);
            const bool negative = bits < (1u << (nb - 1u));
            const uint32_t v = negative ? bits + (0xffffffffu << nb) + 1u : bits, val = (v << al) & 0xffffu;
            WV_EACH {
                if (lane == k2) WV(cf) = val;
            }
            B.nz |= 1ull << k2;
            if (negative) B.neg |= 1ull << k2;
            B.k = k2 + 1u;
            if (!(B.k <= se)) return 0u;
            continue;
        }
        if (kind == 3u) return 1u;
        b.win <<= cons, b.pos -= cons;
        PROGW_COUNT(symbols, 1);
        if (kind == 2u) {
            B.k = k2;
            if (!(B.k <= se)) return 0u;
            continue;
        }
        B.eob_run = (1u << nb) + bits - 1u;
        B.k = se + 1u;
        return 0u;
    }
}
#endif
#if !defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// The symbols of an AC first scan's block, as many in a row as go without help, hand-scheduled (gfx950) like pw_refine_fast below.
// s[40:41] window, s42 valid bits, s43 next dword, s44 its successor's lane, s45 k, s[46:47] / s[48:49] non-zero / negative, s54 se,
// s55 al, s57 end-of-band run, s[82:83] the table's 8-bit lookup in memory, s58 -> 0: the block is through (k > se), 1: the next symbol is
// the portable path's (a code of nine bits and more, a run that leaves the band, a bad symbol), 2: the window is used up.
__device__ __forceinline__ uint32_t pw_first_fast(PwBits &b, const PwTable &tab, WV32 &cf, PwFirstBlock &B, uint32_t se, uint32_t al) {
    uint32_t code, t0;
    // (every scalar the loop takes or gives back: into scalar registers by hand — the compiler keeps some of them in vector registers
    // although every lane holds the same, and "s" then fails with "illegal VGPR to SGPR copy"; a no-op where the value is scalar already)
    const uint32_t se_s = wv_uniform(se), al_s = wv_uniform(al);
    const uint64_t lut8_s = wv_uniform64((uint64_t)(uintptr_t)(const void *)tab.g->lut);
    uint64_t win = wv_uniform64(b.win), nz = wv_uniform64(B.nz), neg = wv_uniform64(B.neg);
    uint32_t pos = wv_uniform(b.pos), nx = wv_uniform(b.nx), dp = wv_uniform(b.dp), k = wv_uniform(B.k), eob = wv_uniform(B.eob_run);
    asm volatile(
        "s_mov_b64 s[40:41], %[win]\n s_mov_b32 s42, %[pos]\n s_mov_b32 s43, %[nx]\n s_mov_b32 s44, %[dp]\n s_mov_b32 s45, %[k]\n"
        "s_mov_b64 s[46:47], %[nz]\n s_mov_b64 s[48:49], %[neg]\n s_mov_b32 s54, %[se]\n s_mov_b32 s55, %[al]\n s_mov_b32 s57, %[eob]\n s_mov_b64 s[82:83], %[lut8]\n"
        "Ltop%=:\n"
        "s_cmp_lt_u32 s42, 32\n"
        "s_cbranch_scc1 Lrefill%=\n"
        "Lsym%=:\n"
        "s_lshr_b32 s60, s41, 26\n"
        "v_readlane_b32 s61, %[lut], s60\n"
        "s_and_b32 s62, s61, 31\n"                  // code length; SCC = (length != 0)
        "s_cbranch_scc0 Lsecond%=\n"
        "s_bfe_u32 s63, s61, 0x50005\n"             // extra bits
        "s_bfe_u32 s64, s61, 0x20011\n"             // kind
        "s_bfe_u32 s65, s61, 0x7000a\n"             // run (ZRL: 16, end of band: 0)
        "Lhave%=:\n"
        "s_lshl_b32 s66, s41, s62\n"
        "s_lshr_b32 s66, s66, 1\n"
        "s_sub_u32 s67, 31, s63\n"
        "s_lshr_b32 s66, s66, s67\n"                // the extra bits' value
        "s_add_u32 s67, s62, s63\n"                 // bits the symbol takes
        "s_add_u32 s68, s45, s65\n"                 // k + run
        "s_cmp_lg_u32 s64, 0\n"
        "s_cbranch_scc1 Lother%=\n"
        "s_cmp_gt_u32 s68, s54\n"                   // a run that leaves the band: the portable path (and the host's business)
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_add_u32 s69, s63, s55\n"
        "s_cmp_gt_u32 s69, 14\n"                    // a magnitude that could make a later correction carry: likewise
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_lshl_b64 s[40:41], s[40:41], s67\n"
        "s_sub_u32 s42, s42, s67\n"
        "s_sub_u32 s69, s63, 1\n"
        "s_lshl_b32 s69, 1, s69\n"                  // 1 << (size - 1): below it the value is negative (src/huffman.rs:165-173)
        "s_lshl_b32 s70, -1, s63\n"
        "s_add_u32 s70, s70, 1\n"
        "s_add_u32 s70, s66, s70\n"
        "s_lshl_b64 s[72:73], 1, s68\n"             // the coefficient's position as a lane mask
        "s_cmp_lt_u32 s66, s69\n"
        "s_cselect_b32 s70, s70, s66\n"
        "s_cselect_b64 s[74:75], s[72:73], 0\n"
        "s_lshl_b32 s70, s70, s55\n"
        "s_and_b32 s70, s70, 0xffff\n"
        "v_mov_b32_e32 %[t0], s70\n"
        "s_or_b64 s[46:47], s[46:47], s[72:73]\n"
        "s_or_b64 s[48:49], s[48:49], s[74:75]\n"
        "v_cndmask_b32_e64 %[cf], %[cf], %[t0], s[72:73]\n"
        "s_add_u32 s45, s68, 1\n"
        "s_cmp_le_u32 s45, s54\n"
        "s_cbranch_scc1 Ltop%=\n"
        "s_mov_b32 s58, 0\n"
        "s_branch Lend%=\n"
        "Lother%=:\n"
        "s_cmp_eq_u32 s64, 3\n"
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_lshl_b64 s[40:41], s[40:41], s67\n"
        "s_sub_u32 s42, s42, s67\n"
        "s_cmp_eq_u32 s64, 2\n"
        "s_cbranch_scc0 Leob%=\n"
        "s_mov_b32 s45, s68\n"                      // ZRL: sixteen zeros
        "s_cmp_le_u32 s45, s54\n"
        "s_cbranch_scc1 Ltop%=\n"
        "s_mov_b32 s58, 0\n"
        "s_branch Lend%=\n"
        "Leob%=:\n"
        "s_lshl_b32 s57, 1, s63\n"                  // end of band: (1 << extra) - 1 + bits more blocks hold nothing of this band
        "s_add_u32 s57, s57, s66\n"
        "s_sub_u32 s57, s57, 1\n"
        "s_add_u32 s45, s54, 1\n"
        "s_mov_b32 s58, 0\n"
        "s_branch Lend%=\n"
        "Lrefill%=:\n"
        "s_cmp_eq_u32 s44, 64\n"
        "s_cbranch_scc1 Lwindow%=\n"
        "s_sub_u32 s60, 32, s42\n"
        "s_mov_b32 s62, s43\n"
        "s_mov_b32 s63, 0\n"
        "s_lshl_b64 s[62:63], s[62:63], s60\n"
        "s_or_b64 s[40:41], s[40:41], s[62:63]\n"
        "s_add_u32 s42, s42, 32\n"
        "v_readlane_b32 s43, %[w], s44\n"
        "s_add_u32 s44, s44, 1\n"
        "s_branch Lsym%=\n"
        "Lsecond%=:\n"                              // a code of seven or eight bits: the 8-bit lookup in memory, through the scalar cache
        "s_lshr_b32 s60, s41, 24\n"
        "s_lshr_b32 s84, s60, 1\n"
        "s_lshl_b32 s84, s84, 2\n"
        "s_load_dword s85, s[82:83], s84\n"
        "s_and_b32 s60, s60, 1\n"
        "s_lshl_b32 s60, s60, 4\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_lshr_b32 s85, s85, s60\n"
        "s_and_b32 s85, s85, 0xffff\n"              // symbol | length << 8
        "s_lshr_b32 s62, s85, 8\n"                  // SCC = (length != 0)
        "s_cbranch_scc0 Lgeneric%=\n"               // longer still: the portable path's walk
        "s_bfe_u32 s65, s85, 0x40004\n"             // run
        "s_and_b32 s63, s85, 15\n"                  // size; SCC = (size != 0)
        "s_cbranch_scc0 Lsecond0%=\n"
        "s_mov_b32 s64, 0\n"                        // a coefficient: `size` magnitude bits behind `run` zeros
        "s_branch Lhave%=\n"
        "Lsecond0%=:\n"
        "s_cmp_eq_u32 s65, 15\n"
        "s_cbranch_scc0 Lsecond1%=\n"
        "s_mov_b32 s64, 2\n"                        // ZRL
        "s_mov_b32 s65, 16\n"
        "s_branch Lhave%=\n"
        "Lsecond1%=:\n"
        "s_mov_b32 s63, s65\n"                      // end of band: `run` low bits of the run of blocks
        "s_mov_b32 s64, 1\n"
        "s_mov_b32 s65, 0\n"
        "s_branch Lhave%=\n"
        "Lgeneric%=:\n"
        "s_mov_b32 s58, 1\n"
        "s_branch Lend%=\n"
        "Lwindow%=:\n"
        "s_mov_b32 s58, 2\n"
        "Lend%=:\n"
        "s_mov_b64 %[win], s[40:41]\n s_mov_b32 %[pos], s42\n s_mov_b32 %[nx], s43\n s_mov_b32 %[dp], s44\n s_mov_b32 %[k], s45\n"
        "s_mov_b64 %[nz], s[46:47]\n s_mov_b64 %[neg], s[48:49]\n s_mov_b32 %[eob], s57\n s_mov_b32 %[code], s58\n"
        : [win] "+s"(win), [pos] "+s"(pos), [nx] "+s"(nx), [dp] "+s"(dp), [k] "+s"(k), [nz] "+s"(nz), [neg] "+s"(neg), [eob] "+s"(eob),
          [code] "=s"(code), [cf] "+v"(cf), [t0] "=&v"(t0)
        : [se] "s"(se_s), [al] "s"(al_s), [lut8] "s"(lut8_s), [lut] "v"(tab.lut6), [w] "v"(b.w)
        : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s54", "s55", "s57", "s58", "s60", "s61", "s62", "s63", "s64", "s65",
          "s66", "s67", "s68", "s69", "s70", "s72", "s73", "s74", "s75", "s82", "s83", "s84", "s85");
    b.win = win, b.pos = pos, b.nx = nx, b.dp = dp, B.k = k, B.nz = nz, B.neg = neg, B.eob_run = eob;
    return code;
}
#endif
#if !defined(PROGW_PORTABLE)
// One call of pw_first_fast on a given state (tests: the device against the C++ twin)
struct PwFirstCase {
    uint64_t win, nz, neg;
    uint32_t pos, nx, dp, k, se, al, eob, code;
    uint32_t lut6[64], w[64], cf[64];
    uint16_t lut8[256];
    const void *table;
};
__device__ inline void pw_first_fast_case(PwFirstCase &c) {
    PwBits b;
    PwTable tab;
    PwFirstBlock B{c.k, c.eob, 0u, c.nz, c.neg};
    WV32 cf;
    b.win = c.win, b.pos = c.pos, b.nx = c.nx, b.dp = c.dp, b.base = 0u, b.src = nullptr, b.n_dwords = 0u;
    tab.g = (const JP_CONST ProgHuffTable *)c.table, tab.mode = 0u;
    WV_EACH { WV(b.w) = c.w[lane], WV(tab.lut6) = c.lut6[lane], WV(cf) = c.cf[lane]; }
    const uint32_t code = pw_first_fast(b, tab, cf, B, c.se, c.al);
    WV_EACH { c.cf[lane] = WV(cf); }
    c.win = b.win, c.pos = b.pos, c.nx = b.nx, c.dp = b.dp, c.k = B.k, c.eob = B.eob_run, c.nz = B.nz, c.neg = B.neg, c.code = code;
}
#endif

// AC first scan (ah == 0; src/decoder.rs:1128-1172)
__device__ inline bool pw_scan_ac_first(const JP_GLOBAL ProgScan &s, uint32_t *status, PwSync &y) {
    PwBits b;
    PwTable tab;
    pw_bits_open(b, s.data, s.n_bytes);
    pw_table_load(tab, s.table[0], 0u);
    int16_t *const coefs = s.comp[0].coefs;
    uint64_t *const masks = s.comp[0].masks;
    // (into scalar registers by hand: byte loads are vector loads, and the hand-scheduled loop takes its state in SGPRs)
    const uint32_t block_w = s.comp[0].block_w, cols = s.cols, total = s.rows * s.cols, ss = wv_uniform(s.ss), se = wv_uniform(s.se), al = wv_uniform(s.al);
    uint32_t eob_run = 0;
    PwWalk at;
    pw_walk_open(at, cols, block_w);
    WV32 cf, unz;
    WV_EACH {
        WV(cf) = 0u;
        WV(unz) = pw_unzig(lane);
    }
    for (uint32_t cb = 0; cb < total; cb += PROGW_CHUNK) {
        const uint32_t n = total - cb < PROGW_CHUNK ? total - cb : PROGW_CHUNK;
        if (!pw_wait_for(y, cb + n)) return false;
        for (uint32_t i = 0; i < n; i++) {
            if (eob_run > 0u) {  // (blocks of an end-of-band run hold nothing of this band: all of the chunk's at once)
                const uint32_t skip = eob_run < n - i ? eob_run : n - i;
                eob_run -= skip;
                i += skip - 1u;
                pw_walk_advance(at, skip);
                continue;
            }
            PROGW_COUNT(blocks, 1);
            PwFirstBlock B{ss, 0u, 0u, 0ull, 0ull};
#if !defined(PROGW_PORTABLE)
            for (;;) {
                const uint32_t code = pw_first_fast(b, tab, cf, B, se, al);
                if (code == 0u) break;
                if (code == 2u) {
                    pw_refill(b);  // (into the next window)
                    continue;
                }
                pw_first_symbol(b, tab, cf, B, se, al);
                if (B.k > se) break;
            }
#else
            do pw_first_symbol(b, tab, cf, B, se, al);
            while (B.k <= se);
#endif
            if (__builtin_expect(B.err != 0u, 0)) {
                pw_flag(status, B.err);
                return false;
            }
            eob_run = B.eob_run;
            const uint64_t nz = B.nz, neg = B.neg;
            if (nz) {
                const size_t blk = pw_walk_block(at);
                WV_EACH {
                    if ((nz >> lane) & 1ull) pw_store16(coefs + blk * 64u + WV(unz), (int16_t)(uint16_t)WV(cf));
                    // (OR, not store: another scan may own other bands of the block; lanes 0 and 1: the two words in one instruction)
                    if (lane < 2u) pw_or64(masks + 2u * blk + lane, lane ? neg : nz);
                }
            }
            pw_walk_advance(at, 1u);
        }
        pw_publish(y, cb + n);
    }
    return true;
}

// AC refinement scan (ah > 0; src/decoder.rs:1192-1298)
struct PwRefine {
    PwBits b;
    uint64_t nz, neg;  // the current block's masks as its producers left them
    WV32 acc;          // lane k: what this scan adds to / stores at zig-zag position k of the block (0: nothing)
    uint32_t bit;      // 1 << al
};
// refine_non_zeroes(start .. end-1, zrl), start < end: a correction bit for every non-zero coefficient until `zrl` zero ones have
// been passed; returns where the walk stopped (the (zrl + 1)-th zero coefficient, or end - 1)
__device__ __forceinline__ uint32_t pw_refine_non_zeroes(PwRefine &R, uint32_t start, uint64_t below_end, uint32_t end, uint32_t zrl) {
    const uint64_t range = below_end & (~0ull << start);
    const uint64_t zeros = range & ~R.nz;
    uint64_t todo = R.nz & range;
    uint32_t stop = end - 1u;
    if (__builtin_expect((uint32_t)__builtin_popcountll(zeros) > zrl, 1)) {
        uint64_t at = zeros;
        if (zrl) {  // the coefficients with exactly `zrl` zero ones below them (inside the range): the last of them is the zero one looked for
            WV_BALLOT(at, wv_rank(zeros, lane) == zrl);
            at &= zeros;
        }
        stop = (uint32_t)__builtin_ctzll(at);
        todo &= ~(~0ull << stop);
    }
    if (todo) {
        uint32_t n = (uint32_t)__builtin_popcountll(todo), done = 0;
        do {  // the correction bits, up to 32 at a time: the first coefficient's bit is the first in the stream
            const uint32_t take = n < 32u ? n : 32u;
            if (__builtin_expect(R.b.pos < take, 0)) pw_refill(R.b);
            const uint32_t corr = pw_look(R.b) >> (32u - take);
            pw_consume(R.b, take);
            PROGW_COUNT(corrections, take);
            WV_EACH {
                const uint32_t j = wv_rank(todo, lane) - done;
                if (((todo >> lane) & 1ull) && j < take && ((corr >> (take - 1u - j)) & 1u)) {
                    // c += sign(c) * bit.  (c & bit) == 0 always: the planner admits only streams whose scans refine a band one bit at
                    // a time (plan_progressive_scans), so every non-zero coefficient is a multiple of 2 * bit here; and |c| < 2^14
                    // (first scans check sz + al <= 14), so the addition cannot carry out of the coefficient's half of the dword.
                    WV(R.acc) = ((R.neg >> lane) & 1ull) ? 0u - R.bit : R.bit;
                }
            }
            n -= take;
            done += take;
        } while (__builtin_expect(n != 0u, 0));
    }
    return stop;
}

// the state of a refinement scan inside a block
struct PwRefineBlock {
    uint32_t k, eob_run, err;
    uint64_t new_nz, new_neg;
};
// ONE symbol of a refinement scan's block, in portable C++ (tests/emu runs this; on the device it is the path of everything the
// hand-scheduled loop below hands back: codes of seven bits and more, more than 32 correction bits at once, a refill at a window's end)
__device__ __forceinline__ void pw_refine_symbol(PwRefine &R, const PwTable &tab, PwRefineBlock &B, uint64_t below_end, uint32_t end) {
    PW_NEED32(R.b);
    const uint32_t look = pw_look(R.b), e = pw_symbol(tab, look);
    const uint32_t len = pw_e_len(e), nb = pw_e_extra(e), kind = pw_e_kind(e);
    const uint32_t bits = pw_field(look, len, nb);  // the sign of a new coefficient / the low bits of an end-of-band run
    pw_consume(R.b, len + nb);
    if (__builtin_expect(kind == PW_KIND_BAD, 0)) {  // no such code / "unexpected huffman code"
        B.err = len ? PROG_ST_REFINE_SYMBOL : PROG_ST_BAD_CODE;
        B.k = end;
        return;
    }
    const uint32_t k = pw_refine_non_zeroes(R, B.k, below_end, end, pw_e_run(e));  // (an end-of-band symbol: 64 — every correction that is left)
    // What the symbol leaves behind, without branches (a taken branch costs this walk 27 cycles, a select 4): a new
    // coefficient at k — for the other kinds the "lane" is 64, which no lane is, and the mask bit falls off the word.
    // (A damaged stream can make the walk end ON a non-zero coefficient — the band's last position, when it runs out of
    // zeros — and the new value then REPLACES it, src/decoder.rs:1251-1256: whatever correction the lane held is
    // overwritten, the sign becomes the new value's.)
#ifdef PROGW_BRANCHY
    if (kind == PW_KIND_COEF) {
        wv_writelane(R.acc, k, bits ? R.bit : 0u - R.bit);
        B.new_nz |= 1ull << k;
        B.new_neg |= (uint64_t)(bits ^ 1u) << k;
    } else if (kind == PW_KIND_EOB) {
        B.eob_run = (1u << nb) - 1u + bits;
    }
#else
    const bool coef = kind == PW_KIND_COEF;
    wv_writelane(R.acc, coef ? k : 64u, bits ? R.bit : 0u - R.bit);
    const uint64_t at = (uint64_t)(coef ? 1u : 0u) << k;
    B.new_nz |= at;
    B.new_neg |= bits ? 0ull : at;
    B.eob_run = kind == PW_KIND_EOB ? (1u << nb) - 1u + bits : B.eob_run;
#endif
    B.k = k + 1u;
}

#if defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// tests/emu: the hand-scheduled loop below, instruction group by instruction group, in C++ — so that the CPU tests walk the same
// fast path / portable path hand-overs as the device does
static inline uint32_t pw_refine_fast(PwRefine &R, const PwTable &tab, PwRefineBlock &B, const WV32 &delta, uint64_t below_end, uint32_t end) {
    PwBits &b = R.b;
    for (;;) {
        if (b.pos < 32u) {  // Lrefill
            if (b.dp == 64u) return 2u;
            b.win |= (uint64_t)b.nx << (32u - b.pos);
            b.pos += 32u;
            b.nx = wv_readlane(b.w, b.dp);
            b.dp++;
        }
        const uint32_t hi = (uint32_t)(b.win >> 32), e = wv_readlane(tab.lut6, hi >> 26);
        uint32_t len = e & 31u, nb = (e >> 5) & 31u, kind = (e >> 17) & 3u, zrl = (e >> 10) & 127u;
        if (len == 0u) {  // Lsecond: codes of seven and eight bits, from the 8-bit lookup in memory
            const uint32_t idx8 = hi >> 24, e8 = (reinterpret_cast<const JP_CONST uint32_t *>(tab.g->lut)[idx8 >> 1] >> (16u * (idx8 & 1u))) & 0xffffu;
            len = e8 >> 8;
            if (len == 0u) return 1u;
            const uint32_t sz = e8 & 15u, r = (e8 >> 4) & 15u;
            if (sz == 1u) nb = 1u, kind = 0u, zrl = r;
            else if (sz != 0u) return 1u;
            else if (r == 15u) nb = 0u, kind = 2u, zrl = 15u;
            else nb = r, kind = 1u, zrl = 64u;
        }
        if (kind == 3u) return 1u;
        const uint32_t bits = ((hi << len) >> 1) >> (31u - nb), cons = len + nb;
        uint64_t win2 = b.win << cons;
        uint32_t pos2 = b.pos - cons;
        const uint64_t range = (~0ull << B.k) & below_end, zeros = range & ~R.nz;
        uint64_t todo = range & R.nz;
        uint32_t stop = end - 1u;
        if ((uint32_t)__builtin_popcountll(zeros) > zrl) {
            uint64_t at = zeros;
            if (zrl) {
                WV_BALLOT(at, wv_rank(zeros, lane) == zrl);
                at &= zeros;
            }
            stop = (uint32_t)__builtin_ctzll(at);
            todo &= ~(~0ull << stop);
        }
        if (todo) {
            const uint32_t n = (uint32_t)__builtin_popcountll(todo);
            if (n > 32u) return 1u;
            if (pos2 < n) {  // Lrefill2
                if (b.dp == 64u) return 1u;
                win2 |= (uint64_t)b.nx << (32u - pos2);
                pos2 += 32u;
                b.nx = wv_readlane(b.w, b.dp);
                b.dp++;
            }
            const uint32_t corr = (uint32_t)(win2 >> 32) >> (32u - n);
            win2 <<= n;
            pos2 -= n;
            PROGW_COUNT(corrections, n);
            WV_EACH {
                const uint32_t t = (corr >> ((n - 1u - wv_rank(todo, lane)) & 31u)) & 1u;
                if (t == 1u && ((todo >> lane) & 1ull)) WV(R.acc) = WV(delta);
            }
        }
        PROGW_COUNT(symbols, 1);
        b.win = win2, b.pos = pos2;
        if (kind == 1u) B.eob_run = (1u << nb) + bits - 1u;
        const uint64_t at = kind == 0u ? 1ull << stop : 0ull;
        const uint32_t val = bits ? R.bit : 0u - R.bit;
        B.new_nz |= at;
        B.new_neg |= bits ? 0ull : at;
        WV_EACH {
            if ((at >> lane) & 1ull) WV(R.acc) = val;
        }
        B.k = stop + 1u;
        if (!(B.k < end)) return 0u;
    }
}
#endif
#if !defined(JPGPU_HOST_EMULATION) && !defined(PROGW_PORTABLE)
// The same step, as many in a row as go without help, hand-scheduled (gfx950): what one wave pays per instruction of a dependent chain
// is 4.1 cycles for a scalar instruction, ~20 more when a value crosses from the vector to the scalar side (v_readlane, v_cmp) and 27
// for a taken branch (tools/ubench_scalar_chain.hip, profiles/round6/01_*) — the compiler's version of the step above is ~130
// instructions, ~9 taken branches and ~30 register copies (it keeps the window registers in loop-carried copies): 940 cycles per
// symbol; this one is ~65 instructions and two taken branches.  Lane masks ARE bit masks here: lane k is zig-zag position k, so an
// SGPR pair such as `todo` or `1 << stop` is used directly as the condition of a v_cndmask.
// State in fixed scalar registers (the halves of a 64-bit inline-asm operand cannot be named): s[40:41] window, s42 valid bits, s43 next
// dword, s44 its successor's lane, s45 k, s[46:47] / s[48:49] new non-zero / new negative, s[50:51] non-zero, s[52:53] below_end,
// s54 end, s55 / s56 +bit / -bit, s57 end-of-band run, s[82:83] the table's 8-bit lookup in memory, s58 -> 0: the block is through, 1: the next symbol is the portable path's, 2: the
// window is used up (refill there); s60-s81 scratch.  Nothing is committed before a symbol's corrections are known to fit.
__device__ __forceinline__ uint32_t pw_refine_fast(PwRefine &R, const PwTable &tab, PwRefineBlock &B, const WV32 &delta, uint64_t below_end, uint32_t end) {
    uint32_t code, t0;
    // (values the compiler keeps in vector registers although every lane holds the same: into scalar ones, or "s" gets a VGPR)
    const uint32_t pbit = wv_uniform(R.bit), nbit = wv_uniform(0u - R.bit), end_s = wv_uniform(end);
    const uint64_t bend_s = ((uint64_t)wv_uniform((uint32_t)(below_end >> 32)) << 32) | wv_uniform((uint32_t)below_end);
    const uint64_t nz_s = wv_uniform64(R.nz), lut8_s = wv_uniform64((uint64_t)(uintptr_t)(const void *)tab.g->lut);
    uint64_t win = wv_uniform64(R.b.win), nnz = wv_uniform64(B.new_nz), nneg = wv_uniform64(B.new_neg);
    uint32_t pos = wv_uniform(R.b.pos), nx = wv_uniform(R.b.nx), dp = wv_uniform(R.b.dp), k = wv_uniform(B.k), eob = wv_uniform(B.eob_run);
    asm volatile(
        "s_mov_b64 s[40:41], %[win]\n s_mov_b32 s42, %[pos]\n s_mov_b32 s43, %[nx]\n s_mov_b32 s44, %[dp]\n s_mov_b32 s45, %[k]\n"
        "s_mov_b64 s[46:47], %[nnz]\n s_mov_b64 s[48:49], %[nneg]\n s_mov_b64 s[50:51], %[nz]\n s_mov_b64 s[52:53], %[bend]\n"
        "s_mov_b32 s54, %[end]\n s_mov_b32 s55, %[pbit]\n s_mov_b32 s56, %[nbit]\n s_mov_b32 s57, %[eob]\n s_mov_b64 s[82:83], %[lut8]\n"
        "Ltop%=:\n"
        "s_cmp_lt_u32 s42, 32\n"
        "s_cbranch_scc1 Lrefill%=\n"
        "Lsym%=:\n"
        "s_lshr_b32 s60, s41, 26\n"
        "v_readlane_b32 s61, %[lut], s60\n"
        "s_and_b32 s62, s61, 31\n"                  // code length; SCC = (length != 0)
        "s_cbranch_scc0 Lsecond%=\n"
        "s_bfe_u32 s63, s61, 0x50005\n"             // extra bits
        "s_bfe_u32 s64, s61, 0x20011\n"             // kind
        "s_bfe_u32 s65, s61, 0x7000a\n"             // zeros to pass
        "s_cmp_eq_u32 s64, 3\n"
        "s_cbranch_scc1 Lgeneric%=\n"
        "Lhave%=:\n"
        "s_lshl_b32 s66, s41, s62\n"
        "s_lshr_b32 s66, s66, 1\n"
        "s_sub_u32 s67, 31, s63\n"
        "s_lshr_b32 s66, s66, s67\n"                // the extra bits' value
        "s_add_u32 s67, s62, s63\n"
        "s_lshl_b64 s[68:69], s[40:41], s67\n"      // the window and its count behind the symbol (not committed yet)
        "s_sub_u32 s70, s42, s67\n"
        "s_lshl_b64 s[72:73], -1, s45\n"
        "s_and_b64 s[72:73], s[72:73], s[52:53]\n"  // positions k .. end-1
        "s_andn2_b64 s[74:75], s[72:73], s[50:51]\n"  // the zero ones of them
        "s_and_b64 s[72:73], s[72:73], s[50:51]\n"    // the non-zero ones: a correction bit each, up to where the walk stops
        "s_bcnt1_i32_b64 s71, s[74:75]\n"
        "s_sub_u32 s76, s54, 1\n"                   // where the walk stops if it runs out of zeros: end - 1
        "s_cmp_le_u32 s71, s65\n"
        "s_cbranch_scc1 Lnohit%=\n"
        "s_cmp_eq_u32 s65, 0\n"
        "s_cbranch_scc1 Lzero%=\n"
        "v_mbcnt_lo_u32_b32 %[t0], s74, 0\n"
        "v_mbcnt_hi_u32_b32 %[t0], s75, %[t0]\n"    // zero coefficients below each lane
        "v_cmp_eq_u32_e32 vcc, s65, %[t0]\n"
        "s_and_b64 s[74:75], vcc, s[74:75]\n"       // the zero coefficient with exactly `run` zero ones below it
        "Lzero%=:\n"
        "s_ff1_i32_b64 s76, s[74:75]\n"
        "s_lshl_b64 s[74:75], -1, s76\n"
        "s_andn2_b64 s[72:73], s[72:73], s[74:75]\n"
        "Lnohit%=:\n"
        "s_cmp_eq_u64 s[72:73], 0\n"
        "s_cbranch_scc1 Lplace%=\n"
        "s_bcnt1_i32_b64 s77, s[72:73]\n"
        "s_cmp_gt_u32 s77, 32\n"
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_cmp_lt_u32 s70, s77\n"
        "s_cbranch_scc1 Lrefill2%=\n"
        "Lcorr%=:\n"
        "s_sub_u32 s78, 32, s77\n"
        "s_lshr_b32 s78, s69, s78\n"                // the correction bits, the first coefficient's in bit n - 1
        "s_lshl_b64 s[68:69], s[68:69], s77\n"
        "s_sub_u32 s70, s70, s77\n"
        "v_mbcnt_lo_u32_b32 %[t0], s72, 0\n"
        "v_mbcnt_hi_u32_b32 %[t0], s73, %[t0]\n"    // which of the bits is this lane's
        "s_sub_u32 s79, s77, 1\n"
        "v_sub_u32_e32 %[t0], s79, %[t0]\n"
        "v_lshrrev_b32_e64 %[t0], %[t0], s78\n"
        "v_and_b32_e32 %[t0], 1, %[t0]\n"
        "v_cmp_eq_u32_e32 vcc, 1, %[t0]\n"
        "s_and_b64 vcc, vcc, s[72:73]\n"
        "v_cndmask_b32_e32 %[acc], %[acc], %[delta], vcc\n"
        "Lplace%=:\n"
        "s_mov_b64 s[40:41], s[68:69]\n"            // committed
        "s_mov_b32 s42, s70\n"
        "s_lshl_b32 s78, 1, s63\n"                  // (an end-of-band symbol's run: (1 << extra) - 1 + bits)
        "s_add_u32 s78, s78, s66\n"
        "s_sub_u32 s78, s78, 1\n"
        "s_cmp_eq_u32 s64, 1\n"
        "s_cselect_b32 s57, s78, s57\n"
        "s_lshl_b64 s[74:75], 1, s76\n"
        "s_cmp_eq_u32 s64, 0\n"
        "s_cselect_b64 s[74:75], s[74:75], 0\n"     // a new coefficient where the walk stopped — or none
        "s_cmp_lg_u32 s66, 0\n"
        "s_cselect_b32 s78, s55, s56\n"
        "s_cselect_b64 s[80:81], 0, s[74:75]\n"
        "v_mov_b32_e32 %[t0], s78\n"
        "s_or_b64 s[46:47], s[46:47], s[74:75]\n"
        "s_or_b64 s[48:49], s[48:49], s[80:81]\n"
        "v_cndmask_b32_e64 %[acc], %[acc], %[t0], s[74:75]\n"
        "s_add_u32 s45, s76, 1\n"
        "s_cmp_lt_u32 s45, s54\n"
        "s_cbranch_scc1 Ltop%=\n"
        "s_mov_b32 s58, 0\n"
        "s_branch Lend%=\n"
        "Lrefill%=:\n"
        "s_cmp_eq_u32 s44, 64\n"
        "s_cbranch_scc1 Lwindow%=\n"
        "s_sub_u32 s60, 32, s42\n"
        "s_mov_b32 s62, s43\n"
        "s_mov_b32 s63, 0\n"
        "s_lshl_b64 s[62:63], s[62:63], s60\n"
        "s_or_b64 s[40:41], s[40:41], s[62:63]\n"
        "s_add_u32 s42, s42, 32\n"
        "v_readlane_b32 s43, %[w], s44\n"
        "s_add_u32 s44, s44, 1\n"
        "s_branch Lsym%=\n"
        "Lrefill2%=:\n"                             // (behind the symbol, in front of its correction bits; nothing committed)
        "s_cmp_eq_u32 s44, 64\n"
        "s_cbranch_scc1 Lgeneric%=\n"
        "s_sub_u32 s60, 32, s70\n"
        "s_mov_b32 s80, s43\n"                      // (s62 / s63 hold the symbol's length and extra-bit count here: the end-of-band run needs them)
        "s_mov_b32 s81, 0\n"
        "s_lshl_b64 s[80:81], s[80:81], s60\n"
        "s_or_b64 s[68:69], s[68:69], s[80:81]\n"
        "s_add_u32 s70, s70, 32\n"
        "v_readlane_b32 s43, %[w], s44\n"
        "s_add_u32 s44, s44, 1\n"
        "s_branch Lcorr%=\n"
        "Lsecond%=:\n"                              // a code of seven or eight bits (4 % of the symbols): the 8-bit lookup in memory, through the scalar cache
        "s_lshr_b32 s60, s41, 24\n"
        "s_lshr_b32 s84, s60, 1\n"
        "s_lshl_b32 s84, s84, 2\n"
        "s_load_dword s85, s[82:83], s84\n"
        "s_and_b32 s60, s60, 1\n"
        "s_lshl_b32 s60, s60, 4\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_lshr_b32 s85, s85, s60\n"
        "s_and_b32 s85, s85, 0xffff\n"              // symbol | length << 8
        "s_lshr_b32 s62, s85, 8\n"                  // SCC = (length != 0)
        "s_cbranch_scc0 Lgeneric%=\n"               // longer still: the portable path's walk
        "s_and_b32 s60, s85, 15\n"                  // size
        "s_bfe_u32 s65, s85, 0x40004\n"             // run
        "s_cmp_eq_u32 s60, 1\n"
        "s_cbranch_scc0 Lsecond0%=\n"
        "s_mov_b32 s63, 1\n"                        // a new coefficient: one sign bit, run zeros to pass
        "s_mov_b32 s64, 0\n"
        "s_branch Lhave%=\n"
        "Lsecond0%=:\n"
        "s_cmp_eq_u32 s60, 0\n"
        "s_cbranch_scc0 Lgeneric%=\n"               // "unexpected huffman code"
        "s_cmp_eq_u32 s65, 15\n"
        "s_cbranch_scc0 Lsecond1%=\n"
        "s_mov_b32 s63, 0\n"                        // ZRL
        "s_mov_b32 s64, 2\n"
        "s_branch Lhave%=\n"
        "Lsecond1%=:\n"
        "s_mov_b32 s63, s65\n"                      // end of band: run low bits, every correction that is left
        "s_mov_b32 s64, 1\n"
        "s_mov_b32 s65, 64\n"
        "s_branch Lhave%=\n"
        "Lgeneric%=:\n"
        "s_mov_b32 s58, 1\n"
        "s_branch Lend%=\n"
        "Lwindow%=:\n"
        "s_mov_b32 s58, 2\n"
        "Lend%=:\n"
        "s_mov_b64 %[win], s[40:41]\n s_mov_b32 %[pos], s42\n s_mov_b32 %[nx], s43\n s_mov_b32 %[dp], s44\n s_mov_b32 %[k], s45\n"
        "s_mov_b64 %[nnz], s[46:47]\n s_mov_b64 %[nneg], s[48:49]\n s_mov_b32 %[eob], s57\n s_mov_b32 %[code], s58\n"
        : [win] "+s"(win), [pos] "+s"(pos), [nx] "+s"(nx), [dp] "+s"(dp), [k] "+s"(k), [nnz] "+s"(nnz), [nneg] "+s"(nneg),
          [eob] "+s"(eob), [code] "=s"(code), [acc] "+v"(R.acc), [t0] "=&v"(t0)
        : [nz] "s"(nz_s), [bend] "s"(bend_s), [end] "s"(end_s), [pbit] "s"(pbit), [nbit] "s"(nbit), [lut8] "s"(lut8_s), [lut] "v"(tab.lut6), [w] "v"(R.b.w), [delta] "v"(delta)
        : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58",
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85");
    R.b.win = win, R.b.pos = pos, R.b.nx = nx, R.b.dp = dp, B.k = k, B.new_nz = nnz, B.new_neg = nneg, B.eob_run = eob;
    return code;
}
#endif

#if !defined(PROGW_PORTABLE)
// One call of pw_refine_fast on a given state (tests: the hand-scheduled loop on the device against its C++ twin on the CPU —
// tests/test_gpu_progw_asm.py, tests/emu/emu_prog.cpp)
struct PwFastCase {
    uint64_t win, nz, neg, new_nz, new_neg;
    uint32_t pos, nx, dp, k, end, al, eob, code;
    uint32_t lut6[64], w[64], acc[64];
    uint16_t lut8[256];  // the head of a ProgHuffTable: symbol | length << 8 per 8-bit prefix
    const void *table;   // where lut8 lies for whoever runs the case
};
__device__ inline void pw_refine_fast_case(PwFastCase &c) {
    PwRefine R;
    PwTable tab;
    PwRefineBlock B{c.k, c.eob, 0u, c.new_nz, c.new_neg};
    WV32 delta;
    R.b.win = c.win, R.b.pos = c.pos, R.b.nx = c.nx, R.b.dp = c.dp, R.b.base = 0u, R.b.src = nullptr, R.b.n_dwords = 0u;
    R.nz = c.nz, R.neg = c.neg, R.bit = 1u << c.al;
    tab.g = (const JP_CONST ProgHuffTable *)c.table, tab.mode = 1u;
    WV_EACH {
        WV(R.b.w) = c.w[lane], WV(tab.lut6) = c.lut6[lane], WV(R.acc) = c.acc[lane];
        WV(delta) = ((R.neg >> lane) & 1ull) ? 0u - R.bit : R.bit;
    }
    const uint64_t below_end = c.end >= 64u ? ~0ull : ((1ull << c.end) - 1ull);
    const uint32_t code = pw_refine_fast(R, tab, B, delta, below_end, c.end);
    WV_EACH { c.acc[lane] = WV(R.acc); }
    c.win = R.b.win, c.pos = R.b.pos, c.nx = R.b.nx, c.dp = R.b.dp, c.k = B.k, c.eob = B.eob_run, c.new_nz = B.new_nz, c.new_neg = B.new_neg, c.code = code;
}
#endif

__device__ inline bool pw_scan_ac_refine(const JP_GLOBAL ProgScan &s, uint32_t *status, PwSync &y) {
    PwRefine R;
    PwTable tab;
    pw_bits_open(R.b, s.data, s.n_bytes);
    pw_table_load(tab, s.table[0], 1u);
    int16_t *const coefs = s.comp[0].coefs;
    uint64_t *const masks = s.comp[0].masks;
    // (into scalar registers once, by hand: byte loads are vector loads, and the hand-scheduled loop takes these in SGPRs at every call)
    const uint32_t block_w = s.comp[0].block_w, cols = s.cols, total = s.rows * s.cols, ss = wv_uniform(s.ss), end = wv_uniform((uint32_t)s.se + 1u);
    const uint64_t below_end = end >= 64u ? ~0ull : ((1ull << end) - 1ull), band = below_end & (~0ull << ss);
    R.bit = wv_uniform(1u << s.al);
    uint32_t eob_run = 0;
    uint32_t n_calls = 0, n_generic = 0, n_window = 0;  // (reported with the scan's time)
    PwWalk at;
    pw_walk_open(at, cols, block_w);
    WV32 unz, m0, m1, m2, m3;
    WV_EACH {
        WV(R.acc) = 0u;
        WV(unz) = pw_unzig(lane);
    }
    for (uint32_t cb = 0; cb < total; cb += PROGW_CHUNK) {
        const uint32_t n = total - cb < PROGW_CHUNK ? total - cb : PROGW_CHUNK;
        // the masks of the chunk's blocks, one block per lane — what the scans this one depends on have left behind
        if (!pw_wait_for(y, cb + n)) return false;
        WV_EACH {
            WV(m0) = WV(m1) = WV(m2) = WV(m3) = 0u;
            if (lane < n) {
                const JP_GLOBAL uint32_t *mp = (const JP_GLOBAL uint32_t *)(masks + 2u * pw_ac_block(cb + lane, cols, block_w));
#ifdef JPGPU_HOST_EMULATION
                WV(m0) = mp[0], WV(m1) = mp[1], WV(m2) = mp[2], WV(m3) = mp[3];
#else
                // (agent scope: the words were written by atomics of waves that may run on another XCD)
                const uint64_t a = __hip_atomic_load((const JP_GLOBAL uint64_t *)mp, __ATOMIC_RELAXED, PW_SCOPE);
                const uint64_t c = __hip_atomic_load((const JP_GLOBAL uint64_t *)mp + 1, __ATOMIC_RELAXED, PW_SCOPE);
                m0 = (uint32_t)a, m1 = (uint32_t)(a >> 32), m2 = (uint32_t)c, m3 = (uint32_t)(c >> 32);
#endif
            }
        }
#ifndef JPGPU_HOST_EMULATION
        // The masks have arrived — said HERE, once per chunk: left to itself the compiler waits at their first use, inside the block loop,
        // with s_waitcnt vmcnt(0) (it cannot count the loop's own stores and atomics), and on this architecture that is a wait for the
        // atomics of the block before: ~1 us per block, 4 of a scan's 10 ms.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#endif
        for (uint32_t i = 0; i < n; i++, pw_walk_advance(at, 1u)) {
            PROGW_COUNT(blocks, 1);
            R.nz = ((uint64_t)wv_readlane(m1, i) << 32) | wv_readlane(m0, i);
            R.neg = ((uint64_t)wv_readlane(m3, i) << 32) | wv_readlane(m2, i);
            PwRefineBlock B{ss, eob_run, 0u, 0ull, 0ull};
            if (eob_run > 0u) {
                eob_run--;
                if ((R.nz & band) == 0ull) continue;  // (nothing of this band in the block: nothing to correct, nothing in the stream)
                pw_refine_non_zeroes(R, ss, below_end, end, 64u);
            } else {
#if !defined(PROGW_PORTABLE)
                WV32 delta;  // what a correction adds to each position's coefficient: -bit where it is negative
                WV_EACH { WV(delta) = ((R.neg >> lane) & 1ull) ? 0u - R.bit : R.bit; }
                for (;;) {
                    n_calls++;
                    const uint32_t code = pw_refine_fast(R, tab, B, delta, below_end, end);
                    if (code == 0u) break;
                    if (code == 2u) {
                        n_window++;
                        pw_refill(R.b);  // (into the next window)
                        continue;
                    }
                    n_generic++;
                    pw_refine_symbol(R, tab, B, below_end, end);
                    if (B.k >= end) break;
                }
#else
                do pw_refine_symbol(R, tab, B, below_end, end);
                while (B.k < end);
#endif
                if (__builtin_expect(B.err != 0u, 0)) {
                    pw_flag(status, B.err);
                    return false;
                }
                eob_run = B.eob_run;
            }
            const uint64_t new_nz = B.new_nz, new_neg = B.new_neg;
            // A new value where the block has a coefficient already: only a damaged stream does that (the walk ran out of zeros and ended
            // ON a non-zero coefficient, which the reference then REPLACES: src/decoder.rs:1251-1256) — the host's business, like every
            // other oddity; what is left is additions only.
            if (__builtin_expect((new_nz & R.nz) != 0ull, 0)) {
                pw_flag(status, PROG_ST_REPLACED);
                return false;
            }
            {
                const size_t blk = pw_walk_block(at);
                WV_EACH {
                    const uint32_t a = WV(R.acc), z = WV(unz);
                    // One atomic add per touched coefficient, on its dword: a correction +-bit in the low half goes in sign-extended (the
                    // coefficient keeps its sign, so the borrow of a -bit out of the low half and the extension's 0xffff in the high half
                    // cancel), a NEW value lands on a zero half and goes in zero-extended (no carry to cancel).
                    if (a) pw_add32(reinterpret_cast<uint32_t *>(coefs + blk * 64u) + (z >> 1), (z & 1u) ? a << 16 : (((new_nz >> lane) & 1ull) ? a & 0xffffu : a));
                    WV(R.acc) = 0u;
                    // Atomic OR, never a store of the whole word: a mask word covers all 63 AC positions of the block, a scan
                    // only its band — with a script such as Y 1-5 | Y 6-63 | refine 1-5 | refine 6-63 the wave of "refine 1-5" runs beside
                    // the wave of "6-63 first" on the same blocks.  (Lanes 0 and 1: both words in one instruction.)
                    if (new_nz && lane < 2u) pw_or64(masks + 2u * blk + lane, lane ? new_neg : new_nz);
                }
            }
        }
        pw_publish(y, cb + n);
    }
#ifndef JPGPU_HOST_EMULATION
    if (threadIdx.x == 0u) {
        uint32_t *rep = const_cast<uint32_t *>((const uint32_t *)s.report);
        rep[1] = n_calls, rep[2] = n_generic, rep[3] = n_window;
    }
#else
    (void)n_calls, (void)n_generic, (void)n_window;
#endif
    return true;
}

// ---- one wave: its scan (pipelined frames) or its track (the scans of a track one after the other) ------------------------------------------
__device__ inline void progw_run_track(const ProgTrack &tr) {
    for (uint32_t i = 0; i < tr.n_scans; i++) {
        const JP_GLOBAL ProgScan &s = *(const JP_GLOBAL ProgScan *)(tr.scans + i);
        PwSync y;
        pw_sync_open(y, s, tr.status);
        bool ok;
#ifndef JPGPU_HOST_EMULATION
        const uint64_t t_start = wall_clock64();  // (100 MHz; JPGPU_PROG_TIMES=1 prints what every scan of the first frame took)
#endif
        if (s.ss == 0u) ok = pw_scan_dc(s, tr.status, y);
        else if (s.ah == 0u) ok = pw_scan_ac_first(s, tr.status, y);
        else ok = pw_scan_ac_refine(s, tr.status, y);
#ifndef JPGPU_HOST_EMULATION
        if (threadIdx.x == 0u) const_cast<uint32_t *>((const uint32_t *)tr.scans[i].report)[0] = (uint32_t)(wall_clock64() - t_start);
#endif
        pw_publish(y, PROG_DONE);  // (also when the scan gave up: whoever waits for it must not wait for good — the image is the host's by then)
        if (!ok) return;
    }
}

}  // namespace jpgpu
