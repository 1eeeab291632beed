// huff_prog_core.hpp — the device decoder for progressive frames: one lane walks one track (huff_prog_job.hpp) scan by scan, block by
// block, symbol by symbol, exactly as the reference's decode_block / decode_block_successive_approximation / refine_non_zeroes do
// (src/decoder.rs:1086-1298; the host restatement this file was written against is csrc/host/frontend.cpp: decode_block,
// decode_block_refine, refine_non_zeroes with its bitmap walk).  Compiled by hipcc for huff_prog_kernel (huff.hip) and by g++ for
// tests/emu (one lane at a time).
//
// Anything the reference would answer with an error, and the few places where an INVALID stream makes it do something that depends
// on its table layout (a run that leaves the band, src/decoder.rs:1138-1146 — see decode_block in frontend.cpp), raise the image's
// status word instead: the host decoder, whose behaviour on odd streams is pinned, then decodes that image.
#pragma once
#include "huff_prog_job.hpp"
#include "pixel_math.hpp"

namespace jpgpu {

constexpr uint32_t PROG_RING_DWORDS = 32u;  // a lane's window on its scan: two halves of 16 dwords
struct ProgLds {
    uint32_t tab[64][PROG_LANE_DWORDS];  // per lane: the 8-bit lookup of its current scan's table (AC: 256 x u16; DC: two tables of 256 bytes)
    uint32_t ring[64][PROG_RING_DWORDS + 1u];  // per lane: the next 512-1,024 bits of its scan (skewed by one dword, like `tab`)
    uint8_t unzig[64];
};

// ---- memory operations other lanes / later scans of the same lane must see: past the L1 -------------------------------------------
// (through address-space-1 pointers: a generic pointer makes them flat_* operations, which count as LDS operations as well — every
// wait for a table or ring read would then wait for every store and atomic in flight)
__device__ __forceinline__ void prog_or32(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    (void)__hip_atomic_fetch_or((JP_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (result unused: the no-return form)
#endif
}
__device__ __forceinline__ void prog_add32(uint32_t *p, uint32_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p += v;
#else
    (void)__hip_atomic_fetch_add((JP_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void prog_or64(uint64_t *p, uint64_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p |= v;
#else
    (void)__hip_atomic_fetch_or((JP_GLOBAL uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void prog_and64(uint64_t *p, uint64_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p &= v;
#else
    (void)__hip_atomic_fetch_and((JP_GLOBAL uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ uint64_t prog_load64(const uint64_t *p) {
#ifdef JPGPU_HOST_EMULATION
    return *p;
#else
    return __hip_atomic_load((const JP_GLOBAL uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void prog_store64(uint64_t *p, uint64_t v) {
#ifdef JPGPU_HOST_EMULATION
    *p = v;
#else
    __hip_atomic_store((JP_GLOBAL uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void prog_store16(int16_t *p, int16_t v) { *(JP_GLOBAL int16_t *)p = v; }
__device__ __forceinline__ void prog_flag(uint32_t *status, uint32_t bits) { prog_or32(status, bits | PROG_ST_HOST); }

// (tests/emu only: how many steps of each kind a track takes — the numbers behind DESIGN.md's ns-per-step figures)
#ifdef JPGPU_PROG_COUNTERS
struct ProgCounters {
    unsigned long long symbols, corrections, blocks, refills;
};
extern ProgCounters g_prog_counters;
#define PROG_COUNT(what, n) (g_prog_counters.what += (n))
#else
#define PROG_COUNT(what, n) ((void)0)
#endif

typedef const JP_GLOBAL ProgScan &ProgScanRef;  // (the descriptors are read through address space 1 as well)
typedef const JP_GLOBAL ProgScanComp &ProgScanCompRef;

// ---- bit reader: zeros behind the scan's data ------------------------------------------------------------------------------------------
// A lane's refill takes ONE dword from the lane's ring in LDS; the ring is fed from global memory sixteen dwords at a time, requested
// half a ring ahead and written into the half the reader has just left.  Why not straight from global memory, a dword per refill
// requested one refill ahead (the first version, and what the chunk decoder of sequential scans does): on gfx9 a wait for a load is a
// wait for EVERY memory operation issued before it — and this loop is full of fire-and-forget stores and atomics that take a
// microsecond to retire; a refill every 32 bits paid that latency every few symbols (tower_progressive.jpg: 65 ms for the longest
// track).  LDS reads have a counter of their own; the global loads are waited for once per 512 bits.
struct ProgBits {
    // The unread bits, left-aligned in {hi, lo}; `pos` of them are held.  (A 32-bit window at a bit position — one v_alignbit_b32 per
    // look instead of 64-bit shifts — measured 3 % SLOWER: profiles/round5/08_progressive_reader_ab.txt; 64-bit shifts issue like
    // 32-bit ones on this part, round 4's 02_ubench_valu64.txt.)
    uint32_t hi, lo, nx;     // nx: the next dword of the stream, read from the ring when the last one was taken (so that a refill never waits for LDS)
    uint32_t pos;
    uint32_t r;              // dwords taken from the ring so far (nx included)
    JP_LDS uint32_t *ring;   // the lane's PROG_RING_DWORDS dwords
    const JP_GLOBAL v4u *src;  // the scan's data, 16 bytes at a time
    uint32_t n16;            // pieces of 16 bytes that hold data (what follows: zeros)
    v4u pre[4];              // the sixteen dwords that go into the ring next (pieces r / 4 + 4 .. + 7 at the time they are stored)
};
__device__ __forceinline__ v4u prog_piece(const ProgBits &b, uint32_t i) { return i < b.n16 ? b.src[i] : v4u{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void prog_ring_put(ProgBits &b, uint32_t half) {  // `pre` -> dwords [16 half, 16 half + 16) of the ring
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        JP_LDS uint32_t *d = b.ring + 16u * half + 4u * j;
        d[0] = b.pre[j].x, d[1] = b.pre[j].y, d[2] = b.pre[j].z, d[3] = b.pre[j].w;
    }
}
// the next dword of the stream (big-endian bit order), the ring fed as a half of it has been read
__device__ __forceinline__ uint32_t prog_next_dword(ProgBits &b) {
    PROG_COUNT(refills, 1);
    const uint32_t w = __builtin_bswap32(b.ring[b.r & (PROG_RING_DWORDS - 1u)]);
    b.r++;
    if ((b.r & 15u) == 0u) {  // a half of the ring has been read: what was requested when the other half was begun goes there ...
        prog_ring_put(b, ((b.r >> 4) - 1u) & 1u);
        const uint32_t p0 = (b.r >> 2) + 8u;  // ... and the half after the next is requested
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) b.pre[j] = prog_piece(b, p0 + j);
    }
    return w;
}
__device__ __forceinline__ void prog_bits_open(ProgBits &b, const uint8_t *data, uint32_t n_bytes, JP_LDS uint32_t *ring) {
    b.pos = 0;
    b.r = 0;
    b.ring = ring;
    b.src = (const JP_GLOBAL v4u *)data;  // (16-byte aligned slots, zero-filled behind the data: huff_stage_segment)
    b.n16 = (n_bytes + 15u) / 16u;
#pragma unroll
    for (uint32_t h = 0; h < 2u; h++) {
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) b.pre[j] = prog_piece(b, 4u * h + j);
        prog_ring_put(b, h);
    }
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) b.pre[j] = prog_piece(b, 8u + j);
    b.hi = b.lo = 0u;
    b.nx = prog_next_dword(b);
}
// afterwards more than 32 bits are held (a step reads at most 16 + 15; a batch of correction bits 32)
__device__ __forceinline__ void prog_refill(ProgBits &b) {
    if (b.pos <= 32u) {
        const uint64_t bits = (((uint64_t)b.hi << 32) | b.lo) | ((uint64_t)b.nx << (32u - b.pos));
        b.hi = (uint32_t)(bits >> 32);
        b.lo = (uint32_t)bits;
        b.pos += 32u;
        b.nx = prog_next_dword(b);
    }
}
__device__ __forceinline__ uint32_t prog_peek(const ProgBits &b, uint32_t n) { return n ? (uint32_t)((((uint64_t)b.hi << 32) | b.lo) >> (64u - n)) : 0u; }
__device__ __forceinline__ void prog_consume(ProgBits &b, uint32_t n) {
    const uint64_t bits = (((uint64_t)b.hi << 32) | b.lo) << n;
    b.hi = (uint32_t)(bits >> 32);
    b.lo = (uint32_t)bits;
    b.pos -= n;
}
__device__ __forceinline__ uint32_t prog_get(ProgBits &b, uint32_t n) {  // n <= 32, after a refill
    const uint32_t v = prog_peek(b, n);
    prog_consume(b, n);
    return v;
}
__device__ __forceinline__ int32_t prog_extend(uint32_t v, uint32_t n) {  // src/huffman.rs:165-173 (n >= 1)
    const int32_t vt = 1 << (n - 1u);
    return (int32_t)v < vt ? (int32_t)v + (int32_t)(0xffffffffu << n) + 1 : (int32_t)v;
}

// ---- a lane's table region --------------------------------------------------------------------------------------------------------
// AC scans: the table's 8-bit lookup (the head of ProgHuffTable: 512 bytes = 32 pieces of 16)
__device__ __forceinline__ void prog_load_table(JP_LDS uint32_t *T, const ProgHuffTable *t) {
    const JP_GLOBAL v4u *src = (const JP_GLOBAL v4u *)t;
    for (uint32_t i = 0; i < 32u; i++) {
        const v4u w = src[i];
        T[4u * i] = w.x, T[4u * i + 1u] = w.y, T[4u * i + 2u] = w.z, T[4u * i + 3u] = w.w;
    }
}
// DC first scans: the tables of ids 0 and 1 — what encoders use — each as 256 bytes: category | code length << 4 (0: the walk; codes of
// nine bits and more are rare in tables of twelve symbols).  Components with table ids 2 and 3 look their codes up in global memory.
__device__ __forceinline__ void prog_load_dc_tables(JP_LDS uint32_t *T, ProgScanRef s) {
    for (uint32_t t = 0; t < 2u; t++) {
        const JP_GLOBAL ProgHuffTable *src = (const JP_GLOBAL ProgHuffTable *)s.table[t];
        for (uint32_t i = 0; i < 64u; i++) {
            uint32_t w = 0;
            if (src)
                for (uint32_t j = 0; j < 4u; j++) {
                    const uint32_t e = src->lut[4u * i + j], len = e >> 8, sym = e & 0xffu;
                    // (a symbol above 15 cannot be packed; the planner refuses such tables — category > 11 is an error anyway)
                    w |= (len && sym < 16u ? (len << 4) | sym : 0u) << (8u * j);
                }
            T[64u * t + i] = w;
        }
    }
}
// the walk for codes the 8-bit lookup does not resolve (src/huffman.rs:44-58); `lds`: the table is the lane's LDS copy
template <class TablePtr>
__device__ __forceinline__ uint32_t prog_walk(ProgBits &b, TablePtr t, bool &bad) {
    const uint32_t b16 = prog_peek(b, 16);
    for (uint32_t i = 8; i < 16u; i++) {
        const int32_t code = (int32_t)(b16 >> (15u - i));
        if (code <= t->maxcode[i]) {
            prog_consume(b, i + 1u);
            const int32_t index = code + t->delta[i];
            if (index < 0 || index >= t->nvalues) {  // ("reference would panic": the host reports it)
                bad = true;
                return 0u;
            }
            return t->values[index];
        }
    }
    bad = true;  // "failed to decode huffman code"
    return 0u;
}
__device__ __forceinline__ uint32_t prog_decode_ac(ProgBits &b, const JP_LDS uint32_t *T, const JP_GLOBAL ProgHuffTable *gt, bool &bad) {
    PROG_COUNT(symbols, 1);
    const uint32_t e = reinterpret_cast<const JP_LDS uint16_t *>(T)[prog_peek(b, 8)];
    if (e >> 8) {
        prog_consume(b, e >> 8);
        return e & 0xffu;
    }
    return prog_walk(b, gt, bad);
}
__device__ __forceinline__ uint32_t prog_decode_dc(ProgBits &b, const JP_LDS uint32_t *T, uint32_t table, ProgScanRef s, bool &bad) {
    const JP_GLOBAL ProgHuffTable *gt = (const JP_GLOBAL ProgHuffTable *)s.table[table];
    if (table < 2u) {
        const uint32_t e = reinterpret_cast<const JP_LDS uint8_t *>(T)[256u * table + prog_peek(b, 8)];
        if (e >> 4) {
            prog_consume(b, e >> 4);
            return e & 15u;
        }
    } else {  // (table ids 2 and 3: CMYK files at most)
        const uint32_t e = gt->lut[prog_peek(b, 8)];
        if (e >> 8) {
            prog_consume(b, e >> 8);
            return e & 0xffu;
        }
    }
    return prog_walk(b, gt, bad);
}

// ---- scans of a track pipelined over lanes: stay behind the scans this one depends on, tell the ones that depend on this one ---------
#ifndef PROG_SPIN_SLEEP  // (A/B builds)
#define PROG_SPIN_SLEEP 120  // x 64 cycles: ~3 us
#endif
#ifndef PROG_SPIN_NAPS_MAX
#define PROG_SPIN_NAPS_MAX 16u
#endif
#ifndef PROG_PUBLISH_EVERY
#define PROG_PUBLISH_EVERY 64u
#endif
struct ProgSync {
    uint32_t *progress;
    const uint32_t *wait[3];
    uint32_t seen[3];   // what wait[i] said last (blocks below it are complete)
    uint32_t whole;
    uint32_t *status;
};
__device__ __forceinline__ void prog_sync_open(ProgSync &y, ProgScanRef s, uint32_t *status) {
    y.progress = s.progress;
    y.whole = s.wait_whole;
    y.status = status;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        y.wait[i] = s.wait[i];
        y.seen[i] = 0u;
    }
}
// may the lane work on block `bi` (its blocks counted in walk order)?  Spins until the scans it depends on are past it; false: it
// gave up (a producer that never moves: cannot happen while workgroups are dispatched in launch order — the launch puts producers in
// front — but a lane that spins for good would hang the device; the image then goes to the host).
__device__ __forceinline__ bool prog_wait_for(ProgSync &y, uint32_t bi) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (y.wait[i] == nullptr) continue;
        const uint32_t need = ((y.whole >> i) & 1u) ? PROG_DONE - 1u : bi;  // (a value ABOVE need lets the lane go: PROG_DONE always does)
        if (y.seen[i] > need) continue;
#ifdef JPGPU_HOST_EMULATION
        y.seen[i] = *y.wait[i];  // (tests/emu runs the lanes one after the other, producers first)
        if (y.seen[i] <= need) {
            prog_flag(y.status, PROG_ST_WAIT);
            return false;
        }
#else
        // (every look is a read at the L2; tens of thousands of lanes looking every microsecond slowed the walks of everybody — 4,096
        // frames 86 ms against 68 with rarer looks, profiles/round5/11_*: a lane that finds its producer behind sleeps longer each time,
        // up to ~50 us — by then the producer has moved a dozen blocks and the next look covers them all)
        uint32_t spins = 0, naps = 1;
        for (;;) {
            y.seen[i] = __hip_atomic_load((const JP_GLOBAL uint32_t *)y.wait[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (y.seen[i] > need) break;
            spins += naps;
            if (spins > (1u << 20)) {  // (seconds: a producer that never moves — cannot happen, see above — must not hang the device)
                prog_flag(y.status, PROG_ST_WAIT);
                return false;
            }
            for (uint32_t k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(PROG_SPIN_SLEEP);
            naps = naps < PROG_SPIN_NAPS_MAX ? naps * 2u : naps;
        }
#endif
    }
    return true;
}
// `done` blocks are complete: everything this lane has stored into them is made visible first (release)
__device__ __forceinline__ void prog_publish(ProgSync &y, uint32_t done) {
    if (y.progress == nullptr || (done & (PROG_PUBLISH_EVERY - 1u)) != 0u) return;
#ifdef JPGPU_HOST_EMULATION
    *y.progress = done;
#else
    __hip_atomic_store((JP_GLOBAL uint32_t *)y.progress, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void prog_publish_end(ProgSync &y) {
    if (y.progress == nullptr) return;
#ifdef JPGPU_HOST_EMULATION
    *y.progress = PROG_DONE;
#else
    __hip_atomic_store((JP_GLOBAL uint32_t *)y.progress, PROG_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// where block (mx, my) x (hp, vp) of scan component c lies
__device__ __forceinline__ size_t prog_block_index(ProgScanCompRef c, uint32_t mx, uint32_t my, uint32_t hp, uint32_t vp) {
    return (size_t)(my * c.v + vp) * c.block_w + (mx * c.h + hp);
}

// ---- DC scans (ss == se == 0; one to four components, src/decoder.rs:1100-1126 and :1181-1190) ------------------------------------
// Returns false if the scan raised the status word.
__device__ inline bool prog_scan_dc(ProgScanRef s, JP_LDS uint32_t *T, JP_LDS uint32_t *ring, uint32_t *status, ProgSync &y) {
    ProgBits b;
    uint32_t bi = 0;
    prog_bits_open(b, s.data, s.n_bytes, ring);
    const bool first = s.ah == 0;
    const uint32_t al = s.al, rows = s.rows, cols = s.cols, ncomp = s.ncomp;
    if (first) prog_load_dc_tables(T, s);
    uint64_t pred = 0;  // four 16-bit predictors (wrapping_add on i16 in the reference)
    bool bad = false;
    for (uint32_t my = 0; my < rows; my++)
        for (uint32_t mx = 0; mx < cols; mx++)
            for (uint32_t c = 0; c < ncomp; c++) {
                ProgScanCompRef sc = s.comp[c];
                const uint32_t ch = sc.h, cv = sc.v, ctable = sc.table;
                for (uint32_t vp = 0; vp < cv; vp++)
                    for (uint32_t hp = 0; hp < ch; hp++) {
                        int16_t *co = sc.coefs + prog_block_index(sc, mx, my, hp, vp) * 64u;
                        if (!prog_wait_for(y, bi)) return false;
                        prog_refill(b);
                        if (first) {
                            const uint32_t cat = prog_decode_dc(b, T, ctable, s, bad);
                            if (bad || cat > 11u) {  // "invalid DC difference magnitude category"
                                prog_flag(status, bad ? PROG_ST_BAD_CODE : PROG_ST_BAD_DC);
                                return false;
                            }
                            uint32_t diff = 0;
                            if (cat) {
                                prog_refill(b);
                                diff = (uint32_t)prog_extend(prog_get(b, cat), cat);
                            }
                            const uint32_t p = (uint32_t)((pred >> (16u * c)) + diff) & 0xffffu;
                            pred = (pred & ~(0xffffull << (16u * c))) | ((uint64_t)p << (16u * c));
                            prog_store16(co, (int16_t)(uint16_t)(p << al));
                        } else if (prog_get(b, 1)) {
                            prog_or32(reinterpret_cast<uint32_t *>(co), 1u << al);  // co[0] |= bit (the low half of the block's first dword)
                        }
                        prog_publish(y, ++bi);
                    }
            }
    return true;
}

// ---- AC first scan (one component, ah == 0, ss >= 1; src/decoder.rs:1128-1172) -----------------------------------------------------
__device__ inline bool prog_scan_ac_first(ProgScanRef s, JP_LDS uint32_t *T, JP_LDS uint32_t *ring, const JP_LDS uint8_t *unzig, uint32_t *status, ProgSync &y) {
    ProgBits b;
    uint32_t bi = 0;
    prog_bits_open(b, s.data, s.n_bytes, ring);
    prog_load_table(T, s.table[0]);
    const JP_GLOBAL ProgHuffTable *const gt = (const JP_GLOBAL ProgHuffTable *)s.table[0];
    // (everything the loop needs of the descriptor, once: one component, h = v = 1)
    int16_t *const coefs = s.comp[0].coefs;
    uint64_t *const masks = s.comp[0].masks;
    const uint32_t block_w = s.comp[0].block_w, rows = s.rows, cols = s.cols, ss = s.ss, se = s.se, al = s.al;
    uint32_t eob_run = 0;
    bool bad = false;
    for (uint32_t my = 0; my < rows; my++)
        for (uint32_t mx = 0; mx < cols; mx++) {
            if (eob_run > 0u) {
                eob_run--;
                prog_publish(y, ++bi);
                continue;
            }
            if (!prog_wait_for(y, bi)) return false;
            const size_t blk = (size_t)my * block_w + mx;
            int16_t *co = coefs + blk * 64u;
            uint64_t nz = 0, neg = 0;
            uint32_t k = ss;
            while (k <= se) {
                prog_refill(b);
                const uint32_t rs = prog_decode_ac(b, T, gt, bad), r = rs >> 4, sz = rs & 15u;
                if (bad) {
                    prog_flag(status, PROG_ST_BAD_CODE);
                    return false;
                }
                if (sz == 0u) {
                    if (r == 15u) {
                        k += 16u;
                        continue;
                    }
                    eob_run = (1u << r) - 1u;
                    if (r) {
                        prog_refill(b);
                        eob_run += prog_get(b, r);
                    }
                    break;
                }
                k += r;
                // a run that leaves the band: what the reference then does with the magnitude bits depends on its table layout
                // (frontend.cpp, decode_block) — the host's business; so is a magnitude that could make a later correction carry
                if (k > se || sz + al > 14u) {
                    prog_flag(status, k > se ? PROG_ST_BAND : PROG_ST_RANGE);
                    return false;
                }
                prog_refill(b);
                const int32_t v = prog_extend(prog_get(b, sz), sz);
                prog_store16(co + unzig[k], (int16_t)(uint16_t)((uint32_t)v << al));
                nz |= 1ull << k;
                if (v < 0) neg |= 1ull << k;
                k++;
            }
            if (nz) {  // (OR, not store: another first scan of this track may own other bands of the block)
                prog_or64(masks + 2u * blk, nz);
                if (neg) prog_or64(masks + 2u * blk + 1u, neg);
            }
            prog_publish(y, ++bi);
        }
    return true;
}

// ---- AC refinement scan (one component, ah > 0, ss >= 1; src/decoder.rs:1192-1298) -------------------------------------------------
struct ProgRefine {
    ProgBits b;
    uint64_t nz, neg;  // the current block's masks as they were when the scan reached it
    int16_t *co;
    uint32_t bit;      // 1 << al
};
// refine_non_zeroes(start .. end-1, zrl): a correction bit for every non-zero coefficient until `zrl` zero ones have been passed;
// returns where the walk stopped (the (zrl + 1)-th zero coefficient, or end - 1)
__device__ __forceinline__ uint32_t prog_refine_non_zeroes(ProgRefine &R, const JP_LDS uint8_t *unzig, uint32_t start, uint32_t end, uint32_t zrl) {
    if (start >= end) return end - 1u;
    const uint64_t below_end = end >= 64u ? ~0ull : ((1ull << end) - 1ull), range = below_end & ~((1ull << start) - 1ull);
    uint64_t zeros = ~R.nz & range;
    uint32_t stop = end;
    bool hit = false;
    if ((uint32_t)__builtin_popcountll(zeros) > zrl) {
        for (uint32_t skip = 0; skip < zrl; skip++) zeros &= zeros - 1ull;
        stop = (uint32_t)__builtin_ctzll(zeros);
        hit = true;
    }
    uint64_t todo = R.nz & range & (stop >= 64u ? ~0ull : ((1ull << stop) - 1ull));
    uint32_t n = (uint32_t)__builtin_popcountll(todo);
    while (n) {  // the correction bits, up to 32 at a time: the first coefficient's bit is the first in the stream
        const uint32_t take = n < 32u ? n : 32u;
        prog_refill(R.b);
        const uint32_t corr = prog_get(R.b, take);
        PROG_COUNT(corrections, take);
        for (uint32_t j = 0; j < take; j++) {
            const uint32_t i = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            if ((corr >> (take - 1u - j)) & 1u) {
                // c += sign(c) * bit.  (c & bit) == 0 always: the planner admits only streams whose scans refine a band one bit at
                // a time (plan_progressive_scans), so every non-zero coefficient is a multiple of 2 * bit here; and |c| < 2^14
                // (first scans check sz + al <= 14), so the addition cannot carry out of the coefficient's half of the dword.
                const uint32_t z = unzig[i];
                const uint32_t delta = ((R.neg >> i) & 1ull) ? 0u - R.bit : R.bit;
                prog_add32(reinterpret_cast<uint32_t *>(R.co) + (z >> 1), (z & 1u) ? delta << 16 : delta);
            }
        }
        n -= take;
    }
    return hit ? stop : end - 1u;
}

__device__ inline bool prog_scan_ac_refine(ProgScanRef s, JP_LDS uint32_t *T, JP_LDS uint32_t *ring, const JP_LDS uint8_t *unzig, uint32_t *status, ProgSync &y) {
    ProgRefine R;
    uint32_t bi = 0;
    prog_bits_open(R.b, s.data, s.n_bytes, ring);
    prog_load_table(T, s.table[0]);
    const JP_GLOBAL ProgHuffTable *const gt = (const JP_GLOBAL ProgHuffTable *)s.table[0];
    int16_t *const coefs = s.comp[0].coefs;
    uint64_t *const masks = s.comp[0].masks;
    const uint32_t block_w = s.comp[0].block_w, rows = s.rows, cols = s.cols, ss = s.ss;
    R.bit = 1u << s.al;
    uint32_t eob_run = 0;
    bool bad = false;
    const uint32_t end = (uint32_t)s.se + 1u;
    // the masks of the block after this one are requested while this one is decoded
    // (pipelined scans: only once the scans this one depends on are past that block — its masks are what they leave behind)
    uint64_t nz_next = 0, neg_next = 0;
    if (rows && cols) {
        if (!prog_wait_for(y, 0u)) return false;
        nz_next = prog_load64(masks);
        neg_next = prog_load64(masks + 1u);
    }
    for (uint32_t my = 0; my < rows; my++)
        for (uint32_t mx = 0; mx < cols; mx++) {
            const size_t blk = (size_t)my * block_w + mx;
            PROG_COUNT(blocks, 1);
            R.co = coefs + blk * 64u;
            R.nz = nz_next;
            R.neg = neg_next;
            {
                uint32_t nx = mx + 1u, ny = my;
                if (nx == cols) nx = 0u, ny++;
                if (ny < rows) {
                    if (!prog_wait_for(y, bi + 1u)) return false;
                    const size_t nb = (size_t)ny * block_w + nx;
                    nz_next = prog_load64(masks + 2u * nb);
                    neg_next = prog_load64(masks + 2u * nb + 1u);
                }
            }
            uint64_t new_nz = 0, new_neg = 0;
            if (eob_run > 0u) {
                eob_run--;
                prog_refine_non_zeroes(R, unzig, ss, end, 64u);
                prog_publish(y, ++bi);
                continue;
            }
            uint32_t k = ss;
            while (k < end) {
                prog_refill(R.b);
                const uint32_t rs = prog_decode_ac(R.b, T, gt, bad), r = rs >> 4, sz = rs & 15u;
                if (bad) {
                    prog_flag(status, PROG_ST_BAD_CODE);
                    return false;
                }
                uint32_t zrl = r;
                int32_t value = 0;
                if (sz == 0u) {
                    if (r != 15u) {
                        eob_run = (1u << r) - 1u;
                        if (r) {
                            prog_refill(R.b);
                            eob_run += prog_get(R.b, r);
                        }
                        zrl = 64u;
                    }
                } else if (sz == 1u) {
                    prog_refill(R.b);
                    value = prog_get(R.b, 1) ? (int32_t)R.bit : -(int32_t)R.bit;
                } else {  // "unexpected huffman code"
                    prog_flag(status, PROG_ST_REFINE_SYMBOL);
                    return false;
                }
                k = prog_refine_non_zeroes(R, unzig, k, end, zrl);
                if (value != 0) {
                    prog_store16(R.co + unzig[k], (int16_t)value);
                    new_nz |= 1ull << k;
                    if (value < 0) new_neg |= 1ull << k;
                }
                k++;
            }
            if (new_nz) {
                // Atomic OR, never a store of the whole word: a mask word covers all 63 AC positions of the block, a scan
                // only its band — with a script such as Y 1-5 | Y 6-63 | refine 1-5 | refine 6-63 the lane of "refine 1-5" runs beside the
                // lane of "6-63 first" on the same blocks, and a store of `R.nz | new_nz` (R.nz read a block ahead) would wipe out what
                // the other lane ORed in between; the later refinement of 6-63 would then count too few non-zero coefficients.
                prog_or64(masks + 2u * blk, new_nz);
                // (a damaged stream can make the walk end ON a non-zero coefficient — at the band's last position, when it runs out of
                // zeros — and the new value then REPLACES it, src/decoder.rs:1251-1256: the sign is the new value's)
                if (new_neg) prog_or64(masks + 2u * blk + 1u, new_neg);
                const uint64_t flip = R.neg & new_nz & ~new_neg;  // (bits of this scan's own band only)
                if (flip) prog_and64(masks + 2u * blk + 1u, ~flip);
            }
            prog_publish(y, ++bi);
        }
    return true;
}

// ---- one lane: its track --------------------------------------------------------------------------------------------------------------
__device__ inline void prog_run_track(JP_LDS ProgLds &L, uint32_t lane, const ProgTrack &tr) {
    JP_LDS uint32_t *T = L.tab[lane], *ring = L.ring[lane];
    for (uint32_t i = 0; i < tr.n_scans; i++) {
        ProgScanRef s = *(const JP_GLOBAL ProgScan *)(tr.scans + i);
        ProgSync y;
        prog_sync_open(y, s, tr.status);
        bool ok;
        if (s.ss == 0u) ok = prog_scan_dc(s, T, ring, tr.status, y);
        else if (s.ah == 0u) ok = prog_scan_ac_first(s, T, ring, L.unzig, tr.status, y);
        else ok = prog_scan_ac_refine(s, T, ring, L.unzig, tr.status, y);
        prog_publish_end(y);  // (also when the scan gave up: whoever waits for it must not wait for good — the image is the host's by then)
        if (!ok) return;
    }
}

}  // namespace jpgpu
