// frontend.cpp — see frontend.hpp.  Plain C++17, no HIP: this translation unit is the host half
// of the decoder (marker loop + entropy decoding) and is also exercised on GPU-less machines
// through jpgpu_decoder_decode_coefficients.
#include "frontend.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace jpgpu {
namespace host {

namespace {

[[noreturn]] void fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[200];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw DecodeError{code, buf};
}

// src/decoder.rs:27-36
const uint8_t kUnzigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                               41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                               30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- byte source: std::io::Read over an in-memory stream (src/lib.rs:56-66) --------------------
struct ByteSource {
    const uint8_t *p = nullptr;
    size_t len = 0, pos = 0;
    uint8_t u8() {
        if (pos >= len) fail(JPGPU_ERR_IO, "failed to fill whole buffer");
        return p[pos++];
    }
    uint16_t u16be() {
        if (len - pos < 2) {
            pos = len;
            fail(JPGPU_ERR_IO, "failed to fill whole buffer");
        }
        uint16_t v = (uint16_t)((p[pos] << 8) | p[pos + 1]);
        pos += 2;
        return v;
    }
    const uint8_t *take(size_t n) {
        if (len - pos < n) {
            pos = len;
            fail(JPGPU_ERR_IO, "failed to fill whole buffer");
        }
        const uint8_t *q = p + pos;
        pos += n;
        return q;
    }
};

// ---- markers (src/marker.rs) ----------------------------------------------------------------------
enum class Mk : uint8_t { SOF, JPG, DHT, DAC, RST, SOI, EOI, SOS, DQT, DNL, DRI, DHP, EXP, APP, JPGn, COM, TEM, RES };
struct Marker {
    Mk kind;
    uint8_t n;
};
Marker marker_from(uint8_t b) {  // b != 0x00, 0xFF (src/marker.rs:63-135)
    if (b == 0x01) return {Mk::TEM, 0};
    if (b <= 0xBF) return {Mk::RES, 0};
    switch (b) {
    case 0xC4: return {Mk::DHT, 0};
    case 0xC8: return {Mk::JPG, 0};
    case 0xCC: return {Mk::DAC, 0};
    case 0xD8: return {Mk::SOI, 0};
    case 0xD9: return {Mk::EOI, 0};
    case 0xDA: return {Mk::SOS, 0};
    case 0xDB: return {Mk::DQT, 0};
    case 0xDC: return {Mk::DNL, 0};
    case 0xDD: return {Mk::DRI, 0};
    case 0xDE: return {Mk::DHP, 0};
    case 0xDF: return {Mk::EXP, 0};
    case 0xFE: return {Mk::COM, 0};
    default: break;
    }
    if (b <= 0xCF) return {Mk::SOF, (uint8_t)(b - 0xC0)};
    if (b <= 0xD7) return {Mk::RST, (uint8_t)(b - 0xD0)};
    if (b >= 0xE0 && b <= 0xEF) return {Mk::APP, (uint8_t)(b - 0xE0)};
    return {Mk::JPGn, (uint8_t)(b - 0xF0)};
}

// ---- Huffman tables (src/huffman.rs:181-285) ---------------------------------------------------
inline int16_t extend(uint16_t value, uint8_t count) {  // :165-173
    const uint16_t vt = (uint16_t)(1u << (count - 1));
    return value < vt ? (int16_t)((int)value + (int)(int16_t)(uint16_t)(0xFFFFu << count) + 1) : (int16_t)value;
}

// Lookahead of the code LUT and of the fused AC table.  The reference uses 8 bits (src/huffman.rs:16); the tables are
// caches of the canonical decoding procedure, so a wider lookahead decodes every valid stream to the same symbols
// with fewer trips through the bit-serial tail (n1 of SURVEY §8f: "wider LUTs").
constexpr int kLutBits = 10;
constexpr int kLutSize = 1 << kLutBits;

struct HuffTable {
    bool present = false;
    bool is_ac = false;
    int nvalues = 0;
    uint8_t bits[16];  // code counts per length: with `values` and `is_ac` everything else here is a function of them
    uint8_t values[256];
    int32_t delta[16], maxcode[16];
    uint8_t lut_value[kLutSize], lut_size[kLutSize];
    int16_t ac_value[kLutSize];
    uint8_t ac_run_size[kLutSize];

    void build(const uint8_t bits[16], const uint8_t *vals, int n, bool ac) {
        uint8_t size_of[256];
        uint16_t code_of[256];
        int count = 0;
        for (int len = 1; len <= 16; len++)
            for (int k = 0; k < bits[len - 1]; k++) {
                if (count >= 256) fail(JPGPU_ERR_FORMAT, "bad huffman table");
                size_of[count++] = (uint8_t)len;
            }
        if (count == 0 || count != n) fail(JPGPU_ERR_FORMAT, "bad huffman table");
        // derive_huffman_codes, :256-285
        uint32_t code = 0;
        uint8_t cur = size_of[0];
        for (int i = 0; i < count; i++) {
            while (cur < size_of[i]) {
                code <<= 1;
                cur++;
            }
            if (code >= (1u << size_of[i])) fail(JPGPU_ERR_FORMAT, "bad huffman code length");
            code_of[i] = (uint16_t)code++;
        }
        uint8_t bits_copy[16];
        memcpy(bits_copy, bits, 16);  // (`bits` may point into this object)
        memset(this, 0, sizeof(*this));
        present = true;
        is_ac = ac;
        nvalues = n;
        memcpy(this->bits, bits_copy, 16);
        memcpy(values, vals, (size_t)n);
        int j = 0;
        for (int i = 0; i < 16; i++) {
            delta[i] = 0;
            maxcode[i] = -1;
            if (bits[i]) {
                delta[i] = j - (int32_t)code_of[j];
                j += bits[i];
                maxcode[i] = code_of[j - 1];
            }
        }
        // The reference's 8-bit tables (src/huffman.rs:190-243), then the wide ones as an exact cache of its decoding
        // PROCEDURE — 8-bit LUT first, then the maxcode walk (:31-58) — evaluated for every kLutBits-bit prefix.
        // Derived this way (not from the code list) the wide tables also agree with the reference on malformed
        // tables, where the maxcode walk accepts bit patterns that are not codes (tests/golden/crashtest).
        uint8_t lut8_value[256], lut8_size[256];
        memset(lut8_value, 0, sizeof(lut8_value));
        memset(lut8_size, 0, sizeof(lut8_size));
        for (int i = 0; i < count; i++)
            if (size_of[i] <= 8) {
                const int pad = 8 - size_of[i], first = code_of[i] << pad;
                for (int b = 0; b < (1 << pad); b++) {
                    lut8_value[first + b] = vals[i];
                    lut8_size[first + b] = size_of[i];
                }
            }
        for (int idx = 0; idx < kLutSize; idx++) {
            const int i8 = idx >> (kLutBits - 8);
            if (lut8_size[i8]) {
                lut_value[idx] = lut8_value[i8];
                lut_size[idx] = lut8_size[i8];
                continue;
            }
            for (int i = 8; i < kLutBits; i++) {
                const int32_t code = idx >> (kLutBits - 1 - i);
                if (code <= maxcode[i]) {
                    const int32_t index = code + delta[i];
                    if (index >= 0 && index < nvalues) {  // else: left to the walk at decode time, which reports it
                        lut_value[idx] = values[index];
                        lut_size[idx] = (uint8_t)(i + 1);
                    }
                    break;
                }
            }
        }
        if (ac)  // fused run/size/value table, :224-243: the reference's entries (code + magnitude within 8 bits), then
                 // what its fallback — decode() followed by receive_extend() — yields when both fit the wide lookahead
            for (int idx = 0; idx < kLutSize; idx++) {
                const uint8_t v = lut_value[idx], sz = lut_size[idx], run = v >> 4, cat = v & 15;
                if (sz > 0 && cat > 0 && sz + cat <= kLutBits) {
                    const uint16_t raw = (uint16_t)((((unsigned)idx << sz) & (kLutSize - 1)) >> (kLutBits - cat));
                    ac_value[idx] = extend(raw, cat);
                    ac_run_size[idx] = (uint8_t)((run << 4) | (sz + cat));
                }
            }
    }
};

// A batch of files from one encoder repeats the same four tables in every file (most encoders write Annex K's): the tables of
// the previous files parsed by this thread are kept, and a table whose definition — code counts, values, class — matches one of
// them is copied instead of derived (a 1080p file's header phase: 15 -> 5 us; 4,096 files per call spend it before anything
// else can start).  Malformed definitions are never cached: build() throws before the entry is written.
inline bool same_definition(const HuffTable &t, const uint8_t bits[16], const uint8_t *vals, int n, bool ac) {
    return t.present && t.is_ac == ac && t.nvalues == n && memcmp(t.bits, bits, 16) == 0 && memcmp(t.values, vals, (size_t)n) == 0;
}
// Where the tables live (round 4; rounds 2-3 kept eight 5.5 kB copies per calling thread — the C API is driven by a thousand
// decoding threads in tests/test_gpu_concurrency.py, and whatever a thread-local holds stays until its thread ends):
// ONE immutable copy per definition in a process-wide registry, and per thread POINTERS to the tables it used last — the hit
// path takes no lock (a first version with a reader-writer lock around the registry made the header phase of 4,096 files 18 ms
// instead of 1.8: 32 threads x 16 k lock operations on one cache line).
// Round 5: the registry is cut into 64 shards by a hash of the definition, a mutex and four tables each.  With ONE mutex and 64
// tables behind it a batch of progressive files from a real encoder — twelve OPTIMISED tables per file, no two files alike — sent
// every thread through that mutex two dozen times per file: the header phase of 4,096 such files took 108 ms on 16 CPUs (10 ms of
// work), 8 threads planned 87 k files/s where one plans 26 k.
inline uint32_t definition_hash(const uint8_t bits[16], const uint8_t *vals, int n, bool ac) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)(ac ? 1 : 0);
    auto mix = [&](uint64_t x) { h = (h ^ x) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; };
    uint64_t w;
    memcpy(&w, bits, 8), mix(w);
    memcpy(&w, bits + 8, 8), mix(w);
    int i = 0;
    for (; i + 8 <= n; i += 8) memcpy(&w, vals + i, 8), mix(w);
    w = 0;
    if (i < n) memcpy(&w, vals + i, (size_t)(n - i)), mix(w);
    mix((uint64_t)n);
    return (uint32_t)(h >> 32);
}
// -> the registry's own (immutable) copy of the table: callers keep the pointer, nothing of a table's 5.5 kB is copied or kept per thread
inline std::shared_ptr<const HuffTable> build_cached(const uint8_t bits[16], const uint8_t *vals, int n, bool ac) {
    constexpr int kSlots = 32, kShards = 64, kPerShard = 4;  // (slots: a progressive file of libjpeg's default script defines twelve tables)
    thread_local std::shared_ptr<const HuffTable> mine[kSlots];
    thread_local uint32_t mine_hash[kSlots];
    thread_local int next = 0;
    const bool sane = n > 0 && n <= 256;
    const uint32_t hash = sane ? definition_hash(bits, vals, n, ac) : 0u;
    if (sane)
        for (int i = 0; i < kSlots; i++)
            if (mine[i] && mine_hash[i] == hash && same_definition(*mine[i], bits, vals, n, ac)) return mine[i];
    // (the registry is leaked on purpose: pool threads of a host that is shutting down may still be in here while statics are
    // destroyed — ADVICE r4)
    struct alignas(64) Shard {
        std::mutex m;
        std::shared_ptr<const HuffTable> e[kPerShard];
        uint32_t hash[kPerShard] = {0, 0, 0, 0};
        int next = 0;
    };
    static Shard *const shards = new Shard[kShards];
    Shard &sh = shards[hash % kShards];
    std::shared_ptr<const HuffTable> found;
    if (sane) {
        std::lock_guard<std::mutex> g(sh.m);
        for (int i = 0; i < kPerShard && !found; i++)
            if (sh.e[i] && sh.hash[i] == hash && same_definition(*sh.e[i], bits, vals, n, ac)) found = sh.e[i];
    }
    if (!found) {
        auto fresh = std::make_shared<HuffTable>();
        fresh->build(bits, vals, n, ac);  // (throws on a malformed definition: nothing is cached)
        found = fresh;
        std::lock_guard<std::mutex> g(sh.m);
        sh.e[sh.next] = found;
        sh.hash[sh.next] = hash;
        sh.next = (sh.next + 1) % kPerShard;
    }
    mine[next] = found;
    mine_hash[next] = hash;
    next = (next + 1) % kSlots;
    return found;
}

// Annex K default tables for MJPEG (src/huffman.rs:295-346)
const uint8_t kK3Bits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kK4Bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kK5Bits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D};
const uint8_t kK6Bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
// The 162 AC symbols of K.5 / K.6 are generated rather than listed: both tables contain exactly
// the run/size symbols (r,s) with s in 1..10 for r in 0..15, plus EOB (0x00) and ZRL (0xF0), in
// the code-length order given by Annex K.  The orders are stored compactly below.
const uint8_t kK5Vals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1, 0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3,
    0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3,
    0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE1, 0xE2,
    0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA};
const uint8_t kK6Vals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xA1, 0xB1, 0xC1, 0x09, 0x23, 0x33, 0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1,
    0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19, 0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A,
    0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA,
    0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA,
    0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA};

// ---- bit reader (src/huffman.rs:14-161) -----------------------------------------------------------
// Same refill policy as the reference (fill while num_bits <= 56 whenever a consumer finds too
// few bits), because *when* the end of the stream is discovered decides between Ok and Io error.
struct BitReader {
    uint64_t bits = 0;
    uint8_t nbits = 0;
    bool has_marker = false;
    Marker marker{Mk::RES, 0};

    void refill(ByteSource &src) {
        // eight bytes at once while the stream is ordinary entropy-coded data (no 0xFF among them, no pending
        // marker, not near the end); anything else goes through the byte loop below, which holds the marker logic
        if (!has_marker && nbits <= 56 && src.len - src.pos >= 8) {
            uint64_t x;
            memcpy(&x, src.p + src.pos, 8);
            const uint64_t nx = ~x;  // a 0xFF byte <=> a zero byte here
            if (((nx - 0x0101010101010101ull) & ~nx & 0x8080808080808080ull) == 0) {
                const uint32_t k = (64u - nbits) >> 3;  // bytes the loop would append (until nbits > 56)
                const uint64_t be = __builtin_bswap64(x);
                bits |= (be >> (64u - 8u * k)) << (64u - nbits - 8u * k);
                nbits = (uint8_t)(nbits + 8u * k);
                src.pos += k;
                return;
            }
        }
        while (nbits <= 56) {
            uint8_t byte = has_marker ? 0 : src.u8();
            if (byte == 0xFF) {
                uint8_t next = src.u8();
                if (next != 0x00) {
                    while (next == 0xFF) next = src.u8();
                    if (next == 0x00) fail(JPGPU_ERR_FORMAT, "FF 00 found where marker was expected");
                    marker = marker_from(next);
                    has_marker = true;
                    continue;
                }
            }
            bits |= (uint64_t)byte << (56 - nbits);
            nbits = (uint8_t)(nbits + 8);
        }
    }
    uint16_t peek(uint8_t n) const { return n ? (uint16_t)((bits >> (64 - n)) & ((1u << n) - 1)) : 0; }
    void consume(uint8_t n) {
        bits <<= n;
        nbits = (uint8_t)(nbits - n);
    }
    uint8_t decode(ByteSource &src, const HuffTable &t) {  // :31-58
        if (nbits < 16) refill(src);
        const uint16_t idx = peek(kLutBits);
        if (const uint8_t size = t.lut_size[idx]) {
            consume(size);
            return t.lut_value[idx];
        }
        const uint16_t b16 = peek(16);
        for (int i = 8; i < 16; i++) {  // (from 8: prefixes the wide table left unresolved include the walk's own error cases)
            const int32_t code = b16 >> (15 - i);
            if (code <= t.maxcode[i]) {
                consume((uint8_t)(i + 1));
                const int32_t index = code + t.delta[i];
                if (index < 0 || index >= t.nvalues) fail(JPGPU_ERR_INTERNAL, "reference would panic: huffman value index");
                return t.values[index];
            }
        }
        fail(JPGPU_ERR_FORMAT, "failed to decode huffman code");
    }
    // :60-78, on the wide table.  Does not consume: the caller needs `code_bits` for entries the reference's 8-bit
    // table does not have (`total_bits` > 8) — there the reference decodes the symbol first and reads the magnitude
    // bits only if the coefficient index is still inside the band (decode_block).
    bool peek_fast_ac(ByteSource &src, const HuffTable &t, int16_t &value, uint8_t &run, uint8_t &total_bits, uint8_t &code_bits) {
        if (!t.is_ac) return false;
        if (nbits < kLutBits) refill(src);
        const uint16_t idx = peek(kLutBits);
        const uint8_t rs = t.ac_run_size[idx];
        if (!rs) return false;
        run = rs >> 4;
        total_bits = rs & 15;
        code_bits = t.lut_size[idx];
        value = t.ac_value[idx];
        return true;
    }
    uint16_t get_bits(ByteSource &src, uint8_t n) {  // :80-90
        if (nbits < n) refill(src);
        const uint16_t v = peek(n);
        consume(n);
        return v;
    }
    int16_t receive_extend(ByteSource &src, uint8_t n) { return extend(get_bits(src, n), n); }  // :92-96
    bool take_marker(ByteSource &src, Marker &m) {  // :103-105
        refill(src);
        const bool had = has_marker;
        m = marker;
        has_marker = false;
        return had;
    }
    void reset() {
        bits = 0;
        nbits = 0;
    }
};

struct FrameInfo {  // src/parser.rs:49-61
    bool is_baseline = false, is_differential = false, arithmetic = false;
    int coding_process = JPGPU_CODING_DCT_SEQUENTIAL;
    uint8_t precision = 0;
    uint16_t image_w = 0, image_h = 0, output_w = 0, output_h = 0, mcu_w = 0, mcu_h = 0;
    std::vector<jpgpu_component> components;
};
struct ScanInfo {  // src/parser.rs:63-73
    int n = 0;
    int component_indices[4], dc_tables[4], ac_tables[4];
    uint8_t ss_start = 0, ss_end = 0, ah = 0, al = 0;
};

uint16_t ceil_div(uint32_t x, uint32_t y) {  // src/parser.rs:282-290
    if (x == 0 || y == 0) fail(JPGPU_ERR_FORMAT, "invalid dimensions");
    return (uint16_t)(1 + (x - 1) / y);
}
void update_component_sizes(uint16_t w, uint16_t h, std::vector<jpgpu_component> &comps, uint16_t &mcu_w, uint16_t &mcu_h) {
    uint32_t h_max = 0, v_max = 0;  // src/parser.rs:292-310
    for (auto &c : comps) {
        h_max = std::max<uint32_t>(h_max, c.horizontal_sampling_factor);
        v_max = std::max<uint32_t>(v_max, c.vertical_sampling_factor);
    }
    mcu_w = ceil_div(w, h_max * 8);
    mcu_h = ceil_div(h, v_max * 8);
    for (auto &c : comps) {
        c.size_width = ceil_div((uint32_t)w * c.horizontal_sampling_factor * c.dct_scale, h_max * 8);
        c.size_height = ceil_div((uint32_t)h * c.vertical_sampling_factor * c.dct_scale, v_max * 8);
        c.block_width = (uint16_t)(mcu_w * c.horizontal_sampling_factor);
        c.block_height = (uint16_t)(mcu_h * c.vertical_sampling_factor);
    }
}

}  // namespace

// The accumulation planes of progressive frames (2 bytes per coefficient: 6 MB for a 1080p image) come from a small pool:
// fresh vectors of that size are mmap'ed, page-faulted in and munmap'ed again by the allocator — with a few dozen decoder
// threads that was 30-45 % of the time a batch of progressive files took (it showed up in the NEXT call's header phase,
// where the previous call's front-ends are destroyed).  JPGPU_HOST_POOL_MB bounds what the pool keeps (default 1024).
namespace {
class CoefPool {
public:
    std::vector<int16_t> take(size_t n) {
        std::vector<int16_t> v;
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].capacity() >= n && free_[i].capacity() <= n + n / 4 + 4096) {
                    v.swap(free_[i]);
                    bytes_ -= v.capacity() * sizeof(int16_t);
                    free_[i].swap(free_.back());
                    free_.pop_back();
                    break;
                }
        }
        v.assign(n, 0);
        return v;
    }
    void trim() {
        std::vector<std::vector<int16_t>> gone;
        std::lock_guard<std::mutex> g(m_);
        gone.swap(free_);
        bytes_ = 0;
    }
    void give(std::vector<int16_t> &v) {
        const size_t b = v.capacity() * sizeof(int16_t);
        if (b < (256u << 10)) return;  // small ones are the allocator's business
        std::vector<std::vector<int16_t>> evicted;  // (freed after the lock is released)
        std::lock_guard<std::mutex> g(m_);
        if (b > limit()) return;
        while (bytes_ + b > limit() && !free_.empty()) {  // the oldest go first: sizes nobody asks for any more do not pile up
            bytes_ -= free_.front().capacity() * sizeof(int16_t);
            evicted.emplace_back();
            evicted.back().swap(free_.front());
            free_.erase(free_.begin());
        }
        bytes_ += b;
        free_.emplace_back();
        free_.back().swap(v);
    }

private:
    static size_t limit() {
        static const size_t l = [] {
            const char *e = getenv("JPGPU_HOST_POOL_MB");
            const long mb = e ? atol(e) : 1024;
            return (size_t)(mb < 0 ? 0 : mb) << 20;
        }();
        return l;
    }
    std::mutex m_;
    std::vector<std::vector<int16_t>> free_;
    size_t bytes_ = 0;
};
CoefPool &coef_pool() {
    static CoefPool *p = new CoefPool;  // (never destroyed: front-ends may outlive static destruction order)
    return *p;
}
}  // namespace

void trim_coefficient_pool() { coef_pool().trim(); }

struct Frontend::Impl {
    std::vector<uint8_t> bytes;
    ByteSource src;
    bool has_frame = false;
    FrameInfo frame;
    HuffTable dc[4], ac[4];
    bool has_qt[4] = {false, false, false, false};
    uint16_t qt[4][64];
    uint16_t restart_interval = 0;
    bool has_adobe = false;
    int adobe = 0;  // 0 Unknown, 1 YCbCr, 2 YCCK
    int ct_override = -1;
    bool is_jfif = false, is_mjpeg = false;
    size_t buffer_limit = (size_t)-1;
    std::vector<int16_t> coefficients[JPGPU_MAX_COMPONENTS];  // progressive accumulation
    ~Impl() {
        for (auto &c : coefficients) coef_pool().give(c);
    }
    bool have_coefficients = false;
    uint64_t finished_mask[JPGPU_MAX_COMPONENTS] = {0, 0, 0, 0};
    bool plane_present[JPGPU_MAX_COMPONENTS] = {false, false, false, false};
    uint16_t plane_qt[JPGPU_MAX_COMPONENTS][64];
    bool has_exif = false, has_xmp = false;
    std::vector<uint8_t> exif, xmp;
    std::vector<IccChunk> icc;
    std::vector<PlannedScan> *plan = nullptr;  // plan_device_scans: describe scans instead of decoding them
    ProgPlan *prog_plan = nullptr;             // plan_progressive_scans: the same for the scans of a progressive frame
    uint8_t prog_state[JPGPU_MAX_COMPONENTS][64];  // per coefficient: 0 = no scan yet, else (Al of the last scan that covered it) + 1
    uint16_t prog_cell_track[JPGPU_MAX_COMPONENTS][64];  // union-find over (component, coefficient) cells: parent cell (c * 64 + k)
    // Per block of a progressive frame: which coefficients (zig-zag positions) are non-zero.  The refinement scans walk
    // bands of up to 63 coefficients per symbol to find the few that exist (src/decoder.rs:1260-1298 does it one by one:
    // 20 of the 27 ms a 1080p progressive image took); with the bitmap the walk costs the non-zero ones only.
    std::vector<uint64_t> nzmask[JPGPU_MAX_COMPONENTS];
    uint64_t *nz = nullptr;  // the current block's bitmap (null outside progressive frames)
    inline void put(int16_t &c, int16_t v, uint8_t zz) {  // zz: the coefficient's zig-zag position
        if (nz) *nz = v ? (*nz | (1ull << zz)) : (*nz & ~(1ull << zz));
        c = v;
    }

    size_t read_length() {  // src/parser.rs:137-147
        const uint16_t l = src.u16be();
        if (l < 2) fail(JPGPU_ERR_FORMAT, "encountered marker with invalid length %u", l);
        return (size_t)l - 2;
    }
    Marker read_marker() {  // src/decoder.rs:766-791
        for (;;) {
            while (src.u8() != 0xFF) {
            }
            uint8_t b = src.u8();
            while (b == 0xFF) b = src.u8();
            if (b != 0x00) return marker_from(b);
        }
    }

    void parse_sof(uint8_t n) {  // src/parser.rs:161-280 + checks of src/decoder.rs:340-379
        const size_t length = read_length();
        if (length <= 6) fail(JPGPU_ERR_FORMAT, "invalid length in SOF");
        FrameInfo f;
        f.is_baseline = n == 0;
        f.is_differential = (n >= 5 && n <= 7) || n >= 13;
        f.coding_process = (n == 0 || n == 1 || n == 5 || n == 9 || n == 13)     ? JPGPU_CODING_DCT_SEQUENTIAL
                           : (n == 2 || n == 6 || n == 10 || n == 14) ? JPGPU_CODING_DCT_PROGRESSIVE
                                                                       : JPGPU_CODING_LOSSLESS;
        f.arithmetic = n >= 9;
        f.precision = src.u8();
        if (f.precision == 8) {
        } else if (f.precision == 12) {
            if (f.is_baseline) fail(JPGPU_ERR_FORMAT, "12 bit sample precision is not allowed in baseline");
        } else if (f.coding_process != JPGPU_CODING_LOSSLESS || f.precision > 16) {
            fail(JPGPU_ERR_FORMAT, "invalid precision %u in frame header", f.precision);
        }
        const uint16_t height = src.u16be(), width = src.u16be();
        if (height == 0) fail(JPGPU_ERR_UNSUPPORTED, "DNL");
        if (width == 0) fail(JPGPU_ERR_FORMAT, "zero width in frame header");
        const uint8_t ncomp = src.u8();
        if (ncomp == 0) fail(JPGPU_ERR_FORMAT, "zero component count in frame header");
        if (f.coding_process == JPGPU_CODING_DCT_PROGRESSIVE && ncomp > 4)
            fail(JPGPU_ERR_FORMAT, "progressive frame with more than 4 components");
        if (length != 6 + 3 * (size_t)ncomp) fail(JPGPU_ERR_FORMAT, "invalid length in SOF");
        for (int i = 0; i < ncomp; i++) {
            jpgpu_component c{};
            c.identifier = src.u8();
            for (auto &o : f.components)
                if (o.identifier == c.identifier) fail(JPGPU_ERR_FORMAT, "duplicate frame component identifier %u", c.identifier);
            const uint8_t hv = src.u8();
            c.horizontal_sampling_factor = hv >> 4;
            c.vertical_sampling_factor = hv & 15;
            if (c.horizontal_sampling_factor == 0 || c.horizontal_sampling_factor > 4)
                fail(JPGPU_ERR_FORMAT, "invalid horizontal sampling factor %u", c.horizontal_sampling_factor);
            if (c.vertical_sampling_factor == 0 || c.vertical_sampling_factor > 4)
                fail(JPGPU_ERR_FORMAT, "invalid vertical sampling factor %u", c.vertical_sampling_factor);
            c.quantization_table_index = src.u8();
            if (c.quantization_table_index > 3 || (f.coding_process == JPGPU_CODING_LOSSLESS && c.quantization_table_index != 0))
                fail(JPGPU_ERR_FORMAT, "invalid quantization table index %u", c.quantization_table_index);
            c.dct_scale = 8;
            f.components.push_back(c);
        }
        f.image_w = f.output_w = width;
        f.image_h = f.output_h = height;
        update_component_sizes(width, height, f.components, f.mcu_w, f.mcu_h);
        // src/decoder.rs:346-379
        if (f.is_differential) fail(JPGPU_ERR_UNSUPPORTED, "Hierarchical");
        if (f.arithmetic) fail(JPGPU_ERR_UNSUPPORTED, "ArithmeticEntropyCoding");
        if (f.precision != 8 && f.coding_process != JPGPU_CODING_LOSSLESS) fail(JPGPU_ERR_UNSUPPORTED, "SamplePrecision(%u)", f.precision);
        if (f.precision < 2 || f.precision > 16) fail(JPGPU_ERR_UNSUPPORTED, "SamplePrecision(%u)", f.precision);
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) fail(JPGPU_ERR_UNSUPPORTED, "ComponentCount(%u)", ncomp);
        // Upsampler::new dry run: only a non-integer ratio can fail (src/upsampler.rs:76-105)
        uint32_t h_max = 0, v_max = 0;
        for (auto &c : f.components) {
            h_max = std::max<uint32_t>(h_max, c.horizontal_sampling_factor);
            v_max = std::max<uint32_t>(v_max, c.vertical_sampling_factor);
        }
        for (auto &c : f.components) {
            const uint32_t h = c.horizontal_sampling_factor, v = c.vertical_sampling_factor;
            const bool h1 = h == h_max || width == 1, v1 = v == v_max || height == 1, h2 = h * 2 == h_max, v2 = v * 2 == v_max;
            if (!((h1 || h2) && (v1 || v2)) && (h_max % h != 0 || v_max % v != 0))
                fail(JPGPU_ERR_UNSUPPORTED, "NonIntegerSubsamplingRatio");
        }
        frame = f;
        has_frame = true;
    }

    ScanInfo parse_sos() {  // src/parser.rs:332-482
        const FrameInfo &f = frame;
        const size_t length = read_length();
        if (length == 0) fail(JPGPU_ERR_FORMAT, "zero length in SOS");
        const uint8_t n = src.u8();
        if (n == 0 || n > 4) fail(JPGPU_ERR_FORMAT, "invalid component count %u in scan header", n);
        if (length != 4 + 2 * (size_t)n) fail(JPGPU_ERR_FORMAT, "invalid length in SOS");
        ScanInfo s;
        uint32_t blocks_per_mcu = 0;
        for (int i = 0; i < n; i++) {
            const uint8_t id = src.u8();
            int ci = -1;
            for (size_t k = 0; k < f.components.size(); k++)
                if (f.components[k].identifier == id) {
                    ci = (int)k;
                    break;
                }
            if (ci < 0) fail(JPGPU_ERR_FORMAT, "scan component identifier %u does not match any of the component identifiers defined in the frame", id);
            int prev_max = 0;
            for (int k = 0; k < i; k++) {
                if (s.component_indices[k] == ci) fail(JPGPU_ERR_FORMAT, "duplicate scan component identifier %u", id);
                prev_max = std::max(prev_max, s.component_indices[k]);
            }
            if (ci < prev_max) fail(JPGPU_ERR_FORMAT, "the scan component order does not follow the order in the frame header");
            const uint8_t tb = src.u8(), dci = tb >> 4, aci = tb & 15;
            if (dci > 3 || (f.is_baseline && dci > 1)) fail(JPGPU_ERR_FORMAT, "invalid dc table index %u", dci);
            if (aci > 3 || (f.is_baseline && aci > 1)) fail(JPGPU_ERR_FORMAT, "invalid ac table index %u", aci);
            s.component_indices[i] = ci;
            s.dc_tables[i] = dci;
            s.ac_tables[i] = aci;
            blocks_per_mcu += (uint32_t)f.components[ci].horizontal_sampling_factor * f.components[ci].vertical_sampling_factor;
        }
        s.n = n;
        if (n > 1 && blocks_per_mcu > 10) fail(JPGPU_ERR_FORMAT, "scan with more than one component and more than 10 blocks per MCU");
        const uint8_t ss = src.u8();
        uint8_t se = src.u8();
        const uint8_t a = src.u8(), ah = a >> 4, al = a & 15;
        if (al >= f.precision) fail(JPGPU_ERR_FORMAT, "invalid point transform, must be less than the frame precision");
        if (f.coding_process == JPGPU_CODING_DCT_PROGRESSIVE) {
            if (se > 63 || ss > se || (ss == 0 && se != 0))
                fail(JPGPU_ERR_FORMAT, "invalid spectral selection parameters: ss=%u, se=%u", ss, se);
            if (ss != 0 && n != 1) fail(JPGPU_ERR_FORMAT, "spectral selection scan with AC coefficients can't have more than one component");
            if (ah > 13 || al > 13) fail(JPGPU_ERR_FORMAT, "invalid successive approximation parameters: ah=%u, al=%u", ah, al);
            if (ah != 0 && ah != al + 1) fail(JPGPU_ERR_FORMAT, "successive approximation scan with more than one bit of improvement");
        } else if (f.coding_process == JPGPU_CODING_LOSSLESS) {
            if (se != 0) fail(JPGPU_ERR_FORMAT, "spectral selection end shall be zero in lossless scan");
            if (ah != 0) fail(JPGPU_ERR_FORMAT, "successive approximation high shall be zero in lossless scan");
            if (ss > 7) fail(JPGPU_ERR_FORMAT, "invalid predictor selection value: %u", ss);
        } else {
            if (se == 0) se = 63;
            if (ss != 0 || se != 63) fail(JPGPU_ERR_FORMAT, "spectral selection is not allowed in non-progressive scan");
            if (ah != 0 || al != 0) fail(JPGPU_ERR_FORMAT, "successive approximation is not allowed in non-progressive scan");
        }
        s.ss_start = ss;
        s.ss_end = (uint8_t)(se + 1);
        s.ah = ah;
        s.al = al;
        return s;
    }

    void parse_dqt() {  // src/parser.rs:485-532 + un-zigzag of src/decoder.rs:485-498
        size_t length = read_length();
        uint16_t parsed[4][64];
        bool got[4] = {false, false, false, false};
        while (length > 0) {
            const uint8_t pq = src.u8();
            const size_t precision = pq >> 4, index = pq & 15;
            if (precision > 1) fail(JPGPU_ERR_FORMAT, "invalid precision %zu in DQT", precision);
            if (index > 3) fail(JPGPU_ERR_FORMAT, "invalid destination identifier %zu in DQT", index);
            if (length < 65 + 64 * precision) fail(JPGPU_ERR_FORMAT, "invalid length in DQT");
            for (int i = 0; i < 64; i++) parsed[index][i] = precision ? src.u16be() : src.u8();
            for (int i = 0; i < 64; i++)
                if (parsed[index][i] == 0) fail(JPGPU_ERR_FORMAT, "quantization table contains element with a zero value");
            got[index] = true;
            length -= 65 + 64 * precision;
        }
        for (int t = 0; t < 4; t++)
            if (got[t]) {
                for (int j = 0; j < 64; j++) qt[t][kUnzigzag[j]] = parsed[t][j];
                has_qt[t] = true;
            }
    }

    void parse_dht() {  // src/parser.rs:536-589, merge of src/decoder.rs:501-518
        size_t length = read_length();
        // The segment's tables as POINTERS to the registry's copies until the whole segment has parsed (the reference merges nothing of
        // a segment that fails, src/decoder.rs:501-518): 128 bytes of stack.  History: round 4 had eight tables — 44 kB — on the stack of
        // whichever thread called the decoder (too much for small-stack threads); round 5 a 44 kB heap block per THREAD, kept
        // until the thread ends — 44 MB on a host with a thousand decoding threads.
        std::shared_ptr<const HuffTable> ndc[4], nac[4];
        while (length > 17) {
            const uint8_t tc = src.u8(), cls = tc >> 4;
            const size_t index = tc & 15;
            if (cls > 1) fail(JPGPU_ERR_FORMAT, "invalid class %u in DHT", cls);
            if (has_frame && frame.is_baseline && index > 1)
                fail(JPGPU_ERR_FORMAT, "a maximum of two huffman tables per class are allowed in baseline");
            if (index > 3) fail(JPGPU_ERR_FORMAT, "invalid destination identifier %zu in DHT", index);
            const uint8_t *counts = src.take(16);
            size_t size = 0;
            for (int i = 0; i < 16; i++) size += counts[i];
            if (size == 0) fail(JPGPU_ERR_FORMAT, "encountered table with zero length in DHT");
            if (size > 256) fail(JPGPU_ERR_FORMAT, "encountered table with excessive length in DHT");
            if (size > length - 17) fail(JPGPU_ERR_FORMAT, "invalid length in DHT");
            const uint8_t *vals = src.take(size);
            (cls == 0 ? ndc[index] : nac[index]) = build_cached(counts, vals, (int)size, cls == 1);
            length -= 17 + size;
        }
        if (length != 0) fail(JPGPU_ERR_FORMAT, "invalid length in DHT");
        for (int i = 0; i < 4; i++) {
            if (ndc[i]) dc[i] = *ndc[i];
            if (nac[i]) ac[i] = *nac[i];
        }
    }

    void parse_app(uint8_t n) {  // src/parser.rs:614-710
        const size_t length = read_length();
        size_t used = 0;
        if (n == 0) {
            if (length >= 5) {
                const uint8_t *b = src.take(5);
                used = 5;
                if (!memcmp(b, "JFIF\0", 5)) is_jfif = true;
                else if (!memcmp(b, "AVI1\0", 5)) is_mjpeg = true;
            }
        } else if (n == 1) {
            const uint8_t *b = src.take(length);
            used = length;
            if (length >= 6 && !memcmp(b, "Exif\0\0", 6)) {
                exif.assign(b + 6, b + length);
                has_exif = true;
            } else if (length >= 29 && !memcmp(b, "http://ns.adobe.com/xap/1.0/\0", 29)) {
                xmp.assign(b + 29, b + length);
                has_xmp = true;
            }
        } else if (n == 2) {
            if (length > 14) {
                const uint8_t *b = src.take(14);
                used = 14;
                if (!memcmp(b, "ICC_PROFILE\0", 12)) {
                    IccChunk c;
                    c.seq_no = b[12];
                    c.num_markers = b[13];
                    const uint8_t *d = src.take(length - 14);
                    c.data.assign(d, d + (length - 14));
                    used = length;
                    icc.push_back(std::move(c));
                }
            }
        } else if (n == 13) {
            if (length >= 14) {
                const uint8_t *b = src.take(14);
                used = 14;
                if (!memcmp(b, "Photoshop 3.0\0", 14)) {
                    src.take(length - 14);
                    used = length;
                }
            }
        } else if (n == 14) {
            if (length >= 12) {
                const uint8_t *b = src.take(12);
                used = 12;
                if (!memcmp(b, "Adobe\0", 6)) {
                    if (b[11] > 2) fail(JPGPU_ERR_FORMAT, "invalid color transform in adobe app segment");
                    has_adobe = true;
                    adobe = b[11];
                }
            }
        }
        if (src.len - src.pos < length - used) {  // skip_bytes, src/parser.rs:149-158
            src.pos = src.len;
            fail(JPGPU_ERR_IO, "unexpected end of file");
        }
        src.pos += length - used;
    }

    // ---- block decoding (src/decoder.rs:1086-1298) ------------------------------------------------
    void decode_block(int16_t *co, BitReader &br, const HuffTable *dct, const HuffTable *act, const ScanInfo &s,
                      uint16_t &eob_run, int16_t &pred) {
        if (s.ss_start == 0) {
            const uint8_t cat = br.decode(src, *dct);
            int16_t diff = 0;
            if (cat > 11) fail(JPGPU_ERR_FORMAT, "invalid DC difference magnitude category");
            if (cat) diff = br.receive_extend(src, cat);
            pred = (int16_t)((uint16_t)pred + (uint16_t)diff);  // wrapping_add
            put(co[0], (int16_t)((uint16_t)pred << s.al), 0);
        }
        uint8_t k = std::max<uint8_t>(s.ss_start, 1);
        if (k < s.ss_end && eob_run > 0) {
            eob_run--;
            return;
        }
        while (k < s.ss_end) {
            int16_t v;
            uint8_t run, total_bits, code_bits;
            if (br.peek_fast_ac(src, *act, v, run, total_bits, code_bits)) {
                k = (uint8_t)(k + run);
                if (k >= s.ss_end) {  // (invalid stream) the reference's fused entries took the magnitude bits, its
                    br.consume(total_bits > 8 ? code_bits : total_bits);  // symbol-then-magnitude path did not
                    break;
                }
                br.consume(total_bits);
                put(co[kUnzigzag[k]], (int16_t)((uint16_t)v << s.al), k);
                k++;
                continue;
            }
            const uint8_t rs = br.decode(src, *act), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r == 15) {
                    k = (uint8_t)(k + 16);
                    continue;
                }
                eob_run = (uint16_t)((1u << r) - 1);
                if (r) eob_run = (uint16_t)(eob_run + br.get_bits(src, r));
                break;
            }
            k = (uint8_t)(k + r);
            if (k >= s.ss_end) break;
            {
                const int16_t v2 = (int16_t)((uint16_t)br.receive_extend(src, sz) << s.al);
                put(co[kUnzigzag[k]], v2, k);
                k++;
            }
        }
    }

    uint8_t refine_non_zeroes(int16_t *co, BitReader &br, uint8_t start, uint8_t end, uint8_t zrl, int16_t bit) {
        // :1260-1298 — for i in start..end: a zero coefficient ends the walk once `zrl` of them have been passed; a
        // non-zero one takes a correction bit.  Same order of bit reads, found through the block's bitmap.
        if (start >= end) return (uint8_t)(end - 1);
        const uint64_t below_end = end >= 64 ? ~0ull : ((1ull << end) - 1ull), range = below_end & ~((1ull << start) - 1ull);
        uint64_t zeros = ~*nz & range;
        uint8_t stop = end;
        bool hit = false;
        if ((uint32_t)__builtin_popcountll(zeros) > zrl) {  // the walk ends AT zero number zrl + 1
            for (uint8_t skip = 0; skip < zrl; skip++) zeros &= zeros - 1ull;
            stop = (uint8_t)__builtin_ctzll(zeros);
            hit = true;
        }
        uint64_t todo = *nz & range & (stop >= 64 ? ~0ull : ((1ull << stop) - 1ull));
        while (todo) {
            const uint8_t i = (uint8_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            int16_t &c = co[kUnzigzag[i]];
            // (the correction bits are as good as random: no branch on them.  c is non-zero and stays so.)
            const int32_t apply = (int32_t)br.get_bits(src, 1) & (int32_t)((c & bit) == 0);
            const int32_t v = (int32_t)c + (((int32_t)c >> 31) | 1) * (int32_t)bit * apply;
            if (v > 32767 || v < -32768) fail(JPGPU_ERR_FORMAT, "Coefficient overflow");
            c = (int16_t)v;
        }
        return hit ? stop : (uint8_t)(end - 1);
    }

    void decode_block_refine(int16_t *co, BitReader &br, const HuffTable *act, const ScanInfo &s, uint16_t &eob_run) {
        const int16_t bit = (int16_t)(1 << s.al);  // :1174-1258
        if (s.ss_start == 0) {
            if (br.get_bits(src, 1) == 1) put(co[0], (int16_t)(co[0] | bit), 0);
            return;
        }
        if (eob_run > 0) {
            eob_run--;
            refine_non_zeroes(co, br, s.ss_start, s.ss_end, 64, bit);
            return;
        }
        uint8_t k = s.ss_start;
        while (k < s.ss_end) {
            const uint8_t rs = br.decode(src, *act), r = rs >> 4, sz = rs & 15;
            uint8_t zrl = r;
            int16_t value = 0;
            if (sz == 0) {
                if (r != 15) {
                    eob_run = (uint16_t)((1u << r) - 1);
                    if (r) eob_run = (uint16_t)(eob_run + br.get_bits(src, r));
                    zrl = 64;
                }
            } else if (sz == 1) {
                value = br.get_bits(src, 1) == 1 ? bit : (int16_t)-bit;
            } else {
                fail(JPGPU_ERR_FORMAT, "unexpected huffman code");
            }
            k = refine_non_zeroes(co, br, k, s.ss_end, zrl, bit);
            if (value != 0) put(co[kUnzigzag[k]], value, k);
            k++;
        }
    }

    // ---- decode_scan (src/decoder.rs:794-1082) ----------------------------------------------------
    // returns true and sets `pending` when a marker was captured at the end of the scan
    struct NotEligible {
        int where;  // which check of plan_scan refused (diagnostics: JPGPU_PLAN_TRACE)
    };
    // plan_device_scans: the checks decode_scan makes before it touches the entropy data, then cut the data into
    // restart segments instead of decoding it
    bool plan_scan(const ScanInfo &scan, const bool (&finished)[JPGPU_MAX_COMPONENTS], Marker &pending) {
        const FrameInfo &f = frame;
        const int nc = scan.n;
        if (f.coding_process != JPGPU_CODING_DCT_SEQUENTIAL || f.precision != 8) throw NotEligible{1};
        if (scan.ss_start != 0 || scan.ss_end != 64 || scan.ah != 0 || scan.al != 0) throw NotEligible{2};
        // one scan carrying all components (what encoders write for sequential files; per-component scans stay on the host)
        if ((size_t)nc != f.components.size()) throw NotEligible{13};
        jpgpu_component comps[JPGPU_MAX_COMPONENTS];
        for (int i = 0; i < nc; i++) {
            const int ci = scan.component_indices[i];
            if (!finished[i] || plane_present[ci]) throw NotEligible{3};
            comps[i] = f.components[ci];
            if (!has_qt[comps[i].quantization_table_index]) throw NotEligible{4};
        }
        if (is_mjpeg) {  // fill_default_mjpeg_tables, as in decode_scan
            bool d0 = false, d1 = false, a0 = false, a1 = false;
            for (int i = 0; i < nc; i++) {
                d0 |= scan.dc_tables[i] == 0;
                d1 |= scan.dc_tables[i] == 1;
                a0 |= scan.ac_tables[i] == 0;
                a1 |= scan.ac_tables[i] == 1;
            }
            if (d0 && !dc[0].present) dc[0].build(kK3Bits, kDcVals, 12, false);
            if (d1 && !dc[1].present) dc[1].build(kK4Bits, kDcVals, 12, false);
            if (a0 && !ac[0].present) ac[0].build(kK5Bits, kK5Vals, 162, true);
            if (a1 && !ac[1].present) ac[1].build(kK6Bits, kK6Vals, 162, true);
        }
        for (int i = 0; i < nc; i++)
            if (!dc[scan.dc_tables[i]].present || !ac[scan.ac_tables[i]].present) throw NotEligible{5};
        PlannedScan ps;
        const bool interleaved = nc > 1;
        const uint32_t max_x = interleaved ? f.mcu_w : comps[0].block_width, max_y = interleaved ? f.mcu_h : comps[0].block_height;
        // the MCU loops of decode_scan stop at the image edge (my * 8 >= image_h, mx * 8 >= image_w)
        ps.cols = std::min<uint32_t>(max_x, ((uint32_t)f.image_w + 7u) / 8u);
        const uint32_t rows = std::min<uint32_t>(max_y, ((uint32_t)f.image_h + 7u) / 8u);
        ps.n_mcu = ps.cols * rows;
        ps.ri = restart_interval;  // 0: no restart markers — one segment, decoded by the self-synchronising chunk decoder
        ps.ncomp = (uint32_t)nc;
        if (ps.n_mcu == 0) throw NotEligible{6};
        {  // the chunk decoder keeps per-block-of-the-MCU tables of 16 entries (the standard allows 10 blocks per MCU)
            uint32_t bpm = 0;
            for (int i = 0; i < nc; i++) bpm += interleaved ? (uint32_t)comps[i].horizontal_sampling_factor * comps[i].vertical_sampling_factor : 1u;
            if (bpm > 16u) throw NotEligible{14};
        }
        for (int i = 0; i < nc; i++) {
            ps.comp[i].frame_index = (uint32_t)scan.component_indices[i];
            ps.comp[i].block_w = comps[i].block_width;
            ps.comp[i].h = interleaved ? comps[i].horizontal_sampling_factor : 1u;
            ps.comp[i].v = interleaved ? comps[i].vertical_sampling_factor : 1u;
            ps.comp[i].dc = (uint32_t)scan.dc_tables[i];
            ps.comp[i].ac = (uint32_t)scan.ac_tables[i];
        }
        {
            // The device form of the eight tables: the sets built last are kept with the definitions they came from (code counts,
            // values, class: everything else in a table is a function of them) and handed out again as they are.
            // (shared by the PROCESS since round 4 — a batch's files come from a few encoders and share their sets; the key holds the
            // definitions only, 280 bytes per table.  Rounds 2-3: one set per thread, each with eight whole host tables as its key.)
            struct Key {
                bool present = false, is_ac = false;
                int nvalues = 0;
                uint8_t bits[16], values[256];
                bool matches(const HuffTable &h) const {
                    return h.present ? (present && is_ac == h.is_ac && nvalues == h.nvalues && memcmp(bits, h.bits, 16) == 0 &&
                                        memcmp(values, h.values, (size_t)h.nvalues) == 0)
                                     : !present;
                }
            };
            struct Last {
                Key key[8];
                std::shared_ptr<const PlannedScan::TableSet> set;
            };
            // (per thread: a POINTER to the set it used last — compared without a lock; the sets themselves are shared, immutable,
            // and registered under a mutex that only a thread's first file of an encoder takes)
            constexpr int kSets = 8;
            static std::mutex &last_m = *new std::mutex;  // (leaked on purpose, like build_cached's registry)
            static std::shared_ptr<const Last> *const last_sets = new std::shared_ptr<const Last>[kSets];
            static int last_next = 0;
            thread_local std::shared_ptr<const Last> mine;
            auto matches = [&](const Last &l) {
                bool same = l.set != nullptr;
                for (int t = 0; t < 8 && same; t++) same = l.key[t].matches(t < 4 ? dc[t] : ac[t - 4]);
                return same;
            };
            std::shared_ptr<const PlannedScan::TableSet> found;
            if (mine && matches(*mine)) found = mine->set;
            if (!found) {
                std::lock_guard<std::mutex> last_lock(last_m);
                for (int e = 0; e < kSets && !found; e++)
                    if (last_sets[e] && matches(*last_sets[e])) {
                        mine = last_sets[e];
                        found = mine->set;
                    }
            }
            if (!found) {
                auto fresh_p = std::make_shared<Last>();
                Last *last = fresh_p.get();
                auto set = std::make_shared<PlannedScan::TableSet>();
                memset(set.get(), 0, sizeof(*set));
                for (int t = 0; t < 8; t++) {
                    const HuffTable &h = t < 4 ? dc[t] : ac[t - 4];
                    DevHuffTable &d = set->t[huff_table_slot(t < 4 ? 0u : 1u, (uint32_t)(t & 3))];
                    Key &k = last->key[t];
                    k.present = h.present;
                    if (!h.present) continue;
                    static_assert(HUFF_LUT_BITS >= 8 && HUFF_LUT_BITS <= kLutBits && sizeof(d.values) == sizeof(h.values), "table layouts");
                    // the device table is the host's wide table cut to HUFF_LUT_BITS: a code that fits has the same entry under every
                    // longer prefix; one that does not is left to the walk (which starts at 9 bits like the reference's, src/huffman.rs:31-58)
                    int n_sub = 0;
                    for (int i = 0; i < (1 << HUFF_LUT_BITS); i++) {
                        const int w = i << (kLutBits - HUFF_LUT_BITS);
                        if (h.lut_size[w] && h.lut_size[w] <= HUFF_LUT_BITS) {
                            d.lut[i] = (uint16_t)(huff_sym_info(h.is_ac ? 1u : 0u, h.lut_value[w]) | ((uint32_t)h.lut_size[w] << SYM_LEN_SHIFT));
                            continue;
                        }
                        // not resolved within the lookahead: the reference's maxcode walk (src/huffman.rs:31-58, from length 9 as
                        // huff_walk does it) evaluated for the 64 continuations of this prefix — if any of them is a code
                        d.lut[i] = (uint16_t)HUFF_SUB_NONE;
                        if (n_sub >= HUFF_SUB_TABLES) continue;
                        uint16_t sub[1 << HUFF_SUB_BITS];
                        bool any = false;
                        for (int x = 0; x < (1 << HUFF_SUB_BITS); x++) {
                            const uint32_t b16 = ((uint32_t)i << HUFF_SUB_BITS) | (uint32_t)x;
                            sub[x] = (uint16_t)SYM_BAD;
                            for (int len = 9; len <= 16; len++) {
                                const int32_t code = (int32_t)(b16 >> (16 - len));
                                if (code <= h.maxcode[len - 1]) {
                                    const int32_t index = code + h.delta[len - 1];
                                    if (index >= 0 && index < h.nvalues) {
                                        sub[x] = (uint16_t)(huff_sym_info(h.is_ac ? 1u : 0u, h.values[index]) | ((uint32_t)(len - 1) << SYM_LEN_SHIFT));
                                        any = true;
                                    }
                                    break;
                                }
                            }
                        }
                        if (!any) continue;  // (nothing but rejections under this prefix: the walk says so at decode time)
                        memcpy(d.lut2[n_sub], sub, sizeof(sub));
                        d.lut[i] = (uint16_t)n_sub++;
                    }
                    memcpy(d.maxcode, h.maxcode, sizeof(d.maxcode));
                    memcpy(d.delta, h.delta, sizeof(d.delta));
                    memcpy(d.values, h.values, sizeof(d.values));
                    d.nvalues = h.nvalues;
                    k.is_ac = h.is_ac;
                    k.nvalues = h.nvalues;
                    memcpy(k.bits, h.bits, 16);
                    memcpy(k.values, h.values, sizeof(h.values));
                }
                last->set = std::move(set);
                found = last->set;
                mine = fresh_p;
                std::lock_guard<std::mutex> last_lock(last_m);
                last_sets[last_next] = std::move(fresh_p);
                last_next = (last_next + 1) % kSets;
            }
            ps.tables = found;
        }
        // cut the entropy-coded data at the RSTn markers: exactly one every `ri` MCUs, numbered 0..7 cyclically
        // (src/decoder.rs:920-956), 0xFF00 pairs inside, one other marker right after the last segment
        const uint32_t n_seg = ps.ri ? (ps.n_mcu + ps.ri - 1u) / ps.ri : 1u;
        ps.data_off = src.pos;
        ps.seg_off.reserve((size_t)n_seg + 1u);
        ps.seg_off.push_back(0u);
        const uint8_t *p = src.p;
        size_t pos = src.pos;
        uint32_t expected_rst = 0;
        if (ps.ri == 0) {
            // No restart markers: the scan of an eligible stream runs up to the EOI marker, and an 0xFF 0xD9 pair cannot occur
            // inside entropy-coded data.  Take the last one of the stream as the end without walking the data (a pass over
            // every byte of every file before anything else could start: 2 of the 4 ms the header phase of 1,024 files took);
            // the staging copy walks the bytes anyway and refuses the image if it meets anything but 0xFF00 pairs — other
            // markers, fill bytes, a second EOI in trailing garbage — which then goes to the host decoder as before.
            size_t end = src.len;
            bool found = false;
            while (end >= pos + 2) {
                const void *d9 = memrchr(p + pos + 1, 0xD9, end - pos - 1);
                if (!d9) break;
                const size_t at = (size_t)(static_cast<const uint8_t *>(d9) - p);
                if (p[at - 1] == 0xFF) {
                    end = at - 1;
                    found = true;
                    break;
                }
                end = at;
            }
            if (!found) throw NotEligible{7};
            if (end - ps.data_off > 0xFFFFFFF0u) throw NotEligible{10};
            {  // a block costs at least two bits (a DC code and an end-of-block code): headers that announce far more blocks
               // than the data can hold (truncated or hostile files) are not worth planes and launches on the device
                uint64_t blocks = 0;
                for (int i = 0; i < nc; i++) blocks += interleaved ? (uint64_t)comps[i].horizontal_sampling_factor * comps[i].vertical_sampling_factor : 1u;
                blocks *= ps.n_mcu;
                if ((uint64_t)(end - ps.data_off) * 8u < blocks * 2u) throw NotEligible{15};
            }
            ps.seg_off.push_back((uint32_t)(end - ps.data_off));
            ps.check_at_staging = true;
            pending = marker_from(0xD9);
            src.pos = end + 2;
        } else
        for (;;) {
            const void *ff = pos < src.len ? memchr(p + pos, 0xFF, src.len - pos) : nullptr;
            if (!ff) throw NotEligible{7};  // ran off the end without a marker
            pos = (size_t)(static_cast<const uint8_t *>(ff) - p);
            if (pos + 1 >= src.len) throw NotEligible{8};
            const uint8_t nb = p[pos + 1];
            if (nb == 0x00) {  // stuffed byte
                pos += 2;
                continue;
            }
            if (nb == 0xFF) throw NotEligible{9};  // fill bytes: legal, rare, host path
            if (pos - ps.data_off > 0xFFFFFFF0u) throw NotEligible{10};
            if (nb >= 0xD0 && nb <= 0xD7) {
                if ((uint32_t)(nb - 0xD0) != expected_rst || ps.seg_off.size() >= n_seg) throw NotEligible{11};
                expected_rst = (expected_rst + 1u) % 8u;
                ps.seg_off.push_back((uint32_t)(pos - ps.data_off));      // end of this segment
                pos += 2;
                // (the next segment starts after the marker: keep one offset list by storing segment starts shifted)
                seg_start_after.push_back((uint32_t)(pos - ps.data_off));
                continue;
            }
            // any other marker ends the scan
            if (ps.seg_off.size() != n_seg) throw NotEligible{12};
            ps.seg_off.push_back((uint32_t)(pos - ps.data_off));
            pending = marker_from(nb);
            src.pos = pos + 2;
            break;
        }
        // seg_off so far holds [0, end_0, end_1, ..., end_last]; starts of segments 1.. are the bytes after the markers.
        // The device wants start/end per segment: interleave into 2 * n_seg entries [start_0, end_0, start_1, end_1, ...]
        {
            std::vector<uint32_t> se;
            se.reserve(2u * n_seg);
            uint64_t bpm = 0;  // blocks per MCU
            for (int i = 0; i < nc; i++) bpm += interleaved ? (uint64_t)comps[i].horizontal_sampling_factor * comps[i].vertical_sampling_factor : 1u;
            for (uint32_t s = 0; s < n_seg; s++) {
                const uint32_t first = s == 0 ? 0u : seg_start_after[s - 1], last = ps.seg_off[s + 1];
                // as for scans without restart markers: a block costs at least two bits, and a segment that cannot hold its
                // MCUs would send a lane of the segment decoder through the padding behind it
                const uint64_t mcus = std::min<uint64_t>(ps.ri, (uint64_t)ps.n_mcu - (uint64_t)s * ps.ri);
                if (last < first || (uint64_t)(last - first) * 8u < mcus * bpm * 2u) throw NotEligible{15};
                se.push_back(first);
                se.push_back(last);
            }
            ps.seg_off.swap(se);
            seg_start_after.clear();
        }
        for (int i = 0; i < nc; i++) {
            const int ci = scan.component_indices[i];
            memcpy(plane_qt[ci], qt[comps[i].quantization_table_index], 128);
            plane_present[ci] = true;
        }
        plan->push_back(std::move(ps));
        return true;  // `pending` holds the marker that ended the scan
    }
    std::vector<uint32_t> seg_start_after;

    // ---- plan_progressive_scans: one scan of a progressive frame --------------------------------------------------------------------
    uint16_t prog_find(uint16_t cell) {
        while (prog_cell_track[cell >> 6][cell & 63] != cell) {
            const uint16_t up = prog_cell_track[cell >> 6][cell & 63];
            prog_cell_track[cell >> 6][cell & 63] = prog_cell_track[up >> 6][up & 63];  // (path halving)
            cell = up;
        }
        return cell;
    }
    static std::shared_ptr<const ProgHuffTable> prog_table_of(const HuffTable &h) {
        auto t = std::make_shared<ProgHuffTable>();
        memset(t.get(), 0, sizeof(*t));
        // the reference's 8-bit table (src/huffman.rs:190-222): codes of up to eight bits under every prefix that begins with them —
        // read off the host's wider cache of the same procedure; longer codes are left to the walk, as in the reference
        for (int i = 0; i < 256; i++) {
            const int w = i << (kLutBits - 8);
            if (h.lut_size[w] && h.lut_size[w] <= 8) t->lut[i] = (uint16_t)(h.lut_value[w] | ((uint32_t)h.lut_size[w] << 8));
        }
        memcpy(t->maxcode, h.maxcode, sizeof(t->maxcode));
        memcpy(t->delta, h.delta, sizeof(t->delta));
        memcpy(t->values, h.values, sizeof(t->values));
        t->nvalues = h.nvalues;
        return t;
    }
    bool plan_prog_scan(const ScanInfo &scan, const bool (&finished)[JPGPU_MAX_COMPONENTS], Marker &pending) {
        const FrameInfo &f = frame;
        const int nc = scan.n;
        if (f.coding_process != JPGPU_CODING_DCT_PROGRESSIVE || f.precision != 8) throw NotEligible{21};
        if (restart_interval != 0) throw NotEligible{22};
        if (prog_plan->scans.size() >= 256u) throw NotEligible{23};
        jpgpu_component comps[JPGPU_MAX_COMPONENTS];
        for (int i = 0; i < nc; i++) {
            comps[i] = f.components[scan.component_indices[i]];
            if (!has_qt[comps[i].quantization_table_index]) throw NotEligible{24};  // decode_scan: "use of unset quantization table"
        }
        if (is_mjpeg) throw NotEligible{25};
        const bool dc_scan = scan.ss_start == 0;
        const uint8_t se = (uint8_t)(scan.ss_end - 1);
        if (dc_scan)
            for (int i = 0; i < nc; i++)
                if (!dc[scan.dc_tables[i]].present) throw NotEligible{26};
        if (scan.ss_end > 1)
            for (int i = 0; i < nc; i++)
                if (!ac[scan.ac_tables[i]].present) throw NotEligible{27};
        if (!dc_scan && nc != 1) throw NotEligible{28};  // (parse_sos has refused it already)
        // successive approximation, one bit at a time from a first scan on, per coefficient
        uint16_t first_cell = 0xffffu;
        for (int i = 0; i < nc; i++) {
            const int ci = scan.component_indices[i];
            for (int k = scan.ss_start; k <= se; k++) {
                uint8_t &st = prog_state[ci][k];
                if (scan.ah == 0) {
                    if (st != 0) throw NotEligible{29};
                } else if (st != scan.ah + 1 || scan.al + 1 != scan.ah) {
                    throw NotEligible{30};
                }
                st = (uint8_t)(scan.al + 1);
                const uint16_t cell = prog_find((uint16_t)(ci * 64 + k));
                if (first_cell == 0xffffu) first_cell = cell;
                else if (cell != first_cell) prog_cell_track[cell >> 6][cell & 63] = first_cell;
            }
        }
        for (int i = 0; i < nc; i++)
            if (finished[i]) memcpy(plane_qt[scan.component_indices[i]], qt[comps[i].quantization_table_index], 128);
        ProgPlannedScan ps;
        const bool interleaved = nc > 1;
        const uint32_t max_x = interleaved ? f.mcu_w : comps[0].block_width, max_y = interleaved ? f.mcu_h : comps[0].block_height;
        ps.cols = std::min<uint32_t>(max_x, ((uint32_t)f.image_w + 7u) / 8u);  // (the MCU loops of decode_scan stop at the image edge)
        ps.rows = std::min<uint32_t>(max_y, ((uint32_t)f.image_h + 7u) / 8u);
        if (ps.cols == 0 || ps.rows == 0) throw NotEligible{31};
        ps.ss = scan.ss_start;
        ps.se = se;
        ps.ah = scan.ah;
        ps.al = scan.al;
        ps.ncomp = (uint32_t)nc;
        ps.track = first_cell;  // (a cell for now: numbered when the stream is through)
        for (int i = 0; i < nc; i++) {
            ps.comp[i].frame_index = (uint32_t)scan.component_indices[i];
            ps.comp[i].block_w = comps[i].block_width;
            ps.comp[i].h = interleaved ? comps[i].horizontal_sampling_factor : 1u;
            ps.comp[i].v = interleaved ? comps[i].vertical_sampling_factor : 1u;
            ps.comp[i].table = dc_scan ? (uint32_t)scan.dc_tables[i] : 0u;
            // (what decode_scan answers with "reference would panic: coefficient index")
            if ((uint64_t)ps.rows * ps.comp[i].v > comps[i].block_height || (uint64_t)ps.cols * ps.comp[i].h > comps[i].block_width) throw NotEligible{32};
        }
        if (dc_scan && scan.ah == 0) {
            for (int i = 0; i < nc; i++) {
                const HuffTable &h = dc[scan.dc_tables[i]];
                for (int v = 0; v < h.nvalues; v++)
                    if (h.values[v] > 15) throw NotEligible{33};  // (the lanes keep DC symbols in four bits; a category above 11 is an error anyway)
                if (!ps.table[scan.dc_tables[i]]) ps.table[scan.dc_tables[i]] = prog_table_of(h);
            }
        } else if (!dc_scan) {
            ps.table[0] = prog_table_of(ac[scan.ac_tables[0]]);
        }
        // the scan's data: up to the first 0xFF that is not followed by its stuffing zero — a marker, which ends the scan
        const uint8_t *p = src.p;
        size_t pos = src.pos;
        ps.data_off = pos;
        for (;;) {
            const void *ff = pos < src.len ? memchr(p + pos, 0xFF, src.len - pos) : nullptr;
            if (!ff) throw NotEligible{34};  // ran off the end without a marker: the host reports it
            pos = (size_t)(static_cast<const uint8_t *>(ff) - p);
            if (pos + 1 >= src.len) throw NotEligible{35};
            const uint8_t nb = p[pos + 1];
            if (nb == 0x00) {
                pos += 2;
                continue;
            }
            if (nb == 0xFF || (nb >= 0xD0 && nb <= 0xD7)) throw NotEligible{36};  // fill bytes, a restart marker without an interval
            if (pos - ps.data_off > 0x0FFFFFF0u) throw NotEligible{37};
            ps.stuffed_bytes = (uint32_t)(pos - ps.data_off);
            pending = marker_from(nb);
            src.pos = pos + 2;
            break;
        }
        prog_plan->scans.push_back(std::move(ps));
        return true;
    }

    bool decode_scan(const ScanInfo &scan, const bool (&finished)[JPGPU_MAX_COMPONENTS], RowSink &sink, Marker &pending) {
        const FrameInfo &f = frame;
        jpgpu_component comps[JPGPU_MAX_COMPONENTS];
        const int nc = scan.n;
        for (int i = 0; i < nc; i++) comps[i] = f.components[scan.component_indices[i]];
        for (int i = 0; i < nc; i++)
            if (!has_qt[comps[i].quantization_table_index]) fail(JPGPU_ERR_FORMAT, "use of unset quantization table");
        if (is_mjpeg) {  // fill_default_mjpeg_tables, src/huffman.rs:295-346
            bool d0 = false, d1 = false, a0 = false, a1 = false;
            for (int i = 0; i < nc; i++) {
                d0 |= scan.dc_tables[i] == 0;
                d1 |= scan.dc_tables[i] == 1;
                a0 |= scan.ac_tables[i] == 0;
                a1 |= scan.ac_tables[i] == 1;
            }
            if (d0 && !dc[0].present) dc[0].build(kK3Bits, kDcVals, 12, false);
            if (d1 && !dc[1].present) dc[1].build(kK4Bits, kDcVals, 12, false);
            if (a0 && !ac[0].present) ac[0].build(kK5Bits, kK5Vals, 162, true);
            if (a1 && !ac[1].present) ac[1].build(kK6Bits, kK6Vals, 162, true);
        }
        if (scan.ss_start == 0)
            for (int i = 0; i < nc; i++)
                if (!dc[scan.dc_tables[i]].present) fail(JPGPU_ERR_FORMAT, "scan makes use of unset dc huffman table");
        if (scan.ss_end > 1)
            for (int i = 0; i < nc; i++)
                if (!ac[scan.ac_tables[i]].present) fail(JPGPU_ERR_FORMAT, "scan makes use of unset ac huffman table");

        for (int i = 0; i < nc; i++)
            if (finished[i]) {
                const int ci = scan.component_indices[i];
                memcpy(plane_qt[ci], qt[comps[i].quantization_table_index], 128);
                sink.frame_slot_hint((uint32_t)i, (uint32_t)ci);
                sink.start((uint32_t)i, comps[i], qt[comps[i].quantization_table_index]);
            }

        const bool progressive = f.coding_process == JPGPU_CODING_DCT_PROGRESSIVE;
        const bool interleaved = nc > 1;
        struct NzOff {  // (the block decoders must not write into a bitmap of another frame)
            Impl *self;
            ~NzOff() { self->nz = nullptr; }
        } nz_off{this};
        int16_t dummy[64];
        memset(dummy, 0, sizeof(dummy));
        BitReader br;
        int16_t pred[JPGPU_MAX_COMPONENTS] = {0, 0, 0, 0};
        uint16_t until_restart = restart_interval, eob_run = 0;
        int expected_rst = 0;
        size_t per_row[JPGPU_MAX_COMPONENTS];
        std::vector<int16_t> rowbuf[JPGPU_MAX_COMPONENTS];
        uint32_t hs[JPGPU_MAX_COMPONENTS], vs[JPGPU_MAX_COMPONENTS];
        for (int i = 0; i < nc; i++) {
            per_row[i] = (size_t)comps[i].block_width * comps[i].vertical_sampling_factor * 64;
            if (!progressive && finished[i]) rowbuf[i].assign(per_row[i], 0);
            hs[i] = interleaved ? comps[i].horizontal_sampling_factor : 1;
            vs[i] = interleaved ? comps[i].vertical_sampling_factor : 1;
        }
        const uint32_t max_x = interleaved ? f.mcu_w : comps[0].block_width;
        const uint32_t max_y = interleaved ? f.mcu_h : comps[0].block_height;

        for (uint32_t my = 0; my < max_y; my++) {
            if (my * 8 >= f.image_h) break;
            for (uint32_t mx = 0; mx < max_x; mx++) {
                if (mx * 8 >= f.image_w) break;
                if (restart_interval > 0) {
                    if (until_restart == 0) {
                        Marker m;
                        if (!br.take_marker(src, m)) fail(JPGPU_ERR_FORMAT, "no marker found where RST%d was expected", expected_rst);
                        if (m.kind != Mk::RST) fail(JPGPU_ERR_FORMAT, "found marker inside scan where RST%d was expected", expected_rst);
                        if (m.n != expected_rst) fail(JPGPU_ERR_FORMAT, "found RST%d where RST%d was expected", m.n, expected_rst);
                        br.reset();
                        memset(pred, 0, sizeof(pred));
                        eob_run = 0;
                        expected_rst = (expected_rst + 1) % 8;
                        until_restart = restart_interval;
                    }
                    until_restart--;
                }
                for (int i = 0; i < nc; i++) {
                    const jpgpu_component &c = comps[i];
                    for (uint32_t vp = 0; vp < vs[i]; vp++)
                        for (uint32_t hp = 0; hp < hs[i]; hp++) {
                            int16_t *co;
                            if (progressive) {
                                const size_t by = (size_t)my * vs[i] + vp, bx = (size_t)mx * hs[i] + hp;
                                const size_t off = (by * c.block_width + bx) * 64;
                                auto &store = coefficients[scan.component_indices[i]];
                                if (off + 64 > store.size()) fail(JPGPU_ERR_INTERNAL, "reference would panic: coefficient index");
                                co = store.data() + off;
                                nz = &nzmask[scan.component_indices[i]][off / 64];
                            } else if (finished[i]) {
                                const uint32_t batch_row = interleaved ? 0 : my % c.vertical_sampling_factor;
                                const size_t by = (size_t)batch_row * vs[i] + vp, bx = (size_t)mx * hs[i] + hp;
                                const size_t off = (by * c.block_width + bx) * 64;
                                if (off + 64 > per_row[i]) fail(JPGPU_ERR_INTERNAL, "reference would panic: row coefficient index");
                                co = rowbuf[i].data() + off;
                            } else {
                                co = dummy;
                            }
                            if (scan.ah == 0) decode_block(co, br, &dc[scan.dc_tables[i]], &ac[scan.ac_tables[i]], scan, eob_run, pred[i]);
                            else decode_block_refine(co, br, &ac[scan.ac_tables[i]], scan, eob_run);
                        }
                }
            }
            for (int i = 0; i < nc; i++) {  // hand the finished MCU row over, :1019-1059
                if (!finished[i]) continue;
                const jpgpu_component &c = comps[i];
                if (!interleaved && (my + 1) * 8 < f.image_h && (my + 1) % c.vertical_sampling_factor > 0) continue;
                if (progressive) {
                    const uint32_t wy = interleaved ? my : my / c.vertical_sampling_factor;
                    const size_t off = (size_t)wy * per_row[i];
                    auto &store = coefficients[scan.component_indices[i]];
                    if (off + per_row[i] > store.size()) fail(JPGPU_ERR_INTERNAL, "reference would panic: coefficient row slice");
                    sink.append_row((uint32_t)i, store.data() + off, per_row[i]);
                } else {
                    sink.append_row((uint32_t)i, rowbuf[i].data(), per_row[i]);
                    std::fill(rowbuf[i].begin(), rowbuf[i].end(), (int16_t)0);
                }
            }
        }
        nz = nullptr;
        if (progressive)
            for (int i = 0; i < nc; i++) sink.scan_finished((uint32_t)scan.component_indices[i]);
        Marker m;
        bool has = br.take_marker(src, m);
        while (has && m.kind == Mk::RST) {  // :1063-1066  marker = self.read_marker().ok()
            try {
                m = read_marker();
            } catch (const DecodeError &) {
                has = false;
            }
        }
        for (int i = 0; i < nc; i++)
            if (finished[i]) {
                const int ci = scan.component_indices[i];
                sink.finish((uint32_t)i, (uint32_t)ci);
                // src/decoder.rs:465-475: the plane is kept only if the component is complete
                if (finished_mask[ci] == ~(uint64_t)0) plane_present[ci] = true;
            }
        pending = m;
        return has;
    }

    // decode_internal, src/decoder.rs:297-615
    void run(bool stop_after_metadata, RowSink *sink) {
        if (stop_after_metadata && has_frame) return;
        if (!has_frame) {
            if (src.u8() != 0xFF) fail(JPGPU_ERR_FORMAT, "first two bytes are not an SOI marker");
            const uint8_t b = src.u8();
            if (b != 0xD8) fail(JPGPU_ERR_FORMAT, "first two bytes are not an SOI marker");
        }
        Marker previous{Mk::SOI, 0}, pending{Mk::RES, 0};
        bool has_pending = false;
        int scans = 0;
        for (;;) {
            Marker m = has_pending ? pending : read_marker();
            has_pending = false;
            switch (m.kind) {
            case Mk::SOF:
                if (has_frame) fail(JPGPU_ERR_UNSUPPORTED, "Hierarchical");
                parse_sof(m.n);
                if (stop_after_metadata) return;
                break;
            case Mk::SOS: {
                if (!has_frame) fail(JPGPU_ERR_FORMAT, "scan encountered before frame");
                const ScanInfo scan = parse_sos();
                const FrameInfo &f = frame;
                if (prog_plan && f.coding_process != JPGPU_CODING_DCT_PROGRESSIVE) throw NotEligible{20};
                if (f.coding_process == JPGPU_CODING_DCT_PROGRESSIVE && !have_coefficients && !prog_plan) {
                    for (size_t i = 0; i < f.components.size(); i++)
                        coefficients[i] = coef_pool().take((size_t)f.components[i].block_width * f.components[i].block_height * 64);
                    for (size_t i = 0; i < f.components.size(); i++)
                        nzmask[i].assign((size_t)f.components[i].block_width * f.components[i].block_height, 0);
                    have_coefficients = true;
                }
                if (f.coding_process == JPGPU_CODING_LOSSLESS)
                    fail(JPGPU_ERR_UNSUPPORTED, "lossless JPEG (SOF3) is decoded by the reference's CPU pipeline only (src/decoder/lossless.rs)");
                bool finished[JPGPU_MAX_COMPONENTS] = {false, false, false, false};
                if (scan.al == 0)
                    for (int k = 0; k < scan.n; k++) {
                        const int i = scan.component_indices[k];
                        if (finished_mask[i] == ~(uint64_t)0) continue;
                        for (int j = scan.ss_start; j < scan.ss_end; j++) finished_mask[i] |= (uint64_t)1 << j;
                        if (finished_mask[i] == ~(uint64_t)0) finished[k] = true;
                    }
                has_pending = prog_plan ? plan_prog_scan(scan, finished, pending) : (plan ? plan_scan(scan, finished, pending) : decode_scan(scan, finished, *sink, pending));
                scans++;
                break;
            }
            case Mk::DQT: parse_dqt(); break;
            case Mk::DHT: parse_dht(); break;
            case Mk::DAC: fail(JPGPU_ERR_UNSUPPORTED, "ArithmeticEntropyCoding");
            case Mk::DRI: {
                if (read_length() != 2) fail(JPGPU_ERR_FORMAT, "DRI with invalid length");
                restart_interval = src.u16be();
                break;
            }
            case Mk::COM: src.take(read_length()); break;
            case Mk::APP: parse_app(m.n); break;
            case Mk::RST:
                if (previous.kind != Mk::SOS) fail(JPGPU_ERR_FORMAT, "RST found outside of entropy-coded data");
                break;
            case Mk::DNL:
                if (previous.kind != Mk::SOS || scans != 1) fail(JPGPU_ERR_FORMAT, "DNL is only allowed immediately after the first scan");
                fail(JPGPU_ERR_UNSUPPORTED, "DNL");
            case Mk::DHP:
            case Mk::EXP: fail(JPGPU_ERR_UNSUPPORTED, "Hierarchical");
            case Mk::EOI: goto done;
            default: fail(JPGPU_ERR_FORMAT, "marker found where not allowed");
            }
            previous = m;
        }
    done:
        if (!has_frame) fail(JPGPU_ERR_FORMAT, "end of image encountered before frame");
        // decode_planes, src/decoder.rs:617-696
        const FrameInfo &f = frame;
        const size_t ncomp = f.components.size();
        {
            const unsigned __int128 need = (unsigned __int128)ncomp * f.output_w * f.output_h;
            if (need > (unsigned __int128)buffer_limit) fail(JPGPU_ERR_FORMAT, "size of decoded image exceeds maximum allowed size");
        }
        if (prog_plan) {  // (plan_progressive_scans: every component gets a plane — finished in some scan, or rendered as it stands)
            if (prog_plan->scans.empty()) throw NotEligible{38};
            for (size_t i = 0; i < ncomp; i++) {
                const jpgpu_component &c = f.components[i];
                if (finished_mask[i] != ~(uint64_t)0) {
                    if (!has_qt[c.quantization_table_index]) throw NotEligible{39};  // (its plane would be missing: "not all components have data")
                    memcpy(plane_qt[i], qt[c.quantization_table_index], 128);
                }
                plane_present[i] = true;
            }
        } else if (f.coding_process == JPGPU_CODING_DCT_PROGRESSIVE && have_coefficients)
            for (size_t i = 0; i < ncomp; i++) {  // render what we have of unfinished components
                if (finished_mask[i] == ~(uint64_t)0) continue;
                const jpgpu_component &c = f.components[i];
                if (!has_qt[c.quantization_table_index]) continue;
                memcpy(plane_qt[i], qt[c.quantization_table_index], 128);
                sink->frame_slot_hint((uint32_t)i, (uint32_t)i);
                sink->start((uint32_t)i, c, qt[c.quantization_table_index]);
                const size_t per_row = (size_t)c.block_width * c.vertical_sampling_factor * 64;
                for (uint32_t my = 0; my < f.mcu_h; my++) {
                    const size_t off = (size_t)my * per_row;
                    if (off + per_row > coefficients[i].size()) fail(JPGPU_ERR_INTERNAL, "reference would panic: coefficient row slice");
                    sink->append_row((uint32_t)i, coefficients[i].data() + off, per_row);
                }
                sink->finish((uint32_t)i, (uint32_t)i);
                plane_present[i] = true;
            }
        for (size_t i = 0; i < ncomp; i++)  // compute_image, src/decoder.rs:1306-1308
            if (!plane_present[i]) fail(JPGPU_ERR_FORMAT, "not all components have data");
    }

    int determine_color_transform() const {  // src/decoder.rs:698-764
        if (ct_override >= 0) return ct_override;
        const auto &c = frame.components;
        if (c.size() == 1) return JPGPU_CT_GRAYSCALE;
        if (c.size() == 3) {
            const uint8_t a = c[0].identifier, b = c[1].identifier, d = c[2].identifier;
            if (a == 1 && b == 2 && d == 3) return JPGPU_CT_YCBCR;
            if (a == 1 && b == 34 && d == 35) return JPGPU_CT_JCS_BG_YCC;
            if (a == 82 && b == 71 && d == 66) return JPGPU_CT_RGB;
            if (a == 114 && b == 103 && d == 98) return JPGPU_CT_JCS_BG_RGB;
            if (is_jfif) return JPGPU_CT_YCBCR;
        }
        if (has_adobe) {
            if (adobe == 0) {
                if (c.size() == 3) return JPGPU_CT_RGB;
                if (c.size() == 4) return JPGPU_CT_CMYK;
            } else {
                return adobe == 1 ? JPGPU_CT_YCBCR : JPGPU_CT_YCCK;
            }
        } else if (c.size() == 4) {
            return JPGPU_CT_CMYK;
        }
        if (c.size() == 4) return JPGPU_CT_YCCK;
        if (c.size() == 3) return JPGPU_CT_YCBCR;
        return JPGPU_CT_UNKNOWN;
    }
};

Frontend::Frontend(const uint8_t *data, size_t len) : impl_(new Impl) {
    impl_->bytes.assign(data, data + len);
    impl_->src.p = impl_->bytes.data();
    impl_->src.len = len;
}
Frontend::Frontend(const uint8_t *data, size_t len, Borrowed) : impl_(new Impl) {
    impl_->src.p = data;
    impl_->src.len = len;
}
Frontend::~Frontend() {}

const uint8_t *Frontend::stream_bytes(size_t *len) const {
    if (len) *len = impl_->src.len;
    return impl_->src.p;
}

void Frontend::read_info() { impl_->run(true, nullptr); }

void Frontend::scale(uint16_t req_w, uint16_t req_h, uint16_t &out_w, uint16_t &out_h) {  // src/decoder.rs:278-290
    read_info();
    FrameInfo &f = impl_->frame;
    // choose_idct_size, src/idct.rs:14-28
    uint32_t idct = 8;
    for (uint32_t s : {1u, 2u, 4u}) {
        const uint16_t sw = (uint16_t)(((uint32_t)f.image_w * s - 1) / 8 + 1), sh = (uint16_t)(((uint32_t)f.image_h * s - 1) / 8 + 1);
        if (sw >= req_w || sh >= req_h) {
            idct = s;
            break;
        }
    }
    for (auto &c : f.components) c.dct_scale = idct;  // FrameInfo::update_idct_size, src/parser.rs:119-134
    update_component_sizes(f.image_w, f.image_h, f.components, f.mcu_w, f.mcu_h);
    f.output_w = (uint16_t)std::ceil((float)f.image_w * (float)idct / 8.0f);
    f.output_h = (uint16_t)std::ceil((float)f.image_h * (float)idct / 8.0f);
    out_w = f.output_w;
    out_h = f.output_h;
}

void Frontend::decode_to(RowSink &sink) { impl_->run(false, &sink); }
bool Frontend::plan_device_scans(std::vector<PlannedScan> &scans) {
    scans.clear();
    impl_->plan = &scans;
    bool ok = true;
    try {
        impl_->run(false, nullptr);
        ok = !scans.empty();
    } catch (const Impl::NotEligible &ne) {
        if (getenv("JPGPU_PLAN_TRACE")) fprintf(stderr, "plan_device_scans: not eligible (check %d)\n", ne.where);
        ok = false;
    } catch (const DecodeError &) {
        ok = false;  // the host decoder reports it
    }
    impl_->plan = nullptr;
    if (!ok) scans.clear();
    return ok;
}
bool Frontend::plan_progressive_scans(ProgPlan &plan) {
    plan.scans.clear();
    plan.n_tracks = 0;
    Impl &im = *impl_;
    im.prog_plan = &plan;
    memset(im.prog_state, 0, sizeof(im.prog_state));
    for (int c = 0; c < JPGPU_MAX_COMPONENTS; c++)
        for (int k = 0; k < 64; k++) im.prog_cell_track[c][k] = (uint16_t)(c * 64 + k);
    bool ok = true;
    try {
        im.run(false, nullptr);
        ok = !plan.scans.empty();
    } catch (const Impl::NotEligible &ne) {
        if (getenv("JPGPU_PLAN_TRACE")) fprintf(stderr, "plan_progressive_scans: not eligible (check %d)\n", ne.where);
        ok = false;
    } catch (const DecodeError &) {
        ok = false;  // the host decoder reports it
    }
    im.prog_plan = nullptr;
    if (ok) {  // cells -> track numbers, in order of first appearance
        uint16_t roots[256];
        uint32_t n = 0;
        for (ProgPlannedScan &ps : plan.scans) {
            const uint16_t root = im.prog_find((uint16_t)ps.track);
            uint32_t t = 0;
            while (t < n && roots[t] != root) t++;
            if (t == n) roots[n++] = root;
            ps.track = t;
        }
        plan.n_tracks = n;
    } else {
        plan.scans.clear();
    }
    return ok;
}
bool Frontend::has_frame() const { return impl_->has_frame; }

jpgpu_image_info Frontend::info() const {  // src/decoder.rs:170-197
    const FrameInfo &f = impl_->frame;
    jpgpu_image_info i{};
    i.width = f.output_w;
    i.height = f.output_h;
    i.coding_process = f.coding_process;
    const size_t n = f.components.size();
    i.pixel_format = n == 1 ? (f.precision <= 8 ? JPGPU_PIXEL_L8 : JPGPU_PIXEL_L16) : (n == 3 ? JPGPU_PIXEL_RGB24 : JPGPU_PIXEL_CMYK32);
    return i;
}
int Frontend::color_transform() const { return impl_->determine_color_transform(); }
uint32_t Frontend::ncomp() const { return (uint32_t)impl_->frame.components.size(); }
const jpgpu_component *Frontend::components() const { return impl_->frame.components.data(); }
uint16_t Frontend::output_width() const { return impl_->frame.output_w; }
uint16_t Frontend::output_height() const { return impl_->frame.output_h; }
const bool *Frontend::planes_present() const { return impl_->plane_present; }
const uint16_t *Frontend::qtable_of_component(uint32_t c) const { return impl_->plane_qt[c]; }
void Frontend::set_color_transform(int ct) { impl_->ct_override = ct; }
void Frontend::set_max_decoding_buffer_size(size_t n) { impl_->buffer_limit = n; }
size_t Frontend::max_decoding_buffer_size() const { return impl_->buffer_limit; }
const std::vector<uint8_t> *Frontend::exif() const { return impl_->has_exif ? &impl_->exif : nullptr; }
const std::vector<uint8_t> *Frontend::xmp() const { return impl_->has_xmp ? &impl_->xmp : nullptr; }

bool Frontend::icc_profile(std::vector<uint8_t> &out) const {  // src/decoder.rs:213-243
    const auto &chunks = impl_->icc;
    const size_t n = chunks.size();
    if (n == 0 || n >= 255) return false;
    const IccChunk *present[256] = {nullptr};
    for (const auto &c : chunks) {
        if (c.num_markers != n || c.seq_no == 0 || present[c.seq_no]) return false;
        present[c.seq_no] = &c;
    }
    out.clear();
    for (size_t i = 1; i <= n; i++) {
        if (!present[i]) return false;
        out.insert(out.end(), present[i]->data.begin(), present[i]->data.end());
    }
    return true;
}

}  // namespace host
}  // namespace jpgpu
