// huff.hip — kernels of the device entropy decoder (huff_sync_core.hpp: sync passes with speculative emission, block numbering,
// expansion of the entry lists into whole blocks, DC sums of scans whose components share their tables) and the range scan for
// coefficients a caller's own kernels put into an arena (jpgpu_batch_classify_on_device / _scan_ranges, the Worker's fused route).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "huff.hpp"
#include "huff_core.hpp"
#include "huff_prog_wave.hpp"
#include "huff_unstuff_core.hpp"
#include "huff_sync_core.hpp"
#include "range_stats.hpp"

namespace jpgpu {

// the lanes' range by-product -> the image's statistics
__device__ __forceinline__ void publish_range(uint32_t *stats, HuffRange rg) { stat_publish_wave(stats, rg.dc, rg.ac); }

// max |c*q| and the largest block-column sum of |c*q| per plane (the two quantities behind the range classes of
// include/jpgpu.h).  Eight lanes per block, one 16-byte row each: a wave reads 1 KB of consecutive coefficients per load
// (one lane per block — every lane on its own cache line — ran at 1.2 TB/s).  The column sums are folded across the
// eight lanes by halving (4 + 2 + 1 exchanges), the maxima per wave, then atomicMax.
struct RangeView {  // what the scan needs of a RangeJob (the table through a pointer: the job's own copy, or device memory)
    const int16_t *coefs;
    uint32_t n_blocks, slot;
    const uint16_t *q;
};
__device__ __forceinline__ void range_scan_body(const RangeView job, uint32_t *__restrict__ stats) {
    const uint32_t row = threadIdx.x & 7u, first = blockIdx.x * 256u;
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[RS_WORDS * job.slot + RS_COL_EXACT] = 1u;  // whole blocks in view: exact column sums
    if (first >= job.n_blocks) return;
    uint32_t q[8];
    {
        const v4u qv = ((const JP_GLOBAL v4u *)job.q)[row];
        const uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) q[k] = (w[k >> 1] >> (16u * (k & 1u))) & 0xffffu;
    }
    uint32_t max_abs = 0, max_col = 0;
    const uint32_t last = min(first + 256u, job.n_blocks);
    // all eight rows this lane is going to look at are requested before the first one is used (one load per iteration, each
    // waited for, kept the kernel at 1.8 TB/s: 93 % of the wave cycles were spent waiting for memory)
    v4u rows[8];
#pragma unroll
    for (uint32_t it = 0; it < 8u; it++) {
        const uint32_t blk = first + (threadIdx.x >> 3) + 32u * it;
        rows[it] = v4u{0u, 0u, 0u, 0u};
        if (blk < last) rows[it] = stream_load((const JP_GLOBAL v4u *)(job.coefs + (size_t)blk * 64u) + row);
    }
#pragma unroll
    for (uint32_t it = 0; it < 8u; it++) {  // (uniform trip count: shuffles inside)
        const uint32_t blk = first + (threadIdx.x >> 3) + 32u * it;
        uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (blk < last) {
            const v4u v = rows[it];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const int32_t c = (int16_t)(uint16_t)(w[k >> 1] >> (16u * (k & 1u)));
                const uint32_t p = (uint32_t)(c < 0 ? -c : c) * q[k];  // <= 32768 * 65535 < 2^31
                max_abs = p > max_abs ? p : max_abs;
                a[k] = p < 0x00ffffffu ? p : 0x00ffffffu;               // saturate the addends: 8 of them cannot wrap
            }
        }
        // rows -> column sums: after the three steps lane `row` holds the sum of column (row bit-reversed... any one column)
        uint32_t b4[4], b2[2];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t keep = (row & 4u) ? a[k + 4u] : a[k], give = (row & 4u) ? a[k] : a[k + 4u];
            b4[k] = keep + (uint32_t)__shfl_xor((int)give, 4);
        }
#pragma unroll
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t keep = (row & 2u) ? b4[k + 2u] : b4[k], give = (row & 2u) ? b4[k] : b4[k + 2u];
            b2[k] = keep + (uint32_t)__shfl_xor((int)give, 2);
        }
        const uint32_t keep = (row & 1u) ? b2[1] : b2[0], give = (row & 1u) ? b2[0] : b2[1];
        const uint32_t col = keep + (uint32_t)__shfl_xor((int)give, 1);
        max_col = col > max_col ? col : max_col;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_abs = max(max_abs, (uint32_t)__shfl_xor((int)max_abs, off));
        max_col = max(max_col, (uint32_t)__shfl_xor((int)max_col, off));
    }
    // one pair of atomics per workgroup (not per wave: atomics — and even plain L2-scope reads — of a few hundred neighbouring
    // words by every wave of the launch are what this kernel would otherwise spend its time on)
    __shared__ uint32_t wg_max[2][4];
    if ((threadIdx.x & 63u) == 0u) {
        wg_max[0][threadIdx.x >> 6] = max_abs;
        wg_max[1][threadIdx.x >> 6] = max_col;
    }
    __syncthreads();
    if (threadIdx.x < 2u) {  // (exact column sums: the image's RS_COL_EXACT word is set by the launcher's fill)
        const uint32_t v = max(max(wg_max[threadIdx.x][0], wg_max[threadIdx.x][1]), max(wg_max[threadIdx.x][2], wg_max[threadIdx.x][3]));
        stat_raise(&stats[RS_WORDS * job.slot + (threadIdx.x ? RS_MAX_COL : RS_MAX_AC)], v);
    }
}
__global__ __launch_bounds__(256) void range_scan_kernel(const RangeJob *__restrict__ jobs, uint32_t *__restrict__ stats) {
    const RangeJob &job = jobs[blockIdx.y];
    range_scan_body(RangeView{job.coefs, job.n_blocks, job.slot, job.q}, stats);
}
// one plane, its table in device memory (the Worker's fused route: jpgpu.cpp)
__global__ __launch_bounds__(256) void range_scan_one_kernel(const int16_t *__restrict__ coefs, uint32_t n_blocks, const uint16_t *__restrict__ q,
                                                             uint32_t *__restrict__ stats) {
    range_scan_body(RangeView{coefs, n_blocks, 0u, q}, stats);
}

// ---- "host light": the staging pass on the device (huff_unstuff_core.hpp) ---------------------------------------------------------------
// grid = (ceil(max pieces), jobs), 256 lanes x 16 bytes per workgroup.  A lane's 16-byte piece, the byte in front of it and the byte
// behind it (neighbours' registers through LDS; the workgroup's first and last lane read theirs from memory).
struct UnstuffView {
    uint32_t w[4];
    uint32_t keep;   // which of the 16 bytes stay
    uint32_t first, last;
    bool bad;
};
__device__ __forceinline__ UnstuffView unstuff_view(const UnstuffJob &job, uint32_t piece, JP_LDS uint8_t *edge /* [2][256] */) {
    const uint32_t lead = (uint32_t)((uintptr_t)job.raw & 15u), total = lead + job.raw_bytes;  // bytes from the aligned base to the scan's end
    const JP_GLOBAL v4u *base = (const JP_GLOBAL v4u *)(job.raw - lead);
    const uint32_t c = piece * 256u + threadIdx.x, lo = c * 16u;  // this lane's 16 bytes: [lo, lo + 16) from the aligned base
    UnstuffView v;
    v.bad = false;
    v4u x = v4u{0u, 0u, 0u, 0u};
    if (lo < total) x = stream_load(base + c);
    v.w[0] = x.x, v.w[1] = x.y, v.w[2] = x.z, v.w[3] = x.w;
    edge[threadIdx.x] = (uint8_t)(x.w >> 24);          // my last byte, for the lane behind me
    edge[256u + threadIdx.x] = (uint8_t)(x.x & 0xffu);  // my first byte, for the lane in front of me
    __syncthreads();
    const JP_GLOBAL uint8_t *bytes = (const JP_GLOBAL uint8_t *)(job.raw - lead);
    uint32_t prev = 0, next = 0;
    if (lo > lead && lo < total) prev = threadIdx.x ? edge[threadIdx.x - 1u] : bytes[lo - 1u];
    const bool has_next = lo + 16u < total;
    if (has_next) next = threadIdx.x < 255u ? edge[256u + threadIdx.x + 1u] : bytes[lo + 16u];
    v.first = lo >= lead ? 0u : min(16u, lead - lo);
    v.last = lo >= total ? 0u : min(16u, total - lo);
    v.keep = unstuff_piece_flags(v.w, prev, next, has_next, v.first, v.last, v.bad);
    __syncthreads();
    return v;
}
__global__ __launch_bounds__(256) void huff_unstuff_count_kernel(const UnstuffJob *__restrict__ jobs) {
    __shared__ uint8_t edge[512];
    __shared__ uint32_t wave_sum[4];
    const UnstuffJob &job = jobs[blockIdx.y];
    if (blockIdx.x >= job.n_pieces) return;
    const UnstuffView v = unstuff_view(job, blockIdx.x, (JP_LDS uint8_t *)edge);
    uint32_t kept = (uint32_t)__popc(v.keep);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += (uint32_t)__shfl_xor((int)kept, off);
    if ((threadIdx.x & 63u) == 0u) wave_sum[threadIdx.x >> 6] = kept;
    const bool any_bad = __syncthreads_or(v.bad);
    if (threadIdx.x == 0) {
        job.piece_kept[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        if (any_bad) atomicOr(job.status, 1u | 16u);  // something other than 0xFF00 pairs inside the scan: the host decodes this image
    }
}
// one workgroup per job: piece_kept -> exclusive prefix sums (in place, n_pieces + 1 entries: the last is the unstuffed length), and the
// job record's length fields (what the host's staging task fills in on the other route: batch.cpp)
__global__ __launch_bounds__(256) void huff_unstuff_scan_kernel(const UnstuffJob *__restrict__ jobs) {
    __shared__ uint32_t wave_tot[4];
    const UnstuffJob &job = jobs[blockIdx.x];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < job.n_pieces; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < job.n_pieces ? job.piece_kept[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
            if ((threadIdx.x & 63u) >= (uint32_t)off) incl += o;
        }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) {
            before += w < (threadIdx.x >> 6) ? wave_tot[w] : 0u;
            total += wave_tot[w];
        }
        if (i < job.n_pieces) job.piece_kept[i] = carry + before + incl - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        job.piece_kept[job.n_pieces] = carry;
        const bool refused = (*job.status & 1u) != 0u;
        HuffSyncJob *sj = job.job;
        sj->n_bits = refused ? 0u : carry * 8u;
        uint32_t n_chunks = (uint32_t)(((uint64_t)carry * 8u + (1u << sj->chunk_shift) - 1u) >> sj->chunk_shift);  // (huff_sync_chunks)
        if (n_chunks == 0u) n_chunks = 1u;
        sj->n_chunks = refused ? 0u : n_chunks;
        sj->data_dwords = (carry + 3u) / 4u;  // (the weave reads nothing beyond: what follows the data counts as zeros)
    }
}
// The bytes a piece keeps, gathered in LDS where they will lie in the slot — shifted by the slot offset's low two bits, so that LDS
// dword d IS slot dword (o0 / 4 + d) — and written out as whole dwords, coalesced; the (up to three) bytes in front of the first whole
// dword and behind the last go out one by one: the neighbouring pieces' workgroups write the other bytes of those dwords.  (First
// version: every lane stored its up to sixteen bytes one by one, 0.1 G byte stores per 256 files: 177 us against 50 for the count pass
// that reads the same bytes.)
__global__ __launch_bounds__(256) void huff_unstuff_compact_kernel(const UnstuffJob *__restrict__ jobs) {
    __shared__ uint8_t edge[512];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t gathered[UNSTUFF_PIECE / 4u + 2u];
    const UnstuffJob &job = jobs[blockIdx.y];
    if (blockIdx.x >= job.n_pieces || (*job.status & 1u)) return;
    const UnstuffView v = unstuff_view(job, blockIdx.x, (JP_LDS uint8_t *)edge);
    const uint32_t mine = (uint32_t)__popc(v.keep);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if ((threadIdx.x & 63u) >= (uint32_t)off) incl += o;
    }
    if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4u; w++) before += w < (threadIdx.x >> 6) ? wave_tot[w] : 0u;
    const uint32_t o0 = job.piece_kept[blockIdx.x], n = job.piece_kept[blockIdx.x + 1u] - o0, sh = o0 & 3u;
    JP_LDS uint8_t *g = (JP_LDS uint8_t *)gathered;
    {
        uint32_t at = sh + before + incl - mine, keep = v.keep;
        if (keep == 0xffffu && (at & 3u) == 0u) {  // (nineteen pieces in twenty hold no 0xFF at all)
            JP_LDS uint32_t *gw = (JP_LDS uint32_t *)(g + at);
            gw[0] = v.w[0], gw[1] = v.w[1], gw[2] = v.w[2], gw[3] = v.w[3];
        } else {
            while (keep) {
                const uint32_t j = (uint32_t)__builtin_ctz(keep);
                keep &= keep - 1u;
                g[at++] = (uint8_t)(v.w[j >> 2] >> (8u * (j & 3u)));
            }
        }
    }
    __syncthreads();
    JP_GLOBAL uint8_t *dst = (JP_GLOBAL uint8_t *)job.dst;
    JP_GLOBAL uint32_t *dstw = (JP_GLOBAL uint32_t *)job.dst + (o0 >> 2);  // (slots are 16-byte aligned)
    const uint32_t end = sh + n, whole0 = sh ? 1u : 0u, whole1 = end >> 2;  // LDS dwords [whole0, whole1) are all this piece's
    for (uint32_t d = whole0 + threadIdx.x; d < whole1; d += 256u) dstw[d] = gathered[d];
    if (threadIdx.x < 4u) {  // in front of the first whole dword
        const uint32_t bpos = threadIdx.x;
        if (sh && bpos >= sh && bpos < (end < 4u ? end : 4u)) dst[(o0 - sh) + bpos] = g[bpos];
    } else if (threadIdx.x < 8u) {  // behind the last one
        const uint32_t t0 = whole1 * 4u > (sh ? 4u : 0u) ? whole1 * 4u : (sh ? 4u : 0u), bpos = t0 + (threadIdx.x - 4u);
        if (bpos < end) dst[(o0 - sh) + bpos] = g[bpos];
    }
    // behind the scan's last byte: zeros up to the dword boundary (the weave reads whole dwords)
    const uint32_t total = job.piece_kept[job.n_pieces];
    if (threadIdx.x == 0u && n && o0 + n == total)
        for (uint32_t z = total; z & 3u; z++) dst[z] = 0u;
}
hipError_t launch_huff_unstuff(const UnstuffJob *d_jobs, uint32_t n_jobs, uint32_t max_pieces, hipStream_t stream) {
    if (n_jobs == 0 || max_pieces == 0) return hipSuccess;
    huff_unstuff_count_kernel<<<dim3(max_pieces, n_jobs), dim3(256), 0, stream>>>(d_jobs);
    huff_unstuff_scan_kernel<<<dim3(n_jobs), dim3(256), 0, stream>>>(d_jobs);
    huff_unstuff_compact_kernel<<<dim3(max_pieces, n_jobs), dim3(256), 0, stream>>>(d_jobs);
    return hipGetLastError();
}

// ---- the weave (huff_job.hpp): the staged scans, 64 chunks side by side -------------------------------------------------
// grid = (ceil(max chunks / 64), sync jobs): one workgroup per tile, 64 rows at a time through a 64 x 64 transposition in LDS —
// every read and every write of the arena is a wave's 256 consecutive bytes.  Reads 1 x the scans, writes 1 x: ~0.2 GB per 256
// 1080p files against the 1.4 GB of misses it spares the passes.
__global__ __launch_bounds__(256) void huff_weave_kernel(const HuffSyncJob *__restrict__ jobs) {
    __shared__ uint32_t t[64][65];
    __shared__ uint32_t first_dword[64];
    const HuffSyncJob &job = jobs[blockIdx.y];
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (tile * HUFF_WEAVE_LANES >= job.n_chunks) return;
    const uint32_t H = huff_weave_height(job.chunk_shift);
    if (threadIdx.x < 64u) {
        const uint32_t i = tile * HUFF_WEAVE_LANES + threadIdx.x;
        first_dword[threadIdx.x] = i < job.n_chunks ? huff_chunk_span(job, i).start >> 5 : 0xffffffffu;  // (none: zeros)
    }
    __syncthreads();
    JP_GLOBAL uint32_t *dst = (JP_GLOBAL uint32_t *)job.weave + (size_t)tile * H * HUFF_WEAVE_LANES;
    for (uint32_t r0 = 0; r0 < H; r0 += 64u) {
        for (uint32_t c = wave; c < 64u; c += 4u) {  // column c: 64 consecutive dwords of chunk c
            const uint32_t f = first_dword[c];
            t[c][lane] = (f != 0xffffffffu && r0 + lane < H) ? huff_weave_value(job, f, r0 + lane) : 0u;
        }
        __syncthreads();
        for (uint32_t r = wave; r < 64u; r += 4u)
            if (r0 + r < H) dst[(size_t)(r0 + r) * HUFF_WEAVE_LANES + lane] = t[lane][r];
        __syncthreads();
    }
}

// ---- scans without restart markers: the self-synchronising chunk decoder (huff_sync_core.hpp) -------------------------
// grid = (ceil(max chunks / 256), sync jobs); lane = chunk.  `changed` of a job holds three counters used in rotation by
// consecutive launches: launch t counts into slot t % 3, reads slot (t-1) % 3 (zero: the job has settled, nothing to do)
// and clears slot (t+1) % 3, so the host can enqueue a fixed number of launches without looking at the device in between.
constexpr uint32_t SYNC_NT = HUFF_SYNC_LANES;

template <uint32_t NT, uint32_t TABLES = 8u>
__device__ __forceinline__ void sync_load_lds(JP_LDS HuffSyncLds &L, const HuffSyncJob *gj) {
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)gj;
        JP_LDS uint32_t *dst = (JP_LDS uint32_t *)&L.job;
        for (uint32_t i = threadIdx.x; i < sizeof(HuffSyncJob) / 4u; i += NT) dst[i] = src[i];
    }
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)gj->tables;
        JP_LDS uint32_t *dst = (JP_LDS uint32_t *)L.tables;
        for (uint32_t i = threadIdx.x; i < TABLES * sizeof(DevHuffTable) / 4u; i += NT) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t l = threadIdx.x; l < 512u; l += NT) huff_sync_fill_lds(L, l);
    __syncthreads();
}

// does chunk i have a start state it has not decoded from yet?  (what huff_sync_chunk decides itself, ahead of the call)
__device__ __forceinline__ bool sync_chunk_has_work(const HuffSyncJob *gj, uint32_t i, uint32_t pass) {
    if (i >= gj->n_chunks) return false;
    if (pass == 0u) return true;
    const HuffChunkSpan span = huff_chunk_span(*gj, i);
    uint32_t p = span.start, qk = 0u;  // (a chunk at the start of the scan or of a restart segment: the truth)
    if (!span.first) {
        p = huff_load_shared(gj->out_pos + (i - 1u));
        qk = huff_load_shared(gj->out_qk + (i - 1u));
    }
    if (gj->uniform) qk &= 0xffu;
    const uint32_t first = span.start;  // (huff_sync_state_plausible)
    if (!span.first && !(p >= first && p - first <= 32u && (qk >> 8) < gj->bpm && (qk & 0xffu) < 64u)) return false;
    if (gj->emit != nullptr) qk |= QK_EMITTED;  // (a state decoded from in pass 0, without emission, is work again)
    return p != gj->in_pos[i] || qk != gj->in_qk[i];
}

// TABLES = 4: every job of the launch uses Huffman table ids 0 and 1 only — the LDS image ends behind their four slots
// (HUFF_SYNC_LDS_COMPACT_BYTES: 24 kB, six workgroups per CU where the full 40 kB allow four).
template <uint32_t TABLES>
__global__ __launch_bounds__(SYNC_NT, 4) void huff_sync_pass_kernel(const HuffSyncJob *__restrict__ jobs, uint32_t launch, uint32_t first_pass,
                                                                                          uint32_t iters) {
    static_assert(TABLES == 4u || TABLES == 8u, "");
    __shared__ alignas(16) uint8_t L_raw[TABLES == 8u ? sizeof(HuffSyncLds) : HUFF_SYNC_LDS_COMPACT_BYTES];
    JP_LDS HuffSyncLds &L = *(JP_LDS HuffSyncLds *)L_raw;
    __shared__ uint16_t todo[SYNC_NT];              // chunks (relative to the workgroup's first) with work, packed to the front
    __shared__ uint32_t wave_cnt[SYNC_NT / 64u];
    const HuffSyncJob *gj = &jobs[blockIdx.y];
    uint32_t *cnt = gj->changed;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[(launch + 1u) % 3u] = 0u;
    if (blockIdx.x * SYNC_NT >= gj->n_chunks) return;
    if (launch > 0u && cnt[(launch - 1u) % 3u] == 0u) return;
    const uint32_t i = blockIdx.x * SYNC_NT + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // does any lane of this workgroup have a new start state?  (after the first launches most do not: skip the table load)
    if (!__syncthreads_or(sync_chunk_has_work(gj, i, first_pass))) return;
    sync_load_lds<SYNC_NT, TABLES>(L, gj);
    bool published = false;
    for (uint32_t it = 0; it < iters; it++) {
        // Lanes with work are packed into as few waves as possible: a wave costs the same with one busy lane as with 64, and
        // from the third pass on a few per cent of the chunks are still being corrected.
        const bool need = sync_chunk_has_work(gj, i, first_pass + it);
        const uint64_t m = __ballot(need);
        if (lane == 0u) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < SYNC_NT / 64u; w++) {
            const uint32_t c = wave_cnt[w];
            before += w < wave ? c : 0u;
            total += c;
        }
        if (need) todo[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)threadIdx.x;
        __syncthreads();
        if (threadIdx.x < total) published |= huff_sync_chunk(L, blockIdx.x * SYNC_NT + todo[threadIdx.x], first_pass + it);
        __syncthreads();
    }
    const uint32_t n_pub = (uint32_t)__syncthreads_count(published);
    if (threadIdx.x == 0 && n_pub) atomicAdd(cnt + launch % 3u, n_pub);
}

// Late launches (passes >= 2): a few per cent of the chunks are still being corrected — nearly every workgroup of the full grid would
// hold one or two of them and sit on its wave slots and 23 kB of LDS for a whole chunk walk (0.4 ms) with one busy lane, and with a
// dozen sub-batches in flight those mostly idle workgroups fill the device's slots and keep the FULL passes of other sub-batches
// waiting (per-dispatch counters of a late launch: 70 % of the wave-cycles of a full one for 3 % of its work;
// tools/attic/probe_concurrency.hip: 1,792 workgroups of this footprint fit the device).  So a late launch gives one workgroup
// SYNC_LATE_SPAN blocks of 256 chunks: it gathers the chunks with work from all of them, 256 at a time, and walks those.
constexpr uint32_t SYNC_LATE_SPAN = 8u;
template <uint32_t TABLES>
__global__ __launch_bounds__(SYNC_NT, 4) void huff_sync_late_kernel(const HuffSyncJob *__restrict__ jobs, uint32_t launch, uint32_t first_pass, uint32_t iters) {
    static_assert(TABLES == 4u || TABLES == 8u, "");
    __shared__ alignas(16) uint8_t L_raw[TABLES == 8u ? sizeof(HuffSyncLds) : HUFF_SYNC_LDS_COMPACT_BYTES];
    JP_LDS HuffSyncLds &L = *(JP_LDS HuffSyncLds *)L_raw;
    __shared__ uint32_t todo[SYNC_NT];  // chunks with work, gathered from the span
    __shared__ uint32_t wave_cnt[SYNC_NT / 64u];
    const HuffSyncJob *gj = &jobs[blockIdx.y];
    uint32_t *cnt = gj->changed;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[(launch + 1u) % 3u] = 0u;
    const uint32_t first = blockIdx.x * (SYNC_LATE_SPAN * SYNC_NT), n = gj->n_chunks;
    if (first >= n) return;
    if (launch > 0u && cnt[(launch - 1u) % 3u] == 0u) return;
    const uint32_t last = min(n, first + SYNC_LATE_SPAN * SYNC_NT), lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool published = false, loaded = false;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t pass = first_pass + it;
        uint32_t filled = 0;  // work items gathered so far (the same in every lane)
        for (uint32_t base = first; base < last || filled; base += SYNC_NT) {
            bool need = false;
            uint32_t before = 0, total = 0;
            if (base < last) {
                need = base + threadIdx.x < last && sync_chunk_has_work(gj, base + threadIdx.x, pass);
                const uint64_t m = __ballot(need);
                if (lane == 0u) wave_cnt[wave] = (uint32_t)__popcll(m);
                __syncthreads();
#pragma unroll
                for (uint32_t w = 0; w < SYNC_NT / 64u; w++) {
                    const uint32_t c = wave_cnt[w];
                    before += w < wave ? c : 0u;
                    total += c;
                }
                before += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            }
            if (filled + total > SYNC_NT || (base >= last && filled)) {  // the list is full (or the span is through): walk what it holds
                if (!loaded) {
                    sync_load_lds<SYNC_NT, TABLES>(L, gj);
                    loaded = true;
                }
                if (threadIdx.x < filled) published |= huff_sync_chunk(L, todo[threadIdx.x], pass, threadIdx.x);
                __syncthreads();
                filled = 0;
            }
            if (need) todo[filled + before] = base + threadIdx.x;
            filled += total;
            __syncthreads();
        }
    }
    const uint32_t n_pub = (uint32_t)__syncthreads_count(published);
    if (threadIdx.x == 0 && n_pub) atomicAdd(cnt + launch % 3u, n_pub);
}

// per job: unsettled after the last launch -> host; else blocks per chunk -> number of each chunk's first block
__global__ __launch_bounds__(SYNC_NT) void huff_sync_scan_kernel(const HuffSyncJob *__restrict__ jobs, uint32_t last_launch) {
    __shared__ uint32_t wave_tot[SYNC_NT / 64u];
    const HuffSyncJob &job = jobs[blockIdx.x];
    if (job.changed[last_launch % 3u] != 0u) {
        if (threadIdx.x == 0) atomicOr(job.status, 1u | 64u);
        return;
    }
    uint32_t carry = 0, bad = 0;
    const bool emits = job.emit != nullptr;
    if (job.n_seg > 1u) {  // restart segments: each is numbered on its own, by one thread (huff_emit_segment_scan)
        for (uint32_t seg = threadIdx.x; seg < job.n_seg; seg += SYNC_NT) bad |= huff_emit_segment_scan(job, seg);
        if (bad) atomicOr(job.status, bad);
        return;
    }
    for (uint32_t base = 0; base < job.n_chunks; base += SYNC_NT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < job.n_chunks ? job.n_blocks[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
            if ((threadIdx.x & 63u) >= (uint32_t)off) incl += o;
        }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < SYNC_NT / 64u; w++) {
            const uint32_t t = wave_tot[w];
            before += w < (threadIdx.x >> 6) ? t : 0u;
            total += t;
        }
        if (i < job.n_chunks) job.n_blocks[i] = carry + before + incl - v;
        if (emits && i < job.n_chunks) bad |= huff_emit_chunk_status(job, i, carry + before + incl);
        carry += total;
        __syncthreads();
    }
    if (emits) {
        if (threadIdx.x == 0) bad |= huff_emit_final_status(job, carry);
        if (bad) atomicOr(job.status, bad);
    }
    if (job.uniform) return;
    // sums of DC differences per chunk and component -> the predictors every chunk starts from (mod 2^16)
    uint32_t dcarry[4] = {0, 0, 0, 0};
    for (uint32_t base = 0; base < job.n_chunks; base += SYNC_NT) {
        const uint32_t i = base + threadIdx.x;
        uint32_t w0 = 0, w1 = 0;
        if (i < job.n_chunks) {
            w0 = job.dc_sum[2u * i];
            w1 = job.dc_sum[2u * i + 1u];
        }
        const uint32_t v4[4] = {w0 & 0xffffu, w0 >> 16, w1 & 0xffffu, w1 >> 16};
        uint32_t excl[4];
#pragma unroll
        for (uint32_t f = 0; f < 4; f++) {
            uint32_t incl = v4[f];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                if ((threadIdx.x & 63u) >= (uint32_t)off) incl += o;
            }
            if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < SYNC_NT / 64u; w++) {
                const uint32_t t = wave_tot[w];
                before += w < (threadIdx.x >> 6) ? t : 0u;
                total += t;
            }
            excl[f] = (dcarry[f] + before + incl - v4[f]) & 0xffffu;
            dcarry[f] += total;
            __syncthreads();
        }
        if (i < job.n_chunks) {
            job.dc_sum[2u * i] = excl[0] | (excl[1] << 16);
            job.dc_sum[2u * i + 1u] = excl[2] | (excl[3] << 16);
        }
    }
}

// ---- speculative emission -> whole blocks (HuffSyncJob::emit) ----------------------------------------------------------
// One WAVE per chunk (no workgroup barriers inside): it reads the chunk's entries 64 at a time, numbers the blocks they belong
// to (a ballot of the "first of a block" bits), scatters the values into a ring of block images in LDS and writes every block
// that is complete as one 128-byte line, eight lanes each — the arena gets every block of the scan exactly once, zeros
// included, so nobody has to clear it first, and the 2-byte scatter happens in LDS instead of HBM.  A block
// belongs to the chunk it STARTS in; the wave follows it through the leading entries of the chunks after.
// All of a chunk's entries are requested before the first is used (up to EXP_LOADS x 64 per round): a wave that waits for
// every 256 bytes in turn would leave the memory system idle.
// EXP_THR: complete blocks waiting in the ring before a store round (8: full rounds only — store rounds after every batch of
// entries measured 0.67 ms against 0.61); EXP_CHUNKS: chunks per wave; EXP_LOADS x 64: entries requested ahead per chunk
// (profiles/round3/14_emission_path.txt).
constexpr uint32_t EXP_WAVES = 4, EXP_CHUNKS = 8, EXP_SLOT = 68, EXP_THR = 8, EXP_LOADS = 16;
constexpr uint32_t EXP_SLOTS = EXP_THR + 65u;  // complete blocks that may wait + 1 open + 64 new
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct ExpandLds {
    HuffSyncJob job;
    HuffBlockDst q_dst[16];
    uint32_t wg_rg[2][EXP_WAVES];
    uint8_t unzig[64];  // an entry carries its coefficient's zig-zag index (one table read less per symbol in the sync passes' chain)
    alignas(16) uint16_t ring[EXP_WAVES][EXP_SLOTS * EXP_SLOT];  // per wave: block images of 136 bytes (8 more than a block: the DC
                                                                 // entries of a run of flat blocks do not all meet in one bank)
};
static_assert(sizeof(HuffSyncJob) % 4 == 0, "copied by dwords");

// Block numbers -> MCU coordinates without a division per block: the wave knows where block S, the first one that starts in
// its chunk, lies (real divisions, once per chunk), every other block is S + d with d < 2^16, and n / x for small n is the
// high half of n * (2^32 / x + 1) (exact while n * x < 2^32; x = 1 has no such factor in 32 bits and is tested for).
struct ExpandAt {
    uint32_t inv_bpm, inv_cols;
    uint32_t bpm, cols, q0, mx0, my0;  // block S = block q0 of MCU (mx0, my0)
};
__device__ __forceinline__ uint32_t small_div(uint32_t n, uint32_t x, uint32_t inv) { return x == 1u ? n : __umulhi(n, inv); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// slots [s_first, s_first + n) of the wave's ring hold blocks S + d0 .. S + d0 + n - 1: written out and cleared, eight lanes per block
__device__ __forceinline__ void expand_store_blocks(JP_LDS ExpandLds &E, JP_LDS uint16_t *ring, const ExpandAt &at, uint32_t S, uint32_t d0, uint32_t n,
                                                    uint32_t total) {
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, grp = lane >> 3;
    for (uint32_t s0 = 0; s0 < n; s0 += 8u) {
        const uint32_t s = s0 + grp;
        if (s < n) {
            JP_LDS v2u *row = (JP_LDS v2u *)(ring + s * EXP_SLOT + sub * 8u);
            const v2u a = row[0], b = row[1];
            row[0] = v2u{0u, 0u};
            row[1] = v2u{0u, 0u};
            if (S + d0 + s < total) {
                const uint32_t t = at.q0 + d0 + s, dm = small_div(t, at.bpm, at.inv_bpm), q = (t - dm * at.bpm) & 15u;
                const uint32_t x = at.mx0 + dm, dy = small_div(x, at.cols, at.inv_cols), mx = x - dy * at.cols, my = at.my0 + dy;
                const uint64_t dst = E.q_dst[q].base + (uint64_t)my * E.q_dst[q].row_stride + (uint64_t)mx * E.q_dst[q].mcu_stride;
                ((JP_GLOBAL v4u *)(uintptr_t)dst)[sub] = v4u{a.x, a.y, b.x, b.y};
            }
        }
    }
}

// one entry of block S + d into image `slot` of the ring; returns |value * quantization value| (0 for lanes without an entry).
// UNIFORM: the entry does not say which component its block belongs to (the lane that wrote it could not know)
template <bool UNIFORM>
__device__ __forceinline__ uint32_t expand_put(JP_LDS ExpandLds &E, JP_LDS uint16_t *ring, const ExpandAt &at, bool valid, uint32_t ent, uint32_t d, uint32_t slot,
                                               uint32_t w0, uint32_t w1) {
    uint32_t c = (ent >> 22) & 3u;
    if (UNIFORM) {
        const uint32_t t = at.q0 + d;
        c = E.job.q_comp[(t - small_div(t, at.bpm, at.inv_bpm) * at.bpm) & 15u] & 3u;
    }
    const uint32_t z = E.unzig[(ent >> 16) & 63u];
    uint32_t v = ent & 0xffffu;
    if (!UNIFORM && huff_entry_is_dc(ent)) {  // the chunk's running sum + what the chunks before it add up to
        const uint32_t w = c < 2u ? w0 : w1;
        v = (v + ((c & 1u) ? w >> 16 : w)) & 0xffffu;
    }
    if (!valid || slot >= EXP_SLOTS) return 0u;  // (the second condition never holds for lists the sync passes wrote)
    ring[slot * EXP_SLOT + z] = (uint16_t)v;
    const int32_t sv = (int16_t)(uint16_t)v;
    return (uint32_t)(sv < 0 ? -sv : sv) * E.job.q[c][z];
}

// What the wave needs to know of a chunk before it can ask for its entries (scalar registers; huff_expand_kernel reads these for all of
// a wave's chunks at once, so that the entries of chunk i + 1 are on their way while chunk i is being assembled)
struct ExpandMeta {
    uint32_t cw, nblk, qk_before, w0, w1;  // emit_cnt[i], n_blocks[i], out_qk[i - 1] (0 for the first chunk), dc_sum[2i], dc_sum[2i + 1]
};
// the first EXP_LOADS x 64 entries of chunk i behind its leading ones (all there is of an ordinary chunk): requested, not waited for
__device__ __forceinline__ void expand_request(const JP_LDS HuffSyncJob &job, uint32_t i, uint32_t stride, const ExpandMeta &m, uint32_t (&ent)[EXP_LOADS]) {
    const uint32_t lane = threadIdx.x & 63u, cnt = min(m.cw & 0xffffu, stride), lead = min(m.cw >> 16, cnt);
    const JP_GLOBAL uint32_t *buf = (const JP_GLOBAL uint32_t *)(job.emit + (size_t)i * stride);
#pragma unroll
    for (uint32_t r = 0; r < EXP_LOADS; r++) {
        const uint32_t e = lead + 64u * r + lane;
        ent[r] = e < cnt ? stream_load(buf + e) : 0u;
    }
}

// One chunk of the scan -> its blocks in the arena.  Complete blocks wait in the ring until eight of them can go out together
// (eight lanes per block: a store instruction for fewer leaves lanes idle).  `first`: the chunk's first round of entries
// (expand_request); longer lists are fetched round by round here.
template <bool UNIFORM>
__device__ __forceinline__ void expand_chunk(JP_LDS ExpandLds &E, JP_LDS uint16_t *ring, ExpandAt &at, uint32_t i, const ExpandMeta &meta, uint32_t (&first)[EXP_LOADS],
                                             uint32_t n_chunks, uint32_t total, uint32_t stride, uint32_t &rg_dc, uint32_t &rg_ac) {
    const JP_LDS HuffSyncJob &job = E.job;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t cnt = min(meta.cw & 0xffffu, stride), lead = min(meta.cw >> 16, cnt);
    if (lead >= cnt) return;  // no block starts in this chunk
    const uint32_t S = meta.nblk + ((meta.qk_before & 0xffu) ? 1u : 0u);  // number of the first block that starts here
    uint32_t seg_end_chunk = n_chunks;  // blocks and lists end with the restart segment, if there are any
    if (job.n_seg > 1u) {
        const uint32_t seg = i / job.seg_chunks;
        uint32_t fb, nb;
        huff_segment_blocks(job, seg, fb, nb);
        total = rfl(fb + nb);
        seg_end_chunk = rfl((seg + 1u) * job.seg_chunks);
    }
    if (S >= total) return;  // (what a stream holds after its last block)
    {
        const uint32_t m = S / at.bpm;
        at.q0 = S - m * at.bpm;
        at.my0 = m / at.cols;
        at.mx0 = m - at.my0 * at.cols;
    }
    const uint32_t w0 = UNIFORM ? 0u : meta.w0, w1 = UNIFORM ? 0u : meta.w1;
    const JP_GLOBAL uint32_t *buf = (const JP_GLOBAL uint32_t *)(job.emit + (size_t)i * stride);
    uint32_t started = 0, base = 0;  // blocks started so far; which of them sits in slot 0
    for (uint32_t e0 = lead; e0 < cnt; e0 += 64u * EXP_LOADS) {
        uint32_t ent[EXP_LOADS];
#pragma unroll
        for (uint32_t r = 0; r < EXP_LOADS; r++) {
            if (e0 == lead) {
                ent[r] = first[r];
            } else {
                const uint32_t e = e0 + 64u * r + lane;
                ent[r] = e < cnt ? stream_load(buf + e) : 0u;
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < EXP_LOADS; r++) {
            if (e0 + 64u * r >= cnt) break;
            const bool valid = e0 + 64u * r + lane < cnt, flag = valid && huff_entry_is_dc(ent[r]);
            const uint64_t m = __ballot(flag);
            const uint32_t local = started + (uint32_t)__popcll(m & lt) + (flag ? 1u : 0u) - 1u;  // the entry's block, counted from S
            const uint32_t a = expand_put<UNIFORM>(E, ring, at, valid, ent[r], local, local - base, w0, w1);
            if (S + local < total) {
                if (flag) rg_dc = UNIFORM ? rg_dc : max(rg_dc, a);  // (uniform scans: huff_dc_prefix_kernel ranges the finished values)
                else rg_ac = max(rg_ac, a);
            }
            started += (uint32_t)__popcll(m);
            const uint32_t pending = started ? started - 1u - base : 0u;  // every block but the last one started is complete
            if (pending >= EXP_THR) {
                const uint32_t out = EXP_THR >= 8u ? pending & ~7u : pending, keep = pending - out + 1u;  // the rest, and the open block, move to the front
                __builtin_amdgcn_wave_barrier();
                expand_store_blocks(E, ring, at, S, base, out, total);
                if ((lane >> 3) < keep) {
                    JP_LDS v2u *from = (JP_LDS v2u *)(ring + (out + (lane >> 3)) * EXP_SLOT + (lane & 7u) * 8u), *to = (JP_LDS v2u *)(ring + lane * 8u + (lane >> 3) * (EXP_SLOT - 64u));
                    const v2u x = from[0], y = from[1];
                    from[0] = v2u{0u, 0u};
                    from[1] = v2u{0u, 0u};
                    to[0] = x;
                    to[1] = y;
                }
                __builtin_amdgcn_wave_barrier();
                base += out;
            }
        }
    }
    if (!started) return;
    // the last block: its remaining entries lead the lists of the chunks that follow
    const uint32_t last = started - 1u;  // (counted from S)
    for (uint32_t j = i + 1u; S + last < total && j < seg_end_chunk; j++) {
        const uint32_t cj = rfl(job.emit_cnt[j]), cntj = min(cj & 0xffffu, stride), leadj = min(cj >> 16, cntj);
        const JP_GLOBAL uint32_t *bj = (const JP_GLOBAL uint32_t *)(job.emit + (size_t)j * stride);
        for (uint32_t e = lane; e < leadj; e += 64u) rg_ac = max(rg_ac, expand_put<UNIFORM>(E, ring, at, true, stream_load(bj + e), last, last - base, 0u, 0u));
        if (leadj < cntj) break;  // a block starts in chunk j: ours ended there
    }
    __builtin_amdgcn_wave_barrier();
    expand_store_blocks(E, ring, at, S, base, started - base, total);
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(EXP_WAVES * 64) void huff_expand_kernel(const HuffSyncJob *__restrict__ jobs) {
    __shared__ ExpandLds E_;
    JP_LDS ExpandLds &E = *(JP_LDS ExpandLds *)&E_;
    const HuffSyncJob *gj = &jobs[blockIdx.y];
    const uint32_t first_chunk = blockIdx.x * (EXP_WAVES * EXP_CHUNKS);
    if (first_chunk >= gj->n_chunks || gj->emit == nullptr || gj->keep_lists || *gj->status != 0u) return;  // (flagged: the host decodes the image)
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)gj;
        JP_LDS uint32_t *dst = (JP_LDS uint32_t *)&E.job;
        for (uint32_t t = threadIdx.x; t < sizeof(HuffSyncJob) / 4u; t += EXP_WAVES * 64u) dst[t] = src[t];
    }
    const uint32_t lane = threadIdx.x & 63u, wave = rfl(threadIdx.x >> 6);
    JP_LDS uint16_t *ring = E.ring[wave];
    for (uint32_t t = lane; t < EXP_SLOTS * EXP_SLOT / 2u; t += 64u) ((JP_LDS uint32_t *)ring)[t] = 0u;
    __syncthreads();
    if (threadIdx.x < 16u) huff_fill_block_dst(E.job, E.q_dst, threadIdx.x);
    huff_fill_unzigzag((JP_LDS uint8_t *)E.unzig, threadIdx.x);
    __syncthreads();
    const JP_LDS HuffSyncJob &job = E.job;
    // (what steers the wave is the same in all its lanes: kept in scalar registers, branches instead of lane masks)
    const uint32_t bpm = rfl(job.bpm), cols = rfl(job.cols), n_chunks = rfl(job.n_chunks), uniform = rfl(job.uniform);
    const uint32_t total = rfl(job.n_mcu) * bpm, stride = rfl(job.emit_stride);
    ExpandAt at;
    at.bpm = bpm;
    at.cols = cols;
    at.inv_bpm = bpm > 1u ? 0xffffffffu / bpm + 1u : 0u;
    at.inv_cols = cols > 1u ? 0xffffffffu / cols + 1u : 0u;
    at.q0 = at.mx0 = at.my0 = 0u;
    uint32_t rg_dc = 0, rg_ac = 0;
    // lanes 0 .. EXP_CHUNKS - 1 fetch what the wave has to know of its chunks (one wait for all of them) ...
    const uint32_t i0 = first_chunk + wave * EXP_CHUNKS;
    uint32_t v_cw = 0, v_nblk = 0, v_qk = 0, v_w0 = 0, v_w1 = 0;
    {
        const uint32_t mine = i0 + (lane % EXP_CHUNKS);
        if (mine < n_chunks) {
            v_cw = job.emit_cnt[mine];
            v_nblk = job.n_blocks[mine];
            v_qk = huff_chunk_span(job, mine).first ? 0u : job.out_qk[mine - 1u];
            if (!uniform) {
                v_w0 = job.dc_sum[2u * mine];
                v_w1 = job.dc_sum[2u * mine + 1u];
            }
        }
    }
    auto meta_of = [&](uint32_t ci) {
        return ExpandMeta{(uint32_t)__builtin_amdgcn_readlane((int)v_cw, (int)ci), (uint32_t)__builtin_amdgcn_readlane((int)v_nblk, (int)ci),
                          (uint32_t)__builtin_amdgcn_readlane((int)v_qk, (int)ci), (uint32_t)__builtin_amdgcn_readlane((int)v_w0, (int)ci),
                          (uint32_t)__builtin_amdgcn_readlane((int)v_w1, (int)ci)};
    };
    // ... and the entries of chunk ci + 1 are requested before chunk ci is assembled
    uint32_t cur[EXP_LOADS], nxt[EXP_LOADS];
    if (i0 < n_chunks) expand_request(job, i0, stride, meta_of(0u), cur);
    for (uint32_t ci = 0; ci < EXP_CHUNKS; ci++) {
        const uint32_t i = i0 + ci;
        if (i >= n_chunks) break;
        const ExpandMeta m = meta_of(ci);
        const bool more = ci + 1u < EXP_CHUNKS && i + 1u < n_chunks;
        if (more) expand_request(job, i + 1u, stride, meta_of(ci + 1u), nxt);
        if (uniform) expand_chunk<true>(E, ring, at, i, m, cur, n_chunks, total, stride, rg_dc, rg_ac);
        else expand_chunk<false>(E, ring, at, i, m, cur, n_chunks, total, stride, rg_dc, rg_ac);
        if (more) {
#pragma unroll
            for (uint32_t r = 0; r < EXP_LOADS; r++) cur[r] = nxt[r];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        rg_dc = max(rg_dc, (uint32_t)__shfl_xor((int)rg_dc, off));
        rg_ac = max(rg_ac, (uint32_t)__shfl_xor((int)rg_ac, off));
    }
    if (lane == 0u) {
        E.wg_rg[0][wave] = rg_dc;
        E.wg_rg[1][wave] = rg_ac;
    }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t v = 0;
        for (uint32_t w = 0; w < EXP_WAVES; w++) v = max(v, E.wg_rg[threadIdx.x][w]);
        if (threadIdx.x == 0u) stat_mark_inexact(job.stats);
        stat_raise(job.stats + (threadIdx.x ? RS_MAX_AC : RS_MAX_DC), v);
    }
}

// The strip index of the entry-list pixel path (fused_entries.hpp): for every (MCU row k, strip s) of a job's 4:2:0 walk, where in the
// lists the run of MCUs [max(s * tx - 1, 0), ...) of row k begins — the chunk in which its first block STARTS (the last chunk whose
// first-block number is not above the block's: a 64-ary search over the chunks) and the place of that block's DC entry in the chunk's
// list (a count of DC entries).  One wave per pair.
__global__ __launch_bounds__(256) void huff_strip_index_kernel(const HuffSyncJob *__restrict__ jobs, const EntryIndexJob *__restrict__ ijobs) {
    const EntryIndexJob ij = ijobs[blockIdx.y];
    const HuffSyncJob &job = jobs[ij.job];
    const uint32_t lane = threadIdx.x & 63u, item = rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (item >= ij.rows * ij.tiles_x || *job.status != 0u) return;
    const uint32_t k = item / ij.tiles_x, s = item - k * ij.tiles_x, a = s ? s * ij.tx - 1u : 0u;
    const uint32_t B0 = job.bpm * (k * job.cols + a), n_chunks = job.n_chunks, stride = job.emit_stride;
    // (restart segments: a segment's first chunk continues nothing, huff_chunk_span(..).first; block numbers run on across segments)
    // ... and a segment's blocks end with its restart interval: what its last chunk made of the bits behind the last block (a state
    // that says "inside a block", handed on through the segment's empty slots) is nobody's — without the clamp the numbers of those
    // slots lie one above the next segment's first block, and the search below needs them in order
    const uint32_t seg_chunks = job.n_seg > 1u ? job.seg_chunks : 0u, seg_blocks = job.ri * job.bpm, all_blocks = job.n_mcu * job.bpm;
    auto first_block = [&](uint32_t c) {
        const bool continues = c && !(seg_chunks && c % seg_chunks == 0u) && (job.out_qk[c - 1u] & 0xffu);
        const uint32_t f = job.n_blocks[c] + (continues ? 1u : 0u);
        return seg_chunks ? min(f, min((c / seg_chunks + 1u) * seg_blocks, all_blocks)) : f;
    };
    uint32_t lo = 0, hi = n_chunks;  // the answer lies in [lo, hi); first_block(lo) <= B0 (chunk 0 starts block 0)
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) / 64u, c = lo + lane * step;
        const bool le = c < hi && first_block(c) <= B0;
        const uint32_t t = (uint32_t)__popcll(__ballot(le));  // (monotone: the lanes that say yes are the first t; lane 0 always does)
        const uint32_t nlo = lo + (t - 1u) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    const uint32_t c0 = lo, cw = job.emit_cnt[c0], cnt = min(cw & 0xffffu, stride);
    uint32_t want = B0 - first_block(c0), e0 = cnt;  // the (want + 1)-th DC entry of the list
    const uint32_t *buf = job.emit + (size_t)c0 * stride;
    constexpr uint32_t IR = 8;  // rounds of 64 entries requested at once (a wave that waits for every 256 bytes in turn: 59 us per 256 images)
    for (uint32_t eb = min(cw >> 16, cnt); eb < cnt && e0 == cnt; eb += 64u * IR) {
        uint32_t ent[IR];
#pragma unroll
        for (uint32_t r = 0; r < IR; r++) ent[r] = buf[min(eb + 64u * r + lane, cnt - 1u)];
#pragma unroll
        for (uint32_t r = 0; r < IR; r++) {
            if (eb + 64u * r >= cnt) break;
            const bool flag = eb + 64u * r + lane < cnt && huff_entry_is_dc(ent[r]);
            const uint64_t m = __ballot(flag);
            const uint32_t n = (uint32_t)__popcll(m);
            if (want < n) {
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint64_t hit = __ballot(flag && before == want);
                e0 = eb + 64u * r + (uint32_t)__builtin_ctzll(hit);
                break;
            }
            want -= n;
        }
    }
    if (lane == 0u) {
        if (e0 >= cnt) atomicOr(job.status, 1u | 1024u);  // (lists that do not hold the block their numbering promises: the host decodes the image)
        ij.tab[2u * item] = c0;
        ij.tab[2u * item + 1u] = e0;
    }
}

// DC differences -> DC values: a running sum (i16 wrapping, src/decoder.rs:1095-1099) per component over its blocks in
// the order the stream has them.  grid = (4, sync jobs), one workgroup per component plane walking it in tiles; the walk
// is a chain of load -> scan -> store round trips, so the tile is as large as a workgroup gets (1,024 lanes x 8 blocks:
// 4 tiles for the luma plane of a 1080p image instead of 32 with 256 x 4 — 0.77 ms -> see profiles/).
constexpr uint32_t DC_NT = 1024, DC_E = 8;

__global__ __launch_bounds__(DC_NT) void huff_dc_prefix_kernel(const HuffSyncJob *__restrict__ jobs) {
    __shared__ uint32_t wave_tot[DC_NT / 64u];
    __shared__ uint32_t s_all[DC_NT * DC_E];  // restart intervals: the tile's running sums, for the value just before a segment's first block
    const HuffSyncJob &job = jobs[blockIdx.y];
    const uint32_t c = blockIdx.x;
    if (c >= job.ncomp || *job.status != 0u || !job.uniform) return;  // (other scans: the write pass stored DC values)
    const HuffScanComp sc = job.comp[c];
    const uint32_t hv = sc.h * sc.v, n = job.n_mcu * hv, cols = job.cols;
    const uint32_t q0 = job.q[c][0];
    // With restart markers the predictor starts again at every segment (src/decoder.rs:928-931): element x of the component's
    // blocks in stream order belongs to the segment that starts at element (x / P) * P, P = blocks of this component per
    // restart interval, and its value is the running sum minus the running sum just before that element.
    const uint32_t P = job.n_seg > 1u ? job.ri * hv : 0u;
    uint32_t carry = 0, carry_b = 0, max_dc = 0;  // carry_b: the running sum just before the segment that holds the tile's first element
    for (uint32_t base = 0; base < n; base += DC_NT * DC_E) {
        const uint32_t s0 = base + threadIdx.x * DC_E;
        JP_GLOBAL int16_t *addr[DC_E];
        uint32_t d[DC_E];
        {
            // the first element by division, the rest by stepping through the MCU
            uint32_t m = s0 / hv, sub = s0 - m * hv, my = m / cols, mx = m - my * cols;
#pragma unroll
            for (uint32_t e = 0; e < DC_E; e++) {
                d[e] = 0;
                addr[e] = nullptr;
                if (s0 + e < n) {
                    const uint32_t vp = sub / sc.h, hp = sub - vp * sc.h;
                    addr[e] = (JP_GLOBAL int16_t *)(sc.dst + ((size_t)(my * sc.v + vp) * sc.block_w + (mx * sc.h + hp)) * 64u);
                    d[e] = (uint16_t)*addr[e];
                }
                if (++sub == hv) {
                    sub = 0;
                    if (++mx == cols) {
                        mx = 0;
                        my++;
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t e = 1; e < DC_E; e++) d[e] += d[e - 1u];
        uint32_t incl = d[DC_E - 1u];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
            if ((threadIdx.x & 63u) >= (uint32_t)off) incl += o;
        }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < DC_NT / 64u; w++) {
            const uint32_t t = wave_tot[w];
            before += w < (threadIdx.x >> 6) ? t : 0u;
            total += t;
        }
        const uint32_t off = carry + before + incl - d[DC_E - 1u];
        if (P) {
#pragma unroll
            for (uint32_t e = 0; e < DC_E; e++) s_all[threadIdx.x * DC_E + e] = d[e] + off;
            __syncthreads();
        }
#pragma unroll
        for (uint32_t e = 0; e < DC_E; e++)
            if (addr[e]) {
                uint32_t sum = d[e] + off;
                if (P) {
                    const uint32_t first = ((s0 + e) / P) * P;  // the segment's first element
                    sum -= first == 0u ? 0u : (first > base ? s_all[first - 1u - base] : (first == base ? carry : carry_b));
                }
                const int32_t v = (int16_t)(uint16_t)sum;
                *addr[e] = (int16_t)v;
                max_dc = max(max_dc, (uint32_t)(v < 0 ? -v : v) * q0);
            }
        if (P) {  // for the next tile: the running sum just before the segment its first element lies in
            const uint32_t next = base + DC_NT * DC_E, first = (next / P) * P;
            if (first > base && first < next) carry_b = s_all[first - 1u - base];
            else if (first == base) carry_b = carry;
        }
        carry += total;
        __syncthreads();
    }
    publish_range(job.stats, HuffRange{max_dc, 0u});
}

// The status words of a launch -> pinned host memory, by a kernel (one workgroup) instead of a device-to-host copy: a copy command queues up
// behind whatever the copy engine of that direction has in flight — with JPGPU_PIPELINE_DOWNLOAD that is hundreds of megabytes of
// pixels per sub-batch, and the host waited 300 ms for 512 bytes of status words before it could finish a sub-batch (round 5).
__global__ __launch_bounds__(256) void copy_words_kernel(uint32_t *__restrict__ dst_host, const uint32_t *__restrict__ src, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += 256u) dst_host[i] = src[i];
    __threadfence_system();
}
hipError_t launch_copy_words_to_host(uint32_t *dst_host_mapped, const uint32_t *d_src, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    copy_words_kernel<<<dim3(1), dim3(256), 0, stream>>>(dst_host_mapped, d_src, n);
    return hipGetLastError();
}

// A sub-batch's pixels -> pinned host memory (JPGPU_PIPELINE_DOWNLOAD), by a kernel of a few workgroups that writes the mapped host
// block itself (16-byte non-temporal stores): tools/attic/probe_d2h2.hip measures 55 GB/s for it — what the copy engine gives the same
// copies ALONE (57) — where hipMemcpyAsync inside jpgpu_pipeline_decode reached 33 with a host core busy the whole time.
__global__ __launch_bounds__(256) void copy_to_host_kernel(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16, uint8_t *__restrict__ dst_tail,
                                                           const uint8_t *__restrict__ src_tail, uint32_t tail) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) __builtin_nontemporal_store(src[i], dst + i);
    if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
// ... and the other way: staged entropy-coded bytes out of pinned host memory (A/B partner of the copy engine for the uploads of the
// device-entropy route; JPGPU_UPLOAD_BY_KERNEL).  Reads over the link are not posted: 128 workgroups keep enough of them in flight.
__global__ __launch_bounds__(256) void copy_from_host_kernel(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = __builtin_nontemporal_load(src + i);
}
hipError_t launch_copy_from_host(void *d_dst, const void *src_host_mapped, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const size_t n16 = (bytes + 15u) / 16u;  // (both blocks are 16-byte aligned and padded by their owner)
    const uint32_t wgs = (uint32_t)std::min<size_t>(128u, (n16 + 255u) / 256u);
    copy_from_host_kernel<<<dim3(wgs), dim3(256), 0, stream>>>((v4u *)d_dst, (const v4u *)src_host_mapped, n16);
    return hipGetLastError();
}
hipError_t launch_copy_to_host(void *dst_host_mapped, const void *d_src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const size_t n16 = bytes / 16u;
    copy_to_host_kernel<<<dim3(64), dim3(256), 0, stream>>>((v4u *)dst_host_mapped, (const v4u *)d_src, n16, (uint8_t *)dst_host_mapped + n16 * 16u,
                                                          (const uint8_t *)d_src + n16 * 16u, (uint32_t)(bytes - n16 * 16u));
    return hipGetLastError();
}

// ---- progressive frames, round 6: one WAVE per scan (huff_prog_wave.hpp) -----------------------------------------------------------------
// grid = one workgroup of one wave per table entry; no LDS.  Entry i belongs to XCD i mod 8 (workgroups are handed to the XCDs
// round-robin in launch order): the host puts all scans of a frame on one XCD, producers in front — whatever a wave waits for was
// dispatched before it by the same XCD's dispatcher, so it is resident or done however oversubscribed the launch is.
__global__ __launch_bounds__(64) void huff_progw_kernel(const ProgTrack *__restrict__ tracks, uint32_t n_tracks) {
    const uint32_t t = blockIdx.x;
    if (t >= n_tracks) return;
    const ProgTrack tr = tracks[t];
    if (tr.scans == nullptr || tr.n_scans == 0u) return;  // (padding of the shorter XCD lists)
    progw_run_track(tr);
}
#if !defined(PROGW_PORTABLE)
// (test hook, not part of the C ABI's headers: the hand-scheduled refinement loop on given states — tests/test_gpu_progw_asm.py)
__global__ __launch_bounds__(64) void progw_refine_fast_case_kernel(PwFastCase *cases, uint32_t n) {
    if (blockIdx.x >= n) return;
    PwFastCase *g = cases + blockIdx.x;
    PwFastCase c;
    c.win = g->win, c.nz = g->nz, c.neg = g->neg, c.new_nz = g->new_nz, c.new_neg = g->new_neg;
    c.pos = g->pos, c.nx = g->nx, c.dp = g->dp, c.k = g->k, c.end = g->end, c.al = g->al, c.eob = g->eob, c.code = 0u;
    // (per-lane arrays: every lane reads and writes its own element only)
    const uint32_t lane = threadIdx.x;
    c.lut6[lane] = g->lut6[lane], c.w[lane] = g->w[lane], c.acc[lane] = g->acc[lane];
    c.table = g->lut8;
    pw_refine_fast_case(c);
    g->acc[lane] = c.acc[lane];
    if (lane == 0u) {
        g->win = c.win, g->pos = c.pos, g->nx = c.nx, g->dp = c.dp, g->k = c.k, g->eob = c.eob, g->new_nz = c.new_nz, g->new_neg = c.new_neg, g->code = c.code;
    }
}
__global__ __launch_bounds__(64) void progw_first_fast_case_kernel(PwFirstCase *cases, uint32_t n) {
    if (blockIdx.x >= n) return;
    PwFirstCase *g = cases + blockIdx.x;
    PwFirstCase c;
    c.win = g->win, c.nz = g->nz, c.neg = g->neg, c.pos = g->pos, c.nx = g->nx, c.dp = g->dp, c.k = g->k, c.se = g->se, c.al = g->al, c.eob = g->eob, c.code = 0u;
    const uint32_t lane = threadIdx.x;
    c.lut6[lane] = g->lut6[lane], c.w[lane] = g->w[lane], c.cf[lane] = g->cf[lane];
    c.table = g->lut8;
    pw_first_fast_case(c);
    g->cf[lane] = c.cf[lane];
    if (lane == 0u) g->win = c.win, g->pos = c.pos, g->nx = c.nx, g->dp = c.dp, g->k = c.k, g->eob = c.eob, g->nz = c.nz, g->neg = c.neg, g->code = c.code;
}
extern "C" int jpgpu_selftest_first_fast(void *host_cases, uint32_t n) {
    PwFirstCase *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * sizeof(PwFirstCase)) != hipSuccess) return 1;
    int rc = 0;
    if (hipMemcpy(d, host_cases, (size_t)n * sizeof(PwFirstCase), hipMemcpyHostToDevice) != hipSuccess) rc = 2;
    if (!rc) {
        progw_first_fast_case_kernel<<<dim3(n), dim3(64)>>>(d, n);
        if (hipDeviceSynchronize() != hipSuccess) rc = 3;
    }
    if (!rc && hipMemcpy(host_cases, d, (size_t)n * sizeof(PwFirstCase), hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    (void)hipFree(d);
    return rc;
}
__global__ __launch_bounds__(64) void progw_dc_fast_case_kernel(PwDcCase *cases, uint32_t n) {
    if (blockIdx.x >= n) return;
    PwDcCase *g = cases + blockIdx.x;
    PwDcCase c;
    c.win = g->win, c.pred = g->pred, c.cm0 = g->cm0, c.cm1 = g->cm1, c.ts0 = g->ts0, c.ts1 = g->ts1;
    c.pos = g->pos, c.nx = g->nx, c.dp = g->dp, c.i = g->i, c.n = g->n, c.al = g->al, c.code = 0u;
    const uint32_t lane = threadIdx.x;
    c.lut0[lane] = g->lut0[lane], c.lut1[lane] = g->lut1[lane], c.w[lane] = g->w[lane], c.val[lane] = g->val[lane];
    pw_dc_fast_case(c);
    g->val[lane] = c.val[lane];
    if (lane == 0u) g->win = c.win, g->pos = c.pos, g->nx = c.nx, g->dp = c.dp, g->i = c.i, g->pred = c.pred, g->code = c.code;
}
extern "C" int jpgpu_selftest_dc_fast(void *host_cases, uint32_t n) {
    PwDcCase *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * sizeof(PwDcCase)) != hipSuccess) return 1;
    int rc = 0;
    if (hipMemcpy(d, host_cases, (size_t)n * sizeof(PwDcCase), hipMemcpyHostToDevice) != hipSuccess) rc = 2;
    if (!rc) {
        progw_dc_fast_case_kernel<<<dim3(n), dim3(64)>>>(d, n);
        if (hipDeviceSynchronize() != hipSuccess) rc = 3;
    }
    if (!rc && hipMemcpy(host_cases, d, (size_t)n * sizeof(PwDcCase), hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    (void)hipFree(d);
    return rc;
}
// (timing: the same state walked `reps` times by one wave; -> milliseconds for the lot)
__global__ __launch_bounds__(64) void progw_refine_fast_bench_kernel(PwFastCase *cases, uint32_t reps) {
    PwFastCase *g = cases;
    const uint32_t lane = threadIdx.x;
    auto u32 = [](uint32_t x) { return wv_uniform(x); };
    auto u64 = [](uint64_t x) { return ((uint64_t)wv_uniform((uint32_t)(x >> 32)) << 32) | wv_uniform((uint32_t)x); };
    PwFastCase c0;  // (every lane holds the same: into scalar registers once)
    c0.win = u64(g->win), c0.nz = u64(g->nz), c0.neg = u64(g->neg), c0.new_nz = u64(g->new_nz), c0.new_neg = u64(g->new_neg);
    c0.pos = u32(g->pos), c0.nx = u32(g->nx), c0.dp = u32(g->dp), c0.k = u32(g->k), c0.end = u32(g->end), c0.al = u32(g->al), c0.eob = u32(g->eob);
    const uint32_t lut = g->lut6[lane], w = g->w[lane], acc = g->acc[lane];
    uint32_t sink = 0;
    for (uint32_t r = 0; r < reps; r++) {
        PwFastCase c;
        c.win = c0.win, c.nz = c0.nz, c.neg = c0.neg, c.new_nz = c0.new_nz, c.new_neg = c0.new_neg;
        c.pos = c0.pos, c.nx = c0.nx, c.dp = c0.dp, c.k = c0.k, c.end = c0.end, c.al = c0.al, c.eob = c0.eob, c.code = 0u;
        c.lut6[lane] = lut, c.w[lane] = w, c.acc[lane] = acc;
        c.table = g->lut8;
        pw_refine_fast_case(c);
        sink += c.acc[lane];
    }
    g->acc[lane] = sink;
}
extern "C" float jpgpu_selftest_refine_fast_ms(void *host_case, uint32_t reps, uint32_t waves) {
    PwFastCase *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(PwFastCase)) != hipSuccess) return -1.f;
    float ms = -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    if (hipMemcpy(d, host_case, sizeof(PwFastCase), hipMemcpyHostToDevice) == hipSuccess) {
        progw_refine_fast_bench_kernel<<<dim3(waves), dim3(64)>>>(d, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        progw_refine_fast_bench_kernel<<<dim3(waves), dim3(64)>>>(d, reps);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0), hipEventDestroy(e1);
    (void)hipFree(d);
    return ms;
}
extern "C" int jpgpu_selftest_refine_fast(void *host_cases, uint32_t n) {
    PwFastCase *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * sizeof(PwFastCase)) != hipSuccess) return 1;
    int rc = 0;
    if (hipMemcpy(d, host_cases, (size_t)n * sizeof(PwFastCase), hipMemcpyHostToDevice) != hipSuccess) rc = 2;
    if (!rc) {
        progw_refine_fast_case_kernel<<<dim3(n), dim3(64)>>>(d, n);
        if (hipDeviceSynchronize() != hipSuccess) rc = 3;
    }
    if (!rc && hipMemcpy(host_cases, d, (size_t)n * sizeof(PwFastCase), hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    (void)hipFree(d);
    return rc;
}
#endif

hipError_t launch_huff_progw(const ProgTrack *d_tracks, uint32_t n_tracks, hipStream_t stream) {
    if (n_tracks == 0) return hipSuccess;
    huff_progw_kernel<<<dim3(n_tracks), dim3(64), 0, stream>>>(d_tracks, n_tracks);
    return hipGetLastError();
}

hipError_t launch_range_scan_one(const int16_t *d_coefs, uint32_t n_blocks, const uint16_t *d_q, uint32_t *d_stats, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    range_scan_one_kernel<<<dim3((n_blocks + 255u) / 256u), dim3(256), 0, stream>>>(d_coefs, n_blocks, d_q, d_stats);
    return hipGetLastError();
}

hipError_t launch_range_scan(const RangeJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t *d_stats, hipStream_t stream) {
    if (n_jobs == 0 || max_blocks == 0) return hipSuccess;
    range_scan_kernel<<<dim3((max_blocks + 255u) / 256u, n_jobs), dim3(256), 0, stream>>>(d_jobs, d_stats);
    return hipGetLastError();
}

// Everything for a sub-batch's scans, enqueued blind: a fixed number of sync launches (settled jobs cost an empty workgroup
// each), block numbering, the expansion of the entry lists and the DC sums of `uniform` scans.
hipError_t launch_huff_sync(const HuffSyncJob *d_jobs, uint32_t n_jobs, uint32_t max_chunks, uint32_t launches, uint32_t iters, hipStream_t stream,
                            hipEvent_t after_sync, bool low_table_ids, const EntryIndexJob *d_index, uint32_t n_index, uint32_t max_index_items) {
    if (n_jobs == 0 || max_chunks == 0 || launches == 0 || iters == 0) {
        if (after_sync) (void)hipEventRecord(after_sync, stream);
        return hipSuccess;
    }
    huff_weave_kernel<<<dim3((max_chunks + HUFF_WEAVE_LANES - 1u) / HUFF_WEAVE_LANES, n_jobs), dim3(256), 0, stream>>>(d_jobs);
    const dim3 grid((max_chunks + SYNC_NT - 1u) / SYNC_NT, n_jobs);
    const dim3 late_grid((max_chunks + SYNC_LATE_SPAN * SYNC_NT - 1u) / (SYNC_LATE_SPAN * SYNC_NT), n_jobs);
    for (uint32_t l = 0; l < launches; l++) {
        const bool late = l * iters >= 2u;  // (passes 0 and 1 are every lane's)
        if (late && low_table_ids) huff_sync_late_kernel<4u><<<late_grid, dim3(SYNC_NT), 0, stream>>>(d_jobs, l, l * iters, iters);
        else if (late) huff_sync_late_kernel<8u><<<late_grid, dim3(SYNC_NT), 0, stream>>>(d_jobs, l, l * iters, iters);
        else if (low_table_ids) huff_sync_pass_kernel<4u><<<grid, dim3(SYNC_NT), 0, stream>>>(d_jobs, l, l * iters, iters);
        else huff_sync_pass_kernel<8u><<<grid, dim3(SYNC_NT), 0, stream>>>(d_jobs, l, l * iters, iters);
    }
    huff_sync_scan_kernel<<<dim3(n_jobs), dim3(SYNC_NT), 0, stream>>>(d_jobs, launches - 1u);
    if (after_sync) (void)hipEventRecord(after_sync, stream);
    if (n_index < n_jobs) {  // (every job's lists kept as lists: nothing to expand, and such jobs are never `uniform`)
        huff_expand_kernel<<<dim3((max_chunks + EXP_WAVES * EXP_CHUNKS - 1u) / (EXP_WAVES * EXP_CHUNKS), n_jobs), dim3(EXP_WAVES * 64u), 0, stream>>>(d_jobs);
        huff_dc_prefix_kernel<<<dim3(4, n_jobs), dim3(DC_NT), 0, stream>>>(d_jobs);
    }
    if (d_index && n_index && max_index_items) huff_strip_index_kernel<<<dim3((max_index_items + 3u) / 4u, n_index), dim3(256), 0, stream>>>(d_jobs, d_index);
    return hipGetLastError();
}

}  // namespace jpgpu
