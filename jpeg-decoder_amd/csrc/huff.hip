// huff.hip — device entropy decoding of restart-marker streams (huff_core.hpp) and the range scan that classifies the
// coefficients it produced (the host never sees them).
#include <hip/hip_runtime.h>

#include "huff.hpp"
#include "huff_core.hpp"
#include "huff_sync_core.hpp"

namespace jpgpu {

// grid = (ceil(max segments / 64), scan jobs), one wave per workgroup: lanes diverge (every lane walks its own bit
// stream), so small workgroups spread the segments over as many SIMDs as possible
// The scan's job record and Huffman tables are copied to LDS first: every symbol costs dependent table reads.
__global__ __launch_bounds__(64) void huff_segments_kernel(const HuffScanJob *__restrict__ jobs) {
    __shared__ HuffLds L;
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)&jobs[blockIdx.y];
        uint32_t *dst = reinterpret_cast<uint32_t *>(&L.job);
        if (threadIdx.x < sizeof(HuffScanJob) / 4u) dst[threadIdx.x] = src[threadIdx.x];
    }
    huff_fill_unzigzag((JP_LDS uint8_t *)L.unzig, threadIdx.x);
    __syncthreads();
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)L.job.tables;
        uint32_t *dst = reinterpret_cast<uint32_t *>(L.tables);
        for (uint32_t i = threadIdx.x; i < 8u * sizeof(DevHuffTable) / 4u; i += 64u) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t seg = blockIdx.x * 64u + threadIdx.x;
    if (seg >= L.job.n_seg) return;
    huff_decode_segment(*(JP_LDS HuffLds *)&L, seg);
}

// one lane per block: max |c*q| and the largest block-column sum of |c*q| (the two quantities behind the range classes of
// include/jpgpu.h), reduced per wave and merged with atomicMax
__global__ __launch_bounds__(256) void range_scan_kernel(const RangeJob *__restrict__ jobs, uint32_t *__restrict__ stats) {
    const RangeJob &job = jobs[blockIdx.y];
    const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
    uint32_t max_abs = 0, max_col = 0;
    if (blk < job.n_blocks) {
        const JP_GLOBAL v4u *p = (const JP_GLOBAL v4u *)(job.coefs + (size_t)blk * 64u);
        uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const v4u v = p[r];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const int32_t c = (int16_t)(uint16_t)(w[k >> 1] >> (16u * (k & 1u)));
                const uint32_t a = (uint32_t)(c < 0 ? -c : c) * (uint32_t)job.q[r * 8u + k];  // <= 32768 * 65535 < 2^31
                max_abs = a > max_abs ? a : max_abs;
                col[k] += a < 0x00ffffffu ? a : 0x00ffffffu;  // saturate the addends: 8 of them cannot wrap
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) max_col = col[k] > max_col ? col[k] : max_col;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_abs = max(max_abs, (uint32_t)__shfl_xor((int)max_abs, off));
        max_col = max(max_col, (uint32_t)__shfl_xor((int)max_col, off));
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicMax(&stats[2u * job.slot], max_abs);
        atomicMax(&stats[2u * job.slot + 1u], max_col);
    }
}


// ---- scans without restart markers: the self-synchronising chunk decoder (huff_sync_core.hpp) -------------------------
// grid = (ceil(max chunks / 256), sync jobs); lane = chunk.  `changed` of a job holds three counters used in rotation by
// consecutive launches: launch t counts into slot t % 3, reads slot (t-1) % 3 (zero: the job has settled, nothing to do)
// and clears slot (t+1) % 3, so the host can enqueue a fixed number of launches without looking at the device in between.
constexpr uint32_t SYNC_NT = 256;

__device__ __forceinline__ void sync_load_lds(JP_LDS HuffSyncLds &L, const HuffSyncJob *gj) {
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)gj;
        JP_LDS uint32_t *dst = (JP_LDS uint32_t *)&L.job;
        static_assert(sizeof(HuffSyncJob) / 4u <= SYNC_NT, "one word per lane");
        if (threadIdx.x < sizeof(HuffSyncJob) / 4u) dst[threadIdx.x] = src[threadIdx.x];
    }
    huff_fill_unzigzag((JP_LDS uint8_t *)L.unzig, threadIdx.x & 63u);
    {
        const JP_GLOBAL uint32_t *src = (const JP_GLOBAL uint32_t *)gj->tables;
        JP_LDS uint32_t *dst = (JP_LDS uint32_t *)L.tables;
        for (uint32_t i = threadIdx.x; i < 8u * sizeof(DevHuffTable) / 4u; i += SYNC_NT) dst[i] = src[i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(SYNC_NT) void huff_sync_pass_kernel(const HuffSyncJob *__restrict__ jobs, uint32_t launch, uint32_t first_pass,
                                                                uint32_t iters) {
    __shared__ HuffSyncLds L;
    const HuffSyncJob *gj = &jobs[blockIdx.y];
    uint32_t *cnt = gj->changed;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[(launch + 1u) % 3u] = 0u;
    const uint32_t n_chunks = gj->n_chunks;
    if (blockIdx.x * SYNC_NT >= n_chunks) return;
    if (launch > 0u && cnt[(launch - 1u) % 3u] == 0u) return;
    const uint32_t i = blockIdx.x * SYNC_NT + threadIdx.x;
    if (launch > 0u) {  // does any lane of this workgroup have a new start state?  (most do not: skip the table load)
        bool need = false;
        if (i > 0u && i < n_chunks) {
            const uint32_t p = huff_load_shared(gj->out_pos + (i - 1u));
            uint32_t qk = huff_load_shared(gj->out_qk + (i - 1u));
            if (gj->uniform) qk &= 0xffu;
            const uint32_t first = i << gj->chunk_shift;  // (huff_sync_state_plausible)
            need = p >= first && p - first <= 32u && (qk >> 8) < gj->bpm && (qk & 0xffu) < 64u && (p != gj->in_pos[i] || qk != gj->in_qk[i]);
        }
        if (!__syncthreads_or(need)) return;
    }
    sync_load_lds(*(JP_LDS HuffSyncLds *)&L, gj);
    bool published = false;
    for (uint32_t it = 0; it < iters; it++) {
        if (i < n_chunks) published |= huff_sync_chunk<false>(*(JP_LDS HuffSyncLds *)&L, i, first_pass + it);
        __syncthreads();
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
    uint32_t carry = 0;
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
        carry += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SYNC_NT) void huff_sync_write_kernel(const HuffSyncJob *__restrict__ jobs) {
    __shared__ HuffSyncLds L;
    const HuffSyncJob *gj = &jobs[blockIdx.y];
    if (blockIdx.x * SYNC_NT >= gj->n_chunks) return;
    if (*gj->status != 0u) return;
    sync_load_lds(*(JP_LDS HuffSyncLds *)&L, gj);
    const uint32_t i = blockIdx.x * SYNC_NT + threadIdx.x;
    if (i < L.job.n_chunks) huff_sync_chunk<true>(*(JP_LDS HuffSyncLds *)&L, i, 0u);
}

// DC differences -> DC values: a running sum (i16 wrapping, src/decoder.rs:1095-1099) per component over its blocks in
// the order the stream has them.  grid = (4, sync jobs), four consecutive blocks per lane and tile.
__global__ __launch_bounds__(SYNC_NT) void huff_dc_prefix_kernel(const HuffSyncJob *__restrict__ jobs) {
    __shared__ uint32_t wave_tot[SYNC_NT / 64u];
    const HuffSyncJob &job = jobs[blockIdx.y];
    const uint32_t c = blockIdx.x;
    if (c >= job.ncomp || *job.status != 0u) return;
    const HuffScanComp sc = job.comp[c];
    const uint32_t hv = sc.h * sc.v, n = job.n_mcu * hv, cols = job.cols;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += SYNC_NT * 4u) {
        const uint32_t s0 = base + threadIdx.x * 4u;
        JP_GLOBAL int16_t *addr[4];
        uint32_t d[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++) {
            const uint32_t s = s0 + e;
            d[e] = 0;
            addr[e] = nullptr;
            if (s < n) {
                const uint32_t m = s / hv, sub = s - m * hv, my = m / cols, mx = m - my * cols, vp = sub / sc.h, hp = sub - vp * sc.h;
                addr[e] = (JP_GLOBAL int16_t *)(sc.dst + ((size_t)(my * sc.v + vp) * sc.block_w + (mx * sc.h + hp)) * 64u);
                d[e] = (uint16_t)*addr[e];
            }
        }
        d[1] += d[0];
        d[2] += d[1];
        d[3] += d[2];
        uint32_t incl = d[3];
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
        const uint32_t off = carry + before + incl - d[3];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            if (addr[e]) *addr[e] = (int16_t)(uint16_t)(d[e] + off);
        carry += total;
        __syncthreads();
    }
}

hipError_t launch_huff_segments(const HuffScanJob *d_jobs, uint32_t n_jobs, uint32_t max_segments, hipStream_t stream) {
    if (n_jobs == 0 || max_segments == 0) return hipSuccess;
    huff_segments_kernel<<<dim3((max_segments + 63u) / 64u, n_jobs), dim3(64), 0, stream>>>(d_jobs);
    return hipGetLastError();
}

hipError_t launch_range_scan(const RangeJob *d_jobs, uint32_t n_jobs, uint32_t max_blocks, uint32_t *d_stats, hipStream_t stream) {
    if (n_jobs == 0 || max_blocks == 0) return hipSuccess;
    range_scan_kernel<<<dim3((max_blocks + 255u) / 256u, n_jobs), dim3(256), 0, stream>>>(d_jobs, d_stats);
    return hipGetLastError();
}

// Everything for the jobs without restart markers, enqueued blind: a fixed number of sync launches (settled jobs cost an
// empty workgroup each), block numbering, the write pass and the DC sums.
hipError_t launch_huff_sync(const HuffSyncJob *d_jobs, uint32_t n_jobs, uint32_t max_chunks, uint32_t launches, uint32_t iters, hipStream_t stream) {
    if (n_jobs == 0 || max_chunks == 0 || launches == 0 || iters == 0) return hipSuccess;
    const dim3 grid((max_chunks + SYNC_NT - 1u) / SYNC_NT, n_jobs);
    for (uint32_t l = 0; l < launches; l++) huff_sync_pass_kernel<<<grid, dim3(SYNC_NT), 0, stream>>>(d_jobs, l, l * iters, iters);
    huff_sync_scan_kernel<<<dim3(n_jobs), dim3(SYNC_NT), 0, stream>>>(d_jobs, launches - 1u);
    huff_sync_write_kernel<<<grid, dim3(SYNC_NT), 0, stream>>>(d_jobs);
    huff_dc_prefix_kernel<<<dim3(4, n_jobs), dim3(SYNC_NT), 0, stream>>>(d_jobs);
    return hipGetLastError();
}

}  // namespace jpgpu
