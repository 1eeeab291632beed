// pipeline.cpp — jpgpu_pipeline_* (include/jpgpu_decoder.h): many JPEG streams -> pixels.
// Host entropy decoding (csrc/host/frontend.cpp, the restatement of src/decoder.rs:794-1298) on a thread pool,
// one image per task, rows staged in pinned memory, per-image async H2D, then the batch kernels (batch.cpp).
#include <hip/hip_runtime.h>
#include <sched.h>
#include <time.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "compact.hpp"
#include "host/frontend.hpp"
#include "host_common.hpp"

using jpgpu::host::DecodeError;
using jpgpu::host::Frontend;
using jpgpu::host::RowSink;

namespace {

constexpr uint32_t kCopyStreams = 4, kD2HStreams = 2;

// Default pool size: one thread per physical core (SMT siblings plus the uploader thread made 256 threads slower and
// erratic on the 2 x 64-core host), but no more than twice the CPUs a cgroup quota grants the process: with
// cpu.max = 16 CPUs, 32 threads sustained 3.8 k 1080p images/s, 128 threads 2.9 k (throttled mid-image).
uint32_t default_threads() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency() / 2u);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long period = 0;
        if (fscanf(f, "%31s %lu", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const unsigned long cpus = (strtoul(quota, nullptr, 10) + period - 1) / period;
            if (cpus > 0) n = std::min<uint32_t>(n, (uint32_t)std::max(2ul, 2ul * cpus));
        }
        fclose(f);
    }
    return n;
}

// CPUs the process may really use: its affinity mask, capped by a cgroup quota (the progressive dispatcher's host side)
uint32_t granted_cpus() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<uint32_t>(n, (uint32_t)std::max(1, CPU_COUNT(&set)));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        unsigned long period = 0;
        if (fscanf(f, "%31s %lu", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const unsigned long cpus = (strtoul(quota, nullptr, 10) + period - 1) / period;
            if (cpus > 0) n = std::min<uint32_t>(n, (uint32_t)cpus);
        }
        fclose(f);
    }
    return n;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// CPU time the whole process has used so far, all threads (jpgpu_pipeline_timings::cpu_ms: what a call costs the host)
double process_cpu_ms() {
    struct timespec ts;
    if (clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) != 0) return 0.0;
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

// Minimal fork-join pool: run(n, fn) calls fn(i) for i in [0, n) on the workers and returns when all are done.
class Pool {
public:
    explicit Pool(uint32_t n) {
        for (uint32_t t = 0; t < n; t++) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto &w : workers_) w.join();
    }
    uint32_t size() const { return (uint32_t)workers_.size(); }
    void run(uint32_t n, const std::function<void(uint32_t)> &fn) {
        if (n == 0) return;
        std::lock_guard<std::mutex> one_at_a_time(run_m_);
        std::unique_lock<std::mutex> g(m_);
        fn_ = &fn;
        n_ = n;
        pending_ = n;
        generation_++;
        // the claim counter carries the generation it belongs to: a worker that read (fn, n) of an earlier run and only
        // now gets to claim an index finds another generation in the counter and backs off — it can neither take an item of
        // this run with the old n nor call the old, by now destroyed, function (use-after-free + a lost item)
        next_.store((uint64_t)(uint32_t)generation_ << 32);
        cv_work_.notify_all();
        cv_done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(uint32_t)> *fn;
            uint32_t n;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_work_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                fn = fn_;
                n = n_;
            }
            if (!fn) continue;  // that run is over already
            uint32_t done = 0;
            uint64_t cur = next_.load();
            for (;;) {
                if ((uint32_t)(cur >> 32) != (uint32_t)seen) break;  // a later run owns the counter
                const uint32_t i = (uint32_t)cur;
                if (i >= n) break;
                if (!next_.compare_exchange_weak(cur, cur + 1)) continue;  // (cur reloaded)
                (*fn)(i);
                done++;
                cur = next_.load();
            }
            if (done) {
                std::lock_guard<std::mutex> g(m_);
                pending_ -= done;
                if (pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_m_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(uint32_t)> *fn_ = nullptr;
    uint32_t n_ = 0, pending_ = 0;
    std::atomic<uint64_t> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// Worker of one image: rows go straight to their final place in the pinned staging memory (the frame slot of a worker
// index is known from frame_slot_hint), so a coefficient is written once on the host — either as it is (dense mode:
// the staging memory mirrors the coefficient arena) or in the compact transport form (compact.hpp), which is what
// the pipeline sends by default.
class StageSink : public RowSink {
public:
    // dense: off/len = the component's place in the arena image of `stage`; compact: off = its compact region
    // (compact_max_bytes(len / 128)) in `stage`
    StageSink(uint8_t *stage, const size_t (&off)[4], const size_t (&len)[4], bool compact) : stage_(stage), compact_(compact) {
        for (int c = 0; c < 4; c++) {
            off_[c] = off[c];
            len_[c] = len[c];
            range_[c] = 0;
            done_[c] = false;
            slot_of_[c] = (uint32_t)c;
            written_[c] = 0;
            bytes_[c] = 0;
        }
    }
    void frame_slot_hint(uint32_t index, uint32_t slot) override { slot_of_[index] = slot; }
    void start(uint32_t index, const jpgpu_component &, const uint16_t qt[64]) override {
        written_[index] = 0;
        memcpy(qt_[index], qt, 128);
        if (compact_) {
            const uint32_t slot = slot_of_[index];
            writer_[index].reset(new jpgpu::CompactWriter(stage_ + off_[slot], len_[slot] / 128, qt_[index]));
        }
    }
    void append_row(uint32_t index, const int16_t *co, size_t len) override {
        if (compact_) {
            writer_[index]->add_blocks(co, len / 64);
            return;
        }
        const uint32_t slot = slot_of_[index];
        const size_t room = len_[slot] - written_[index], bytes = len * sizeof(int16_t);
        const size_t n = bytes < room ? bytes : room;  // rows past the plane are dropped like the Worker drops them
        memcpy(stage_ + off_[slot] + written_[index], co, n);
        written_[index] += n;
    }
    void finish(uint32_t index, uint32_t slot) override {
        if (slot != slot_of_[index]) throw DecodeError{JPGPU_ERR_INTERNAL, "pipeline: plane finished under another frame slot"};
        // a scan may end early (src/decoder.rs:1000-1006 breaks out of the MCU loops at the image edge): the
        // plane keeps zeros where no row was appended, like the Worker's zero-initialised plane
        if (compact_) {
            bytes_[slot] = writer_[index]->finish(&range_[slot]);
            writer_[index].reset();
        } else {
            if (written_[index] < len_[slot]) memset(stage_ + off_[slot] + written_[index], 0, len_[slot] - written_[index]);
            range_[slot] = jpgpu_range_class(reinterpret_cast<const int16_t *>(stage_ + off_[slot]), len_[slot] / sizeof(int16_t), qt_[index]);
            bytes_[slot] = len_[slot];
        }
        memcpy(slot_qt_[slot], qt_[index], 128);
        done_[slot] = true;
    }
    int range_class(uint32_t slot) const { return range_[slot]; }
    bool done(uint32_t slot) const { return done_[slot]; }
    const uint16_t *qt(uint32_t slot) const { return slot_qt_[slot]; }
    size_t bytes(uint32_t slot) const { return bytes_[slot]; }  // to send for this component

private:
    uint8_t *stage_;
    bool compact_;
    size_t off_[4], len_[4], written_[4], bytes_[4];
    uint32_t slot_of_[4];
    uint16_t qt_[4][64], slot_qt_[4][64];
    int range_[4];
    bool done_[4];
    std::unique_ptr<jpgpu::CompactWriter> writer_[4];
};

// Finished images are handed to one uploader thread: hipMemcpyAsync calls from hundreds of threads contend in the
// runtime, one caller keeps the copy queues busy.  skip = the image failed and will never be uploaded.
struct UploadQueue {
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::pair<uint32_t, int>> items;  // (image, 0 failed / 1 staged: upload it / 2 decode its entropy data on the device)
    void push(uint32_t i, int kind) {
        {
            std::lock_guard<std::mutex> g(m);
            items.emplace_back(i, kind);
        }
        cv.notify_one();
    }
};

bool same_geometry(const jpgpu_image_desc &a, const jpgpu_image_desc &b) {
    if (a.ncomp != b.ncomp || a.out_w != b.out_w || a.out_h != b.out_h || a.color_transform != b.color_transform) return false;
    for (uint32_t c = 0; c < a.ncomp; c++)
        if (memcmp(&a.components[c], &b.components[c], sizeof(jpgpu_component)) != 0) return false;
    return true;
}

// The images of a call are cut into sub-batches (own jpgpu_batch, arenas and pinned staging each): as soon as the
// last image of a sub-batch is uploaded its kernels and its download are enqueued, while the pool is still busy with
// the entropy decoding of the following sub-batches.
struct SubBatch {
    jpgpu_batch *batch = nullptr;
    std::vector<jpgpu_image_desc> descs;
    uint8_t *h_coef = nullptr, *h_out = nullptr;
    size_t h_coef_bytes = 0, h_out_bytes = 0;
    bool compact = false;             // layout of h_coef: compact regions (stage_off) or a mirror of the arena
    std::vector<size_t> stage_off;    // compact mode: [image*4 + comp] offset of the component's region in h_coef
    uint32_t remaining = 0;  // images not yet uploaded / failed (uploader thread only)
    hipEvent_t ready[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t decoded = nullptr;  // recorded on the compute stream behind the sub-batch's pixel kernels: downloads and the gather wait for it
    void drop() {
        if (batch) jpgpu_batch_destroy(batch);
        batch = nullptr;
        if (h_coef) (void)hipHostFree(h_coef);
        if (h_out) (void)hipHostFree(h_out);
        h_coef = h_out = nullptr;
        h_coef_bytes = h_out_bytes = 0;
        descs.clear();
        stage_off.clear();
    }
};

// Up to 32 sub-batches per call (half of them at most for the device-entropy route, 256 images each by default), eight compute
// streams shared round robin.  Round 3: with 8 sub-batches a call of 4,096 files made four device sub-batches of 1,024, whose
// write pass took 11 us per image against 7 us in sub-batches of 256 (a 6.4 GB arena per sub-batch instead of 1.6 GB:
// profiles/round3/08_subbatch_size.txt).
constexpr uint32_t kSubBatchImages = 64, kMaxSubBatches = 64, kComputeStreams = 32, kComputeStreamsDefault = 12;  // (round 4: 12, not 16 — equal without a
    // collective library in the process (54.8 vs 55.9 ms per 4,096 files), but next to RCCL's own streams 16 measure 69-72 ms against 55.5: profiles/round4/08_sub_batch_sizes.txt)
     // (more than 16 in use is slower: 24 streams 109 ms, 32 streams 75 ms per 4,096 files against 53-60, whatever GPU_MAX_HW_QUEUES says — tools/gpu_streams.sh)

}  // namespace

struct jpgpu_pipeline {
    int device = 0;
    std::string err;
    std::unique_ptr<Pool> pool;
    std::unique_ptr<Pool> stage_pool;  // the staging copies of the device entropy route (the uploader thread's parallel-for)
    // results of the last call
    uint32_t n = 0;
    std::vector<std::unique_ptr<Frontend>> fes;
    std::vector<uint8_t> has_frame;  // per image: a frame header was parsed (jpgpu_pipeline_image_info)
    std::vector<int> status;
    std::vector<std::string> errors;
    std::vector<jpgpu_image_info> infos;
    std::vector<int32_t> sub_of, slot;  // image -> sub-batch / index in it, -1 if it never got there
    std::vector<std::vector<jpgpu::host::PlannedScan>> plans;  // per image: scans for the device entropy decoder (empty: host)
    std::vector<jpgpu::host::ProgPlan> prog_plans;             // per image: the scans of a PROGRESSIVE frame for the device (no scans: host)
    std::vector<SubBatch> subs;          // kept across calls while the geometry sequence repeats
    uint32_t n_subs = 0;                 // sub-batches used by the last call
    std::string path;
    bool downloaded = false;  // the last call copied the pixels to host memory
    hipStream_t copy_streams[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t compute[kComputeStreams] = {};
    // JPGPU_PIPELINE_DOWNLOAD: the device-to-host copies have streams of their own (round 5; on the compute stream a sub-batch's
    // download held back the kernels of the next sub-batch that shares the stream, and the copies of a call did not run back to back)
    hipStream_t d2h[kD2HStreams] = {};
    uint16_t req_w = 0, req_h = 0;  // jpgpu_pipeline_set_scale (0 x 0: full size)
    int color_transform = -1;       // jpgpu_pipeline_set_color_transform (< 0: what every image says itself)
    size_t max_bytes = SIZE_MAX;    // jpgpu_pipeline_set_max_decoding_buffer_size
    uint32_t n_compute = kComputeStreamsDefault;  // streams in use (JPGPU_PIPE_STREAMS: tuning knob, up to kComputeStreams)
    jpgpu::DeviceScratch scratch[kComputeStreams];  // work space of the chunk decoder, one per compute stream (launches on a stream run in turn)
    jpgpu_pipeline_timings t{};
    // ---- a child of a multi-device pipeline with JPGPU_PIPELINE_GATHER in force: behind every sub-batch's pixel kernels its pixel arena
    // is copied to `gather_device` (peer-to-peer; a plain device copy when that is this device) on a stream of THIS device — the copy
    // of one sub-batch runs while the next ones decode, every child drives its own link (round 5; round 4 copied everything from one
    // stream of the first device after all children had joined)
    bool gather_on = false;
    int gather_device = -1;
    uint8_t *d_gather = nullptr;      // on gather_device
    size_t gather_cap = 0;
    std::vector<size_t> gather_off;   // [sub-batch] -> offset of its arena in d_gather
    hipStream_t gather_stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gather_ev;  // [sub-batch] timing events around its copy
    uint64_t gather_bytes = 0;
    // ---- several devices behind one object (jpgpu_pipeline_create_multi): this pipeline only deals the images of a call to its
    // children (one ordinary pipeline per listed device, image i -> child i mod n) and maps the per-image accessors back
    std::vector<jpgpu_pipeline *> children;
    std::vector<std::vector<int>> child_cpus;  // CPUs of each child's share (empty: no pinning)
    uint32_t multi_n = 0;                      // images of the last call
    bool gathered = false;                     // JPGPU_PIPELINE_GATHER: the children's pixel arenas were copied to devices[0] (each child's d_gather)
    std::vector<std::string> peer_error;       // [child] why its device cannot reach devices[0] (empty: it can)
};

static const jpgpu_pipeline *child_of(const jpgpu_pipeline *p, uint32_t &image) {
    const uint32_t n = (uint32_t)p->children.size();
    const jpgpu_pipeline *c = p->children[image % n];
    image /= n;
    return c;
}
static int multi_decode(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n, uint32_t flags);
// Up to how many scans does a call walk its progressive frames a WAVE per scan (pipelined, huff_prog_job.hpp)?  Always: the launch
// order keeps a frame's waves on one XCD, producers in front (csrc/batch.cpp), so an oversubscribed launch cannot starve a producer
// (round 5's lanes had to fit the device at once).  Beyond the limit: a wave per TRACK, its scans one after the other —
// JPGPU_PROG_LANES_MAX (tests, A/B; read per call) / JPGPU_PROG_SERIAL.
static uint64_t prog_lanes_max() {
    if (getenv("JPGPU_PROG_SERIAL")) return 0;
    const char *e = getenv("JPGPU_PROG_LANES_MAX");
    return e ? (uint64_t)std::max<long>(atol(e), 0) : ~0ull;
}

static uint32_t progressive_share_for_the_device(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n);
static void pin_to(const std::vector<int> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof(set), &set);  // (best effort: a refused mask leaves the thread where it was)
}

// An image the device entropy decoder handed back: the pinned behaviour is the host decoder's — decode it here (uploader
// thread; rare) and upload it densely, or record the error it raises.
// An image the device entropy decoder handed back: decode it on the host (dense planes, plain upload).  Two steps so that
// several of them can be decoded side by side: host_redecode_stage (any thread), host_redecode_upload (the uploader).
struct Redecode {
    uint32_t image = 0, nc = 0;
    size_t off[4] = {0, 0, 0, 0}, ln[4] = {0, 0, 0, 0};
    std::vector<uint8_t> planes;
    uint16_t qt[4][64];
    int status = JPGPU_OK;
    std::string error;
};

static void read_info_with_options(const jpgpu_pipeline *p, Frontend &fe);

static void host_redecode_stage(jpgpu_pipeline *p, SubBatch &sb, Redecode &r, const uint8_t *data, size_t len) {
    const uint32_t bi = (uint32_t)p->slot[r.image];
    try {
        Frontend fe(data, len, Frontend::Borrowed{});
        read_info_with_options(p, fe);
        r.nc = fe.ncomp();
        size_t total = 0;
        for (uint32_t c = 0; c < r.nc; c++) {
            r.ln[c] = jpgpu_batch_coef_bytes(sb.batch, bi, c);
            r.off[c] = total;
            total += r.ln[c];
        }
        r.planes.resize(total);
        StageSink sink(r.planes.data(), r.off, r.ln, false);
        fe.decode_to(sink);
        for (uint32_t c = 0; c < r.nc; c++) {
            if (!sink.done(c) || !fe.planes_present()[c]) throw DecodeError{JPGPU_ERR_FORMAT, "not all components have data"};
            memcpy(r.qt[c], sink.qt(c), 128);
        }
    } catch (const DecodeError &e) {
        r.status = e.code;
        r.error = e.message;
    } catch (const std::exception &e) {
        r.status = JPGPU_ERR_INTERNAL;
        r.error = e.what();
    }
}

static void host_redecode_upload(jpgpu_pipeline *p, SubBatch &sb, Redecode &r) {
    const uint32_t bi = (uint32_t)p->slot[r.image];
    for (uint32_t c = 0; c < r.nc && r.status == JPGPU_OK; c++) {
        jpgpu_batch_set_quantization_table(sb.batch, bi, c, r.qt[c]);
        if (jpgpu_batch_upload(sb.batch, bi, c, reinterpret_cast<const int16_t *>(r.planes.data() + r.off[c]), r.ln[c] / 2) != JPGPU_OK) {
            r.status = JPGPU_ERR_IO;
            r.error = jpgpu_batch_last_error(sb.batch);
        }
    }
    if (r.status != JPGPU_OK) {
        p->status[r.image] = r.status;
        p->errors[r.image] = r.error;
    }
}

 static int pipeline_create_sized(int device, uint32_t n_threads, uint32_t n_stage_threads, jpgpu_pipeline **out);
#define P_HIP(call)                                                                                              \
    do {                                                                                                         \
        hipError_t _e = (call);                                                                                  \
        if (_e != hipSuccess) return jpgpu::set_err(p->err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

extern "C" {

int jpgpu_pipeline_create(int device, uint32_t n_threads, jpgpu_pipeline **out) {
    if (n_threads == 0) n_threads = default_threads();
    return pipeline_create_sized(device, n_threads, std::max<uint32_t>(2u, n_threads / 2u), out);
}
}  // extern "C"

// n_threads entropy / header workers + n_stage_threads for the staging copies of the device-entropy route (jpgpu_pipeline_create:
// half as many again; a child of a multi-device pipeline: both out of its share of the budget)
static int pipeline_create_sized(int device, uint32_t n_threads, uint32_t n_stage_threads, jpgpu_pipeline **out) {
    if (!out) return JPGPU_ERR_FORMAT;
    jpgpu_pipeline *p = new jpgpu_pipeline();
    *out = p;  // returned even on failure so that last_error can be read
    p->device = device;
    int rc = jpgpu::use_device(device, p->err);
    if (rc) return rc;  // no usable MI355X: there is no CPU fallback for the pixel work
    p->pool.reset(new Pool(std::max(1u, n_threads)));
    p->stage_pool.reset(new Pool(std::max(1u, n_stage_threads)));
    p->subs.resize(kMaxSubBatches);
    for (uint32_t k = 0; k < kCopyStreams; k++) P_HIP(hipStreamCreateWithFlags(&p->copy_streams[k], hipStreamNonBlocking));
    if (const char *e = getenv("JPGPU_PIPE_STREAMS")) p->n_compute = (uint32_t)std::min<long>(std::max<long>(atol(e), 1), kComputeStreams);
    for (uint32_t k = 0; k < p->n_compute; k++) P_HIP(hipStreamCreateWithFlags(&p->compute[k], hipStreamNonBlocking));
    for (uint32_t k = 0; k < kD2HStreams; k++) P_HIP(hipStreamCreateWithFlags(&p->d2h[k], hipStreamNonBlocking));
    for (SubBatch &sb : p->subs) {
        for (uint32_t k = 0; k < kCopyStreams; k++) P_HIP(hipEventCreateWithFlags(&sb.ready[k], hipEventDisableTiming));
        P_HIP(hipEventCreateWithFlags(&sb.decoded, hipEventDisableTiming));
    }
    return JPGPU_OK;
}

extern "C" {

void jpgpu_pipeline_destroy(jpgpu_pipeline *p) {
    if (!p) return;
    std::string e;
    if (!p->children.empty()) {
        for (jpgpu_pipeline *c : p->children) jpgpu_pipeline_destroy(c);
        delete p;
        return;
    }
    if (jpgpu::use_device(p->device, e) == JPGPU_OK) {
        (void)hipDeviceSynchronize();
        if (p->gather_stream) (void)hipStreamDestroy(p->gather_stream);
        for (auto &ev : p->gather_ev) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        if (p->d_gather) {  // (lives on the gathering device)
            if (hipSetDevice(p->gather_device) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipFree(p->d_gather);
            }
            (void)hipSetDevice(p->device);
        }
        for (SubBatch &sb : p->subs) {
            sb.drop();
            for (uint32_t k = 0; k < kCopyStreams; k++)
                if (sb.ready[k]) (void)hipEventDestroy(sb.ready[k]);
            if (sb.decoded) (void)hipEventDestroy(sb.decoded);
        }
        for (uint32_t k = 0; k < kD2HStreams; k++)
            if (p->d2h[k]) (void)hipStreamDestroy(p->d2h[k]);
        for (uint32_t k = 0; k < kCopyStreams; k++)
            if (p->copy_streams[k]) (void)hipStreamDestroy(p->copy_streams[k]);
        for (uint32_t k = 0; k < kComputeStreams; k++)
            if (p->compute[k]) (void)hipStreamDestroy(p->compute[k]);
        for (auto &sc : p->scratch)
            if (sc.d) (void)hipFree(sc.d);
    }
    delete p;
}

const char *jpgpu_pipeline_last_error(const jpgpu_pipeline *p) { return p ? p->err.c_str() : ""; }

// Which of the planned streams are better off with the host decoder after all (decided once the headers of a call are read).
static void back_to_the_host(jpgpu_pipeline *p, uint32_t i, const uint8_t *const *data, const size_t *len) {
    p->plans[i].clear();  // (the planning pass spent the front-end: a fresh one for the host path)
    try {
        p->fes[i].reset(new Frontend(data[i], len[i], Frontend::Borrowed{}));
        read_info_with_options(p, *p->fes[i]);
    } catch (const DecodeError &e) {
        p->status[i] = e.code;
        p->errors[i] = e.message;
    }
}

// Decoder options of the pipeline (jpgpu_pipeline_set_*), for every front-end it creates: what read_info() + the setters + scale()
// leave in a reference Decoder
static void read_info_with_options(const jpgpu_pipeline *p, Frontend &fe) {
    fe.read_info();
    if (p->max_bytes != SIZE_MAX) fe.set_max_decoding_buffer_size(p->max_bytes);
    if (p->color_transform >= 0) fe.set_color_transform(p->color_transform);
    if (p->req_w | p->req_h) {  // Decoder::scale: the IDCT size of this image (the coefficients are the same at every scale)
        uint16_t ow, oh;
        fe.scale(p->req_w, p->req_h, ow, oh);
    }
}

static void keep_on_host_what_the_device_would_decode_slower(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n) {
    if (getenv("JPGPU_PIPE_FORCE_DEVICE")) return;  // (tests, A/B runs)
    // (restart-marker streams cost what streams without markers cost since round 3 — every segment in chunk slots of its own —
    // and stay on the device; rounds 1-3 weighed one lane per segment against the host's cores here)
    // Streams without restart markers whose blocks are very long (noise at quality >= 98: no end-of-block symbols at all) keep
    // the chunk decoder re-synchronising for dozens of passes (measured: beyond ~350 bits per block more launches than it is
    // given) — it would flag them in the end; the host decodes them right away instead.
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK && !p->plans[i].empty() && p->plans[i][0].seg_off.size() == 2) {
            const jpgpu::host::PlannedScan &ps = p->plans[i][0];
            uint64_t blocks = 0;
            for (uint32_t c = 0; c < ps.ncomp; c++) blocks += (uint64_t)ps.comp[c].h * ps.comp[c].v;
            blocks *= ps.n_mcu;
            const uint64_t bits = ps.seg_off.size() >= 2 ? (uint64_t)(ps.seg_off[1] - ps.seg_off[0]) * 8u : 0u;
            if (blocks && bits / blocks > 384u) back_to_the_host(p, i, data, len);
        }
}

// Progressive frames: which of a call's eligible frames does the device decode?  All of them or none (a split was built
// in round 5 and removed: with both routes busy the host's threads and the device route's staging team compete for the same cores) —
// decided by a COST MODEL of what the planner has counted, not by timing earlier calls (round 5's wall-clock dispatcher needed a probe
// call and three more before it settled, mis-routed three different ways and made one call's route depend on a neighbour's load:
// VERDICT r5).  The same call takes the same route the first time and the tenth.
//   device = fixed + max(chain, volume) + frames x per_frame
//     chain  : a scan is one dependent walk, one wave (huff_prog_wave.hpp) — the launch lasts at least as long as the call's LONGEST scan:
//              its entropy-coded bytes x kDevChainNsPerByte;
//     volume : all scans of all frames share the device's scalar units: total entropy-coded bytes x kDevVolumeNsPerByte;
//   host   = fixed + total entropy-coded bytes x kHostNsPerByte / min(worker threads, CPUs the process may use).
// The constants are measurements on an MI355X with an EPYC 9575F host (profiles/round6/03_progressive_cost_model.txt; re-measure with
// tools/prog_calls.py --percent 100 / --percent 0): benches/tower_progressive.jpg's longest scan, 16.7 kB, walks in 7.8 ms; 4,096
// such frames (235 MB of scans) take 62-66 ms, 4,096 distinct ones (268 MB) 78-82; 128 / 192 / 256 of them cost 16 host CPUs 9.7 / 13.1 /
// 16.0-16.7 ms (the device: 9.0 / 9.8 / 10.1-10.6).  The device must be ahead by a tenth (the host route is
// the one whose behaviour on odd streams is pinned).  JPGPU_PIPE_PROG_DEVICE_PERCENT pins the device's share (tests, A/B).
// Returns how many frames keep their device plan; the others get a fresh front-end for the host path.
constexpr double kDevFixedMs = 1.1, kDevChainNsPerByte = 470.0, kDevVolumeNsPerByte = 0.19, kDevPerFrameMs = 0.006, kHostFixedMs = 1.5, kHostNsPerByte = 17.0;
static uint32_t progressive_share_for_the_device(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n) {
    std::vector<uint32_t> elig;
    uint64_t scan_bytes = 0, longest_scan = 0, host_bytes = 0;  // entropy-coded bytes: of the eligible frames' scans, of the longest of those scans, of the progressive frames the host decodes anyway
    for (uint32_t i = 0; i < n; i++) {
        if (p->status[i] != JPGPU_OK) continue;
        if (!p->prog_plans[i].scans.empty()) {
            elig.push_back(i);
            for (const auto &ps : p->prog_plans[i].scans) {
                scan_bytes += ps.stuffed_bytes;
                longest_scan = std::max<uint64_t>(longest_scan, ps.stuffed_bytes);
            }
        } else if (p->infos[i].coding_process == JPGPU_CODING_DCT_PROGRESSIVE) {
            host_bytes += len[i];
        }
    }
    const uint32_t e = (uint32_t)elig.size();
    if (e == 0) return 0;
    uint32_t d;
    if (const char *pin = getenv("JPGPU_PIPE_PROG_DEVICE_PERCENT")) {
        d = (uint32_t)((uint64_t)e * (uint64_t)std::min<long>(std::max<long>(atol(pin), 0), 100) / 100u);
    } else {
        static const uint32_t cpus = granted_cpus();
        const double workers = (double)std::max<uint32_t>(1u, std::min<uint32_t>(p->pool->size(), cpus));
        const double host_all = kHostFixedMs + (double)(scan_bytes + host_bytes) * kHostNsPerByte * 1e-6 / workers;
        const double host_rest = (double)host_bytes * kHostNsPerByte * 1e-6 / workers;
        const double device = kDevFixedMs + std::max((double)longest_scan * kDevChainNsPerByte, (double)scan_bytes * kDevVolumeNsPerByte) * 1e-6 + e * kDevPerFrameMs;
        d = std::max(device, host_rest) < 0.9 * host_all ? e : 0u;
        if (getenv("JPGPU_PIPE_TRACE"))
            fprintf(stderr, "pipeline trace: progressive dispatcher: %u eligible frames, %.2f MB of scans, longest %llu bytes: device %.2f ms, host %.2f ms on %.0f workers -> %s\n",
                    e, scan_bytes * 1e-6, (unsigned long long)longest_scan, device, host_all, workers, d ? "device" : "host");
    }
    // the LAST (e - d) eligible frames go back to the host (its threads start from the front of the list)
    for (uint32_t k = d; k < e; k++) {
        const uint32_t i = elig[k];
        p->prog_plans[i].scans.clear();
        p->prog_plans[i].n_tracks = 0;
        try {
            p->fes[i].reset(new Frontend(data[i], len[i], Frontend::Borrowed{}));
            read_info_with_options(p, *p->fes[i]);
        } catch (const DecodeError &err) {
            p->status[i] = err.code;
            p->errors[i] = err.message;
        }
    }
    return d;
}

int jpgpu_pipeline_decode(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n, uint32_t flags) {
    jpgpu::TraceRange roctx_range("jpgpu_pipeline_decode");
    // (unknown bits are refused, not ignored: 8u was round 2-3's JPGPU_PIPELINE_PROGRESSIVE_DELTAS, removed in round 4 — a caller built
    // against that header would otherwise silently get another transport: ADVICE r4)
    constexpr uint32_t kKnownFlags = JPGPU_PIPELINE_DOWNLOAD | JPGPU_PIPELINE_DENSE | JPGPU_PIPELINE_DEVICE_ENTROPY | JPGPU_PIPELINE_GATHER |
                                     JPGPU_PIPELINE_HOST_LIGHT | JPGPU_PIPELINE_HOST_STAGED | JPGPU_PIPELINE_INPUT_PINNED | JPGPU_PIPELINE_PROGRESSIVE_ON_HOST;
    if (p && (flags & ~kKnownFlags)) return jpgpu::set_err(p->err, JPGPU_ERR_FORMAT, "jpgpu_pipeline_decode: unknown flag bits 0x%x (this library: %s)", flags & ~kKnownFlags, jpgpu_version());
    if (p && !p->children.empty()) return (n && (!data || !len)) ? JPGPU_ERR_FORMAT : multi_decode(p, data, len, n, flags);
    if (!p || !p->pool || (n && (!data || !len))) return JPGPU_ERR_FORMAT;
    if ((flags & JPGPU_PIPELINE_GATHER) && !p->gather_on) flags &= ~(uint32_t)JPGPU_PIPELINE_GATHER;  // (one device: the pixels are where a gather would put them)
    int rc = jpgpu::use_device(p->device, p->err);
    if (rc) return rc;
    const double t0 = now_ms(), cpu0 = process_cpu_ms();
    const bool download = (flags & JPGPU_PIPELINE_DOWNLOAD) != 0, compact = (flags & JPGPU_PIPELINE_DENSE) == 0;
    const bool device_entropy = (flags & JPGPU_PIPELINE_DEVICE_ENTROPY) != 0;
    const bool device_progressive = device_entropy && (flags & JPGPU_PIPELINE_PROGRESSIVE_ON_HOST) == 0 && !getenv("JPGPU_PIPE_PROGRESSIVE_ON_HOST");
    p->n = n;
    p->fes.clear();
    p->fes.resize(n);
    p->has_frame.assign(n, 0);
    p->status.assign(n, JPGPU_ERR_INTERNAL);
    p->errors.assign(n, std::string());
    p->infos.assign(n, jpgpu_image_info{});
    p->sub_of.assign(n, -1);
    p->plans.clear();
    p->plans.resize(n);
    p->prog_plans.clear();
    p->prog_plans.resize(n);
    p->slot.assign(n, -1);
    p->n_subs = 0;
    p->downloaded = download;
    p->path.clear();
    p->t = jpgpu_pipeline_timings{};
    p->t.threads = p->pool->size() + p->stage_pool->size();  // (+ one uploader thread per call, which mostly waits)
    std::vector<jpgpu_image_desc> cand(n);

    // 1. headers
    p->pool->run(n, [&](uint32_t i) {
        try {
            p->fes[i].reset(new Frontend(data[i], len[i], Frontend::Borrowed{}));
            Frontend &fe = *p->fes[i];
            read_info_with_options(p, fe);
            p->infos[i] = fe.info();
            jpgpu_image_desc d;
            memset(&d, 0, sizeof(d));
            d.ncomp = fe.ncomp();
            for (uint32_t c = 0; c < d.ncomp; c++) {
                d.components[c] = fe.components()[c];
                for (int k = 0; k < 64; k++) d.quantization_tables[c][k] = 1;  // set once the scans are parsed
            }
            d.out_w = fe.output_width();
            d.out_h = fe.output_height();
            d.color_transform = fe.color_transform();
            {
                // A block costs at least two bits.  A header that announces far more blocks than the file can hold (truncated or
                // hostile: 65535 x 65535 in 200 bytes) would have the sub-batch allocate gigabytes of arenas and pinned staging
                // — 2.5 s for one such file — before the entropy decoder finds out: let it find out first, into a sink that
                // keeps nothing.  (A stream that decodes after all goes on as usual.)
                uint64_t blocks = 0;
                for (uint32_t c = 0; c < d.ncomp; c++) blocks += (uint64_t)d.components[c].block_width * d.components[c].block_height;
                if ((uint64_t)len[i] * 8u < blocks * 2u) {
                    struct Nothing : RowSink {
                        void start(uint32_t, const jpgpu_component &, const uint16_t *) override {}
                        void append_row(uint32_t, const int16_t *, size_t) override {}
                        void finish(uint32_t, uint32_t) override {}
                    } nothing;
                    Frontend probe(data[i], len[i], Frontend::Borrowed{});
                    read_info_with_options(p, probe);
                    probe.decode_to(nothing);  // throws what the image's decode() would throw
                }
            }
            {
                // what compute_image would refuse for this frame (an impossible sampling combination, a colour function whose row
                // copy would overrun: src/decoder.rs:1300-1336, src/upsampler.rs:20-45) fails THIS image here — a sub-batch is
                // created from frames the pixel backend takes, one refused frame must not fail its neighbours
                jpgpu::ImageJob probe;
                uint8_t *no_planes[4] = {nullptr, nullptr, nullptr, nullptr};
                size_t out_len = 0;
                std::string why;
                const int v = jpgpu::build_image_job(d.components, d.ncomp, no_planes, d.out_w, d.out_h, d.color_transform, nullptr, probe, out_len, why);
                if (v != JPGPU_OK) {
                    // (the reference decodes the entropy data before it gets to compute_image: a stream that is broken as well reports that)
                    struct Nothing : RowSink {
                        void start(uint32_t, const jpgpu_component &, const uint16_t *) override {}
                        void append_row(uint32_t, const int16_t *, size_t) override {}
                        void finish(uint32_t, uint32_t) override {}
                    } nothing;
                    Frontend whole(data[i], len[i], Frontend::Borrowed{});
                    read_info_with_options(p, whole);
                    whole.decode_to(nothing);
                    throw DecodeError{v, why};
                }
            }
            cand[i] = d;
            if (device_entropy && p->infos[i].coding_process == JPGPU_CODING_DCT_PROGRESSIVE) {
                // a progressive frame: its scans as tracks for the device (huff_prog_wave.hpp), if the stream is plainly eligible;
                // how many of a call's eligible frames really go there is decided below, once all headers are read
                if (device_progressive && fe.plan_progressive_scans(p->prog_plans[i])) {
                    for (uint32_t c = 0; c < d.ncomp; c++) memcpy(cand[i].quantization_tables[c], fe.qtable_of_component(c), 128);
                    p->has_frame[i] = 1;
                    p->fes[i].reset();
                } else if (device_progressive) {
                    p->prog_plans[i].scans.clear();
                    p->fes[i].reset(new Frontend(data[i], len[i], Frontend::Borrowed{}));
                    read_info_with_options(p, *p->fes[i]);
                }
            } else if (device_entropy) {  // eligible for the device entropy decoder?  (the planning pass spends the object)
                if (fe.plan_device_scans(p->plans[i])) {
                    for (uint32_t c = 0; c < d.ncomp; c++) memcpy(cand[i].quantization_tables[c], fe.qtable_of_component(c), 128);
                    // The plan holds all the device route needs: the front-end (45 kB of table space) goes back to the allocator of
                    // this thread right away — 4,096 of them alive until the next call were 180 MB of fresh pages per call.
                    p->has_frame[i] = 1;
                    p->fes[i].reset();
                } else {
                    p->plans[i].clear();
                    p->fes[i].reset(new Frontend(data[i], len[i], Frontend::Borrowed{}));
                    read_info_with_options(p, *p->fes[i]);
                }
            }
            if (p->fes[i]) p->has_frame[i] = p->fes[i]->has_frame() ? 1 : 0;
            p->status[i] = JPGPU_OK;
        } catch (const DecodeError &e) {
            p->status[i] = e.code;
            p->errors[i] = e.message;
            if (p->fes[i]) p->has_frame[i] = p->fes[i]->has_frame() ? 1 : 0;
        } catch (const std::exception &e) {
            p->status[i] = JPGPU_ERR_INTERNAL;
            p->errors[i] = e.what();
        }
    });
    if (device_entropy) keep_on_host_what_the_device_would_decode_slower(p, data, len, n);
    const uint32_t n_prog_dev = device_progressive ? progressive_share_for_the_device(p, data, len, n) : 0u;
    const double t1 = now_ms();

    // 2. sub-batches of the images that have a frame (each kept while its geometry sequence repeats)
    // Images bound for the device entropy decoder go first, in large sub-batches: one lane decodes one restart segment, a
    // launch needs many images to fill the machine (and launches of different sub-batches were observed to run one
    // after the other); host-decoded images follow in sub-batches of about 64 so that uploads, kernels and downloads overlap
    // the remaining entropy decoding.
    std::vector<uint32_t> ok;
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK && !p->plans[i].empty()) ok.push_back(i);
    const uint32_t n_dev = (uint32_t)ok.size();  // sequential streams for the chunk decoder; then the progressive frames for the track walker
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK && !p->prog_plans[i].scans.empty()) ok.push_back(i);
    if ((uint32_t)ok.size() - n_dev != n_prog_dev) return jpgpu::set_err(p->err, JPGPU_ERR_INTERNAL, "pipeline: progressive dispatch count");
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK && p->plans[i].empty() && p->prog_plans[i].scans.empty()) ok.push_back(i);
    if (ok.empty()) {
        p->t.headers_ms = t1 - t0;
        p->t.total_ms = now_ms() - t0;
        return JPGPU_OK;
    }
    std::vector<uint32_t> bounds{0u};  // sub-batch j = ok[bounds[j] .. bounds[j+1])
    uint32_t n_dev_subs = 0;           // the first sub-batches hold the images whose entropy data goes to the device ...
    uint32_t first_prog_sub = 0;       // ... of which those from this one on hold progressive frames (the track walker)
    {
        // one lane per chunk of a scan: 256 images fill the machine — but several sub-batches are in flight at a time (one compute
        // stream each, and a hardware queue each when GPU_MAX_HW_QUEUES allows: jpgpu_process_init), the latency-bound late sync
        // passes of one run next to the full passes of another, and staging, upload and kernels of neighbours overlap: 128 images
        // per sub-batch, up to 32 of them (256 files per call 8.6 -> 7.5 ms, 1,024: 21.1 -> 18.8 ms, 4,096: 68.7 -> 63.5 ms against 256 x 16)
        const long dev_sub_env = getenv("JPGPU_PIPE_DEV_SUB") ? atol(getenv("JPGPU_PIPE_DEV_SUB")) : 0;  // tuning knob (read per call)
        // (round 4: calls of up to 512 files in sub-batches of 64 — four of them for 256 files: the first enters the device after a
        // quarter of the staging, and four chains of latency-bound passes overlap instead of two; 256 files 6.5-7.1 -> 6.15 ms on the
        // same boxes, 1,024 and 4,096 files unchanged: profiles/round4/08_sub_batch_sizes.txt)
        const uint32_t dev_sub_images = dev_sub_env > 0 ? (uint32_t)dev_sub_env : (n_dev <= 512u ? kSubBatchImages : 2u * kSubBatchImages);
        const long dev_cap_env = getenv("JPGPU_PIPE_MAX_DEV_SUBS") ? atol(getenv("JPGPU_PIPE_MAX_DEV_SUBS")) : 0;  // tuning knob (read per call)
        const uint32_t dev_cap = dev_cap_env > 0 ? (uint32_t)std::min<long>(dev_cap_env, kMaxSubBatches / 2u) : kMaxSubBatches / 2u;
        const uint32_t dev_subs = n_dev ? std::min<uint32_t>(dev_cap, (n_dev + dev_sub_images - 1u) / dev_sub_images) : 0u;
        for (uint32_t j = 1; j <= dev_subs; j++) bounds.push_back((uint32_t)((uint64_t)n_dev * j / dev_subs));
        // Progressive frames for the device: FEW, large sub-batches — a launch lasts as long as one lane's walk of the longest track
        // whatever the number of frames (every track has a lane of its own while they fit the machine: 512 waves), so frames split
        // over sub-batches that share a stream would pay that walk once per sub-batch; up to four, for the overlap of one's staging
        // and upload with another's walk
        const long prog_sub_env = getenv("JPGPU_PIPE_PROG_SUBS") ? atol(getenv("JPGPU_PIPE_PROG_SUBS")) : 0;  // tuning knob (read per call)
        const uint32_t prog_subs = n_prog_dev ? std::min<uint32_t>({(uint32_t)(prog_sub_env > 0 ? prog_sub_env : 4), kMaxSubBatches / 2u - std::min(dev_subs, kMaxSubBatches / 2u - 1u),
                                                                    (n_prog_dev + 255u) / 256u}) : 0u;
        first_prog_sub = dev_subs;
        for (uint32_t j = 1; j <= prog_subs; j++) bounds.push_back(n_dev + (uint32_t)((uint64_t)n_prog_dev * j / prog_subs));
        n_dev_subs = dev_subs + prog_subs;
        const uint32_t n_devs = n_dev + n_prog_dev, n_host = (uint32_t)ok.size() - n_devs;
        const uint32_t host_subs = n_host ? std::min<uint32_t>(kMaxSubBatches - n_dev_subs, (n_host + kSubBatchImages - 1u) / kSubBatchImages) : 0u;
        for (uint32_t j = 1; j <= host_subs; j++) bounds.push_back(n_devs + (uint32_t)((uint64_t)n_host * j / host_subs));
    }
    const uint32_t n_subs = (uint32_t)bounds.size() - 1u;
    p->n_subs = n_subs;
    for (uint32_t j = 0; j < n_subs; j++) {
        SubBatch &sb = p->subs[j];
        const uint32_t first = bounds[j], last = bounds[j + 1];
        std::vector<jpgpu_image_desc> descs;
        for (uint32_t k = first; k < last; k++) {
            p->sub_of[ok[k]] = (int32_t)j;
            p->slot[ok[k]] = (int32_t)(k - first);
            descs.push_back(cand[ok[k]]);
        }
        sb.remaining = last - first;
        // Sub-batches of device-entropy images need no pinned staging for coefficients (their entropy-coded bytes are staged by
        // the batch itself; an image the device decoder hands back is decoded into memory of its own): for 4096 x 1080p that
        // would have been 28 GB of pinned host memory.
        const bool staged = j >= n_dev_subs;
        bool reuse = sb.batch && descs.size() == sb.descs.size() && sb.compact == compact && (sb.h_coef != nullptr) == staged;
        for (size_t k = 0; reuse && k < descs.size(); k++) reuse = same_geometry(descs[k], sb.descs[k]);
        if (!reuse) {
            sb.drop();
            rc = jpgpu_batch_create(p->device, descs.data(), (uint32_t)descs.size(), JPGPU_BATCH_DEFAULT, &sb.batch);
            if (rc) {
                // a frame the pixel backend refuses (e.g. an impossible sampling combination) fails the images of
                // its sub-batch the way the reference fails it in compute_image
                const std::string msg = sb.batch ? jpgpu_batch_last_error(sb.batch) : "batch_create";
                for (uint32_t k = first; k < last; k++) {
                    p->status[ok[k]] = rc;
                    p->errors[ok[k]] = msg;
                    p->sub_of[ok[k]] = p->slot[ok[k]] = -1;
                }
                sb.drop();
                sb.remaining = 0;
                continue;
            }
            sb.descs = descs;
            sb.compact = compact;
            sb.h_coef_bytes = jpgpu_batch_coef_arena_bytes(sb.batch);
            if (compact) {  // worst-case compact size per component (12 B/block more than dense), 256-B aligned
                sb.stage_off.assign(descs.size() * 4, 0);
                size_t so = 0;
                for (size_t k = 0; k < descs.size(); k++)
                    for (uint32_t c = 0; c < descs[k].ncomp; c++) {
                        sb.stage_off[k * 4 + c] = so;
                        so += jpgpu::align_up(jpgpu::compact_max_bytes(jpgpu_batch_coef_bytes(sb.batch, (uint32_t)k, c) / 128), 256);
                    }
                sb.h_coef_bytes = std::max<size_t>(so, 256);
            }
            if (staged) P_HIP(hipHostMalloc((void **)&sb.h_coef, sb.h_coef_bytes, hipHostMallocDefault));
            else sb.h_coef_bytes = 0;
        }
        if (download && !sb.h_out) {
            sb.h_out_bytes = jpgpu_batch_out_arena_bytes(sb.batch);
            P_HIP(hipHostMalloc((void **)&sb.h_out, sb.h_out_bytes, hipHostMallocDefault));
        }
        const char *pth = jpgpu_batch_path(sb.batch);
        if (p->path.empty()) p->path = pth;
        else if (p->path != pth) p->path = "mixed";
    }
    p->gather_bytes = 0;
    if (p->gather_on) {  // (a child of a multi-device pipeline: where its sub-batches' pixels go on the gathering device)
        size_t total = 0;
        p->gather_off.assign(n_subs, 0);
        for (uint32_t j = 0; j < n_subs; j++) {
            p->gather_off[j] = total;
            total += p->subs[j].batch ? jpgpu::align_up(jpgpu_batch_out_arena_bytes(p->subs[j].batch), 256) : 0;
        }
        if (total > p->gather_cap) {
            // (allocated on the gathering device by this child's thread: hipSetDevice is per thread)
            if (hipSetDevice(p->gather_device) != hipSuccess) return jpgpu::set_err(p->err, JPGPU_ERR_IO, "gather: cannot select device %d", p->gather_device);
            if (p->d_gather) (void)hipFree(p->d_gather);
            p->d_gather = nullptr;
            p->gather_cap = 0;
            const hipError_t ge = hipMalloc((void **)&p->d_gather, total + total / 8);
            (void)hipSetDevice(p->device);
            if (ge != hipSuccess) return jpgpu::set_err(p->err, JPGPU_ERR_IO, "gather: %zu bytes on device %d: %s", total + total / 8, p->gather_device, hipGetErrorString(ge));
            p->gather_cap = total + total / 8;
        }
        if (!p->gather_stream) P_HIP(hipStreamCreateWithFlags(&p->gather_stream, hipStreamNonBlocking));  // on THIS device: the source drives the copy
        while (p->gather_ev.size() < n_subs) {
            hipEvent_t a = nullptr, b = nullptr;
            P_HIP(hipEventCreate(&a));
            P_HIP(hipEventCreate(&b));
            p->gather_ev.emplace_back(a, b);
        }
    }
    const double t2 = now_ms();

    // 3. entropy decoding (pool) + per-image upload and per-sub-batch kernels / download (one uploader thread)
    std::atomic<uint64_t> jpeg_bytes{0}, coef_bytes{0};
    std::atomic<int> hip_failed{0};
    UploadQueue q;
    uint32_t n_jobs = 0;
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK) n_jobs++;
    std::vector<size_t> first_of(n), bytes_of(n);  // dense mode: one copy per image
    std::vector<size_t> cbytes((size_t)n * 4, 0);  // compact mode: bytes per component
    std::vector<int> crange((size_t)n * 4, 0);
    std::string launch_err;
    double t_last_upload = t2;
    const bool trace = getenv("JPGPU_PIPE_TRACE") != nullptr;
    std::mutex trace_m;
    double busy_sum = 0, busy_max = 0, last_end = 0, prog_host_ms = 0, prog_dev_ms = 0;
    uint64_t prog_host_bytes = 0, prog_dev_bytes = 0;
    uint32_t device_prog_images = 0, prog_host_images = 0, n_host_images = 0, light_images = 0, entry_images = 0;
    // Host light (include/jpgpu_decoder.h) is the DEFAULT at every thread count (round 6); JPGPU_PIPELINE_HOST_STAGED asks for the host's
    // staging pass.  Why no rule by thread count any more (round 5 chose staging above 16 worker threads): host staging swings 2 x with the
    // box and with what else runs on it — 4,096 x 1080p at 16 CPUs / 32 threads: 49-51 ms on quiet boxes, 60.7 ms on the driver's, 115-119 ms
    // there at 2-8 CPUs — while light sat at 48-53 ms in every run on every box (profiles/round5/07_*, round6/).  What
    // staging wins where it wins (1-2 ms of 50) is not worth what it loses where it loses.
    const bool input_pinned = (flags & JPGPU_PIPELINE_INPUT_PINNED) != 0;
    const char *light_env = getenv("JPGPU_PIPE_HOST_LIGHT");  // (tests, fuzzers, A/B: 1 / 0 force the mode for calls that do not say themselves)
    const bool light_default = light_env ? atoi(light_env) != 0 : true;
    const bool host_light = input_pinned || (flags & JPGPU_PIPELINE_HOST_LIGHT) != 0 || ((flags & JPGPU_PIPELINE_HOST_STAGED) == 0 && light_default);
    // JPGPU_PIPE_ENTRY_PIXELS (default 1): 4:2:0 images keep their scan as the chunk decoder's entry lists and the pixel walk reads those
    // (csrc/fused_entries.hpp) instead of whole blocks expanded into the coefficient arena; 0: round 5's expansion kernel for every image
    const char *entry_env = getenv("JPGPU_PIPE_ENTRY_PIXELS");  // (read per call: tests switch it)
    const bool entry_pixels = entry_env ? atoi(entry_env) != 0 : true;
    const uint32_t entropy_mode = (host_light ? jpgpu::DEVICE_ENTROPY_LIGHT : 0u) | (input_pinned ? jpgpu::DEVICE_ENTROPY_INPUT_PINNED : 0u) |
                                  (entry_pixels ? jpgpu::DEVICE_ENTROPY_ENTRY_PIXELS : 0u);
    // Progressive frames for the device: a lane per scan only while ALL the call's lanes fit the device at once (prog_lanes_max())
    uint64_t prog_lanes = 0;
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK) prog_lanes += p->prog_plans[i].scans.size();
    const bool prog_pipelined = prog_lanes <= prog_lanes_max();
    double prog_dev_extra_ms = 0;  // launches of progressive sub-batches: host time of the launch calls + range scan + pixel kernels (device), summed
    uint32_t device_rejected = 0, device_images = 0;
    double dev_ms[4] = {0, 0, 0, 0};  // JPGPU_BATCH_KERNEL_TIMES: phases of the device entropy path, summed over the sub-batches
    bool dev_ms_valid = false;
    // the pool is idle while device-entropy images are staged (its tasks for them return at once): lend it to the copy
    std::mutex par_m;
    // (a team of its own, kept between calls: the pool proper may still be inside its run() of step 3 when the uploader thread gets
    // here, and sixteen fresh threads per launch — 512 thread starts per 4,096-file call — were half of a launch's host time)
    const std::function<void(uint32_t, const std::function<void(uint32_t)> &)> par_for = [&](uint32_t cnt, const std::function<void(uint32_t)> &fn) {
        p->stage_pool->run(cnt, fn);
    };
    (void)par_m;
    // What follows the pixel kernels of sub-batch `sj` on its compute stream `cs`: the download to pinned host memory and / or the
    // copy to the gathering device, each on a stream of its own behind the `decoded` event (the compute stream goes on with the next
    // sub-batch that shares it).  Called again when a sub-batch is decoded a second time (an image the device decoder handed back).
    // `again`: the sub-batch's second decode: the first copies have been waited for by the caller (the second decode writes
    // the arena they read), the second ones replace them — their bytes are counted once, their timing events are the ones that stay.
    auto after_decode = [&](uint32_t sj, hipStream_t cs, bool again = false) -> bool {
        SubBatch &sb = p->subs[sj];
        if (!download && !p->gather_on) return true;
        if (hipEventRecord(sb.decoded, cs) != hipSuccess) return false;
        if (download) {
            static const uint32_t n_d2h = (uint32_t)std::min<long>(std::max<long>(getenv("JPGPU_PIPE_D2H_STREAMS") ? atol(getenv("JPGPU_PIPE_D2H_STREAMS")) : kD2HStreams, 1), kD2HStreams);
            hipStream_t ds = p->d2h[sj % n_d2h];
            const double c0 = trace ? now_ms() : 0.0;
            // (by a copy kernel that writes the pinned block itself: the copy engine's hipMemcpyAsync reached 33 GB/s in here — with a
            // host core busy the whole time — against 57 for the same copies alone: tools/attic/probe_d2h*.hip, profiles/round5)
            if (hipStreamWaitEvent(ds, sb.decoded, 0) != hipSuccess ||
                jpgpu::copy_device_to_pinned_host(sb.h_out, jpgpu_batch_out_arena(sb.batch), sb.h_out_bytes, ds) != JPGPU_OK)
                return false;
            if (trace) fprintf(stderr, "pipeline trace: download of sub-batch %u (%zu MB) enqueued at +%.2f ms, the call took %.2f ms\n", sj, sb.h_out_bytes >> 20, c0 - t2, now_ms() - c0);
        }
        if (p->gather_on && p->d_gather) {
            const size_t bytes = jpgpu_batch_out_arena_bytes(sb.batch);
            uint8_t *dst = p->d_gather + p->gather_off[sj];
            hipStream_t gs = p->gather_stream;
            if (hipStreamWaitEvent(gs, sb.decoded, 0) != hipSuccess || hipEventRecord(p->gather_ev[sj].first, gs) != hipSuccess) return false;
            // (JPGPU_PIPE_FORCE_PEER_COPY=1, tests: hipMemcpyPeerAsync also when the gathering device is this one — legal with identical
            // ordinals —, so that a box with one GPU drives the call and its event bookkeeping, not the plain-copy branch)
            static const bool force_peer = getenv("JPGPU_PIPE_FORCE_PEER_COPY") != nullptr && atoi(getenv("JPGPU_PIPE_FORCE_PEER_COPY")) != 0;
            const hipError_t ce = p->gather_device == p->device && !force_peer
                                      ? hipMemcpyAsync(dst, jpgpu_batch_out_arena(sb.batch), bytes, hipMemcpyDeviceToDevice, gs)  // (a device listed twice: no peer)
                                      : hipMemcpyPeerAsync(dst, p->gather_device, jpgpu_batch_out_arena(sb.batch), p->device, bytes, gs);
            if (ce != hipSuccess || hipEventRecord(p->gather_ev[sj].second, gs) != hipSuccess) return false;
            if (!again) p->gather_bytes += bytes;
        }
        return true;
    };
    // the copies after_decode() started for sub-batch `sj` have left the arena (before it is written a second time)
    auto wait_for_copies = [&](uint32_t sj) -> bool {
        static const uint32_t n_d2h = (uint32_t)std::min<long>(std::max<long>(getenv("JPGPU_PIPE_D2H_STREAMS") ? atol(getenv("JPGPU_PIPE_D2H_STREAMS")) : kD2HStreams, 1), kD2HStreams);
        if (download && hipStreamSynchronize(p->d2h[sj % n_d2h]) != hipSuccess) return false;
        if (p->gather_on && p->d_gather && hipStreamSynchronize(p->gather_stream) != hipSuccess) return false;
        return true;
    };
    std::thread uploader([&] {
        std::string e;
        if (jpgpu::use_device(p->device, e) != JPGPU_OK) hip_failed.store(1);
        uint32_t handled = 0, k = 0;
        std::vector<std::pair<uint32_t, int>> take;
        std::vector<std::vector<uint32_t>> dev_images(p->n_subs);  // per sub-batch: images whose entropy data goes to the device
        std::vector<uint32_t> pending_subs;                         // sub-batches whose device entropy launch is in flight
        while (handled < n_jobs) {
            {
                std::unique_lock<std::mutex> g(q.m);
                q.cv.wait(g, [&] { return !q.items.empty(); });
                take.swap(q.items);
            }
            for (const auto &it : take) {
                const uint32_t i = it.first;
                SubBatch &sb = p->subs[(uint32_t)p->sub_of[i]];
                if (it.second == 2) {
                    dev_images[(uint32_t)p->sub_of[i]].push_back(i);
                    device_images++;
                }
                if (it.second == 1 && !hip_failed.load()) {
                    hipStream_t cps = p->copy_streams[k++ % kCopyStreams];
                    if (sb.compact) {
                        const uint32_t bi = (uint32_t)p->slot[i];
                        for (uint32_t c = 0; c < sb.descs[bi].ncomp; c++)
                            if (jpgpu::batch_upload_compact(sb.batch, bi, c, sb.h_coef + sb.stage_off[(size_t)bi * 4 + c],
                                                            cbytes[(size_t)i * 4 + c], crange[(size_t)i * 4 + c], cps, true) != JPGPU_OK) {
                                launch_err = jpgpu_batch_last_error(sb.batch);
                                hip_failed.store(1);
                            }
                    } else {
                        uint8_t *d_coef = (uint8_t *)jpgpu_batch_coef_arena(sb.batch);
                        if (hipMemcpyAsync(d_coef + first_of[i], sb.h_coef + first_of[i], bytes_of[i], hipMemcpyHostToDevice, cps) != hipSuccess)
                            hip_failed.store(1);
                    }
                }
                handled++;
                if (--sb.remaining == 0 && !hip_failed.load()) {  // sub-batch complete: kernels + download behind its uploads
                    hipStream_t cs = p->compute[(uint32_t)p->sub_of[i] % p->n_compute];
                    bool okk = true;
                    for (uint32_t c = 0; c < kCopyStreams && okk; c++)
                        okk = hipEventRecord(sb.ready[c], p->copy_streams[c]) == hipSuccess &&
                              hipStreamWaitEvent(cs, sb.ready[c], 0) == hipSuccess;
                    std::vector<uint32_t> &dv = dev_images[(uint32_t)p->sub_of[i]];
                    if (okk && !dv.empty()) {
                        // entropy decoding on the device: enqueue now, finish the sub-batch (collect, stragglers, pixel
                        // kernels, download) when nothing else is waiting — the launches of several sub-batches then
                        // run side by side (one wave per SIMD each: they do not compete)
                        const double l0 = now_ms();
                        if ((uint32_t)p->sub_of[i] >= first_prog_sub) {  // progressive frames: one lane per track (huff_prog_wave.hpp)
                            std::vector<jpgpu::DeviceProgressiveImage> list;
                            for (uint32_t di : dv) {
                                list.push_back(jpgpu::DeviceProgressiveImage{(uint32_t)p->slot[di], data[di], &p->prog_plans[di]});
                                prog_dev_bytes += len[di];
                            }
                            device_prog_images += (uint32_t)dv.size();
                            okk = jpgpu::batch_device_progressive_launch(sb.batch, list.data(), (uint32_t)list.size(), cs, &par_for,
                                                                         p->copy_streams[(uint32_t)p->sub_of[i] % kCopyStreams],
                                                                         &p->scratch[(uint32_t)p->sub_of[i] % p->n_compute], prog_pipelined) == JPGPU_OK;
                            prog_dev_extra_ms += now_ms() - l0;
                        } else {
                            std::vector<jpgpu::DeviceEntropyImage> list;
                            for (uint32_t di : dv) list.push_back(jpgpu::DeviceEntropyImage{(uint32_t)p->slot[di], data[di], &p->plans[di]});
                            uint32_t n_light = 0, n_entry = 0;
                            okk = jpgpu::batch_device_entropy_launch(sb.batch, list.data(), (uint32_t)list.size(), cs, &par_for,
                                                                     p->copy_streams[(uint32_t)p->sub_of[i] % kCopyStreams],
                                                                     &p->scratch[(uint32_t)p->sub_of[i] % p->n_compute], n_dev_subs <= 2u, entropy_mode, &n_light, &n_entry) == JPGPU_OK;
                            light_images += n_light;
                            entry_images += n_entry;
                        }
                        if (trace) fprintf(stderr, "pipeline trace: device entropy launch of sub-batch %d at +%.2f ms took %.2f ms (host)\n", p->sub_of[i], l0 - t2, now_ms() - l0);
                        // The pixel kernels follow at once on the same stream: the classes of the decoded coefficients are a
                        // by-product of the write pass and stay on the device (range_stats.hpp), so nothing has to come
                        // back to the host in between.  What does come back, later, is the status word per image: an image
                        // the device decoder refused is decoded here and the sub-batch's kernels run once more.
                        if (okk && jpgpu_batch_decode(sb.batch, cs) != JPGPU_OK) okk = false;
                        if (!okk) launch_err = jpgpu_batch_last_error(sb.batch);
                        else pending_subs.push_back((uint32_t)p->sub_of[i]);
                        // (download / gather at once, on the assumption that the device decoder hands nothing back — the rule; a
                        // sub-batch it does hand an image back from is decoded and copied once more below)
                        if (okk && !after_decode((uint32_t)p->sub_of[i], cs)) okk = false;
                        if (!okk) hip_failed.store(1);
                        continue;
                    }
                    if (okk && jpgpu_batch_decode(sb.batch, cs) != JPGPU_OK) {
                        launch_err = jpgpu_batch_last_error(sb.batch);
                        okk = false;
                    }
                    if (okk) okk = after_decode((uint32_t)p->sub_of[i], cs);
                    if (!okk) hip_failed.store(1);
                }
            }
            take.clear();
            // (only once every image is accounted for: synchronising earlier would hold back the launches of the
            // sub-batches that complete in the meantime)
            if (handled >= n_jobs && !pending_subs.empty() && !hip_failed.load()) {
                for (uint32_t sj : pending_subs) {
                    SubBatch &sb = p->subs[sj];
                    hipStream_t cs = p->compute[sj % p->n_compute];
                    std::vector<uint32_t> &dv = dev_images[sj];
                    std::vector<uint32_t> st(dv.size(), 0);
                    const double s0 = now_ms();
                    bool okk = hipStreamSynchronize(cs) == hipSuccess;
                    if (trace) fprintf(stderr, "pipeline trace: sub-batch %u synchronised at +%.2f ms after waiting %.2f ms\n", sj, now_ms() - t2, now_ms() - s0);
                    okk = okk &&
                               jpgpu::batch_device_entropy_collect(sb.batch, st.data(), (uint32_t)st.size()) == JPGPU_OK;
                    if (const char *dump = getenv("JPGPU_PIPE_DUMP_COEFS")) {  // (debugging aid: the first image's planes as the device decoders left them -> <path>.c<component>.bin)
                        for (uint32_t c = 0; okk && c < 4u; c++) {
                            const size_t nb = jpgpu_batch_coef_bytes(sb.batch, 0, c);
                            if (!nb) continue;
                            std::vector<uint8_t> h(nb);
                            if (hipMemcpy(h.data(), (const uint8_t *)jpgpu_batch_coef_arena(sb.batch) + jpgpu_batch_coef_offset(sb.batch, 0, c), nb, hipMemcpyDeviceToHost) != hipSuccess) break;
                            const std::string name = std::string(dump) + ".c" + std::to_string(c) + ".bin";
                            if (FILE *f = fopen(name.c_str(), "wb")) {
                                fwrite(h.data(), 1, nb, f);
                                fclose(f);
                            }
                        }
                    }
                    if (okk && sj >= first_prog_sub) {  // (what the dispatcher learns: the walk of this launch)
                        float kms = 0.f;
                        if (jpgpu::batch_progressive_kernel_ms(sb.batch, &kms)) prog_dev_ms = std::max(prog_dev_ms, (double)kms);
                        // (what the frames of this sub-batch cost on top of the walk: the host side of the launch — staging, job records —
                        // and the device's range scan + pixel kernels, from the phase events)
                        float ph[4];
                        if (jpgpu::batch_phase_times(sb.batch, ph)) prog_dev_extra_ms += ph[2] + ph[3];
                    }
                    std::vector<Redecode> redo;
                    for (size_t k2 = 0; okk && k2 < dv.size(); k2++)
                        if (st[k2]) {
                            if (trace) fprintf(stderr, "pipeline trace: image %u handed back by the device decoder, status 0x%x\n", dv[k2], st[k2]);
                            device_rejected++;
                            redo.emplace_back();
                            redo.back().image = dv[k2];
                        }
                    {
                        float ms[4];
                        float stamp[6];
                        if (okk && trace && jpgpu::batch_phase_stamps(p->subs[pending_subs[0]].batch, sb.batch, stamp))
                            fprintf(stderr, "pipeline trace: sub-batch %u on the device: entropy launch +%.2f, sync passes %.2f .. %.2f, expansion .. %.2f, pixel kernels %.2f .. %.2f ms\n",
                                    sj, stamp[0], stamp[1], stamp[2], stamp[3], stamp[4], stamp[5]);
                        if (okk && jpgpu::batch_phase_times(sb.batch, ms)) {  // JPGPU_BATCH_KERNEL_TIMES
                            dev_ms[0] += ms[0], dev_ms[1] += ms[1], dev_ms[2] += ms[2], dev_ms[3] += ms[3];
                            dev_ms_valid = true;
                        }
                    }
                    if (!redo.empty()) {
                        const std::function<void(uint32_t)> body = [&](uint32_t r) { host_redecode_stage(p, sb, redo[r], data[redo[r].image], len[redo[r].image]); };
                        if (redo.size() > 1) par_for((uint32_t)redo.size(), body);
                        else body(0);
                        for (Redecode &r : redo) host_redecode_upload(p, sb, r);
                        okk = okk && wait_for_copies(sj);  // (write after read, made explicit: the first download / gather read the arena the kernels now rewrite)
                        if (okk && jpgpu_batch_decode(sb.batch, cs) != JPGPU_OK) {  // (the whole sub-batch once more: rare)
                            launch_err = jpgpu_batch_last_error(sb.batch);
                            okk = false;
                        }
                        if (okk) okk = after_decode(sj, cs, true);
                    }
                    if (!okk) hip_failed.store(1);
                }
                pending_subs.clear();
            }
        }
        t_last_upload = now_ms();
    });
    p->pool->run(n, [&](uint32_t i) {
        if (p->status[i] != JPGPU_OK) return;
        SubBatch &sb = p->subs[(uint32_t)p->sub_of[i]];
        const uint32_t bi = (uint32_t)p->slot[i];
        size_t off[4] = {0, 0, 0, 0}, ln[4] = {0, 0, 0, 0};
        const uint32_t nc = cand[i].ncomp;  // (images planned for the device have no front-end any more)
        for (uint32_t c = 0; c < nc; c++) {
            off[c] = sb.compact ? sb.stage_off[(size_t)bi * 4 + c] : jpgpu_batch_coef_offset(sb.batch, bi, c);
            ln[c] = jpgpu_batch_coef_bytes(sb.batch, bi, c);
        }
        try {
            const double w0 = now_ms();
            if (!p->plans[i].empty() || !p->prog_plans[i].scans.empty()) {  // planned in the headers phase: the entropy-coded bytes go to the device as they are
                for (uint32_t c = 0; c < nc; c++) jpgpu_batch_set_quantization_table(sb.batch, bi, c, cand[i].quantization_tables[c]);
                jpeg_bytes += len[i];
                for (const auto &ps : p->plans[i]) coef_bytes += ps.seg_off.back();  // bytes that cross PCIe for this image
                for (const auto &ps : p->prog_plans[i].scans) coef_bytes += ps.stuffed_bytes;
                q.push(i, 2);
                return;
            }
            Frontend &fe = *p->fes[i];
            StageSink sink(sb.h_coef, off, ln, sb.compact);
            fe.decode_to(sink);
            {
                const double w1 = now_ms();
                std::lock_guard<std::mutex> g(trace_m);
                busy_sum += w1 - w0;
                busy_max = std::max(busy_max, w1 - w0);
                last_end = std::max(last_end, w1);
                n_host_images++;
                if (p->infos[i].coding_process == JPGPU_CODING_DCT_PROGRESSIVE) {
                    prog_host_ms += w1 - w0;
                    prog_host_bytes += len[i];
                    prog_host_images++;
                }
            }
            for (uint32_t c = 0; c < nc; c++)
                if (!sink.done(c) || !fe.planes_present()[c]) throw DecodeError{JPGPU_ERR_FORMAT, "not all components have data"};
            size_t sent = 0;
            for (uint32_t c = 0; c < nc; c++) {
                jpgpu_batch_set_quantization_table(sb.batch, bi, c, sink.qt(c));
                jpgpu_batch_set_range_class(sb.batch, bi, c, sink.range_class(c));
                cbytes[(size_t)i * 4 + c] = sink.bytes(c);
                crange[(size_t)i * 4 + c] = sink.range_class(c);
                sent += sink.bytes(c);
            }
            // dense mode: the planes of one image are consecutive in the arena: one copy
            first_of[i] = off[0];
            bytes_of[i] = off[nc - 1] + ln[nc - 1] - off[0];
            jpeg_bytes += len[i];
            coef_bytes += sb.compact ? sent : bytes_of[i];
            q.push(i, 1);
        } catch (const DecodeError &e) {
            p->status[i] = e.code;
            p->errors[i] = e.message;
            q.push(i, 0);
        } catch (const std::exception &e) {
            p->status[i] = JPGPU_ERR_INTERNAL;
            p->errors[i] = e.what();
            q.push(i, 0);
        }
    });
    const double t3 = now_ms();
    if (trace && device_entropy) fprintf(stderr, "pipeline trace: device entropy decoder handed %u image(s) back to the host\n", device_rejected);
    if (trace) fprintf(stderr, "pipeline trace: workers %.1f ms wall, decode_to sum %.1f ms (avg %.2f, max %.2f), last decode end +%.1f ms\n", t3 - t2, busy_sum, busy_sum / std::max(1u, n_jobs), busy_max, last_end - t2);
    uploader.join();
    if (hip_failed.load())
        return jpgpu::set_err(p->err, JPGPU_ERR_IO, "pipeline: upload / launch failed%s%s", launch_err.empty() ? "" : ": ", launch_err.c_str());

    // 4. drain: whatever kernels and downloads are still in flight
    for (uint32_t k = 0; k < kCopyStreams; k++) P_HIP(hipStreamSynchronize(p->copy_streams[k]));
    for (uint32_t k = 0; k < p->n_compute; k++) P_HIP(hipStreamSynchronize(p->compute[k]));
    if (trace) fprintf(stderr, "pipeline trace: copy and compute streams drained at +%.2f ms\n", now_ms() - t2);
    if (download)
        for (uint32_t k = 0; k < kD2HStreams; k++) {
            P_HIP(hipStreamSynchronize(p->d2h[k]));
            if (trace) fprintf(stderr, "pipeline trace: download stream %u drained at +%.2f ms\n", k, now_ms() - t2);
        }
    // (a child's gather stream is NOT waited for here: the parent does, after every child has decoded — what it then still waits is the exposed part)
    const double t4 = now_ms();
    uint64_t pixel_bytes = 0;
    uint32_t okc = 0;
    for (uint32_t i = 0; i < n; i++)
        if (p->status[i] == JPGPU_OK) {
            okc++;
            pixel_bytes += jpgpu_batch_out_bytes(p->subs[(uint32_t)p->sub_of[i]].batch, (uint32_t)p->slot[i]);
        }
    p->t.headers_ms = t1 - t0;
    p->t.setup_ms = t2 - t1;
    p->t.entropy_and_upload_ms = t3 - t2;
    p->t.kernels_ms = 0.0;  // overlapped: see drain_ms
    p->t.download_ms = t4 - t3;
    p->t.total_ms = t4 - t0;
    p->t.decode_ms = t4 - t0;
    p->t.images_ok = okc;
    p->t.dev_times_valid = dev_ms_valid ? 1u : 0u;
    p->t.dev_fill_ms = dev_ms[0];
    p->t.dev_sync_ms = dev_ms[1];
    p->t.dev_write_ms = dev_ms[2];
    p->t.dev_pixel_ms = dev_ms[3];
    p->t.jpeg_bytes = jpeg_bytes.load();
    p->t.coefficient_bytes = coef_bytes.load();
    p->t.pixel_bytes = pixel_bytes;
    p->t.images_device_entropy = device_images;
    p->t.images_device_rejected = device_rejected;
    p->t.images_device_progressive = device_prog_images;
    p->t.images_host_light = light_images;
    p->t.images_entry_pixels = entry_images;
    p->t.input_pinned = input_pinned ? 1u : 0u;
    if (trace && (prog_host_bytes || device_prog_images))
        fprintf(stderr, "pipeline trace: progressive frames: %u on the device (walk %.2f ms, launches + range scan + pixels %.2f ms), %u on the host (entropy phase %.2f ms, %.1f ns per byte and thread)\n",
                device_prog_images, prog_dev_ms, prog_dev_extra_ms, prog_host_images, t3 - t2, prog_host_bytes ? prog_host_ms * 1e6 / (double)prog_host_bytes : 0.0);
    p->t.cpu_ms = process_cpu_ms() - cpu0;
    (void)t_last_upload;
    return JPGPU_OK;
}

static const SubBatch *sub_of(const jpgpu_pipeline *p, uint32_t i) {
    if (!p || i >= p->n || p->status[i] != JPGPU_OK || p->sub_of[i] < 0) return nullptr;
    const SubBatch &sb = p->subs[(uint32_t)p->sub_of[i]];
    return sb.batch ? &sb : nullptr;
}

// (a pipeline over several devices: image i of the call is image i / n of child i mod n)
#define MULTI(P, I, CALL, NONE)                                   \
    if ((P) && !(P)->children.empty()) {                          \
        if ((I) >= (P)->multi_n) return NONE;                     \
        uint32_t ci = (I);                                        \
        const jpgpu_pipeline *c = child_of((P), ci);              \
        return CALL;                                              \
    }
int jpgpu_pipeline_image_status(const jpgpu_pipeline *p, uint32_t i) {
    MULTI(p, i, jpgpu_pipeline_image_status(c, ci), JPGPU_ERR_FORMAT)
    return (p && i < p->n) ? p->status[i] : JPGPU_ERR_FORMAT;
}
const char *jpgpu_pipeline_image_error(const jpgpu_pipeline *p, uint32_t i) {
    MULTI(p, i, jpgpu_pipeline_image_error(c, ci), "")
    return (p && i < p->n) ? p->errors[i].c_str() : "";
}
int jpgpu_pipeline_image_info(const jpgpu_pipeline *p, uint32_t i, jpgpu_image_info *info) {
    MULTI(p, i, jpgpu_pipeline_image_info(c, ci, info), JPGPU_ERR_FORMAT)
    if (!p || i >= p->n || !info || i >= p->has_frame.size() || !p->has_frame[i]) return JPGPU_ERR_FORMAT;
    *info = p->infos[i];
    return JPGPU_OK;
}
size_t jpgpu_pipeline_pixel_bytes(const jpgpu_pipeline *p, uint32_t i) {
    MULTI(p, i, jpgpu_pipeline_pixel_bytes(c, ci), 0)
    const SubBatch *sb = sub_of(p, i);
    return sb ? jpgpu_batch_out_bytes(sb->batch, (uint32_t)p->slot[i]) : 0;
}
const void *jpgpu_pipeline_pixels_device(const jpgpu_pipeline *p, uint32_t i) {
    if (p && !p->children.empty()) {
        if (i >= p->multi_n) return nullptr;
        uint32_t ci = i;
        const jpgpu_pipeline *c = child_of(p, ci);
        if (!p->gathered) return jpgpu_pipeline_pixels_device(c, ci);
        const SubBatch *sb = sub_of(c, ci);  // the copy on the first device
        if (!sb || !c->d_gather) return nullptr;
        return c->d_gather + c->gather_off[(uint32_t)c->sub_of[ci]] + jpgpu_batch_out_offset(sb->batch, (uint32_t)c->slot[ci]);
    }
    const SubBatch *sb = sub_of(p, i);
    return sb ? (const uint8_t *)jpgpu_batch_out_arena(sb->batch) + jpgpu_batch_out_offset(sb->batch, (uint32_t)p->slot[i]) : nullptr;
}
const uint8_t *jpgpu_pipeline_pixels_host(const jpgpu_pipeline *p, uint32_t i) {
    MULTI(p, i, jpgpu_pipeline_pixels_host(c, ci), nullptr)
    const SubBatch *sb = sub_of(p, i);
    return (sb && sb->h_out && p->downloaded) ? sb->h_out + jpgpu_batch_out_offset(sb->batch, (uint32_t)p->slot[i]) : nullptr;
}
int jpgpu_pipeline_download(jpgpu_pipeline *p, uint32_t i, uint8_t *dst, size_t cap, size_t *len) {
    if (!p) return JPGPU_ERR_FORMAT;
    if (!p->children.empty()) {  // (from the device that decoded the image: the gathered copy is the same bytes)
        if (i >= p->multi_n) return jpgpu::set_err(p->err, JPGPU_ERR_FORMAT, "download: image %u has no pixels", i);
        jpgpu_pipeline *c = p->children[i % p->children.size()];
        const int rc = jpgpu_pipeline_download(c, i / (uint32_t)p->children.size(), dst, cap, len);
        if (rc) p->err = c->err;
        return rc;
    }
    const SubBatch *sb = sub_of(p, i);
    if (!sb) return jpgpu::set_err(p->err, JPGPU_ERR_FORMAT, "download: image %u has no pixels", i);
    const int rc = jpgpu_batch_download(sb->batch, (uint32_t)p->slot[i], dst, cap, len);
    if (rc) p->err = jpgpu_batch_last_error(sb->batch);
    return rc;
}
const char *jpgpu_pipeline_kernel_path(const jpgpu_pipeline *p) { return p ? p->path.c_str() : ""; }
int jpgpu_pipeline_set_max_decoding_buffer_size(jpgpu_pipeline *p, size_t max_bytes) {
    if (!p) return JPGPU_ERR_FORMAT;
    for (jpgpu_pipeline *c : p->children) jpgpu_pipeline_set_max_decoding_buffer_size(c, max_bytes);
    p->max_bytes = max_bytes;
    return JPGPU_OK;
}
int jpgpu_pipeline_set_color_transform(jpgpu_pipeline *p, int color_transform) {
    if (!p) return JPGPU_ERR_FORMAT;
    for (jpgpu_pipeline *c : p->children) jpgpu_pipeline_set_color_transform(c, color_transform);
    p->color_transform = color_transform;
    return JPGPU_OK;
}
int jpgpu_pipeline_set_scale(jpgpu_pipeline *p, uint16_t requested_width, uint16_t requested_height) {
    if (!p) return JPGPU_ERR_FORMAT;
    for (jpgpu_pipeline *c : p->children) jpgpu_pipeline_set_scale(c, requested_width, requested_height);
    p->req_w = requested_width;
    p->req_h = requested_height;
    return JPGPU_OK;
}
uint32_t jpgpu_pipeline_device_count(const jpgpu_pipeline *p) { return !p ? 0u : (p->children.empty() ? 1u : (uint32_t)p->children.size()); }
int jpgpu_pipeline_image_device(const jpgpu_pipeline *p, uint32_t i) {
    if (!p) return -1;
    if (p->children.empty()) return i < p->n ? p->device : -1;
    return i < p->multi_n ? p->children[i % p->children.size()]->device : -1;
}
int jpgpu_pipeline_pixels_device_ordinal(const jpgpu_pipeline *p, uint32_t i) {
    if (!p) return -1;
    if (!p->children.empty() && p->gathered) return i < p->multi_n ? p->device : -1;
    return jpgpu_pipeline_image_device(p, i);
}
// CPUs the calling thread may run on, in order
static std::vector<int> allowed_cpus() {
    std::vector<int> v;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &set)) v.push_back(c);
    return v;
}

// ---- CPU shares by NUMA node ("sub-linear unless feeder threads are pinned per NUMA node") -----------------------------
// A device's feeder threads belong on the socket its PCIe root hangs off: <sysfs>/bus/pci/devices/<bdf>/numa_node names the node,
// <sysfs>/devices/system/node/node<N>/cpulist its CPUs.  Devices of one node split the node's ALLOWED CPUs evenly, in list order.  If
// any device's node is unknown (-1: one socket, a VM) or a node has fewer allowed CPUs than devices, every device falls back to a
// contiguous slice of the allowed list (round 4's rule) — shares are always disjoint.
static std::string read_small_file(const std::string &path) {
    std::string out;
    if (FILE *f = fopen(path.c_str(), "r")) {
        char buf[4096];
        const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        out.assign(buf, n);
    }
    return out;
}
static int numa_node_of(const std::string &sysfs, const char *bdf) {
    if (!bdf || !*bdf) return -1;
    std::string id(bdf);
    for (char &ch : id) ch = (char)tolower((unsigned char)ch);
    const std::string txt = read_small_file(sysfs + "/bus/pci/devices/" + id + "/numa_node");
    if (txt.empty()) return -1;
    char *end = nullptr;
    const long v = strtol(txt.c_str(), &end, 10);
    return end == txt.c_str() ? -1 : (int)v;
}
static std::vector<int> parse_cpulist(const std::string &txt) {  // "0-63,128-191"
    std::vector<int> v;
    const char *q = txt.c_str();
    while (*q) {
        if (*q < '0' || *q > '9') {
            q++;
            continue;
        }
        char *end = nullptr;
        long a = strtol(q, &end, 10), b = a;
        q = end;
        if (*q == '-') {
            b = strtol(q + 1, &end, 10);
            q = end;
        }
        for (long c = a; c <= b && c < 65536 && v.size() < 65536; c++) v.push_back((int)c);
    }
    return v;
}
static std::vector<std::vector<int>> plan_cpu_shares(const std::string &sysfs, const std::vector<std::string> &bdfs, const std::vector<int> &allowed,
                                                     std::vector<int> *nodes_out) {
    const size_t n = bdfs.size();
    std::vector<std::vector<int>> shares(n);
    std::vector<int> nodes(n, -1);
    for (size_t k = 0; k < n; k++) nodes[k] = numa_node_of(sysfs, bdfs[k].c_str());
    if (nodes_out) *nodes_out = nodes;
    if (n == 0 || allowed.size() < n) return shares;  // (fewer CPUs than devices: nobody is pinned)
    bool by_node = true;
    std::vector<std::vector<int>> by(n);
    for (size_t k = 0; k < n && by_node; k++) {
        if (nodes[k] < 0) {
            by_node = false;
            break;
        }
        size_t rank_on_node = 0, on_node = 0;
        for (size_t j = 0; j < n; j++)
            if (nodes[j] == nodes[k]) {
                if (j < k) rank_on_node++;
                on_node++;
            }
        const std::vector<int> node_cpus = parse_cpulist(read_small_file(sysfs + "/devices/system/node/node" + std::to_string(nodes[k]) + "/cpulist"));
        std::vector<int> cand;
        for (int c : allowed)
            if (std::find(node_cpus.begin(), node_cpus.end(), c) != node_cpus.end()) cand.push_back(c);
        if (cand.size() < on_node) {
            by_node = false;
            break;
        }
        const size_t a = cand.size() * rank_on_node / on_node, b = cand.size() * (rank_on_node + 1) / on_node;
        by[k].assign(cand.begin() + (long)a, cand.begin() + (long)b);
    }
    for (size_t k = 0; k < n; k++) {
        if (by_node) {
            shares[k] = by[k];
        } else {
            const size_t a = allowed.size() * k / n, b = allowed.size() * (k + 1) / n;
            shares[k].assign(allowed.begin() + (long)a, allowed.begin() + (long)b);
        }
    }
    return shares;
}
static const char *sysfs_root() {
    const char *e = getenv("JPGPU_SYSFS_ROOT");  // (tests: a fake tree)
    return e && *e ? e : "/sys";
}

int jpgpu_host_alloc(size_t bytes, void **out) {
    if (!out) return JPGPU_ERR_FORMAT;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return JPGPU_ERR_IO;
    }
    return JPGPU_OK;
}
void jpgpu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int jpgpu_device_pci_bus_id(int device, char *buf, size_t cap) {
    if (!buf || cap < 13) return JPGPU_ERR_FORMAT;
    buf[0] = 0;
    if (hipDeviceGetPCIBusId(buf, (int)cap, device) != hipSuccess) {
        (void)hipGetLastError();
        buf[0] = 0;
        return JPGPU_ERR_NO_DEVICE;
    }
    return JPGPU_OK;
}

int jpgpu_plan_cpu_shares(const char *sysfs, const char *const *pci_bus_ids, uint32_t n_devices, const int *allowed, uint32_t n_allowed,
                          int *device_of_cpu, int *numa_nodes) {
    if (!pci_bus_ids || !allowed || !device_of_cpu || n_devices == 0) return JPGPU_ERR_FORMAT;
    std::vector<std::string> bdfs;
    for (uint32_t k = 0; k < n_devices; k++) bdfs.emplace_back(pci_bus_ids[k] ? pci_bus_ids[k] : "");
    const std::vector<int> al(allowed, allowed + n_allowed);
    std::vector<int> nodes;
    const auto shares = plan_cpu_shares(sysfs && *sysfs ? sysfs : sysfs_root(), bdfs, al, &nodes);
    for (uint32_t i = 0; i < n_allowed; i++) device_of_cpu[i] = -1;
    for (uint32_t k = 0; k < n_devices; k++)
        for (int c : shares[k])
            for (uint32_t i = 0; i < n_allowed; i++)
                if (allowed[i] == c) device_of_cpu[i] = (int)k;
    if (numa_nodes)
        for (uint32_t k = 0; k < n_devices; k++) numa_nodes[k] = nodes[k];
    return JPGPU_OK;
}

int jpgpu_pipeline_create_multi(const int *devices, uint32_t n_devices, uint32_t n_threads, uint32_t flags, jpgpu_pipeline **out) {
    if (!out) return JPGPU_ERR_FORMAT;
    jpgpu_pipeline *p = new jpgpu_pipeline();
    *out = p;  // returned even on failure so that last_error can be read
    if (!devices || n_devices == 0 || n_devices > 64) return jpgpu::set_err(p->err, JPGPU_ERR_FORMAT, "create_multi: 1..64 devices, got %u", n_devices);
    if (flags & ~(uint32_t)JPGPU_PIPELINE_MULTI_PIN_CPUS) return jpgpu::set_err(p->err, JPGPU_ERR_FORMAT, "create_multi: unknown flag bits 0x%x", flags & ~(uint32_t)JPGPU_PIPELINE_MULTI_PIN_CPUS);
    p->device = devices[0];
    // Thread budget: `n_threads` host threads IN ALL (0: what one pipeline would take by default — one per physical core, capped by
    // a cgroup CPU quota), dealt evenly; every child keeps at least two.  A child's share covers BOTH its pools (entropy / header
    // workers and the staging team of the device-entropy route: ADVICE r4 — the staging teams came on top of the budget); what is
    // not counted is one uploader thread per child and call, which mostly waits.
    const uint32_t budget = n_threads ? n_threads : default_threads();
    const uint32_t per_child = std::max<uint32_t>(2u, budget / n_devices);
    const uint32_t stage = std::max<uint32_t>(1u, per_child / 3u), workers = std::max<uint32_t>(1u, per_child - stage);
    p->child_cpus.assign(n_devices, std::vector<int>());
    if (flags & JPGPU_PIPELINE_MULTI_PIN_CPUS) {
        std::vector<std::string> bdfs;
        for (uint32_t k = 0; k < n_devices; k++) {
            char id[64] = {0};
            (void)jpgpu_device_pci_bus_id(devices[k], id, sizeof(id));
            bdfs.emplace_back(id);
        }
        p->child_cpus = plan_cpu_shares(sysfs_root(), bdfs, allowed_cpus(), nullptr);
    }
    p->peer_error.assign(n_devices, std::string());
    for (uint32_t k = 0; k < n_devices; k++) {
        jpgpu_pipeline *c = nullptr;
        int rc = JPGPU_OK;
        // (created on a thread of its own that has moved to the child's CPUs first: the pool threads inherit the mask)
        std::thread maker([&] {
            pin_to(p->child_cpus[k]);
            rc = pipeline_create_sized(devices[k], workers, stage, &c);
            // Peer access between this device and the gathering one, both ways (the copy engine of the source writes into the
            // destination's memory; without it the runtime stages through the host).  A refusal is remembered and reported by the
            // first call that asks for a gather — a pipeline without JPGPU_PIPELINE_GATHER does not need it.
            if (rc == JPGPU_OK && devices[k] != devices[0]) {
                int can = 0;
                hipError_t e = hipDeviceCanAccessPeer(&can, devices[k], devices[0]);
                if (e == hipSuccess && can) {
                    e = hipSetDevice(devices[k]) == hipSuccess ? hipDeviceEnablePeerAccess(devices[0], 0) : hipErrorInvalidDevice;
                    if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
                    if (e == hipSuccess) {
                        e = hipSetDevice(devices[0]) == hipSuccess ? hipDeviceEnablePeerAccess(devices[k], 0) : hipErrorInvalidDevice;
                        if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
                    }
                    (void)hipSetDevice(devices[k]);
                }
                (void)hipGetLastError();
                if (e != hipSuccess || !can) {
                    char msg[200];
                    snprintf(msg, sizeof(msg), "device %d cannot reach device %d peer-to-peer (%s)", devices[k], devices[0], e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer says no");
                    p->peer_error[k] = msg;
                }
            }
        });
        maker.join();
        if (c) p->children.push_back(c);
        if (rc) {
            p->err = c ? c->err : "create_multi: out of memory";
            return rc;
        }
    }
    p->t.threads = (workers + stage) * n_devices;
    return JPGPU_OK;
}

int jpgpu_pipeline_last_timings(const jpgpu_pipeline *p, jpgpu_pipeline_timings *t) {
    if (!p || !t) return JPGPU_ERR_FORMAT;
    *t = p->t;
    return JPGPU_OK;
}

}  // extern "C"

// One call over several devices: image i goes to child i mod n (north_star: "a batch of independent images shards one-image-per-GPU");
// the children decode side by side, each on a thread that sits on the child's CPUs; there is no data-path exchange between
// devices.  JPGPU_PIPELINE_GATHER: every child copies each sub-batch's pixel arena to the first device as soon as that sub-batch's
// pixel kernels are done (after_decode: peer-to-peer on a stream of the SOURCE device — each child drives its own xGMI link, and the
// copy of one sub-batch runs while the next ones decode; nothing to reduce — SURVEY 8e); jpgpu_pipeline_pixels_device then points
// into the copies.  gather_ms = what the call still waited for the copies after the last child had decoded (the exposed part),
// gather_copy_ms = the longest child's summed copy time (events around each copy).
static int multi_decode(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n, uint32_t flags) {
    const double t0 = now_ms(), cpu0 = process_cpu_ms();
    const uint32_t nc = (uint32_t)p->children.size();
    const bool gather = (flags & JPGPU_PIPELINE_GATHER) != 0;
    const uint32_t child_flags = flags & ~(uint32_t)JPGPU_PIPELINE_GATHER;
    p->multi_n = n;
    p->gathered = false;
    if (gather)
        for (uint32_t k = 0; k < nc; k++)
            if (!p->peer_error[k].empty()) return jpgpu::set_err(p->err, JPGPU_ERR_IO, "gather: %s", p->peer_error[k].c_str());
    std::vector<std::vector<const uint8_t *>> cd(nc);
    std::vector<std::vector<size_t>> cl(nc);
    for (uint32_t i = 0; i < n; i++) {
        cd[i % nc].push_back(data[i]);
        cl[i % nc].push_back(len[i]);
    }
    std::vector<int> rcs(nc, JPGPU_OK);
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < nc; k++) {
        p->children[k]->gather_on = gather;
        p->children[k]->gather_device = p->device;
        th.emplace_back([&, k] {
            pin_to(p->child_cpus[k]);  // (the call's uploader thread starts from here and inherits the mask)
            rcs[k] = jpgpu_pipeline_decode(p->children[k], cd[k].data(), cl[k].data(), (uint32_t)cd[k].size(), child_flags);
        });
    }
    for (auto &t : th) t.join();
    const double t1 = now_ms();
    p->path.clear();
    p->t = jpgpu_pipeline_timings{};
    for (uint32_t k = 0; k < nc; k++) {
        const jpgpu_pipeline *c = p->children[k];
        if (rcs[k]) {
            p->err = c->err;
            for (uint32_t j = 0; j < nc; j++)  // (never leave copies in flight behind an error)
                if (p->children[j]->gather_stream) (void)hipStreamSynchronize(p->children[j]->gather_stream);
            return rcs[k];
        }
        if (cd[k].empty()) continue;
        if (p->path.empty()) p->path = c->path;
        else if (p->path != c->path) p->path = "mixed";
        p->t.headers_ms = std::max(p->t.headers_ms, c->t.headers_ms);
        p->t.setup_ms = std::max(p->t.setup_ms, c->t.setup_ms);
        p->t.entropy_and_upload_ms = std::max(p->t.entropy_and_upload_ms, c->t.entropy_and_upload_ms);
        p->t.download_ms = std::max(p->t.download_ms, c->t.download_ms);
        p->t.threads += c->t.threads;
        p->t.images_ok += c->t.images_ok;
        p->t.jpeg_bytes += c->t.jpeg_bytes;
        p->t.coefficient_bytes += c->t.coefficient_bytes;
        p->t.pixel_bytes += c->t.pixel_bytes;
        p->t.images_device_entropy += c->t.images_device_entropy;
        p->t.images_device_rejected += c->t.images_device_rejected;
        p->t.images_device_progressive += c->t.images_device_progressive;
        p->t.images_host_light += c->t.images_host_light;
        p->t.images_entry_pixels += c->t.images_entry_pixels;
        p->t.input_pinned |= c->t.input_pinned;
    }
    p->t.decode_ms = t1 - t0;
    if (gather) {
        for (uint32_t k = 0; k < nc; k++) {
            jpgpu_pipeline *c = p->children[k];
            if (!c->gather_stream) continue;
            P_HIP(hipStreamSynchronize(c->gather_stream));
            double copy_ms = 0;
            for (uint32_t j = 0; j < c->n_subs && j < c->gather_ev.size(); j++) {
                float ms = 0.f;
                if (c->subs[j].batch && hipEventElapsedTime(&ms, c->gather_ev[j].first, c->gather_ev[j].second) == hipSuccess) copy_ms += ms;
                else (void)hipGetLastError();
            }
            p->t.gather_copy_ms = std::max(p->t.gather_copy_ms, copy_ms);
            p->t.gather_bytes += c->gather_bytes;
        }
        p->gathered = true;
    }
    p->t.gather_ms = now_ms() - t1;
    p->t.total_ms = now_ms() - t0;
    p->t.cpu_ms = process_cpu_ms() - cpu0;
    return JPGPU_OK;
}
