// host_common.hpp — helpers shared by the C-ABI translation units (internal).
#pragma once
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/jpgpu.h"
#include "jobs.hpp"

namespace jpgpu {

int set_err(std::string &dst, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int use_device(int device, std::string &err);
size_t plane_bytes(const jpgpu_component &c);
int choose_color_fn(uint32_t ncomp, int color_transform, uint32_t &fn, std::string &err);
int build_image_job(const jpgpu_component *comps, uint32_t ncomp, uint8_t *const *d_planes, uint16_t out_w,
                    uint16_t out_h, int color_transform, uint8_t *d_out, ImageJob &job, size_t &out_len,
                    std::string &err);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// roctx range around a host-side phase (the reference has no tracing hooks; rocprofv3 --marker-trace shows these
// next to the kernels).  libroctx64 is looked up at run time (dlopen, once): the library neither links against it nor fails
// to load where ROCm's tracing library is absent or lives elsewhere — the ranges are then no-ops.
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi &roctx_api();  // jpgpu.cpp
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name) : on(roctx_api().push != nullptr) {
        if (on) (void)roctx_api().push(name);
    }
    ~TraceRange() {
        if (on) (void)roctx_api().pop();
    }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

// jpgpu_batch_upload_compact for buffers written by CompactWriter inside this library (no consistency pass)
int batch_upload_compact(jpgpu_batch *b, uint32_t image, uint32_t comp, const void *compact, size_t bytes, int range_class,
                         void *hip_stream, bool trusted);

// `bytes` of device memory -> a pinned host block (its host address), behind what `hip_stream` holds: by a copy kernel that writes the
// mapped block itself (huff.hip, launch_copy_to_host); falls back to the copy engine if the block has no device mapping
int copy_device_to_pinned_host(void *host_pinned, const void *d_src, size_t bytes, void *hip_stream);

// before a worker goes back to the decoder API's pool of idle workers
void worker_recycle(jpgpu_worker *w);


// ---- device entropy decoding of restart-marker streams (huff_core.hpp), used by the pipeline ----------------------
namespace host {
struct PlannedScan;
struct ProgPlan;
}
struct DeviceEntropyImage {
    uint32_t image;                                  // index in the batch
    const uint8_t *file;                             // the stream the plan refers to (host memory)
    const std::vector<host::PlannedScan> *scans;     // Frontend::plan_device_scans
};
// Enqueue on `hip_stream`: upload of the scans' bytes, tables and job records, zero-fill of the images' coefficient
// planes, the segment decoder, the range scan.  Then (after the stream has been synchronised) collect: status[k] != 0
// means image k of the list must be decoded on the host instead; the others have their range classes set.
// `par`: optional parallel-for (count, body) used for the staging copies of the scans' bytes.
// `copy_stream`: optional second stream for the upload (the kernels on `hip_stream` wait for it through an event).
// `scratch`: optional device-only work space of the chunk decoder (per-chunk states, emission buffers: ~16 bytes per byte of
// entropy-coded data) owned by the caller and shared by all launches it enqueues on the SAME stream — they run one after the
// other there; without it every batch keeps its own (32 sub-batches of a 4,096-file call: 27 GB instead of 7).
// `alone`: the launch has the device to itself (a call with one or two sub-batches): what counts is how long its passes take one
// after the other, not how much work they are — the first sync pass walks whole chunks (fewer lanes re-run in the later,
// chain-bound passes: sync passes of 256 files 2.9 -> 2.5 ms) and launches keep two passes each.
struct DeviceScratch {
    uint8_t *d = nullptr;
    size_t cap = 0;
};
// `mode`: DEVICE_ENTROPY_LIGHT — scans without restart markers go up as the file holds them and the device does the staging pass
// (marker check, unstuffing: huff_unstuff_core.hpp); | DEVICE_ENTROPY_INPUT_PINNED — the files lie in page-locked memory: no staging copy
// either, the copy engine reads them.  n_light (optional): how many of the listed images took that route.
// | DEVICE_ENTROPY_ENTRY_PIXELS — 4:2:0 images whose scan qualifies keep their entry lists and the next jpgpu_batch_decode on the same
// stream runs the walk that reads them (fused_entries.hpp): nothing of such an image goes through the coefficient arena.
constexpr uint32_t DEVICE_ENTROPY_LIGHT = 1u, DEVICE_ENTROPY_INPUT_PINNED = 2u, DEVICE_ENTROPY_ENTRY_PIXELS = 4u;
int batch_device_entropy_launch(jpgpu_batch *b, const DeviceEntropyImage *images, uint32_t n, void *hip_stream,
                                const std::function<void(uint32_t, const std::function<void(uint32_t)> &)> *par = nullptr,
                                void *copy_stream = nullptr, DeviceScratch *scratch = nullptr, bool alone = false, uint32_t mode = 0u,
                                uint32_t *n_light = nullptr, uint32_t *n_entry = nullptr);  // n_entry: images on the entry-list pixel path
int batch_device_entropy_collect(jpgpu_batch *b, uint32_t *status, uint32_t n);
// The same for PROGRESSIVE frames (huff_prog_wave.hpp): the scans of every listed image are staged and uploaded, the
// images' planes and non-zero / sign masks zero-filled, ONE launch walks all tracks (a lane per track of dependent scans; the
// coefficients are accumulated in the arena in place), then the range scan classifies the finished planes on the device.  Every image of
// the batch must be listed (the range scan covers the whole arena).  Collect with batch_device_entropy_collect.
// kernel_ms (optional, after the stream has been synchronised): batch_progressive_kernel_ms.
struct DeviceProgressiveImage {
    uint32_t image;
    const uint8_t *file;
    const host::ProgPlan *plan;  // Frontend::plan_progressive_scans
};
// `pipelined`: a lane per SCAN, scans of a band staying behind one another block by block (huff_prog_job.hpp) — only for launches whose
// lanes, together with those of the launches that run beside them, FIT the device (48 k lanes): waiting lanes hold their slots, and two
// oversubscribed launches could keep each other's producers from ever being dispatched (a lane gives up after seconds and flags its
// image, so the answer stays right — but slow).  false: a lane per track, nobody waits for anybody.
int batch_device_progressive_launch(jpgpu_batch *b, const DeviceProgressiveImage *images, uint32_t n, void *hip_stream,
                                    const std::function<void(uint32_t, const std::function<void(uint32_t)> &)> *par = nullptr,
                                    void *copy_stream = nullptr, DeviceScratch *scratch = nullptr, bool pipelined = true);
bool batch_progressive_kernel_ms(jpgpu_batch *b, float *ms);  // duration of the last launch's track kernel (events; stream synchronised)
bool batch_phase_times(jpgpu_batch *b, float ms[4]);  // JPGPU_BATCH_KERNEL_TIMES, batch.cpp
bool batch_phase_stamps(jpgpu_batch *ref, jpgpu_batch *b, float ms[6]);  // (+ JPGPU_PIPE_TRACE) event times relative to ref's first event

}  // namespace jpgpu
