// batch.cpp — batch driver of the C ABI (include/jpgpu.h, jpgpu_batch_*).
//
// A batch is N independent images (the unit the reference decodes one-per-Decoder,
// src/decoder.rs:134-154) laid out in two HBM arenas: all coefficient planes (int16,
// block-raster = the concatenation of each component's append_row buffers)
// and all output pixels.  One decode = a handful of launches over the whole batch.
// Images of a fusable kind (4:2:0 YCbCr, 4:4:4 YCbCr / RGB, gray; any size) are grouped per kind and run the fused
// kernels (fused.hip), one launch group per kind; everything else runs the generic two-kernel path (kernels.hip)
// through device job tables.  jpgpu_batch_path names the kernels: "fused420", ..., "generic", or "mixed".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include <mutex>

#include "compact.hpp"
#include "host/frontend.hpp"
#include "huff.hpp"
#include "fused.hpp"
#include "fused_entries.hpp"
#include "fused_scaled.hpp"
#include "host_common.hpp"
#include "kernels.hpp"
#include "range_stats.hpp"

using namespace jpgpu;

struct jpgpu_batch {
    int device = 0;
    uint32_t flags = 0;
    std::string err;
    std::string path = "generic";
    std::vector<jpgpu_image_desc> descs;
    // arena layout
    std::vector<size_t> coef_off;   // [image*4 + comp]
    std::vector<size_t> coef_len;   // bytes
    std::vector<size_t> plane_off;  // [image*4 + comp] (generic path scratch)
    std::vector<size_t> out_off, out_len;
    size_t coef_bytes = 0, out_bytes = 0, plane_bytes_total = 0;
    uint8_t *d_coef = nullptr, *d_out = nullptr;
    bool own_coef = false, own_out = false;
    uint8_t *d_planes = nullptr;
    uint16_t *d_qt = nullptr;
    PlaneJob *d_plane_jobs = nullptr;
    ImageJob *d_image_jobs = nullptr;
    std::vector<PlaneJob> plane_jobs;
    std::vector<ImageJob> image_jobs;
    std::vector<uint8_t> sane;  // per image*4+comp: 1 if every |c*q| < 2^15 (24-bit path exact)
    uint32_t max_blocks = 0, max_w = 0, max_h = 0;
    bool scales[9] = {false, false, false, false, false, false, false, false, false};
    bool jobs_dirty = true;
    bool qt_dirty = false;
    std::vector<FusedPlan> fused;       // one per fusable kind present in the batch
    std::vector<uint32_t> generic_ids;  // images on the generic path
    // Reduced-size decodes in one launch (fused_scaled.hpp): images whose components all sit at one dct_scale < 8 — their own job
    // tables (PlaneJobs carry the coefficient / table pointers, ImageJobs the upsampler kinds and the output), no u8 planes in HBM
    std::vector<uint32_t> scaled_ids;
    std::vector<ScaledGeom> scaled_geoms;
    std::vector<PlaneJob> s_plane_jobs;
    std::vector<ImageJob> s_image_jobs;
    ScaledGeom *d_scaled_geoms = nullptr;
    PlaneJob *d_s_plane_jobs = nullptr;
    ImageJob *d_s_image_jobs = nullptr;
    uint32_t s_max_tiles_x = 0, s_max_bands = 0, s_lds_bytes = 0;
    bool s_scales[9] = {false, false, false, false, false, false, false, false, false};
    std::string scaled_name;            // path name of the scaled launch group ("fused420-s4", ...; "fusedscaled-mixed")
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // compact transport (compact.hpp): staging area in HBM, allocated at the first jpgpu_batch_upload_compact
    std::mutex compact_mutex;
    uint8_t *d_compact = nullptr;
    std::vector<size_t> compact_off;      // [image*4 + comp]
    std::vector<uint8_t> compact_pending; // [image*4 + comp]: uploaded, to be expanded by the next decode
    ExpandJob *d_expand_jobs = nullptr;
    bool any_compact_pending = false;
    // device entropy decoding (huff.hip): one pinned + one device staging block, grown on demand
    uint8_t *h_entropy = nullptr, *d_entropy = nullptr;
    size_t entropy_cap = 0, entropy_host_cap = 0;
    uint32_t *h_entropy_out = nullptr;  // pinned read-back: status per listed image, then 2 range stats per (image, comp)
    size_t entropy_out_cap = 0;
    hipEvent_t entropy_uploaded = nullptr, entropy_filled = nullptr;
    uint8_t *d_scan = nullptr;     // jpgpu_batch_scan_ranges: stats + job table on the device, kept between calls
    uint32_t *h_scan = nullptr;    // pinned read-back of the stats
    size_t scan_cap = 0;
    bool scan_jobs_valid = false;  // the job table on the device matches the bound arena and the current q-tables
    uint8_t *h_bounce = nullptr;  // pinned: jpgpu_batch_download into pageable memory
    size_t h_bounce_cap = 0;
    std::vector<uint32_t> entropy_images;  // images of the launch in flight
    size_t entropy_out_off = 0;            // offset of the status / stats words inside d_entropy
    // Classes decided ON THE DEVICE (range_stats.hpp): statistics raised by the kernels that write the coefficients, turned
    // into class bits by class_finalize_* in front of the pixel kernels.  cls_src[image * 4 + comp] = 1: that component's
    // class comes from the image's statistics; 0: from `sane` (what the host knows).  dev_classes: some component does, so
    // decodes run the finalize kernels and the `_dyn` pixel kernels instead of one launch per class.
    uint32_t *d_stats = nullptr;        // RS_WORDS per image
    uint8_t *d_host_cls = nullptr;      // per image * 4 + comp: 0 / 1 / 3 or CLS_FROM_DEVICE
    static constexpr int kClsRing = 4;
    uint8_t *h_host_cls = nullptr;      // pinned, kClsRing copies (the upload is asynchronous on the decode stream)
    hipEvent_t cls_sent[kClsRing] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t cls_next = 0;
    uint32_t *d_plane_job_slot = nullptr;  // generic path: plane job -> image * 4 + comp
    std::vector<uint8_t> cls_src;
    bool dev_classes = false;
    bool cls_dirty = true;              // class knowledge changed since the tables / the class table were last sent
    // JPGPU_BATCH_KERNEL_TIMES (diagnostics, jpgpu_pipeline_timings): events around the phases of the device entropy path
    hipEvent_t ev_phase[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool phase_events_valid = false;
    bool progressive_launch = false;  // the last device entropy launch was batch_device_progressive_launch
    // Entry-list pixel path (fused_entries.hpp): images whose last device entropy launch kept their scan as entry lists.  entry_img[image]
    // = 1 until the host uploads coefficients for the image (a re-decode): the dense kernels skip it (CLS_SKIP), the decode that follows
    // the launch on its stream runs s420_entries_kernel over the plan(s) and copies the status words once more behind it.
    std::vector<uint8_t> entry_img;
    bool entries_pending = false;
    const EntrySrc *d_entry_srcs = nullptr;   // per batch image, inside d_entropy
    const uint32_t *d_entry_status = nullptr; // the launch's status words (device) and how many
    uint32_t entry_status_n = 0;
};

#define B_HIP(call)                                                                                     \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) return set_err(b->err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

// first use of the device-side classes: statistics (zeroed), class table, pinned staging
static int batch_enable_dev_classes(jpgpu_batch *b) {
    if (b->d_stats) return JPGPU_OK;
    const size_t n = b->descs.size();
    B_HIP(hipMalloc((void **)&b->d_stats, n * RS_WORDS * sizeof(uint32_t)));
    B_HIP(hipMemset(b->d_stats, 0, n * RS_WORDS * sizeof(uint32_t)));
    B_HIP(hipMalloc((void **)&b->d_host_cls, n * 4));
    B_HIP(hipHostMalloc((void **)&b->h_host_cls, n * 4 * jpgpu_batch::kClsRing, hipHostMallocDefault));
    for (auto &e : b->cls_sent) B_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!b->generic_ids.empty()) B_HIP(hipMalloc((void **)&b->d_plane_job_slot, n * 4 * sizeof(uint32_t)));
    b->jobs_dirty = true;  // (the plane-job slot table goes up with the jobs)
    return JPGPU_OK;
}

// mark component `idx` = image * 4 + comp as classified by the device statistics / by the host (`sane[idx]`)
static void batch_class_source(jpgpu_batch *b, size_t idx, bool from_device) {
    if (b->cls_src[idx] != (from_device ? 1 : 0)) {
        b->cls_src[idx] = from_device ? 1 : 0;
        b->cls_dirty = true;
    }
    if (from_device) b->dev_classes = true;
}
static void batch_set_host_class(jpgpu_batch *b, size_t idx, uint8_t cls) {
    if (idx / 4 < b->entry_img.size() && b->entry_img[idx / 4]) {  // coefficients from the host: the image is a dense one again
        b->entry_img[idx / 4] = 0;
        b->cls_dirty = true;
    }
    if (b->sane[idx] != cls) {
        b->sane[idx] = cls;
        b->cls_dirty = true;
    }
    batch_class_source(b, idx, false);
}

static int batch_refresh_jobs(jpgpu_batch *b, hipStream_t stream = nullptr) {
    // host-side classes are part of the launch tables (one launch per class: fused_bind); with device-side classes they
    // travel in a small table of their own, asynchronously, and the launch tables stay as they are
    const bool need_bind = b->jobs_dirty || (b->cls_dirty && !b->dev_classes);
    if (!need_bind && !b->cls_dirty) return JPGPU_OK;
    if (!b->d_coef || !b->d_out) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    if (b->dev_classes) {
        const size_t n4 = b->descs.size() * 4;
        const uint32_t k = b->cls_next++ % jpgpu_batch::kClsRing;
        uint8_t *h = b->h_host_cls + (size_t)k * n4;
        B_HIP(hipEventSynchronize(b->cls_sent[k]));  // (its previous copy, four refreshes ago: long gone)
        for (size_t i = 0; i < n4; i++) h[i] = (i / 4 < b->entry_img.size() && b->entry_img[i / 4]) ? CLS_SKIP : (b->cls_src[i] ? CLS_FROM_DEVICE : b->sane[i]);
        B_HIP(hipMemcpyAsync(b->d_host_cls, h, n4, hipMemcpyHostToDevice, stream));
        B_HIP(hipEventRecord(b->cls_sent[k], stream));
    }
    b->cls_dirty = false;
    if (!need_bind) return JPGPU_OK;
    const uint32_t n = (uint32_t)b->descs.size();
    b->plane_jobs.clear();
    b->image_jobs.clear();
    for (uint32_t i : b->generic_ids) {
        const jpgpu_image_desc &d = b->descs[i];
        uint8_t *planes[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t c = 0; c < d.ncomp; c++) {
            const jpgpu_component &cc = d.components[c];
            PlaneJob j{};
            j.coefs = reinterpret_cast<const int16_t *>(b->d_coef + b->coef_off[i * 4 + c]);
            j.plane = b->d_planes ? b->d_planes + b->plane_off[i * 4 + c] : nullptr;
            j.qt = b->d_qt + ((size_t)i * 4 + c) * 64;
            j.block_w = cc.block_width;
            j.n_blocks = (uint32_t)cc.block_width * cc.block_height;
            j.scale = cc.dct_scale;
            j.flags = b->sane[i * 4 + c];
            planes[c] = j.plane;
            b->plane_jobs.push_back(j);
        }
        ImageJob ij;
        size_t out_len = 0;
        int rc = build_image_job(d.components, d.ncomp, planes, d.out_w, d.out_h, d.color_transform,
                                 b->d_out + b->out_off[i], ij, out_len, b->err);
        if (rc) return rc;
        b->image_jobs.push_back(ij);
    }
    if (!b->plane_jobs.empty())
        B_HIP(hipMemcpy(b->d_plane_jobs, b->plane_jobs.data(), b->plane_jobs.size() * sizeof(PlaneJob), hipMemcpyHostToDevice));
    if (!b->plane_jobs.empty() && b->d_plane_job_slot) {
        std::vector<uint32_t> slots;
        for (uint32_t i : b->generic_ids)
            for (uint32_t c = 0; c < b->descs[i].ncomp; c++) slots.push_back(i * 4 + c);
        B_HIP(hipMemcpy(b->d_plane_job_slot, slots.data(), slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!b->image_jobs.empty())
        B_HIP(hipMemcpy(b->d_image_jobs, b->image_jobs.data(), b->image_jobs.size() * sizeof(ImageJob), hipMemcpyHostToDevice));
    b->s_plane_jobs.clear();
    b->s_image_jobs.clear();
    for (uint32_t i : b->scaled_ids) {  // (planes: none — the kernel keeps them in LDS; the reduced IDCTs are exact at any class)
        const jpgpu_image_desc &d = b->descs[i];
        uint8_t *no_planes[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t c = 0; c < d.ncomp; c++) {
            PlaneJob j{};
            j.coefs = reinterpret_cast<const int16_t *>(b->d_coef + b->coef_off[i * 4 + c]);
            j.qt = b->d_qt + ((size_t)i * 4 + c) * 64;
            j.block_w = d.components[c].block_width;
            j.n_blocks = (uint32_t)d.components[c].block_width * d.components[c].block_height;
            j.scale = d.components[c].dct_scale;
            b->s_plane_jobs.push_back(j);
        }
        ImageJob ij;
        size_t out_len = 0;
        int rc = build_image_job(d.components, d.ncomp, no_planes, d.out_w, d.out_h, d.color_transform, b->d_out + b->out_off[i], ij, out_len, b->err);
        if (rc) return rc;
        b->s_image_jobs.push_back(ij);
    }
    if (!b->s_plane_jobs.empty()) {
        B_HIP(hipMemcpy(b->d_s_plane_jobs, b->s_plane_jobs.data(), b->s_plane_jobs.size() * sizeof(PlaneJob), hipMemcpyHostToDevice));
        B_HIP(hipMemcpy(b->d_s_image_jobs, b->s_image_jobs.data(), b->s_image_jobs.size() * sizeof(ImageJob), hipMemcpyHostToDevice));
    }
    for (FusedPlan &fp : b->fused) {
        int rc = fused_bind(fp, b->d_coef, b->d_out, b->d_qt, b->coef_off, b->out_off, b->sane, b->err);
        if (rc) return rc;
    }
    if (b->qt_dirty) {
        std::vector<uint16_t> qt((size_t)n * 4 * 64, 1);
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t c = 0; c < b->descs[i].ncomp; c++)
                memcpy(&qt[((size_t)i * 4 + c) * 64], b->descs[i].quantization_tables[c], 128);
        B_HIP(hipMemcpy(b->d_qt, qt.data(), qt.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        b->qt_dirty = false;
    }
    b->jobs_dirty = false;
    return JPGPU_OK;
}

extern "C" {

int jpgpu_batch_create(int device, const jpgpu_image_desc *descs, uint32_t n_images, uint32_t flags,
                       jpgpu_batch **out) {
    if (!out) return JPGPU_ERR_FORMAT;
    *out = nullptr;
    if (!descs || n_images == 0 || n_images > 65535) return JPGPU_ERR_FORMAT;
    jpgpu_batch *b = new jpgpu_batch();
    *out = b;  // returned even on failure so the caller can read last_error, then destroy
    b->device = device;
    b->flags = flags;
    int rc = use_device(device, b->err);
    if (rc) return rc;
    b->descs.assign(descs, descs + n_images);
    b->coef_off.assign((size_t)n_images * 4, 0);
    b->coef_len.assign((size_t)n_images * 4, 0);
    b->plane_off.assign((size_t)n_images * 4, 0);
    b->out_off.assign(n_images, 0);
    b->out_len.assign(n_images, 0);
    b->sane.assign((size_t)n_images * 4, 0);
    b->cls_src.assign((size_t)n_images * 4, 0);
    // path resolution: group the images that can share a fused launch, the rest is generic
    std::vector<uint32_t> kind_key(n_images, 0);
    if (!(flags & JPGPU_BATCH_FORCE_GENERIC))
        for (uint32_t i = 0; i < n_images; i++)
            if (b->descs[i].ncomp >= 1 && b->descs[i].ncomp <= 4) kind_key[i] = fused_kind_key(b->descs[i]);
    size_t co = 0, po = 0, oo = 0;
    for (uint32_t i = 0; i < n_images; i++) {
        const jpgpu_image_desc &d = b->descs[i];
        if (d.ncomp == 0 || d.ncomp > 4) return set_err(b->err, JPGPU_ERR_FORMAT, "image %u: bad component count %u", i, d.ncomp);
        // validate once with dummy plane pointers (same checks as compute_image)
        uint8_t *dummy[4] = {nullptr, nullptr, nullptr, nullptr};
        ImageJob ij;
        size_t out_len = 0;
        rc = build_image_job(d.components, d.ncomp, dummy, d.out_w, d.out_h, d.color_transform, nullptr, ij, out_len, b->err);
        if (rc) return rc;
        // reduced-size decodes (every component at one dct_scale < 8): one launch, planes in LDS (fused_scaled.hpp)
        ScaledGeom sg;
        static const uint32_t scaled_tx = getenv("JPGPU_SCALED_TX") ? (uint32_t)std::max(8, atoi(getenv("JPGPU_SCALED_TX"))) : 64u;  // (tuning / test knob)
        static const uint32_t scaled_ry = getenv("JPGPU_SCALED_RY") ? (uint32_t)std::max(1, atoi(getenv("JPGPU_SCALED_RY"))) : 8u;
        const bool scaled = kind_key[i] == 0 && !(flags & JPGPU_BATCH_FORCE_GENERIC) && scaled_geom_from_job(d.components, d.ncomp, ij, sg, scaled_tx, scaled_ry);
        if (scaled) {
            b->scaled_ids.push_back(i);
            b->scaled_geoms.push_back(sg);
            b->s_max_tiles_x = std::max(b->s_max_tiles_x, sg.tiles_x);
            b->s_max_bands = std::max(b->s_max_bands, sg.bands);
            b->s_lds_bytes = std::max(b->s_lds_bytes, sg.lds_bytes);
            b->s_scales[sg.scale] = true;
            const char *nm = scaled_path_name(sg);
            if (b->scaled_name.empty()) b->scaled_name = nm;
            else if (b->scaled_name != nm) b->scaled_name = "fusedscaled-mixed";
        }
        for (uint32_t c = 0; c < d.ncomp; c++) {
            const jpgpu_component &cc = d.components[c];
            size_t cb = (size_t)cc.block_width * cc.block_height * 64 * sizeof(int16_t);
            b->coef_off[i * 4 + c] = co;
            b->coef_len[i * 4 + c] = cb;
            co += align_up(cb, 256);
            if (kind_key[i] == 0 && !scaled) {  // generic path: intermediate u8 plane, launch extents
                b->plane_off[i * 4 + c] = po;
                po += align_up(plane_bytes(cc), 256);
                b->max_blocks = std::max<uint32_t>(b->max_blocks, (uint32_t)cc.block_width * cc.block_height);
                b->scales[cc.dct_scale] = true;
            }
        }
        b->out_off[i] = oo;
        b->out_len[i] = out_len;
        oo += align_up(out_len, 256);
        if (kind_key[i] == 0 && !scaled) {
            b->generic_ids.push_back(i);
            b->max_w = std::max<uint32_t>(b->max_w, d.ncomp == 1 ? d.components[0].size_width : d.out_w);
            b->max_h = std::max<uint32_t>(b->max_h, d.ncomp == 1 ? d.components[0].size_height : d.out_h);
        }
    }
    b->coef_bytes = std::max<size_t>(co, 256);
    b->out_bytes = std::max<size_t>(oo, 256);
    b->plane_bytes_total = std::max<size_t>(po, 256);

    {
        std::vector<uint32_t> keys;
        for (uint32_t i = 0; i < n_images; i++)
            if (kind_key[i] && std::find(keys.begin(), keys.end(), kind_key[i]) == keys.end()) keys.push_back(kind_key[i]);
        for (uint32_t key : keys) {
            std::vector<jpgpu_image_desc> sub;
            std::vector<uint32_t> ids;
            for (uint32_t i = 0; i < n_images; i++)
                if (kind_key[i] == key) {
                    sub.push_back(b->descs[i]);
                    ids.push_back(i);
                }
            FusedPlan fp;
            std::string why;
            if (!fused_plan(sub, ids, fp, why)) return set_err(b->err, JPGPU_ERR_INTERNAL, "fused plan: %s", why.c_str());
            b->fused.push_back(std::move(fp));
        }
        if (b->fused.empty()) b->path = "generic";
        else if (b->fused.size() == 1 && b->generic_ids.empty()) b->path = b->fused[0].name;
        else b->path = "mixed";
    }
    if (!b->scaled_ids.empty()) b->path = (b->fused.empty() && b->generic_ids.empty()) ? b->scaled_name : "mixed";
    hipError_t e;
#define C_HIP(call)                                                                        \
    if ((e = (call)) != hipSuccess) return set_err(b->err, JPGPU_ERR_IO, "%s: %s", #call, hipGetErrorString(e))
    if (!(flags & JPGPU_BATCH_EXTERNAL_BUFFERS)) {
        C_HIP(hipMalloc((void **)&b->d_coef, b->coef_bytes));
        b->own_coef = true;
        C_HIP(hipMalloc((void **)&b->d_out, b->out_bytes));
        b->own_out = true;
    }
    if (!b->generic_ids.empty()) C_HIP(hipMalloc((void **)&b->d_planes, b->plane_bytes_total));
    for (FusedPlan &fp : b->fused) {
        rc = fused_alloc(fp, b->err);
        if (rc) return rc;
    }
    C_HIP(hipMalloc((void **)&b->d_qt, (size_t)n_images * 4 * 64 * sizeof(uint16_t)));
    {
        std::vector<uint16_t> qt((size_t)n_images * 4 * 64, 1);
        for (uint32_t i = 0; i < n_images; i++)
            for (uint32_t c = 0; c < b->descs[i].ncomp; c++)
                memcpy(&qt[((size_t)i * 4 + c) * 64], b->descs[i].quantization_tables[c], 128);
        C_HIP(hipMemcpy(b->d_qt, qt.data(), qt.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    C_HIP(hipMalloc((void **)&b->d_plane_jobs, (size_t)n_images * 4 * sizeof(PlaneJob)));
    C_HIP(hipMalloc((void **)&b->d_image_jobs, (size_t)n_images * sizeof(ImageJob)));
    if (!b->scaled_ids.empty()) {
        const size_t ns = b->scaled_ids.size();
        uint32_t pj = 0;
        for (size_t k = 0; k < ns; k++) {
            b->scaled_geoms[k].first_plane_job = pj;
            pj += b->descs[b->scaled_ids[k]].ncomp;
        }
        C_HIP(hipMalloc((void **)&b->d_scaled_geoms, ns * sizeof(ScaledGeom)));
        C_HIP(hipMemcpy(b->d_scaled_geoms, b->scaled_geoms.data(), ns * sizeof(ScaledGeom), hipMemcpyHostToDevice));
        C_HIP(hipMalloc((void **)&b->d_s_plane_jobs, ns * 4 * sizeof(PlaneJob)));
        C_HIP(hipMalloc((void **)&b->d_s_image_jobs, ns * sizeof(ImageJob)));
    }
    C_HIP(hipEventCreate(&b->ev0));
    C_HIP(hipEventCreate(&b->ev1));
#undef C_HIP
    return JPGPU_OK;
}

void jpgpu_batch_destroy(jpgpu_batch *b) {
    if (!b) return;
    std::string err;
    if (use_device(b->device, err) == JPGPU_OK) {
        hipDeviceSynchronize();
        if (b->own_coef && b->d_coef) hipFree(b->d_coef);
        if (b->own_out && b->d_out) hipFree(b->d_out);
        if (b->d_planes) hipFree(b->d_planes);
        if (b->d_qt) hipFree(b->d_qt);
        if (b->d_compact) hipFree(b->d_compact);
        if (b->d_expand_jobs) hipFree(b->d_expand_jobs);
        if (b->d_entropy) hipFree(b->d_entropy);
        if (b->h_entropy) hipHostFree(b->h_entropy);
        if (b->h_entropy_out) hipHostFree(b->h_entropy_out);
        if (b->entropy_uploaded) hipEventDestroy(b->entropy_uploaded);
        if (b->entropy_filled) hipEventDestroy(b->entropy_filled);
        if (b->h_bounce) hipHostFree(b->h_bounce);
        if (b->d_scan) hipFree(b->d_scan);
        if (b->h_scan) hipHostFree(b->h_scan);
        if (b->d_stats) hipFree(b->d_stats);
        if (b->d_host_cls) hipFree(b->d_host_cls);
        if (b->h_host_cls) hipHostFree(b->h_host_cls);
        if (b->d_plane_job_slot) hipFree(b->d_plane_job_slot);
        for (auto &e : b->cls_sent)
            if (e) hipEventDestroy(e);
        for (auto &e : b->ev_phase)
            if (e) hipEventDestroy(e);
        if (b->d_plane_jobs) hipFree(b->d_plane_jobs);
        if (b->d_image_jobs) hipFree(b->d_image_jobs);
        if (b->d_scaled_geoms) hipFree(b->d_scaled_geoms);
        if (b->d_s_plane_jobs) hipFree(b->d_s_plane_jobs);
        if (b->d_s_image_jobs) hipFree(b->d_s_image_jobs);
        for (FusedPlan &fp : b->fused) fused_free(fp);
        if (b->ev0) hipEventDestroy(b->ev0);
        if (b->ev1) hipEventDestroy(b->ev1);
    }
    delete b;
}

const char *jpgpu_batch_last_error(const jpgpu_batch *b) { return b ? b->err.c_str() : ""; }
const char *jpgpu_batch_path(const jpgpu_batch *b) { return b ? b->path.c_str() : ""; }
int jpgpu_batch_class_counts(jpgpu_batch *b, uint32_t counts[3]) {
    if (!b || !counts) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    rc = batch_refresh_jobs(b);
    if (rc) return rc;
    counts[0] = counts[1] = counts[2] = 0;
    if (b->dev_classes) {  // the classes live on the device: have them worked out there and read the image tables back
        B_HIP(hipDeviceSynchronize());
        for (FusedPlan &fp : b->fused) {
            B_HIP(fused_finalize_classes(fp, nullptr, b->d_stats, b->d_host_cls));
            std::vector<uint8_t> bits;
            rc = fused_read_classes(fp, bits, b->err);
            if (rc) return rc;
            for (uint8_t f : bits) counts[(f & 2u) ? 2 : ((f & 1u) ? 1 : 0)]++;
        }
        return JPGPU_OK;
    }
    for (const FusedPlan &fp : b->fused)
        for (int c = 0; c < 3; c++) counts[c] += fp.class_images[c];
    return JPGPU_OK;
}
size_t jpgpu_batch_coef_arena_bytes(const jpgpu_batch *b) { return b ? b->coef_bytes : 0; }
size_t jpgpu_batch_out_arena_bytes(const jpgpu_batch *b) { return b ? b->out_bytes : 0; }
size_t jpgpu_batch_coef_offset(const jpgpu_batch *b, uint32_t image, uint32_t comp) {
    return (b && image < b->descs.size() && comp < 4) ? b->coef_off[image * 4 + comp] : 0;
}
size_t jpgpu_batch_coef_bytes(const jpgpu_batch *b, uint32_t image, uint32_t comp) {
    return (b && image < b->descs.size() && comp < 4) ? b->coef_len[image * 4 + comp] : 0;
}
size_t jpgpu_batch_out_offset(const jpgpu_batch *b, uint32_t image) {
    return (b && image < b->descs.size()) ? b->out_off[image] : 0;
}
size_t jpgpu_batch_out_bytes(const jpgpu_batch *b, uint32_t image) {
    return (b && image < b->descs.size()) ? b->out_len[image] : 0;
}
void *jpgpu_batch_coef_arena(const jpgpu_batch *b) { return b ? b->d_coef : nullptr; }
void *jpgpu_batch_out_arena(const jpgpu_batch *b) { return b ? b->d_out : nullptr; }

int jpgpu_batch_bind(jpgpu_batch *b, void *device_coef_arena, void *device_out_arena) {
    if (!b) return JPGPU_ERR_FORMAT;
    if (!(b->flags & JPGPU_BATCH_EXTERNAL_BUFFERS)) return set_err(b->err, JPGPU_ERR_FORMAT, "batch owns its buffers");
    if (!device_coef_arena || !device_out_arena || ((uintptr_t)device_coef_arena & 255) || ((uintptr_t)device_out_arena & 255))
        return set_err(b->err, JPGPU_ERR_FORMAT, "bind: arenas must be non-null and 256-byte aligned");
    b->d_coef = (uint8_t *)device_coef_arena;
    b->d_out = (uint8_t *)device_out_arena;
    b->jobs_dirty = true;
    b->scan_jobs_valid = false;
    return JPGPU_OK;
}

int jpgpu_batch_set_range_hint(jpgpu_batch *b, uint32_t image, int sane) {
    if (!b || image >= b->descs.size()) return JPGPU_ERR_FORMAT;
    for (uint32_t c = 0; c < 4; c++) batch_set_host_class(b, (size_t)image * 4 + c, (uint8_t)(sane & 3));
    return JPGPU_OK;
}

int jpgpu_batch_set_range_class(jpgpu_batch *b, uint32_t image, uint32_t comp, int range_class) {
    if (!b || image >= b->descs.size() || comp >= 4) return JPGPU_ERR_FORMAT;
    batch_set_host_class(b, (size_t)image * 4 + comp, (uint8_t)(range_class & 3));
    return JPGPU_OK;
}

// the range-scan job table on the device (d_scan: [ stats of the blocking scan | RangeJob[] ]); slot = image * 4 + comp
static int batch_scan_jobs(jpgpu_batch *b, uint32_t &n_jobs, uint32_t &max_blocks, size_t &jobs_off) {
    const size_t n_jobs_max = b->descs.size() * 4;
    const size_t stats_bytes = b->descs.size() * 4 * RS_WORDS * sizeof(uint32_t);
    jobs_off = align_up(stats_bytes, 256);
    if (!b->d_scan) {
        B_HIP(hipMalloc((void **)&b->d_scan, jobs_off + 2 * n_jobs_max * sizeof(RangeJob)));
        B_HIP(hipHostMalloc((void **)&b->h_scan, stats_bytes, hipHostMallocDefault));
        b->scan_jobs_valid = false;
    }
    max_blocks = 0, n_jobs = 0;
    for (size_t i = 0; i < b->descs.size(); i++)
        for (uint32_t c = 0; c < b->descs[i].ncomp; c++) {
            max_blocks = std::max(max_blocks, (uint32_t)(b->coef_len[i * 4 + c] / 128));
            n_jobs++;
        }
    if (!b->scan_jobs_valid) {
        // two tables back to back: slots per component (the blocking scan's per-component classes) and per image (the
        // device-side statistics are kept per image: range_stats.hpp)
        std::vector<RangeJob> jobs;
        for (int per_image = 0; per_image < 2; per_image++)
            for (size_t i = 0; i < b->descs.size(); i++)
                for (uint32_t c = 0; c < b->descs[i].ncomp; c++) {
                    RangeJob r;
                    r.coefs = reinterpret_cast<const int16_t *>(b->d_coef + b->coef_off[i * 4 + c]);
                    r.n_blocks = (uint32_t)(b->coef_len[i * 4 + c] / 128);
                    r.slot = per_image ? (uint32_t)i : (uint32_t)(i * 4 + c);
                    memcpy(r.q, b->descs[i].quantization_tables[c], 128);
                    jobs.push_back(r);
                }
        B_HIP(hipMemcpy(b->d_scan + jobs_off, jobs.data(), jobs.size() * sizeof(RangeJob), hipMemcpyHostToDevice));
        b->scan_jobs_valid = true;
    }
    return JPGPU_OK;
}

int jpgpu_batch_scan_ranges(jpgpu_batch *b, void *hip_stream, uint8_t *classes) {
    if (!b) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    hipStream_t s = (hipStream_t)hip_stream;
    // stats and job table live on the device between calls (a call per decode must not allocate: bench.py times it)
    uint32_t max_blocks = 0, n_jobs = 0;
    size_t jobs_off = 0;
    rc = batch_scan_jobs(b, n_jobs, max_blocks, jobs_off);
    if (rc) return rc;
    const size_t stats_bytes = b->descs.size() * 4 * RS_WORDS * sizeof(uint32_t);
    uint32_t *stats = b->h_scan;
    hipError_t e = hipMemsetAsync(b->d_scan, 0, stats_bytes, s);
    if (e == hipSuccess)
        e = launch_range_scan(reinterpret_cast<const RangeJob *>(b->d_scan + jobs_off), n_jobs, max_blocks, reinterpret_cast<uint32_t *>(b->d_scan), s);
    if (e == hipSuccess) e = hipMemcpyAsync(stats, b->d_scan, stats_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return set_err(b->err, JPGPU_ERR_IO, "scan_ranges: %s", hipGetErrorString(e));
    for (size_t i = 0; i < b->descs.size(); i++)
        for (uint32_t c = 0; c < 4; c++) {
            uint8_t cls = 0;
            if (c < b->descs[i].ncomp) {
                const uint32_t *st = stats + (i * 4 + c) * RS_WORDS;
                cls = (uint8_t)range_class_from_stats(st[RS_MAX_DC], st[RS_MAX_AC], st[RS_MAX_COL], 1u);
                batch_set_host_class(b, i * 4 + c, cls);
            }
            if (classes) classes[i * 4 + c] = cls;
        }
    return JPGPU_OK;
}

int jpgpu_batch_classify_on_device(jpgpu_batch *b, void *hip_stream) {
    if (!b) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    hipStream_t s = (hipStream_t)hip_stream;
    rc = batch_enable_dev_classes(b);
    if (rc) return rc;
    uint32_t max_blocks = 0, n_jobs = 0;
    size_t jobs_off = 0;
    rc = batch_scan_jobs(b, n_jobs, max_blocks, jobs_off);
    if (rc) return rc;
    // zero, then the scan (which also marks the column maxima as exact: RS_COL_EXACT)
    B_HIP(hipMemsetAsync(b->d_stats, 0, b->descs.size() * RS_WORDS * sizeof(uint32_t), s));
    B_HIP(launch_range_scan(reinterpret_cast<const RangeJob *>(b->d_scan + jobs_off) + n_jobs, n_jobs, max_blocks, b->d_stats, s));
    for (size_t i = 0; i < b->descs.size(); i++)
        for (uint32_t c = 0; c < b->descs[i].ncomp; c++) {
            b->sane[i * 4 + c] = 0;  // (the host does not know)
            batch_class_source(b, i * 4 + c, true);
        }
    return JPGPU_OK;
}

int jpgpu_batch_set_quantization_table(jpgpu_batch *b, uint32_t image, uint32_t comp, const uint16_t q[64]) {
    if (!b || !q || image >= b->descs.size() || comp >= b->descs[image].ncomp) return JPGPU_ERR_FORMAT;
    if (memcmp(b->descs[image].quantization_tables[comp], q, 128) == 0) return JPGPU_OK;
    memcpy(b->descs[image].quantization_tables[comp], q, 128);
    // the range class of coefficients already uploaded was computed with the old table (|c*q| bounds): unknown again
    // (statistics the device gathered with the old table included)
    batch_set_host_class(b, (size_t)image * 4 + comp, 0);
    b->scan_jobs_valid = false;
    b->qt_dirty = true;
    b->jobs_dirty = true;
    return JPGPU_OK;
}

int jpgpu_batch_upload(jpgpu_batch *b, uint32_t image, uint32_t comp, const int16_t *coefficients, size_t len) {
    if (!b) return JPGPU_ERR_FORMAT;
    if (image >= b->descs.size() || comp >= b->descs[image].ncomp || !coefficients)
        return set_err(b->err, JPGPU_ERR_FORMAT, "upload: bad image/component");
    if (len * sizeof(int16_t) != b->coef_len[image * 4 + comp])
        return set_err(b->err, JPGPU_ERR_FORMAT, "upload: %zu coefficients, geometry needs %zu", len,
                       b->coef_len[image * 4 + comp] / sizeof(int16_t));
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    // range scan (part of H2D staging): per-position max |c| times q must stay below 2^15 for
    // the 24-bit multiply path to be exact (pixel_math.hpp idct8x8<SANE>)
    uint8_t sane = 0;  // bit0: every |c*q| < 2^15; bit1: additionally every column sum of |c*q| <= 5900
    if (!(b->flags & JPGPU_BATCH_ASSUME_HOSTILE))
        sane = (uint8_t)jpgpu_range_class(coefficients, len, b->descs[image].quantization_tables[comp]);
    batch_set_host_class(b, (size_t)image * 4 + comp, sane);
    B_HIP(hipMemcpy(b->d_coef + b->coef_off[image * 4 + comp], coefficients, len * sizeof(int16_t), hipMemcpyHostToDevice));
    return JPGPU_OK;
}

}  // extern "C"

// `trusted`: the buffer comes from CompactWriter in this library (pipeline.cpp) — skip the consistency pass
int jpgpu::batch_upload_compact(jpgpu_batch *b, uint32_t image, uint32_t comp, const void *compact, size_t bytes,
                                int range_class, void *hip_stream, bool trusted) {
    if (!b) return JPGPU_ERR_FORMAT;
    if (image >= b->descs.size() || comp >= b->descs[image].ncomp || !compact)
        return set_err(b->err, JPGPU_ERR_FORMAT, "upload_compact: bad image/component");
    const size_t idx = (size_t)image * 4 + comp, nblk = b->coef_len[idx] / 128;
    // the device trusts the index: check it here (one pass over the fixed part)
    if (bytes < nblk * 12 || ((bytes - nblk * 12) & 1) || bytes > compact_max_bytes(nblk))
        return set_err(b->err, JPGPU_ERR_FORMAT, "upload_compact: %zu bytes do not fit %zu blocks", bytes, nblk);
    if (!trusted) {
        const uint64_t *bm = static_cast<const uint64_t *>(compact);
        const uint32_t *first = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(compact) + nblk * 8);
        size_t n = 0;
        for (size_t k = 0; k < nblk; k++) {
            if (first[k] != n) return set_err(b->err, JPGPU_ERR_FORMAT, "upload_compact: inconsistent value index at block %zu", k);
            n += (size_t)__builtin_popcountll(bm[k]);
        }
        if (n * 2 != bytes - nblk * 12) return set_err(b->err, JPGPU_ERR_FORMAT, "upload_compact: value count does not match the bitmaps");
    }
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    {
        std::lock_guard<std::mutex> g(b->compact_mutex);
        if (!b->d_compact) {
            const size_t n = b->descs.size();
            b->compact_off.assign(n * 4, 0);
            b->compact_pending.assign(n * 4, 0);
            size_t off = 0;
            for (size_t i = 0; i < n; i++)
                for (uint32_t c = 0; c < b->descs[i].ncomp; c++) {
                    b->compact_off[i * 4 + c] = off;
                    off += align_up(compact_max_bytes(b->coef_len[i * 4 + c] / 128), 256);
                }
            B_HIP(hipMalloc((void **)&b->d_compact, std::max<size_t>(off, 256)));
            B_HIP(hipMalloc((void **)&b->d_expand_jobs, n * 4 * sizeof(ExpandJob)));
        }
        // range_class < 0: the sender did not classify — expand_compact_kernel ranges the values while it expands them
        b->compact_pending[idx] = range_class >= 0 ? 1 : 2;
        b->any_compact_pending = true;
        if (range_class >= 0) {
            batch_set_host_class(b, idx, (uint8_t)(range_class & 3));
        } else {
            rc = batch_enable_dev_classes(b);
            if (rc) return rc;
            b->sane[idx] = 0;
            // (the whole plane is replaced and ranged, at expansion time, with the table the device then holds — batch_refresh_jobs
            // runs first; older maxima only over-estimate)
            batch_class_source(b, idx, true);
        }
    }
    B_HIP(hipMemcpyAsync(b->d_compact + b->compact_off[idx], compact, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    return JPGPU_OK;
}

int jpgpu::copy_device_to_pinned_host(void *host_pinned, const void *d_src, size_t bytes, void *hip_stream) {
    static const bool engine = getenv("JPGPU_DOWNLOAD_BY_COPY_ENGINE") != nullptr;  // A/B: hipMemcpyAsync instead of the copy kernel
    void *mapped = nullptr;
    if (!engine && ((uintptr_t)host_pinned & 15u) == 0 && ((uintptr_t)d_src & 15u) == 0 && hipHostGetDevicePointer(&mapped, host_pinned, 0) == hipSuccess)
        return launch_copy_to_host(mapped, d_src, bytes, (hipStream_t)hip_stream) == hipSuccess ? JPGPU_OK : JPGPU_ERR_IO;
    (void)hipGetLastError();
    return hipMemcpyAsync(host_pinned, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream) == hipSuccess ? JPGPU_OK : JPGPU_ERR_IO;
}

// staged bytes of the device-entropy route: pinned block -> its device twin (offsets and lengths are multiples of 16 by construction)
static hipError_t upload_staged(void *d_dst, const void *h_src, size_t bytes, hipStream_t s) {
    static const bool by_kernel = getenv("JPGPU_UPLOAD_BY_KERNEL") != nullptr;
    void *mapped = nullptr;
    if (by_kernel && ((uintptr_t)h_src & 15u) == 0 && ((uintptr_t)d_dst & 15u) == 0 && hipHostGetDevicePointer(&mapped, const_cast<void *>(h_src), 0) == hipSuccess)
        return launch_copy_from_host(d_dst, mapped, bytes, s);
    if (by_kernel) (void)hipGetLastError();
    return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s);
}

// ---- device entropy decoding -----------------------------------------------------------------------------------
// the status words of a launch into b->h_entropy_out (pinned), behind the kernels on `s`: by a kernel, not by the copy engine (huff.hip)
static hipError_t batch_status_to_host(jpgpu_batch *b, const uint32_t *d_status, uint32_t n, hipStream_t s) {
    void *mapped = nullptr;
    hipError_t e = hipHostGetDevicePointer(&mapped, b->h_entropy_out, 0);
    if (e != hipSuccess) {  // (no mapping: the copy engine after all)
        (void)hipGetLastError();
        return hipMemcpyAsync(b->h_entropy_out, d_status, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    }
    return launch_copy_words_to_host(static_cast<uint32_t *>(mapped), d_status, n, s);
}
static uint32_t env_u32(const char *name, uint32_t dflt, uint32_t lo, uint32_t hi) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const long v = atol(e);
    return (uint32_t)std::min<long>(std::max<long>(v, lo), hi);
}

// Staging block layout (same offsets in the pinned and the device copy):
//   [ status: n x u32 | settle counters: 4 x u32 per sync job | HuffSyncJob[] (segment jobs) | HuffSyncJob[] (chunk jobs) |
//     DevHuffTable[8] per scan | segment offsets | scan bytes ]   + device only: per-chunk state of the sync jobs
// The range statistics of the decoded coefficients are a by-product of the kernels that write them (HuffSyncJob::stats ->
// the batch's d_stats; round 2 ran range_scan_kernel over the arena afterwards and read the result back).
namespace {
// A scan as the file holds it, into the pinned staging block (host light): the destination is read by the copy engine and never by
// this CPU, so the stores go past the caches — a plain memcpy of 400 kB reads every destination line before it overwrites it, and
// the staging team of a sub-batch is bound by exactly that traffic (25 MB per 64 files: 0.7-0.8 ms on 16 CPUs, the launches of a
// 256-file call one after the other).  JPGPU_STAGE_PLAIN_MEMCPY=1: memcpy (A/B).
inline void copy_past_the_caches(uint8_t *dst, const uint8_t *src, size_t n) {
#if defined(__x86_64__) && defined(__clang__)
    static const bool plain = getenv("JPGPU_STAGE_PLAIN_MEMCPY") != nullptr;
    if (plain || n < 4096u) {
        memcpy(dst, src, n);
        return;
    }
    typedef long long v2di __attribute__((vector_size(16)));
    const size_t head = (16u - ((uintptr_t)dst & 15u)) & 15u;
    memcpy(dst, src, head);
    dst += head, src += head, n -= head;
    size_t i = 0;
    for (; i + 64u <= n; i += 64u) {
        v2di a, b, c, d;
        memcpy(&a, src + i, 16), memcpy(&b, src + i + 16, 16), memcpy(&c, src + i + 32, 16), memcpy(&d, src + i + 48, 16);
        __builtin_nontemporal_store(a, (v2di *)(dst + i));
        __builtin_nontemporal_store(b, (v2di *)(dst + i + 16));
        __builtin_nontemporal_store(c, (v2di *)(dst + i + 32));
        __builtin_nontemporal_store(d, (v2di *)(dst + i + 48));
    }
    __builtin_ia32_sfence();
    memcpy(dst + i, src + i, n - i);
#else
    memcpy(dst, src, n);
#endif
}
struct LaunchClock {  // JPGPU_PIPE_TRACE: where a slow launch spent its time (host)
    bool on = getenv("JPGPU_PIPE_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    char text[512];
    size_t used = 0;
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        used += (size_t)snprintf(text + used, used < sizeof(text) ? sizeof(text) - used : 0, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - last).count());
        if (used >= sizeof(text)) used = sizeof(text) - 1;
        last = now;
    }
    ~LaunchClock() {
        if (on && std::chrono::duration<double, std::milli>(last - t0).count() > 3.0) fprintf(stderr, "pipeline trace: slow device entropy launch (ms):%s\n", text);
    }
};
}  // namespace

int jpgpu::batch_device_entropy_launch(jpgpu_batch *b, const DeviceEntropyImage *images, uint32_t n, void *hip_stream,
                                       const std::function<void(uint32_t, const std::function<void(uint32_t)> &)> *par, void *copy_stream,
                                       DeviceScratch *scratch, bool alone, uint32_t mode, uint32_t *n_light, uint32_t *n_entry) {
    if (!b || !images || n == 0) return JPGPU_ERR_FORMAT;
    LaunchClock clk;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    hipStream_t s = (hipStream_t)hip_stream;
    // Chunk size of the chunk decoder: ~48 blocks per chunk settle in the fewest passes and give the best throughput when a
    // call fills the device (256 x 1080p: 270 k lanes).  A call with few streams is latency-bound instead — three passes in
    // which every lane walks its whole chunk, ~2.4 us per symbol: one 1080p image is 1,055 lanes and 2.3 ms — so small calls
    // get smaller chunks and more (cheap once settled) passes: 1080p 3.07 -> 1.68 ms, 2160p 4.82 -> 3.02, 512^2 2.25 -> 1.30
    // through jpgpu_pipeline_decode (profiles/round2/09_decoder_latency.txt).  The environment knobs pin the values (A/B).
    static const bool sync_pinned = getenv("JPGPU_SYNC_BLOCKS") || getenv("JPGPU_SYNC_MIN_SHIFT") || getenv("JPGPU_SYNC_LAUNCHES");
    static const uint32_t env_sync_blocks = env_u32("JPGPU_SYNC_BLOCKS", 48, 1, 1024);      // blocks per chunk aimed at
    static const uint32_t env_sync_min_shift = env_u32("JPGPU_SYNC_MIN_SHIFT", 10, 7, 15);  // smallest chunk: 1 << this many bits
    static const uint32_t env_sync_launches = env_u32("JPGPU_SYNC_LAUNCHES", 10, 1, 64);
    uint32_t sync_blocks = env_sync_blocks, sync_min_shift = env_sync_min_shift, sync_launches = env_sync_launches;
    // Passes per launch: a call that fills the device runs one pass per launch — launches cost 5 us once a job has settled, and passes
    // that begin with everything the one before published need fewer of them (4,096 files: 53 ms against 55-57 with two passes per
    // launch) —; small calls, which wait for every launch, keep two.
    static const bool iters_pinned = getenv("JPGPU_SYNC_ITERS") != nullptr;
    static const uint32_t env_iters = env_u32("JPGPU_SYNC_ITERS", 2, 1, 8);
    uint32_t sync_iters = env_iters;
    static const uint32_t env_late = env_u32("JPGPU_SYNC_LATE_PASS", 0, 0, 64);  // (0: chosen here)
    uint32_t late_pass = env_late ? env_late : HUFF_LATE_PASS;
    if (!sync_pinned) {
        uint64_t lanes = 0;  // at the throughput setting
        for (uint32_t k = 0; k < n && images[k].scans; k++)
            for (const host::PlannedScan &ps : *images[k].scans)
                if (ps.seg_off.size() >= 2) {  // (restart segments go through the chunk decoder too: about as many lanes for the same bytes)
                    uint32_t blocks = 0;
                    for (uint32_t c = 0; c < ps.ncomp; c++) blocks += ps.comp[c].h * ps.comp[c].v;
                    uint64_t bytes = 0;
                    for (size_t sg = 0; sg + 1 < ps.seg_off.size(); sg += 2) bytes += ps.seg_off[sg + 1] - ps.seg_off[sg];
                    if (bytes < (1u << 28)) lanes += huff_sync_chunks((uint32_t)bytes, huff_sync_chunk_shift((uint32_t)bytes, blocks * ps.n_mcu, 48u, 10u));
                }
        if (lanes < 16384u) sync_blocks = 12u, sync_min_shift = 9u, sync_launches = 16u;
        else if (lanes < 65536u) sync_blocks = 24u, sync_launches = 12u;
        else {
            if (!iters_pinned) sync_iters = 1u, sync_launches = 16u;
            // one of many sub-batches in flight: what counts is the work, and longer chunks mean fewer lanes that decode their chunk twice
            // (4,096 files: 50.4-55.2 ms against 51.3-60.0 on one box, interleaved; a call of one or two sub-batches waits for the chains
            // of its late passes instead and keeps the shorter ones: tools/gpu_knobs3.sh)
            if (!alone) sync_blocks = 64u;
        }
    }
    // Speculative emission (huff_job.hpp): the sync passes leave entry lists, huff_expand_kernel writes whole blocks — no write
    // pass, and no zero fill for images whose scans cover their planes.
    bool low_table_ids = env_u32("JPGPU_SYNC_COMPACT_TABLES", 1, 0, 1) != 0;  // until a scan uses a Huffman table id above 1 (huff_sync_pass_kernel<4>)
    static const bool tail_pinned = getenv("JPGPU_SYNC_TAIL") != nullptr;
    static const uint32_t env_tail = env_u32("JPGPU_SYNC_TAIL", 3, 1, 8);  // eighths of its chunk a lane walks in the first sync pass
    const uint32_t sync_tail = (alone && !tail_pinned) ? 8u : env_tail;
    if (alone && !iters_pinned) sync_iters = env_iters;
    // Restart-marker streams (huff_job.hpp, HuffSyncJob::seg_chunks): every segment gets chunk slots of its own (`uniform` scans
    // too: huff_dc_prefix_kernel starts its sums again at every segment); a scan whose restart interval covers all its MCUs is ONE
    // segment, i.e. a scan without restart markers.  Bit positions are 32-bit numbers relative to the scan's first slot: a scan
    // whose slots exceed 2^29 bytes is handed back to the host (status bit 8).
    struct DriGeom {
        bool chunked, too_large;
        uint32_t shift, seg_chunks;
    };
    auto dri_geom = [&](const host::PlannedScan &ps) {
        DriGeom g{false, false, 0u, 0u};
        if (ps.ri == 0 || ps.seg_off.size() < 4) return g;
        uint32_t blocks = 0;
        for (uint32_t c = 0; c < ps.ncomp; c++) blocks += ps.comp[c].h * ps.comp[c].v;
        uint64_t stuffed = 0, slots = 0;
        uint32_t longest = 0;
        for (size_t sg = 0; sg + 1 < ps.seg_off.size(); sg += 2) {
            stuffed += ps.seg_off[sg + 1] - ps.seg_off[sg];
            slots += huff_slot_bytes(ps.seg_off[sg + 1] - ps.seg_off[sg]);
            longest = std::max<uint32_t>(longest, ps.seg_off[sg + 1] - ps.seg_off[sg]);
        }
        g.chunked = true;
        if (stuffed >= (1u << 28) || slots >= (1u << 29)) {
            g.too_large = true;
            return g;
        }
        g.shift = huff_sync_chunk_shift((uint32_t)stuffed, blocks * ps.n_mcu, sync_blocks, sync_min_shift);
        g.seg_chunks = huff_sync_chunks(longest, g.shift);
        return g;
    };
    rc = batch_enable_dev_classes(b);
    if (rc) return rc;
    // "Host light" (include/jpgpu_decoder.h): scans without restart markers go up as the file holds them — the staging task is a plain
    // memcpy, or nothing at all when the caller's buffers are pinned (DEVICE_ENTROPY_INPUT_PINNED: the copy engine reads them) — into a
    // MIRROR of the data area, and huff_unstuff_* (huff.hip) does what huff_stage_segment does: marker check, unstuffing into the
    // scan's slot, the job record's lengths.  Scans with restart markers are staged by the host as ever (into the mirror too: their
    // jobs then read them there).
    const bool light = (mode & DEVICE_ENTROPY_LIGHT) != 0, input_pinned = light && (mode & DEVICE_ENTROPY_INPUT_PINNED) != 0;
    // Entry-list pixel path: which 4:2:0 strip walk (if any) a batch image belongs to
    const bool entry_pixels = (mode & DEVICE_ENTROPY_ENTRY_PIXELS) != 0;
    std::vector<const FusedGeom *> walk_geom(entry_pixels ? b->descs.size() : 0, nullptr);
    if (entry_pixels)
        for (const FusedPlan &fp : b->fused)
            if (fp.kind == FUSED_420 && fp.strip)
                for (uint32_t i = 0; i < fp.n_images; i++) walk_geom[fp.ids[i]] = &fp.geoms[i];
    if (b->entry_img.size() != b->descs.size()) b->entry_img.assign(b->descs.size(), 0);
    for (uint8_t &e : b->entry_img)
        if (e) e = 0, b->cls_dirty = true;
    b->entries_pending = false;
    size_t n_raw_jobs = 0;
    uint32_t max_pieces = 0, light_images = 0;
    constexpr size_t PINNED_SPAN_GAP_MAX = 4096u;  // (bytes; below one page: see where the spans are built)
    struct PinnedSpan {
        const uint8_t *start, *end;  // the caller's bytes [start, end): the scans of consecutive files and what lies between them
        size_t mirror_off;           // where `start` lands in the span region of the mirror
    };
    std::vector<PinnedSpan> spans;
    std::vector<size_t> raw_mirror_off;  // per raw scan, in listing order: its offset in the span region
    struct RawScan {
        const uint8_t *src;
        size_t bytes, listed;  // (listed: its place in listing order)
    };
    std::vector<RawScan> raw_scans;
    size_t n_sync_jobs = 0, seg_words = 0, data_bytes = 0, scratch_bytes = 0;
    // Files of one encoder repeat the same Huffman tables (27 kB per scan in device form): a scan whose tables equal those of
    // the scan before it shares that copy — one in the staging block, one upload, one set of lines in the L2.
    const host::PlannedScan::TableSet *prev_tables = nullptr;
    size_t n_table_sets = 0;
    for (uint32_t k = 0; k < n; k++) {
        if (images[k].image >= b->descs.size() || !images[k].scans || !images[k].file) return set_err(b->err, JPGPU_ERR_FORMAT, "device entropy: bad image");
        if (entry_pixels && walk_geom[images[k].image]) scratch_bytes += align_up((size_t)walk_geom[images[k].image]->mcu_h * walk_geom[images[k].image]->tiles_x * 8u, 16);
        for (const host::PlannedScan &ps : *images[k].scans) {
            if (!ps.tables) return set_err(b->err, JPGPU_ERR_FORMAT, "device entropy: scan without tables");
            if (!prev_tables || (prev_tables != ps.tables.get() && memcmp(prev_tables, ps.tables.get(), sizeof(*prev_tables)) != 0)) n_table_sets++;
            prev_tables = ps.tables.get();
            seg_words += ps.seg_off.size();
            size_t stuffed = 0;
            for (size_t sg = 0; sg + 1 < ps.seg_off.size(); sg += 2) {
                data_bytes += huff_slot_bytes(ps.seg_off[sg + 1] - ps.seg_off[sg]);
                stuffed += ps.seg_off[sg + 1] - ps.seg_off[sg];
            }
            n_sync_jobs++;  // every scan is a job of the chunk decoder, with its per-chunk state and entry buffers (device only)
            if (light && ps.check_at_staging && ps.seg_off.size() == 2) {
                n_raw_jobs++;
                const uint32_t pieces = (uint32_t)((stuffed + 15u + UNSTUFF_PIECE - 1u) / UNSTUFF_PIECE);
                max_pieces = std::max(max_pieces, pieces);
                scratch_bytes += align_up(((size_t)pieces + 1u) * 4u, 16);
                if (input_pinned) raw_scans.push_back(RawScan{images[k].file + ps.data_off, stuffed, raw_scans.size()});
            }
            if (const DriGeom g = dri_geom(ps); g.chunked) {
                const size_t chunks = g.too_large ? 0 : (size_t)(ps.seg_off.size() / 2) * g.seg_chunks;
                scratch_bytes += align_up(chunks * 8 * 4, 16) + align_up(chunks * 4, 16) + align_up(chunks * huff_emit_stride(g.shift) * 4, 16) +
                                 align_up(huff_weave_dwords((uint32_t)chunks, g.shift) * 4, 256) + 256;
            } else {  // one segment: a scan without restart markers (or one whose restart interval covers it)
                if (ps.seg_off.size() != 2 || stuffed >= (1u << 28)) return set_err(b->err, JPGPU_ERR_FORMAT, "device entropy: bad plan");
                uint32_t blocks = 0;
                for (uint32_t c = 0; c < ps.ncomp; c++) blocks += ps.comp[c].h * ps.comp[c].v;
                const uint32_t shift = huff_sync_chunk_shift((uint32_t)stuffed, blocks * ps.n_mcu, sync_blocks, sync_min_shift);
                const size_t chunks = huff_sync_chunks((uint32_t)stuffed, shift);
                scratch_bytes += align_up(chunks * 8 * 4, 16) + align_up(chunks * 4, 16) + align_up(chunks * huff_emit_stride(shift) * 4, 16) +
                                 align_up(huff_weave_dwords((uint32_t)chunks, shift) * 4, 256) + 256;
            }
        }
    }
    if (input_pinned && !raw_scans.empty()) {
        // Pinned input: files that follow one another in the caller's memory (a loader's arena) travel in ONE copy, headers and gaps
        // included; a scan's place in the mirror is then its place in the span.  Two scans share a span only if the gap between them is
        // SHORTER THAN ONE PAGE (it was 8 kB): the caller vouches for the bytes of its files only, and a gap of a page or more
        // may hold a page that is not mapped or not pinned; a gap below 4,096 bytes lies in the page of the byte in front of it and the
        // page of the byte behind it, both of which hold bytes of a file (pinning and mapping are per page).  (One
        // hipMemcpyAsync per file: 4,096 calls per call of 4,096 files — 114 ms on 16 CPUs.)  By ADDRESS, not in listing order: the
        // pipeline lists a sub-batch's images as its threads finish their headers, and an arena need not hold files in call order.
        std::sort(raw_scans.begin(), raw_scans.end(), [](const RawScan &a, const RawScan &c) { return a.src < c.src; });
        raw_mirror_off.assign(raw_scans.size(), 0);
        for (const RawScan &r : raw_scans) {
            if (spans.empty() || r.src < spans.back().end || (size_t)(r.src - spans.back().end) >= PINNED_SPAN_GAP_MAX) {
                const size_t at = spans.empty() ? 0 : align_up(spans.back().mirror_off + (size_t)(spans.back().end - spans.back().start), 16) + 16;
                spans.push_back(PinnedSpan{r.src, r.src, at});
            }
            raw_mirror_off[r.listed] = spans.back().mirror_off + (size_t)(r.src - spans.back().start);
            spans.back().end = r.src + r.bytes;
        }
    }
    const size_t off_status = 0, off_cnt = align_up(off_status + (size_t)n * 4, 16);
    const size_t off_jobs = align_up(off_cnt + n_sync_jobs * 16, 16), off_sjobs = off_jobs;
    const size_t off_ujobs = align_up(off_sjobs + n_sync_jobs * sizeof(HuffSyncJob), 16);
    const size_t off_esrc = align_up(off_ujobs + n_raw_jobs * sizeof(UnstuffJob), 16);  // entry-list pixel path: EntrySrc per BATCH image, EntryIndexJob per listed image
    const size_t off_ijobs = align_up(off_esrc + (entry_pixels ? b->descs.size() * sizeof(EntrySrc) : 0), 16);
    const size_t off_tables = align_up(off_ijobs + (entry_pixels ? (size_t)n * sizeof(EntryIndexJob) : 0), 16);
    const size_t off_seg = align_up(off_tables + n_table_sets * 8 * sizeof(DevHuffTable), 16), off_data = align_up(off_seg + seg_words * 4, 16);
    const size_t total = off_data + data_bytes;              // uploaded
    // (light: what is uploaded lands in the mirror — a copy of the data area's layout, and behind it, with pinned input, the spans)
    const size_t span_bytes = spans.empty() ? 0 : spans.back().mirror_off + (size_t)(spans.back().end - spans.back().start) + 64;
    const size_t off_mirror = align_up(total, 256), off_spans = align_up(off_mirror + data_bytes + 64, 256), dev_end = light ? off_spans + span_bytes : total;
    const size_t off_scratch = align_up(dev_end, 256), total_dev = scratch ? dev_end : off_scratch + scratch_bytes;
    if (scratch && scratch_bytes > scratch->cap) {  // (hipFree waits for whatever still uses the block)
        if (scratch->d) (void)hipFree(scratch->d);
        scratch->d = nullptr;
        scratch->cap = 0;
        const size_t cap = scratch_bytes + scratch_bytes / 4;
        B_HIP(hipMalloc((void **)&scratch->d, cap));
        scratch->cap = cap;
    }
    if (total_dev > b->entropy_cap) {
        if (b->d_entropy) (void)hipFree(b->d_entropy);
        b->d_entropy = nullptr;
        b->entropy_cap = 0;
        const size_t cap = total_dev + total_dev / 4;
        B_HIP(hipMalloc((void **)&b->d_entropy, cap));
        b->entropy_cap = cap;
    }
    if (total > b->entropy_host_cap) {
        if (b->h_entropy) (void)hipHostFree(b->h_entropy);
        b->h_entropy = nullptr;
        b->entropy_host_cap = 0;
        const size_t cap = total + total / 4;
        B_HIP(hipHostMalloc((void **)&b->h_entropy, cap, hipHostMallocDefault));
        b->entropy_host_cap = cap;
    }
    const size_t out_words = (size_t)n;
    if (out_words > b->entropy_out_cap) {
        if (b->h_entropy_out) (void)hipHostFree(b->h_entropy_out);
        b->h_entropy_out = nullptr;
        B_HIP(hipHostMalloc((void **)&b->h_entropy_out, (out_words + 64) * 4, hipHostMallocDefault));
        b->entropy_out_cap = out_words + 64;
    }
    clk.mark("buffers");
    uint8_t *h = b->h_entropy, *d = b->d_entropy;
    memset(h, 0, off_jobs);  // status words and settle counters start at zero
    HuffSyncJob *sjobs = reinterpret_cast<HuffSyncJob *>(h + off_sjobs);
    uint8_t *xs = scratch ? scratch->d : d;  // base of the device-only work space
    size_t si = 0, tcur = off_tables, tnext = off_tables, scur = off_seg, dcur = off_data, xcur = scratch ? 0 : off_scratch;
    prev_tables = nullptr;
    uint32_t max_chunks = 0;
    std::vector<uint32_t> stat_images;  // listed images, for the fills that zero their statistics
    struct CopyTask {
        uint8_t *dst;        // first slot of the scan in the pinned block
        uint32_t *seg_table; // its 2 * n_seg words
        uint32_t dst_off;    // offset of dst inside the data area (which slice of the upload the scan belongs to)
        const uint8_t *src;  // the scan's entropy-coded bytes
        const host::PlannedScan *ps;
        HuffSyncJob *sync;   // its job record (a scan of one segment: the unstuffed length goes there)
        uint32_t *h_status;  // the image's status word in the pinned block (set here if the staging pass refuses the stream)
        bool raw;            // host light: the bytes go up as they are (huff_unstuff_* does the rest on the device)
    };
    std::vector<CopyTask> copies;
    UnstuffJob *ujobs = reinterpret_cast<UnstuffJob *>(h + off_ujobs);
    size_t ui = 0;
    EntrySrc *esrc = reinterpret_cast<EntrySrc *>(h + off_esrc);
    EntryIndexJob *ijobs = reinterpret_cast<EntryIndexJob *>(h + off_ijobs);
    uint32_t n_index = 0, max_index_items = 0;
    if (entry_pixels) memset(esrc, 0, b->descs.size() * sizeof(EntrySrc));
    std::vector<std::pair<size_t, size_t>> zero_ranges;  // coefficient planes of the listed images
    b->entropy_images.clear();
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t img = images[k].image;
        b->entropy_images.push_back(img);
        const jpgpu_image_desc &desc = b->descs[img];
        bool needs_zeros = false;  // (the expansion writes every block of a scan, zeros included; planes a scan does not cover: below)
        bool image_is_light = false;
        stat_images.push_back(img);
        for (uint32_t c = 0; c < desc.ncomp; c++) {  // the class of every component: from what the expansion leaves in d_stats
            b->sane[(size_t)img * 4 + c] = 0;
            batch_class_source(b, (size_t)img * 4 + c, true);
        }
        for (const host::PlannedScan &ps : *images[k].scans) {
            // one staging task per scan: its segments, unstuffed, each in its own aligned slot (huff_stage_segment)
            const DriGeom dg = dri_geom(ps);
            HuffSyncJob *sj = &sjobs[si];
            const bool raw_scan = light && ps.check_at_staging && ps.seg_off.size() == 2;
            copies.push_back(CopyTask{h + dcur, reinterpret_cast<uint32_t *>(h + scur), (uint32_t)(dcur - off_data), images[k].file + ps.data_off, &ps, sj,
                                      reinterpret_cast<uint32_t *>(h + off_status) + k, raw_scan});
            if (raw_scan && !image_is_light) {
                image_is_light = true;
                light_images++;
            }
            size_t scan_bytes = 0, stuffed = 0;
            for (size_t sg = 0; sg + 1 < ps.seg_off.size(); sg += 2) {
                scan_bytes += huff_slot_bytes(ps.seg_off[sg + 1] - ps.seg_off[sg]);
                stuffed += ps.seg_off[sg + 1] - ps.seg_off[sg];
            }
            if (!prev_tables || (prev_tables != ps.tables.get() && memcmp(prev_tables, ps.tables.get(), sizeof(*prev_tables)) != 0)) {
                tcur = tnext;
                tnext += 8 * sizeof(DevHuffTable);
                memcpy(h + tcur, ps.tables->t, sizeof(ps.tables->t));
            }
            prev_tables = ps.tables.get();
            HuffScanComp comp[4];
            memset(comp, 0, sizeof(comp));
            uint16_t scan_q[4][64];
            memset(scan_q, 0, sizeof(scan_q));
            for (uint32_t c = 0; c < ps.ncomp; c++) {
                const uint32_t fi = ps.comp[c].frame_index;
                if (fi >= desc.ncomp || ps.comp[c].block_w != desc.components[fi].block_width)
                    return set_err(b->err, JPGPU_ERR_FORMAT, "device entropy: plan does not match the image descriptor");
                memcpy(scan_q[c], desc.quantization_tables[fi], 128);
                comp[c].dst = reinterpret_cast<int16_t *>(b->d_coef + b->coef_off[(size_t)img * 4 + fi]);
                comp[c].block_w = ps.comp[c].block_w;
                comp[c].h = ps.comp[c].h;
                comp[c].v = ps.comp[c].v;
                comp[c].dc = ps.comp[c].dc;
                comp[c].ac = ps.comp[c].ac;
                if (ps.comp[c].dc > 1u || ps.comp[c].ac > 1u) low_table_ids = false;
            }
            {
                memset(sj, 0, sizeof(*sj));
                memcpy(sj->comp, comp, sizeof(comp));
                sj->ncomp = ps.ncomp;
                memcpy(sj->q, scan_q, sizeof(scan_q));
                sj->stats = b->d_stats + (size_t)img * RS_WORDS;
                huff_sync_finish_job(*sj);
                sj->chunk_shift = dg.chunked ? dg.shift : huff_sync_chunk_shift((uint32_t)stuffed, sj->bpm * ps.n_mcu, sync_blocks, sync_min_shift);
                sj->pass0_skip = ((1u << sj->chunk_shift) >> 3) * (8u - sync_tail);
                sj->late_pass = late_pass;
                // (one segment: an upper bound, the staging task sets the real count; restart segments: slots per segment x segments)
                const uint32_t chunks = dg.chunked ? (dg.too_large ? 0u : (uint32_t)(ps.seg_off.size() / 2) * dg.seg_chunks)
                                                   : huff_sync_chunks((uint32_t)stuffed, sj->chunk_shift);
                uint32_t *st = reinterpret_cast<uint32_t *>(xs + xcur);
                // (bit positions and segment offsets are relative to the scan's first slot; light: a scan the host staged itself — restart
                // segments — lies in the mirror, where the uploads of such a launch land)
                sj->data = (light && !raw_scan) ? d + off_mirror + (dcur - off_data) : d + dcur;
                if (dg.chunked) {
                    sj->seg_off = reinterpret_cast<const uint32_t *>(d + scur);
                    sj->n_seg = (uint32_t)(ps.seg_off.size() / 2);
                    sj->ri = ps.ri;
                    sj->seg_chunks = dg.seg_chunks;
                    sj->n_chunks = chunks;
                    if (dg.too_large) reinterpret_cast<uint32_t *>(h + off_status)[k] |= 1u | 256u;  // the host decodes this image
                }
                sj->tables = reinterpret_cast<const DevHuffTable *>(d + tcur);
                sj->status = reinterpret_cast<uint32_t *>(d + off_status) + k;
                sj->changed = reinterpret_cast<uint32_t *>(d + off_cnt) + si * 4;
                sj->in_pos = st;
                sj->in_qk = st + chunks;
                sj->out_pos = st + 2 * (size_t)chunks;
                sj->out_qk = st + 3 * (size_t)chunks;
                sj->n_blocks = st + 4 * (size_t)chunks;
                sj->dc_sum = st + 5 * (size_t)chunks;
                sj->blk_end = st + 7 * (size_t)chunks;
                sj->cols = ps.cols;
                sj->n_mcu = ps.n_mcu;
                max_chunks = std::max(max_chunks, chunks);
                xcur += align_up((size_t)chunks * 8 * 4, 16);
                sj->emit_stride = huff_emit_stride(sj->chunk_shift);
                sj->emit_cnt = reinterpret_cast<uint32_t *>(xs + xcur);
                xcur += align_up((size_t)chunks * 4, 16);
                sj->emit = reinterpret_cast<uint32_t *>(xs + xcur);
                xcur += align_up((size_t)chunks * sj->emit_stride * 4, 16);
                xcur = align_up(xcur, 256);  // the weave: rows of 256 bytes (the block's base is 256-byte aligned)
                sj->weave = reinterpret_cast<const uint32_t *>(xs + xcur);
                sj->data_dwords = (uint32_t)(scan_bytes / 4);
                xcur += align_up(huff_weave_dwords(chunks, sj->chunk_shift) * 4, 256);
                if (raw_scan) {
                    UnstuffJob &uj = ujobs[ui++];
                    memset(&uj, 0, sizeof(uj));
                    // (the data area's layout in the mirror, 16-byte aligned — or the scan's place in its span: the kernels take any alignment)
                    uj.raw = input_pinned ? d + off_spans + raw_mirror_off[ui - 1u] : d + off_mirror + (dcur - off_data);
                    uj.raw_bytes = (uint32_t)stuffed;
                    uj.n_pieces = (uint32_t)((((uintptr_t)uj.raw & 15u) + stuffed + UNSTUFF_PIECE - 1u) / UNSTUFF_PIECE);
                    uj.dst = d + dcur;
                    uj.piece_kept = reinterpret_cast<uint32_t *>(xs + xcur);
                    xcur += align_up(((size_t)uj.n_pieces + 1u) * 4u, 16);
                    uj.job = reinterpret_cast<HuffSyncJob *>(d + off_sjobs) + si;
                    uj.status = sj->status;
                    sj->n_chunks = chunks;  // (an upper bound until huff_unstuff_scan_kernel has counted)
                    sj->n_bits = 0;
                }
                uint32_t block_h[4] = {0, 0, 0, 0};
                for (uint32_t c = 0; c < ps.ncomp; c++) block_h[c] = desc.components[ps.comp[c].frame_index].block_height;
                if (!huff_scan_covers_planes(*sj, block_h)) needs_zeros = true;
                // Entry-list pixel path: the image's ONE scan holds its three components interleaved in frame order, 2x2 / 1x1 / 1x1, with
                // tables of their own for luma and chroma (the entries then carry their component), with or without restart segments, and
                // covers the planes of the 4:2:0 walk the image belongs to.
                if (const FusedGeom *wg = entry_pixels ? walk_geom[img] : nullptr; wg && images[k].scans->size() == 1 && !(dg.chunked && dg.too_large) && !needs_zeros &&
                                                                                    ps.ncomp == 3 && !sj->uniform && sj->bpm == 6u && sj->cols == wg->mcu_w &&
                                                                                    sj->n_mcu == wg->mcu_w * wg->mcu_h && wg->tiles_x * wg->tx >= wg->mcu_w) {
                    bool ok = true;
                    for (uint32_t c = 0; c < 3; c++) ok = ok && ps.comp[c].frame_index == c && ps.comp[c].h == (c ? 1u : 2u) && ps.comp[c].v == (c ? 1u : 2u);
                    if (ok) {
                        sj->keep_lists = 1u;
                        uint32_t *tab = reinterpret_cast<uint32_t *>(xs + xcur);
                        xcur += align_up((size_t)wg->mcu_h * wg->tiles_x * 8u, 16);
                        ijobs[n_index++] = EntryIndexJob{(uint32_t)si, wg->tx, wg->tiles_x, wg->mcu_h, tab};
                        max_index_items = std::max(max_index_items, wg->mcu_h * wg->tiles_x);
                        esrc[img] = EntrySrc{reinterpret_cast<const HuffSyncJob *>(d + off_sjobs) + si, tab};
                        b->entry_img[img] = 1;
                        b->cls_dirty = true;
                    }
                }
                si++;
            }
            dcur += scan_bytes;
            scur += ps.seg_off.size() * 4;
        }
        if (needs_zeros)
            zero_ranges.emplace_back(b->coef_off[(size_t)img * 4], b->coef_off[(size_t)img * 4 + desc.ncomp - 1] + b->coef_len[(size_t)img * 4 + desc.ncomp - 1]);
    }
    {
        hipStream_t raw_stream = (copy_stream && copy_stream != hip_stream) ? (hipStream_t)copy_stream : s;
        bool raw_copy_failed = false;
        const std::function<void(uint32_t)> body = [&](uint32_t t) {
            const CopyTask &ct = copies[t];
            if (ct.raw) {  // host light: as the file holds it — one memcpy, or none (the copy engine reads the caller's pinned buffer)
                const uint32_t nraw = ct.ps->seg_off[1] - ct.ps->seg_off[0];
                ct.seg_table[0] = 0;
                ct.seg_table[1] = nraw;  // (the stuffed length; the job's lengths come from huff_unstuff_scan_kernel)
                if (!input_pinned) copy_past_the_caches(ct.dst, ct.src + ct.ps->seg_off[0], nraw);
                return;  // (pinned input: the copy engine reads the caller's buffer — enqueued below, by this thread alone)
            }
            uint32_t o = 0;
            for (size_t sg = 0; sg + 1 < ct.ps->seg_off.size(); sg += 2) {
                const uint32_t first = ct.ps->seg_off[sg], n = ct.ps->seg_off[sg + 1] - first;
                bool clean = true;
                ct.seg_table[sg] = o;  // (relative to the scan's first slot)
                ct.seg_table[sg + 1] = huff_stage_segment(ct.dst + o, ct.src + first, n, ct.ps->check_at_staging ? &clean : nullptr);
                o += huff_slot_bytes(n);
                if (!clean) *ct.h_status |= 1u | 16u;  // something other than 0xFF00 pairs inside the scan: the host decodes this image
            }
            if (ct.sync) {
                const bool refused = (*ct.h_status & 1u) != 0u;
                if (ct.sync->n_seg > 1u) {  // restart segments in chunk slots: the slots are where they are, the segment table says what they hold
                    if (refused) ct.sync->n_chunks = 0u;
                } else {
                    ct.sync->n_bits = refused ? 0u : ct.seg_table[1] * 8u;
                    ct.sync->n_chunks = refused ? 0u : huff_sync_chunks(ct.seg_table[1], ct.sync->chunk_shift);
                }
            }
        };
        // Staging and upload in slices: while the host threads unstuff the scans of one slice into the pinned block, the
        // DMA engine carries the slice before (round 3: the whole block was staged, 2.5 ms for 256 x 1080p on the 16 CPUs the
        // box grants, and only then uploaded, 2 ms).  In front of the uploads, on the same stream, the fills: the planes start
        // as zeros (the Worker's zero-initialised plane: only non-zero coefficients are written; neighbouring images are
        // cleared with one fill — a fill per image was 1,024 tiny launches = 28 ms per 1,024 images) and so do the images' range
        // statistics.  They run while the host stages the first slice, next to nothing else of this sub-batch; on a second
        // stream (`copy_stream`) they and the uploads also stay clear of the kernels other sub-batches have in flight — behind
        // a fill on the kernels' own stream an upload waited for the machine to drain (2 of 7.5 ms per sub-batch).
        clk.mark("jobs");
        const bool two_streams = copy_stream && copy_stream != hip_stream;
        hipStream_t cps = two_streams ? (hipStream_t)copy_stream : s;
        if (two_streams && !b->entropy_uploaded) B_HIP(hipEventCreateWithFlags(&b->entropy_uploaded, hipEventDisableTiming));
        std::sort(stat_images.begin(), stat_images.end());
        for (size_t z = 0; z < stat_images.size();) {
            const size_t first = stat_images[z];
            size_t last = first;
            for (z++; z < stat_images.size() && stat_images[z] <= last + 1; z++) last = stat_images[z];
            B_HIP(hipMemsetAsync(b->d_stats + first * RS_WORDS, 0, (last - first + 1) * RS_WORDS * sizeof(uint32_t), cps));
        }
        std::sort(zero_ranges.begin(), zero_ranges.end());
        for (size_t z = 0; z < zero_ranges.size();) {
            size_t first = zero_ranges[z].first, last = zero_ranges[z].second;
            for (z++; z < zero_ranges.size() && zero_ranges[z].first <= last + 256; z++) last = std::max(last, zero_ranges[z].second);
            B_HIP(hipMemsetAsync(b->d_coef + first, 0, last - first, cps));
        }
        clk.mark("fills");
        // One parallel-for over all staging tasks; whoever finishes the last task of a slice (~8 MB of the data area) sends
        // that slice on its way.  (A parallel-for per slice spent more on starting threads than the overlap gave back.)
        const uint32_t n_tasks = (uint32_t)copies.size();
        static const uint32_t slice_shift = env_u32("JPGPU_STAGE_SLICE_SHIFT", 23, 20, 31);  // tuning knob: bytes per slice = 1 << this
        const uint32_t n_slices = std::max<uint32_t>(1u, std::min<uint32_t>({16u, n_tasks, (uint32_t)(data_bytes >> slice_shift) + 1u}));
        std::vector<std::atomic<uint32_t>> left(n_slices);
        auto slice_first = [&](uint32_t g) { return (uint32_t)((uint64_t)n_tasks * g / n_slices); };
        auto slice_of = [&](uint32_t t) {
            uint32_t g = (uint32_t)(((uint64_t)t * n_slices) / n_tasks);
            while (g + 1u < n_slices && slice_first(g + 1u) <= t) g++;
            while (g > 0u && slice_first(g) > t) g--;
            return g;
        };
        for (uint32_t g = 0; g < n_slices; g++) left[g].store(slice_first(g + 1u) - slice_first(g));
        std::atomic<int> copy_failed{0};
        std::atomic<uint32_t> max_copy_us{0}, max_task_us{0};
        const int device = b->device;
        const std::function<void(uint32_t)> staged = [&](uint32_t t) {
            const auto b0 = std::chrono::steady_clock::now();
            body(t);
            if (input_pinned && !copies[t].raw) {  // (its slots, out of the pinned block into the mirror)
                size_t bytes = 0;
                for (size_t sg = 0; sg + 1 < copies[t].ps->seg_off.size(); sg += 2) bytes += huff_slot_bytes(copies[t].ps->seg_off[sg + 1] - copies[t].ps->seg_off[sg]);
                if (hipSetDevice(device) != hipSuccess || upload_staged(d + off_mirror + copies[t].dst_off, copies[t].dst, bytes, cps) != hipSuccess) copy_failed.store(1);
            }
            if (clk.on) {
                const uint32_t us = (uint32_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - b0).count();
                uint32_t cur = max_task_us.load();
                while (us > cur && !max_task_us.compare_exchange_weak(cur, us)) {}
            }
            const uint32_t g = slice_of(t);
            if (left[g].fetch_sub(1u) == 1u) {  // the slice is complete
                const uint32_t t0 = slice_first(g), t1 = slice_first(g + 1u);
                const size_t lo = off_data + copies[t0].dst_off, hi = t1 < n_tasks ? off_data + copies[t1].dst_off : total;
                const auto c0 = std::chrono::steady_clock::now();
                // (light: into the mirror.  With pinned input there are no slice uploads: the raw scans have gone up on their own, straight
                // from the caller's buffers, and a slice's copy out of the pinned block would overwrite them with whatever that block
                // holds — the scans the host staged itself, restart segments, go up one by one as well: `staged` below)
                if (!input_pinned && (hipSetDevice(device) != hipSuccess || upload_staged(d + (light ? off_mirror + (lo - off_data) : lo), h + lo, hi - lo, cps) != hipSuccess))
                    copy_failed.store(1);
                if (clk.on) {
                    const uint32_t us = (uint32_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
                    uint32_t cur = max_copy_us.load();
                    while (us > cur && !max_copy_us.compare_exchange_weak(cur, us)) {}
                }
            }
        };
        if (input_pinned) {
            // one caller, one copy per span of adjacent files (4,096 hipMemcpyAsync calls from a team of sixteen threads made the call
            // twice as long as from one: 108 against 53 ms; and from one thread on 16 CPUs still 114 ms)
            for (const PinnedSpan &sp : spans)
                if (hipMemcpyAsync(d + off_spans + sp.mirror_off, sp.start, (size_t)(sp.end - sp.start), hipMemcpyHostToDevice, raw_stream) != hipSuccess) raw_copy_failed = true;
        }
        if (par && n_tasks > 1) (*par)(n_tasks, staged);
        else
            for (uint32_t t = 0; t < n_tasks; t++) staged(t);
        clk.mark("staging+uploads");
        if (clk.on) clk.used += (size_t)snprintf(clk.text + clk.used, clk.used < sizeof(clk.text) ? sizeof(clk.text) - clk.used : 0, " (slowest staging task %.2f, slowest hipMemcpyAsync call %.2f)", max_task_us.load() / 1e3, max_copy_us.load() / 1e3);
        if (copy_failed.load() || raw_copy_failed) return set_err(b->err, JPGPU_ERR_IO, "device entropy: upload of the staged scans failed");
        // the head of the block last: the staging tasks wrote into its job records (unstuffed lengths, chunk counts, status)
        B_HIP(upload_staged(d, h, off_data, cps));
        if (two_streams) {
            B_HIP(hipEventRecord(b->entropy_uploaded, cps));
            B_HIP(hipStreamWaitEvent(s, b->entropy_uploaded, 0));
        }
    }
    clk.mark("head");
    // JPGPU_BATCH_KERNEL_TIMES: events between the phases (fills | sync passes | write pass + DC sums | pixel kernels)
    static const bool phase_times = getenv("JPGPU_BATCH_KERNEL_TIMES") != nullptr;
    b->phase_events_valid = false;
    b->progressive_launch = false;
    if (phase_times) {
        for (auto &e : b->ev_phase)
            if (!e) B_HIP(hipEventCreate(&e));
        B_HIP(hipEventRecord(b->ev_phase[0], s));
    }
    if (phase_times) B_HIP(hipEventRecord(b->ev_phase[1], s));
    if (n_raw_jobs) B_HIP(launch_huff_unstuff(reinterpret_cast<const UnstuffJob *>(d + off_ujobs), (uint32_t)n_raw_jobs, max_pieces, s));
    if (n_light) *n_light = light_images;
    B_HIP(launch_huff_sync(reinterpret_cast<const HuffSyncJob *>(d + off_sjobs), (uint32_t)n_sync_jobs, max_chunks, sync_launches, sync_iters, s,
                           phase_times ? b->ev_phase[2] : nullptr, low_table_ids, reinterpret_cast<const EntryIndexJob *>(d + off_ijobs), n_index, max_index_items));
    if (n_entry) *n_entry = n_index;
    if (n_index) {
        b->entries_pending = true;
        b->d_entry_srcs = reinterpret_cast<const EntrySrc *>(d + off_esrc);
        b->d_entry_status = reinterpret_cast<const uint32_t *>(d + off_status);
        b->entry_status_n = n;
    }
    if (phase_times) {
        B_HIP(hipEventRecord(b->ev_phase[3], s));
        b->phase_events_valid = true;
    }
    clk.mark("kernels");
    // the status words into pinned memory (the only thing the host needs to look at: which images it has to decode itself)
    B_HIP(batch_status_to_host(b, reinterpret_cast<const uint32_t *>(d + off_status), n, s));
    clk.mark("status copy");
    return JPGPU_OK;
}

// JPGPU_BATCH_KERNEL_TIMES: milliseconds of the phases of the last device entropy launch and of the decode that followed it on
// the same stream ([0] fills, [1] restart-segment decoder + sync passes + block numbering, [2] write pass + DC sums,
// [3] class finalize + pixel kernels); false if they were not recorded.  The stream must have been synchronised.
bool jpgpu::batch_phase_times(jpgpu_batch *b, float ms[4]) {
    if (!b || !b->phase_events_valid) return false;
    for (int i = 0; i < 4; i++) ms[i] = 0.f;
    bool ok = hipEventElapsedTime(&ms[0], b->ev_phase[0], b->ev_phase[1]) == hipSuccess;
    ok = ok && hipEventElapsedTime(&ms[1], b->ev_phase[1], b->ev_phase[2]) == hipSuccess;
    ok = ok && hipEventElapsedTime(&ms[2], b->ev_phase[2], b->ev_phase[3]) == hipSuccess;
    if (ok && hipEventQuery(b->ev_phase[5]) == hipSuccess) ok = hipEventElapsedTime(&ms[3], b->ev_phase[4], b->ev_phase[5]) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}

// JPGPU_PIPE_TRACE: when the phase events of `b` fired, in milliseconds after `ref`'s first one (both recorded, streams synchronised)
bool jpgpu::batch_phase_stamps(jpgpu_batch *ref, jpgpu_batch *b, float ms[6]) {
    if (!ref || !b || !ref->phase_events_valid || !b->phase_events_valid) return false;
    bool ok = true;
    for (int i = 0; i < 6 && ok; i++) {
        ms[i] = -1.f;
        if (hipEventQuery(b->ev_phase[i]) == hipSuccess) ok = hipEventElapsedTime(&ms[i], ref->ev_phase[0], b->ev_phase[i]) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    return ok;
}

// ---- progressive frames on the device (huff_prog_wave.hpp) ------------------------------------------------------------------------
// Staging block (same offsets in the pinned and the device copy):
//   [ status: n x u32 | ProgTrack[] | ProgScan[] | ProgHuffTable[] | scan bytes ]     device only: the masks (16 bytes per block)
int jpgpu::batch_device_progressive_launch(jpgpu_batch *b, const DeviceProgressiveImage *images, uint32_t n, void *hip_stream,
                                           const std::function<void(uint32_t, const std::function<void(uint32_t)> &)> *par, void *copy_stream,
                                           DeviceScratch *scratch, bool allow_pipelined) {
    if (!b || !images || n == 0) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_coef) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    if (n != b->descs.size()) return set_err(b->err, JPGPU_ERR_FORMAT, "device progressive: every image of the batch must be listed");
    hipStream_t s = (hipStream_t)hip_stream;
    rc = batch_enable_dev_classes(b);
    if (rc) return rc;
    size_t n_scans = 0, n_tracks = 0, n_tables = 0, data_bytes = 0, mask_bytes = 0;
    for (uint32_t k = 0; k < n; k++) {
        if (images[k].image >= b->descs.size() || !images[k].plan || !images[k].file) return set_err(b->err, JPGPU_ERR_FORMAT, "device progressive: bad image");
        const host::ProgPlan &pl = *images[k].plan;
        n_scans += pl.scans.size();
        n_tracks += pl.n_tracks;
        for (const host::ProgPlannedScan &ps : pl.scans) {
            for (int t = 0; t < 4; t++)
                if (ps.table[t]) n_tables++;
            data_bytes += huff_slot_bytes(ps.stuffed_bytes);
        }
        const jpgpu_image_desc &desc = b->descs[images[k].image];
        for (uint32_t c = 0; c < desc.ncomp; c++) mask_bytes += align_up(b->coef_len[(size_t)images[k].image * 4 + c] / 128 * 16, 256);
    }
    // One WAVE per scan (huff_prog_wave.hpp), the scans of a track pipelined (huff_prog_job.hpp): a progress word each (behind the masks,
    // zeroed with them); the launch order below keeps a frame's waves on one XCD, producers in front, so an oversubscribed launch
    // cannot starve a producer.  JPGPU_PROG_SERIAL=1 / !allow_pipelined (tests): a wave per TRACK, its scans one after the other.
    // (Round 5 walked a LANE per scan — huff_prog_core.hpp, in the git history: 3 x slower at 256 frames, 25 % at 4,096 distinct ones.)
    static const bool serial_env = getenv("JPGPU_PROG_SERIAL") != nullptr;
    const bool serial_tracks = serial_env || !allow_pipelined;
    const size_t progress_off = mask_bytes;
    mask_bytes += align_up(n_scans * 4u, 256);
    // (eight lists that differ by less than one frame's 256 scans)
    const size_t max_lanes = n_tracks + n_scans + 64u * 64u;
    n_tracks = max_lanes;
    const size_t off_status = 0, off_tracks = align_up((size_t)n * 4, 16), off_scans = align_up(off_tracks + n_tracks * sizeof(ProgTrack), 16);
    const size_t off_tables = align_up(off_scans + n_scans * sizeof(ProgScan), 16), off_data = align_up(off_tables + n_tables * sizeof(ProgHuffTable), 16);
    const size_t total = off_data + data_bytes;
    const size_t off_masks = align_up(total, 256), total_dev = scratch ? total : off_masks + mask_bytes;
    if (scratch && mask_bytes > scratch->cap) {
        if (scratch->d) (void)hipFree(scratch->d);
        scratch->d = nullptr;
        scratch->cap = 0;
        B_HIP(hipMalloc((void **)&scratch->d, mask_bytes + mask_bytes / 4));
        scratch->cap = mask_bytes + mask_bytes / 4;
    }
    if (total_dev > b->entropy_cap) {
        if (b->d_entropy) (void)hipFree(b->d_entropy);
        b->d_entropy = nullptr;
        b->entropy_cap = 0;
        B_HIP(hipMalloc((void **)&b->d_entropy, total_dev + total_dev / 4));
        b->entropy_cap = total_dev + total_dev / 4;
    }
    if (total > b->entropy_host_cap) {
        if (b->h_entropy) (void)hipHostFree(b->h_entropy);
        b->h_entropy = nullptr;
        b->entropy_host_cap = 0;
        B_HIP(hipHostMalloc((void **)&b->h_entropy, total + total / 4, hipHostMallocDefault));
        b->entropy_host_cap = total + total / 4;
    }
    if ((size_t)n > b->entropy_out_cap) {
        if (b->h_entropy_out) (void)hipHostFree(b->h_entropy_out);
        b->h_entropy_out = nullptr;
        B_HIP(hipHostMalloc((void **)&b->h_entropy_out, ((size_t)n + 64) * 4, hipHostMallocDefault));
        b->entropy_out_cap = (size_t)n + 64;
    }
    uint8_t *h = b->h_entropy, *d = b->d_entropy, *dm = scratch ? scratch->d : d + off_masks;
    memset(h, 0, off_tracks);  // status words
    ProgTrack *tracks = reinterpret_cast<ProgTrack *>(h + off_tracks);
    ProgScan *scans = reinterpret_cast<ProgScan *>(h + off_scans);
    struct StageTask {
        const host::ProgPlannedScan *ps;
        const uint8_t *src;
        uint8_t *dst;       // the scan's slot in the pinned block
        uint8_t *tables;    // where its tables go in the pinned block
        ProgScan *scan;
        uint32_t *h_status;
    };
    std::vector<StageTask> tasks;
    tasks.reserve(n_scans);
    struct TrackOrder {
        uint32_t first_scan, n_scans, image_k;
        uint64_t weight;  // bytes of entropy-coded data the lane walks
        uint32_t rank;    // pipelined scans: how many scans deep its dependencies go (0: none); serial tracks: 0
        uint32_t kind;    // pipelined scans: which scan of its frame's script it is (band, approximation, first component); serial tracks: 0
        uint64_t chain;   // pipelined scans: the place of this scan's track among its frame's tracks by the bytes along the longest chain of dependent scans (0: the longest; waves: the launch order)
    };
    std::vector<TrackOrder> order;
    order.reserve(n_tracks);
    size_t si = 0, tcur = off_tables, dcur = off_data, mcur = 0;
    b->entropy_images.clear();
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t img = images[k].image;
        b->entropy_images.push_back(img);
        const jpgpu_image_desc &desc = b->descs[img];
        const host::ProgPlan &pl = *images[k].plan;
        uint64_t *mask_of[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t c = 0; c < desc.ncomp; c++) {
            mask_of[c] = reinterpret_cast<uint64_t *>(dm + mcur);
            mcur += align_up(b->coef_len[(size_t)img * 4 + c] / 128 * 16, 256);
        }
        // Which scans does a scan depend on?  For every coefficient it covers, the LAST earlier scan that covered it (that one waited
        // for its own predecessors block by block, so staying behind it is staying behind them all).  More than three, or more than
        // 64 levels: the frame's tracks are walked serially, one lane each.
        const size_t first_si = si;
        const uint32_t ns = (uint32_t)pl.scans.size();
        const host::ProgDependencies pd = serial_tracks ? host::ProgDependencies{} : host::prog_plan_dependencies(pl);
        const bool pipelined = !serial_tracks && pd.ok;
        const std::vector<uint32_t> &rank = pd.rank;
        const std::vector<std::array<int32_t, 3>> &deps = pd.deps;
        // the scans, in stream order (serial tracks: grouped by track, each track's scans contiguous)
        std::vector<uint32_t> scan_order;
        if (pipelined) {
            for (uint32_t j = 0; j < ns; j++) scan_order.push_back(j);
        } else {
            for (uint32_t t = 0; t < pl.n_tracks; t++)
                for (uint32_t j = 0; j < ns; j++)
                    if (pl.scans[j].track == t) scan_order.push_back(j);
        }
        std::vector<size_t> si_of(ns, 0);
        for (uint32_t j : scan_order) {
            const host::ProgPlannedScan &ps = pl.scans[j];
            ProgScan &sc = scans[si];
            si_of[j] = si;
            memset(&sc, 0, sizeof(sc));
            sc.data = d + dcur;
            sc.ss = ps.ss, sc.se = ps.se, sc.ah = ps.ah, sc.al = ps.al;
            sc.ncomp = ps.ncomp, sc.cols = ps.cols, sc.rows = ps.rows;
            for (uint32_t c = 0; c < ps.ncomp; c++) {
                const uint32_t fi = ps.comp[c].frame_index;
                if (fi >= desc.ncomp || ps.comp[c].block_w != desc.components[fi].block_width)
                    return set_err(b->err, JPGPU_ERR_FORMAT, "device progressive: plan does not match the image descriptor");
                sc.comp[c].coefs = reinterpret_cast<int16_t *>(b->d_coef + b->coef_off[(size_t)img * 4 + fi]);
                sc.comp[c].masks = mask_of[fi];
                sc.comp[c].block_w = ps.comp[c].block_w;
                sc.comp[c].h = ps.comp[c].h;
                sc.comp[c].v = ps.comp[c].v;
                sc.comp[c].table = ps.comp[c].table;
            }
            StageTask st{&ps, images[k].file + ps.data_off, h + dcur, h + tcur, &sc, reinterpret_cast<uint32_t *>(h + off_status) + k};
            for (int tb = 0; tb < 4; tb++)
                if (ps.table[tb]) {
                    sc.table[tb] = reinterpret_cast<const ProgHuffTable *>(d + tcur);
                    tcur += sizeof(ProgHuffTable);
                }
            tasks.push_back(st);
            dcur += huff_slot_bytes(ps.stuffed_bytes);
            si++;
        }
        if (pipelined) {
            for (uint32_t j = 0; j < ns; j++) {
                ProgScan &sc = scans[si_of[j]];
                sc.progress = reinterpret_cast<uint32_t *>(dm + progress_off) + si_of[j];
                for (uint32_t w = 0; w < 3u; w++) {
                    if (deps[j][w] < 0) continue;
                    sc.wait[w] = reinterpret_cast<const uint32_t *>(dm + progress_off) + si_of[(uint32_t)deps[j][w]];
                    if (!host::prog_same_walk(pl.scans[j], pl.scans[(uint32_t)deps[j][w]])) sc.wait_whole |= 1u << w;  // (else block for block)
                }
                const host::ProgPlannedScan &pj = pl.scans[j];
                const uint32_t kind = ((uint32_t)pj.ss << 24) | ((uint32_t)pj.se << 16) | ((uint32_t)pj.ah << 12) | ((uint32_t)pj.al << 8) | (pj.comp[0].frame_index << 4) | pj.ncomp;
                order.push_back(TrackOrder{(uint32_t)si_of[j], 1u, k, pj.stuffed_bytes, rank[j], kind, 0u});
            }
            {   // the longest chain of every track: a scan's bytes + the longest chain of the scans that wait for it, the maximum per track
                std::vector<uint64_t> down(ns, 0u), best(pl.n_tracks + 1u, 0u);
                for (uint32_t j = ns; j-- > 0;) {  // (consumers come later in the stream: down[j] is final when the loop reaches j)
                    down[j] += pl.scans[j].stuffed_bytes;
                    for (uint32_t w = 0; w < 3u; w++)
                        if (deps[j][w] >= 0) down[(uint32_t)deps[j][w]] = std::max(down[(uint32_t)deps[j][w]], down[j]);
                }
                for (uint32_t j = 0; j < ns; j++) {
                    uint64_t &bt = best[std::min<uint32_t>(pl.scans[j].track, pl.n_tracks)];
                    bt = std::max(bt, down[j]);
                }
                // ... as the track's PLACE among the frame's tracks (0: the longest chain): frames differ in their bytes, the places compare
                for (uint32_t j = 0; j < ns; j++) {
                    const uint64_t mine = best[std::min<uint32_t>(pl.scans[j].track, pl.n_tracks)];
                    uint32_t place = 0;
                    for (uint32_t t = 0; t < pl.n_tracks; t++)
                        if (best[t] > mine || (best[t] == mine && t < std::min<uint32_t>(pl.scans[j].track, pl.n_tracks))) place++;
                    order[order.size() - ns + j].chain = place;
                }
            }
        } else {
            size_t at = first_si;
            for (uint32_t t = 0; t < pl.n_tracks; t++) {
                TrackOrder to{(uint32_t)at, 0u, k, 0u, 0u, 0u, 0u};
                for (uint32_t j = 0; j < ns; j++)
                    if (pl.scans[j].track == t) {
                        to.weight += pl.scans[j].stuffed_bytes;
                        to.n_scans++;
                    }
                at += to.n_scans;
                if (to.n_scans) order.push_back(to);
            }
        }
        for (uint32_t c = 0; c < desc.ncomp; c++) {  // the classes of the finished planes: from the range scan below
            b->sane[(size_t)img * 4 + c] = 0;
            batch_class_source(b, (size_t)img * 4 + c, true);
        }
    }
    // Lanes in launch order: by dependency rank (producers in front: workgroups are dispatched in order, so whatever a lane waits for is
    // resident or done), every rank starting a wave of its own (a lane never waits for a lane of its own wave), and inside a rank like
    // with like — the same scan of the frames' scripts side by side (a wave whose lanes walk DC, first AC and refinement scans runs the
    // three loops one after the other), heavy lanes first: the 64 lanes of a wave walk scans of the same kind and about the same length
    std::stable_sort(order.begin(), order.end(), [](const TrackOrder &a, const TrackOrder &c) {
        return a.rank != c.rank ? a.rank < c.rank : (a.kind != c.kind ? a.kind < c.kind : a.weight > c.weight);
    });
    size_t n_lanes = 0;
    const auto entry_of = [&](const TrackOrder &o) {
        return ProgTrack{reinterpret_cast<const ProgScan *>(d + off_scans) + o.first_scan, o.n_scans, reinterpret_cast<uint32_t *>(d + off_status) + o.image_k};
    };
    {
        // Waves in launch order (huff.hip, huff_progw_kernel): workgroup i runs on XCD i mod 8 and every XCD dispatches its workgroups in
        // order.  So: every frame's waves on ONE XCD (the frame with the fewest waves so far takes the next frame: lists of equal length),
        // and inside an XCD's list by dependency rank, in groups of JPGPU_PROG_GROUP frames (default: all — rank-major: the waves of a
        // rank run at full occupancy before the next rank's are dispatched; a wave that catches up with its producer sleeps), inside a
        // rank the heavy scans first.  A producer is in front of its consumers in the list of their XCD: when a consumer runs, the
        // producer is resident or done — no deadlock however many waves the launch has, and a frame's planes and masks stay in one L2.
        static const uint32_t group = getenv("JPGPU_PROG_GROUP") ? (uint32_t)std::max(1, atoi(getenv("JPGPU_PROG_GROUP"))) : 0x7fffffffu;
        constexpr uint32_t XCDS = 8u;
        std::vector<uint32_t> xcd_of(n, 0u), seq_of(n, 0u);  // per listed image: its XCD, its number among that XCD's frames
        {
            std::vector<uint32_t> per_image(n, 0u);
            for (const TrackOrder &o : order) per_image[o.image_k]++;
            size_t load[XCDS] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t frames[XCDS] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t k = 0; k < n; k++) {
                uint32_t x = 0;
                for (uint32_t j = 1; j < XCDS; j++)
                    if (load[j] < load[x]) x = j;
                xcd_of[k] = x;
                seq_of[k] = frames[x]++;
                load[x] += per_image[k];
            }
        }
        std::vector<TrackOrder> lists[XCDS];
        for (const TrackOrder &o : order) lists[xcd_of[o.image_k]].push_back(o);
        size_t longest = 0;
        for (auto &l : lists) {
            // (round 6, second half: the tracks with the longest chains FIRST — Y's first scans, their refinements, then the short tracks:
            // with ranks only, the last scans of the long chains were dispatched last and the launch ended with a 9 ms tail of a few
            // waves per SIMD; scans of one track share `chain`, and inside a track the rank keeps producers in front.  JPGPU_PROG_ORDER=rank: as before)
            static const bool by_rank = getenv("JPGPU_PROG_ORDER") && !strcmp(getenv("JPGPU_PROG_ORDER"), "rank");
            std::stable_sort(l.begin(), l.end(), [&](const TrackOrder &a, const TrackOrder &c) {
                const uint32_t ga = seq_of[a.image_k] / group, gc = seq_of[c.image_k] / group;
                if (ga != gc) return ga < gc;
                if (!by_rank && a.chain != c.chain) return a.chain < c.chain;
                return a.rank != c.rank ? a.rank < c.rank : a.weight > c.weight;
            });
            longest = std::max(longest, l.size());
        }
        if (longest * XCDS > max_lanes) return set_err(b->err, JPGPU_ERR_INTERNAL, "device progressive: wave table");
        for (size_t j = 0; j < longest; j++)
            for (uint32_t x = 0; x < XCDS; x++) tracks[n_lanes++] = j < lists[x].size() ? entry_of(lists[x][j]) : ProgTrack{nullptr, 0u, nullptr};
    }
    const bool two_streams = copy_stream && copy_stream != hip_stream;
    hipStream_t cps = two_streams ? (hipStream_t)copy_stream : s;
    if (two_streams && !b->entropy_uploaded) B_HIP(hipEventCreateWithFlags(&b->entropy_uploaded, hipEventDisableTiming));
    // zeros: the planes (the Worker's zero-initialised plane: src/decoder.rs:400-412), the masks, the statistics
    // (the masks on the kernels' own stream: with a caller's `scratch` they are shared by the launches of that stream, which run one
    // after the other there — a fill on the copy stream would run into the previous launch's walk)
    B_HIP(hipMemsetAsync(b->d_coef, 0, b->coef_bytes, cps));
    B_HIP(hipMemsetAsync(dm, 0, mask_bytes, s));
    const std::function<void(uint32_t)> stage = [&](uint32_t t) {
        const StageTask &st = tasks[t];
        bool clean = true;
        st.scan->n_bytes = huff_stage_segment(st.dst, st.src, st.ps->stuffed_bytes, &clean);
        if (!clean) *st.h_status |= PROG_ST_HOST | PROG_ST_STAGING;  // (cannot happen: the planner walked the same bytes)
        uint8_t *tp = st.tables;
        for (int tb = 0; tb < 4; tb++)
            if (st.ps->table[tb]) {
                memcpy(tp, st.ps->table[tb].get(), sizeof(ProgHuffTable));
                tp += sizeof(ProgHuffTable);
            }
    };
    if (par && tasks.size() > 1) (*par)((uint32_t)tasks.size(), stage);
    else
        for (uint32_t t = 0; t < tasks.size(); t++) stage(t);
    B_HIP(upload_staged(d, h, total, cps));
    if (two_streams) {
        B_HIP(hipEventRecord(b->entropy_uploaded, cps));
        B_HIP(hipStreamWaitEvent(s, b->entropy_uploaded, 0));
    }
    // events around the track kernel: always (the pipeline's dispatcher learns the device's latency from them), and the phase events of
    // JPGPU_BATCH_KERNEL_TIMES ([1]..[2] "sync" = the track kernel, [2]..[3] "write" = the range scan)
    for (auto &e : b->ev_phase)
        if (!e) B_HIP(hipEventCreate(&e));
    B_HIP(hipEventRecord(b->ev_phase[0], s));
    B_HIP(hipEventRecord(b->ev_phase[1], s));
    B_HIP(launch_huff_progw(reinterpret_cast<const ProgTrack *>(d + off_tracks), (uint32_t)n_lanes, s));
    B_HIP(hipEventRecord(b->ev_phase[2], s));
    if (getenv("JPGPU_PROG_TIMES")) {  // (debugging aid: a synchronisation inside the launch)
        B_HIP(hipStreamSynchronize(s));
        const uint32_t ns = (uint32_t)images[0].plan->scans.size();
        std::vector<ProgScan> back(ns);
        B_HIP(hipMemcpy(back.data(), d + off_scans, ns * sizeof(ProgScan), hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < ns; j++)
            fprintf(stderr, "prog times: frame 0 scan %u (ss %u se %u ah %u al %u, %u bytes): %.3f ms\n", j, back[j].ss, back[j].se, back[j].ah, back[j].al, back[j].n_bytes, back[j].report[0] * 1e-5);
        for (uint32_t j = 0; j < ns; j++)
            if (back[j].ah && back[j].ss) fprintf(stderr, "prog times: frame 0 scan %u: %u calls of the hand-scheduled loop, %u symbols on the portable path, %u window switches\n", j, back[j].report[1], back[j].report[2], back[j].report[3]);
    }
    {
        const int crc = jpgpu_batch_classify_on_device(b, s);
        if (crc) return crc;
    }
    B_HIP(hipEventRecord(b->ev_phase[3], s));
    b->phase_events_valid = true;
    b->progressive_launch = true;
    B_HIP(batch_status_to_host(b, reinterpret_cast<const uint32_t *>(d + off_status), n, s));
    return JPGPU_OK;
}

bool jpgpu::batch_progressive_kernel_ms(jpgpu_batch *b, float *ms) {
    if (!b || !ms || !b->progressive_launch || !b->ev_phase[1] || !b->ev_phase[2]) return false;
    if (hipEventElapsedTime(ms, b->ev_phase[1], b->ev_phase[2]) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

int jpgpu::batch_device_entropy_collect(jpgpu_batch *b, uint32_t *status, uint32_t n) {
    if (!b || !status || n != b->entropy_images.size()) return JPGPU_ERR_FORMAT;
    // (the classes of the accepted images stay on the device: d_stats; an image the host decodes instead gets its class with
    // its upload)
    for (uint32_t k = 0; k < n; k++) status[k] = b->h_entropy_out[k];
    b->entropy_images.clear();
    return JPGPU_OK;
}

extern "C" {

int jpgpu_batch_upload_compact(jpgpu_batch *b, uint32_t image, uint32_t comp, const void *compact, size_t bytes,
                               int range_class, void *hip_stream) {
    return jpgpu::batch_upload_compact(b, image, comp, compact, bytes, range_class, hip_stream, false);
}

// compact uploads since the last decode -> dense coefficient arena, on the decode stream
static int batch_expand_pending(jpgpu_batch *b, hipStream_t s) {
    std::vector<ExpandJob> jobs;
    std::vector<uint32_t> stat_fresh;
    uint32_t max_blocks = 0;
    {
        std::lock_guard<std::mutex> g(b->compact_mutex);
        if (!b->any_compact_pending) return JPGPU_OK;
        // An image's statistics start afresh when every component they stand for is being re-sent now; otherwise they only
        // grow (sound, possibly pessimistic).
        for (size_t img = 0; img < b->descs.size(); img++) {
            bool any = false, all = true;
            for (uint32_t c = 0; c < b->descs[img].ncomp; c++) {
                const size_t idx = img * 4 + c;
                if (b->compact_pending[idx] == 2 && b->cls_src[idx]) any = true;
                else if (b->cls_src[idx]) all = false;
            }
            if (any && all) stat_fresh.push_back((uint32_t)img);
        }
        for (size_t idx = 0; idx < b->compact_pending.size(); idx++)
            if (b->compact_pending[idx]) {
                ExpandJob j{};
                j.compact = b->d_compact + b->compact_off[idx];
                j.dense = reinterpret_cast<int16_t *>(b->d_coef + b->coef_off[idx]);
                j.n_blocks = (uint32_t)(b->coef_len[idx] / 128);
                if (b->compact_pending[idx] == 2 && b->cls_src[idx]) {  // unclassified by the sender: ranged on the way
                    j.qt = b->d_qt + idx * 64;
                    j.stats = b->d_stats + (idx / 4) * RS_WORDS;
                }
                max_blocks = std::max(max_blocks, j.n_blocks);
                jobs.push_back(j);
                b->compact_pending[idx] = 0;
            }
        b->any_compact_pending = false;
    }
    if (jobs.empty()) return JPGPU_OK;
    for (uint32_t img : stat_fresh) B_HIP(hipMemsetAsync(b->d_stats + (size_t)img * RS_WORDS, 0, RS_WORDS * sizeof(uint32_t), s));
    B_HIP(hipMemcpy(b->d_expand_jobs, jobs.data(), jobs.size() * sizeof(ExpandJob), hipMemcpyHostToDevice));
    B_HIP(launch_expand_compact(b->d_expand_jobs, (uint32_t)jobs.size(), max_blocks, s));
    return JPGPU_OK;
}

int jpgpu_batch_decode(jpgpu_batch *b, void *hip_stream) {
    if (!b) return JPGPU_ERR_FORMAT;
    jpgpu::TraceRange roctx_range("jpgpu_batch_decode");
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    rc = batch_refresh_jobs(b, s);
    if (rc) return rc;
    rc = batch_expand_pending(b, s);
    if (rc) return rc;
    if (b->phase_events_valid) B_HIP(hipEventRecord(b->ev_phase[4], s));
    // device-side classes: statistics -> class bits in the launch tables (class_finalize_*), then the `_dyn` kernels
    const uint32_t *st = b->dev_classes ? b->d_stats : nullptr;
    const uint8_t *hc = b->dev_classes ? b->d_host_cls : nullptr;
    auto entry_images_of = [&](const FusedPlan &fp) {
        uint32_t e = 0;
        if (fp.kind == FUSED_420 && fp.strip && !b->entry_img.empty())
            for (uint32_t id : fp.ids) e += b->entry_img[id];
        return e;
    };
    if (b->entries_pending) {  // the walk that reads the entry lists of the launch in front of this decode, and what it flagged
        b->entries_pending = false;
        for (FusedPlan &fp : b->fused)
            if (entry_images_of(fp)) B_HIP(fused_launch_entries(fp, s, b->d_entry_srcs));
        B_HIP(batch_status_to_host(b, b->d_entry_status, b->entry_status_n, s));
    }
    for (FusedPlan &fp : b->fused)
        if (entry_images_of(fp) < fp.n_images) B_HIP(fused_launch(fp, s, st, hc));  // (a plan of entry-list images only has nothing for the dense kernels)
    if (!b->scaled_ids.empty())
        B_HIP(launch_scaled_fused(b->d_scaled_geoms, b->d_s_image_jobs, b->d_s_plane_jobs, (uint32_t)b->scaled_ids.size(), b->s_max_tiles_x, b->s_max_bands,
                                  b->s_lds_bytes, b->s_scales, s));
    if (!b->generic_ids.empty()) {
        const uint32_t n = (uint32_t)b->image_jobs.size();
        if (b->dev_classes) B_HIP(launch_class_finalize_planes(b->d_plane_jobs, b->d_plane_job_slot, (uint32_t)b->plane_jobs.size(), st, hc, s));
        static const uint32_t kScales[4] = {8, 4, 2, 1};
        for (uint32_t sc : kScales)
            if (b->scales[sc]) B_HIP(launch_idct_planes(b->d_plane_jobs, (uint32_t)b->plane_jobs.size(), b->max_blocks, sc, s));
        B_HIP(launch_upsample_color(b->d_image_jobs, n, b->max_w, b->max_h, s));
    }
    if (b->phase_events_valid) B_HIP(hipEventRecord(b->ev_phase[5], s));
    return JPGPU_OK;
}

int jpgpu_batch_synchronize(jpgpu_batch *b, void *hip_stream) {
    if (!b) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    B_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    return JPGPU_OK;
}

int jpgpu_batch_download(jpgpu_batch *b, uint32_t image, uint8_t *dst, size_t cap, size_t *len) {
    if (!b) return JPGPU_ERR_FORMAT;
    if (image >= b->descs.size()) return set_err(b->err, JPGPU_ERR_FORMAT, "download: bad image");
    const size_t n = b->out_len[image];
    if (len) *len = n;
    if (!dst || cap < n) return set_err(b->err, JPGPU_ERR_FORMAT, "download: destination too small");
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    if (!b->d_out) return set_err(b->err, JPGPU_ERR_FORMAT, "batch has no device buffers bound");
    B_HIP(hipDeviceSynchronize());
    if (n == 0) return JPGPU_OK;
    // straight into pageable memory the copy runs at 0.5 GB/s (the runtime pins the destination page by page): bounce
    // through pinned memory unless the caller's buffer is pinned itself
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned || n < (64u << 10)) {
        B_HIP(hipMemcpy(dst, b->d_out + b->out_off[image], n, hipMemcpyDeviceToHost));
        return JPGPU_OK;
    }
    if (b->h_bounce_cap < n) {
        if (b->h_bounce) (void)hipHostFree(b->h_bounce);
        b->h_bounce = nullptr;
        b->h_bounce_cap = 0;
        B_HIP(hipHostMalloc((void **)&b->h_bounce, n + n / 4, hipHostMallocDefault));
        b->h_bounce_cap = n + n / 4;
    }
    B_HIP(hipMemcpy(b->h_bounce, b->d_out + b->out_off[image], n, hipMemcpyDeviceToHost));
    memcpy(dst, b->h_bounce, n);
    return JPGPU_OK;
}

int jpgpu_batch_time(jpgpu_batch *b, void *hip_stream, uint32_t iters, float *ms_per_decode) {
    if (!b || !ms_per_decode || iters == 0) return JPGPU_ERR_FORMAT;
    int rc = use_device(b->device, b->err);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    rc = jpgpu_batch_decode(b, hip_stream);  // warm-up + job upload
    if (rc) return rc;
    B_HIP(hipStreamSynchronize(s));
    B_HIP(hipEventRecord(b->ev0, s));
    for (uint32_t i = 0; i < iters; i++) {
        rc = jpgpu_batch_decode(b, hip_stream);
        if (rc) return rc;
    }
    B_HIP(hipEventRecord(b->ev1, s));
    B_HIP(hipEventSynchronize(b->ev1));
    float ms = 0.f;
    B_HIP(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *ms_per_decode = ms / (float)iters;
    return JPGPU_OK;
}

}  // extern "C"
