/*
 * jpgpu.h — C ABI of the MI355X (gfx950) pixel-pipeline backend for image-rs/jpeg-decoder.
 *
 * This is the drop-in boundary: exactly what a `src/worker/hip.rs` backend of the crate
 * binds over FFI (see INTEGRATION.md for the Rust `extern "C"` block).  Plain pointers and
 * sizes only; no C++ / torch types.  Every entry point cites the reference interface
 * (paths relative to the reference crate root, v0.3.2) it replaces.
 *
 * Semantics are those of the crate's *scalar* path (`--features platform_independent`):
 * integer IDCT of src/idct.rs, upsamplers of src/upsampler.rs, colour conversion of
 * src/decoder.rs:1391-1508 — bit-exact, including i32 wrap-around on hostile input.
 *
 * Threading: a context object (worker / batch / decoder) is used by one thread at a time
 * (the crate holds its worker behind a RefCell, src/worker/mod.rs:44-46); different
 * contexts may be used concurrently from any threads.  Nothing unwinds across this ABI.
 */
#ifndef JPGPU_H
#define JPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPGPU_MAX_COMPONENTS 4 /* src/decoder.rs:21 MAX_COMPONENTS */

/* Status codes. 1..4 map one-to-one onto src/error.rs:16-48 Error::{Format, Unsupported,
 * Io, Internal}; the Rust shim turns them back into that enum. */
enum {
    JPGPU_OK = 0,
    JPGPU_ERR_FORMAT = 1,      /* Error::Format(String)            */
    JPGPU_ERR_UNSUPPORTED = 2, /* Error::Unsupported(..)           */
    JPGPU_ERR_IO = 3,          /* Error::Io (incl. device/runtime) */
    JPGPU_ERR_INTERNAL = 4,    /* Error::Internal / would-panic    */
    JPGPU_ERR_NO_DEVICE = 5,   /* no usable gfx950 device (maps to Error::Io) */
};

/* ColorTransform, src/decoder.rs:76-98 (same order). */
enum {
    JPGPU_CT_NONE = 0,
    JPGPU_CT_UNKNOWN = 1,
    JPGPU_CT_GRAYSCALE = 2,
    JPGPU_CT_RGB = 3,
    JPGPU_CT_YCBCR = 4,
    JPGPU_CT_CMYK = 5,
    JPGPU_CT_YCCK = 6,
    JPGPU_CT_JCS_BG_YCC = 7,
    JPGPU_CT_JCS_BG_RGB = 8,
};

/* parser::Component, src/parser.rs:76-89 (what RowData carries, src/worker/mod.rs:18-22). */
typedef struct jpgpu_component {
    uint8_t identifier;
    uint8_t horizontal_sampling_factor;
    uint8_t vertical_sampling_factor;
    uint8_t quantization_table_index;
    uint32_t dct_scale;                  /* 8 (full), 4, 2 or 1 */
    uint16_t size_width, size_height;    /* size       */
    uint16_t block_width, block_height;  /* block_size */
} jpgpu_component;

/* ---- library ------------------------------------------------------------------------ */
const char *jpgpu_version(void);
int jpgpu_device_count(int *count);            /* number of visible HIP devices */
const char *jpgpu_status_string(int status);
/* Opt-in, once, BEFORE the process's first HIP call (the HIP runtime reads its environment when it initialises): sets
 * GPU_MAX_HW_QUEUES=24 unless the variable is set already — jpgpu_pipeline_decode runs its sub-batches side by side on ~20 streams,
 * which the runtime otherwise folds onto 4 hardware queues.  Returns 1 if it set the variable, 0 if not.
 * Loading the library has no such side effect. */
int jpgpu_process_init(void);

/* ---- Worker: trait Worker, src/worker/mod.rs:24-35 ----------------------------------- */
/* One worker per decode() (WorkerScope, src/worker/mod.rs:44-95).  Planes live in HBM. */
typedef struct jpgpu_worker jpgpu_worker;

int jpgpu_worker_create(int device, jpgpu_worker **out); /* WorkerScopeInner::Hip(Default) */
void jpgpu_worker_destroy(jpgpu_worker *w);
const char *jpgpu_worker_last_error(const jpgpu_worker *w);
/* Name of the kernels the last jpgpu_compute_image of this worker ran: "generic" (planes -> pixels) or the fused kernel
 * of the frame's kind ("fused420", ... — coefficients -> pixels in one launch, taken when every component reached the
 * frame as a complete plane of coefficients at full scale through finish_plane). Diagnostics / tests. */
const char *jpgpu_worker_last_path(const jpgpu_worker *w);
/* Range class (0, 1 or 3: see jpgpu_batch_set_range_hint) the fused kernel of the last jpgpu_compute_image ran with; -1 if
 * that call took the generic kernels.  The class is worked out on the device from the frame's coefficients (one scan at HBM
 * speed in front of the kernel, nothing read back); this call reads it back (blocking).  Diagnostics / tests. */
int jpgpu_worker_last_class(jpgpu_worker *w);

/* Worker::start(RowData{index, component, quantization_table}) — src/worker/mod.rs:25,
 * src/worker/rayon.rs:40-49.  `quantization_table` is in natural (un-zigzagged) order. */
int jpgpu_worker_start(jpgpu_worker *w, uint32_t index, const jpgpu_component *component,
                       const uint16_t quantization_table[64]);

/* Worker::append_row((index, Vec<i16>)) — src/worker/mod.rs:26, src/worker/rayon.rs:71-132.
 * `len` must equal block_width * vertical_sampling_factor * 64 (the reference asserts).
 * The data is consumed (copied to pinned staging) before the call returns. */
int jpgpu_worker_append_row(jpgpu_worker *w, uint32_t index, const int16_t *coefficients,
                            size_t len);

/* Worker::append_rows(iterator) — src/worker/mod.rs:29-34, src/worker/rayon.rs:140-185:
 * `n_rows` consecutive MCU rows stored back to back. */
int jpgpu_worker_append_rows(jpgpu_worker *w, uint32_t index, const int16_t *coefficients,
                             size_t n_rows);

/* Worker::get_result(index) -> Vec<u8> — src/worker/mod.rs:27, src/worker/rayon.rs:134-137.
 * Runs the IDCT of everything appended, copies the plane (block_w*block_h*dct_scale^2
 * bytes) to `dst`, and (mem::take) forgets the host-visible result; the device plane is
 * kept for jpgpu_compute_image under the same index until the next start(). */
int jpgpu_worker_get_result(jpgpu_worker *w, uint32_t index, uint8_t *dst, size_t cap,
                            size_t *len);

/* Device-resident variant: finish the plane but do not download it.  `plane_slot` is the
 * frame-level component position it will have in jpgpu_compute_image (decode_scan uses
 * scan-local indices, src/decoder.rs:848-852,1072-1075). */
int jpgpu_worker_finish_plane(jpgpu_worker *w, uint32_t index, uint32_t plane_slot);

/* compute_image, src/decoder.rs:1300-1336 == 1-component compaction + compute_image_parallel
 * (src/worker/mod.rs:97-128, src/worker/rayon.rs:193-219).  If host_planes is NULL the
 * device planes left by get_result / finish_plane are used (slot i = components[i]);
 * otherwise host_planes[i] (plane bytes as returned by get_result) are uploaded first.
 * Output: out_w*out_h*ncomp bytes (1 component: size_w*size_h). */
int jpgpu_compute_image(jpgpu_worker *w, const jpgpu_component *components, uint32_t ncomp,
                        const uint8_t *const *host_planes, uint16_t out_w, uint16_t out_h,
                        int color_transform, uint8_t *dst, size_t cap, size_t *len);

/* ---- Batch: N independent images per launch (no reference counterpart: the crate decodes
 * one image per Decoder; this is how a batch shards one-image-per-task across a GPU) ------ */
typedef struct jpgpu_image_desc {
    uint32_t ncomp;
    jpgpu_component components[JPGPU_MAX_COMPONENTS];
    uint16_t quantization_tables[JPGPU_MAX_COMPONENTS][64]; /* per component, natural order */
    uint16_t out_w, out_h;   /* FrameInfo::output_size */
    int32_t color_transform; /* determine_color_transform() result */
} jpgpu_image_desc;

typedef struct jpgpu_batch jpgpu_batch;

enum {
    JPGPU_BATCH_DEFAULT = 0,
    JPGPU_BATCH_EXTERNAL_BUFFERS = 1, /* caller binds device memory (e.g. torch tensors)   */
    JPGPU_BATCH_FORCE_GENERIC = 2,    /* never take the fused fast paths (two-kernel path)  */
    JPGPU_BATCH_ASSUME_HOSTILE = 4,   /* skip the range scan: always use the exact 32-bit path */
};

int jpgpu_batch_create(int device, const jpgpu_image_desc *descs, uint32_t n_images,
                       uint32_t flags, jpgpu_batch **out);
void jpgpu_batch_destroy(jpgpu_batch *b);
const char *jpgpu_batch_last_error(const jpgpu_batch *b);

/* Arena layout (bytes): coefficients of image i, component c start at coef_offset(i,c) and
 * hold block_w*block_h*64 int16 in block-raster order (the concatenation of that
 * component's append_row buffers); pixels of image i start at out_offset(i). */
size_t jpgpu_batch_coef_arena_bytes(const jpgpu_batch *b);
size_t jpgpu_batch_out_arena_bytes(const jpgpu_batch *b);
size_t jpgpu_batch_coef_offset(const jpgpu_batch *b, uint32_t image, uint32_t comp);
size_t jpgpu_batch_coef_bytes(const jpgpu_batch *b, uint32_t image, uint32_t comp);
size_t jpgpu_batch_out_offset(const jpgpu_batch *b, uint32_t image);
size_t jpgpu_batch_out_bytes(const jpgpu_batch *b, uint32_t image);

/* EXTERNAL_BUFFERS: device pointers owned by the caller (>= *_arena_bytes, 256-B aligned). */
int jpgpu_batch_bind(jpgpu_batch *b, void *device_coef_arena, void *device_out_arena);
void *jpgpu_batch_coef_arena(const jpgpu_batch *b); /* device pointer */
void *jpgpu_batch_out_arena(const jpgpu_batch *b);  /* device pointer */

/* Host -> HBM upload of one component's coefficients (also range-scans them, see
 * jpgpu_batch_set_range_hint). Blocking. */
int jpgpu_batch_upload(jpgpu_batch *b, uint32_t image, uint32_t comp, const int16_t *coefficients,
                       size_t len);
/* For coefficients written straight into a bound arena: range class of this image's
 * dequantized coefficients s = coefficient * q (what jpgpu_batch_upload computes itself):
 *   0  unknown / hostile -> wrap-exact kernels (always correct);
 *   1  every |s| < 2^15;
 *   3  additionally, in every 8x8 block, each column's sum of |s| is <= 5900.
 * Higher classes select faster arithmetic that is bit-exact only on such data.
 * Default for never-uploaded images: 0. */
int jpgpu_batch_set_range_hint(jpgpu_batch *b, uint32_t image, int sane);
/* Same, for one component; and the class of a buffer as jpgpu_batch_upload would compute it (pure host function,
 * no device needed) — for feeders that stage coefficients themselves (jpgpu_pipeline_*, jpgpu_decoder.h). */
int jpgpu_batch_set_range_class(jpgpu_batch *b, uint32_t image, uint32_t comp, int range_class);
int jpgpu_range_class(const int16_t *coefficients, size_t len, const uint16_t quantization_table[64]);
/* The same classification done ON THE DEVICE for every image of the batch, from the coefficients as they stand in the
 * arena (written there by the caller's own kernels or copies into a bound arena, or by the device entropy decoder):
 * one pass over the arena at HBM speed on `hip_stream`, blocking; afterwards every component has the range class
 * jpgpu_batch_upload would have given it. If `classes` is not NULL it receives 4 entries per image. */
int jpgpu_batch_scan_ranges(jpgpu_batch *b, void *hip_stream, uint8_t *classes);
/* The classification WITHOUT the host: one pass over the arena on `hip_stream` (asynchronous, nothing is read back) leaves
 * per-image range statistics on the device, and from then on every jpgpu_batch_decode turns them into the images' classes
 * there (a few-microsecond kernel in front of the pixel kernels, which pick their arithmetic per workgroup) — no host
 * synchronisation between whoever wrote the coefficients and the pixel kernels.  The library's own writers leave the same
 * statistics as a by-product and need no pass at all: the device entropy decoder (jpgpu_pipeline_decode with
 * JPGPU_PIPELINE_DEVICE_ENTROPY), jpgpu_batch_upload_compact with range_class < 0.  A class set from
 * the host afterwards (upload, set_range_hint / set_range_class, scan_ranges) takes over again for that component. */
int jpgpu_batch_classify_on_device(jpgpu_batch *b, void *hip_stream);
/* Replace the quantization table given in the image descriptor (RowData.quantization_table of Worker::start,
 * src/worker/mod.rs:18-22): feeders learn it only while parsing the stream. Takes effect at the next decode.  If the table
 * differs from the one in place, the component's range class goes back to 0 (unknown: wrap-exact kernels) — the class of
 * coefficients uploaded earlier was computed with the old table; upload (or jpgpu_batch_set_range_class / scan_ranges) afterwards. */
int jpgpu_batch_set_quantization_table(jpgpu_batch *b, uint32_t image, uint32_t comp, const uint16_t quantization_table[64]);

/* Compact coefficient transport: PCIe carries, per component,
 *     [ n_blocks x u64 bitmap | n_blocks x u32 first-value index | nnz x i16 values ]
 * (bit k of a bitmap = natural-order coefficient k of the block is non-zero; values in ascending k; the index is the
 * block's position in the value array) and a kernel expands it into the coefficient arena at the start of the next
 * decode.  jpgpu_compact_encode converts `n_blocks` dense blocks (pure host function), returns the bytes written
 * (<= jpgpu_compact_max_bytes) and, if asked, the range class of jpgpu_batch_set_range_hint.
 * jpgpu_batch_upload_compact validates the buffer, copies it asynchronously on `hip_stream` (the buffer must stay
 * valid until that stream reaches the copy; use pinned memory for real overlap) and marks the component for expansion;
 * it also sets the component's range class when `range_class` >= 0; with `range_class` < 0 (the sender did not classify)
 * the expansion kernel ranges the values on the device while it has them in registers (jpgpu_batch_classify_on_device). */
size_t jpgpu_compact_max_bytes(size_t n_blocks);
size_t jpgpu_compact_encode(const int16_t *coefficients, size_t n_blocks, const uint16_t quantization_table[64], void *dst,
                            int *range_class);
int jpgpu_batch_upload_compact(jpgpu_batch *b, uint32_t image, uint32_t comp, const void *compact, size_t bytes,
                               int range_class, void *hip_stream);

/* Enqueue the whole batch on `hip_stream` (a hipStream_t; NULL = the null stream). */
int jpgpu_batch_decode(jpgpu_batch *b, void *hip_stream);
int jpgpu_batch_synchronize(jpgpu_batch *b, void *hip_stream);
/* HBM -> host download of one image's pixels. Blocking. */
int jpgpu_batch_download(jpgpu_batch *b, uint32_t image, uint8_t *dst, size_t cap, size_t *len);
/* Timing helper: `iters` back-to-back jpgpu_batch_decode on `hip_stream` bracketed by
 * hipEvents on that stream; returns the average milliseconds per decode of the batch. */
int jpgpu_batch_time(jpgpu_batch *b, void *hip_stream, uint32_t iters, float *ms_per_decode);
/* Name of the kernel path the batch resolved to ("fused420", "generic", ...). */
const char *jpgpu_batch_path(const jpgpu_batch *b);
/* How many images of the batch's fused launch groups run in each arithmetic variant (counts[0]: wrap-exact, counts[1]:
 * range class 1, counts[2]: range class 3) with the range classes as they stand: images of different classes get
 * separate launches, so an image with hostile coefficients costs only itself. Images on the generic path are not
 * counted (they carry their class per component). */
int jpgpu_batch_class_counts(jpgpu_batch *b, uint32_t counts[3]);

#ifdef __cplusplus
}
#endif
#endif /* JPGPU_H */
