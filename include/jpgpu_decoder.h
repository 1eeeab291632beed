/*
 * jpgpu_decoder.h — C ABI of the host front-end: the crate's public `Decoder` surface
 * (src/decoder.rs:101-295, src/lib.rs:39-41) on top of the MI355X pixel backend (jpgpu.h).
 *
 * The reference's front-end is Rust; no Rust toolchain exists in this image, so the marker
 * parser (src/parser.rs), Huffman / progressive entropy decoder (src/huffman.rs,
 * src/decoder.rs:794-1298) and the marker loop (src/decoder.rs:297-615) are restated in C++
 * (jpeg-decoder_amd/csrc/host) and feed exactly what the crate hands to `Worker::append_row`.
 * Entropy decoding runs on the host (inherently serial per stream) for single images and whatever the device routes of the pipeline
 * below do not take; every pixel is produced on the GPU.
 */
#ifndef JPGPU_DECODER_H
#define JPGPU_DECODER_H

#include "jpgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PixelFormat, src/decoder.rs:39-60 */
enum { JPGPU_PIXEL_L8 = 0, JPGPU_PIXEL_L16 = 1, JPGPU_PIXEL_RGB24 = 2, JPGPU_PIXEL_CMYK32 = 3 };
/* CodingProcess, src/parser.rs:24-33 */
enum { JPGPU_CODING_DCT_SEQUENTIAL = 0, JPGPU_CODING_DCT_PROGRESSIVE = 1, JPGPU_CODING_LOSSLESS = 2 };

/* ImageInfo, src/decoder.rs:62-73 */
typedef struct jpgpu_image_info {
    uint16_t width, height;
    int32_t pixel_format;
    int32_t coding_process;
} jpgpu_image_info;

typedef struct jpgpu_decoder jpgpu_decoder;

/* Decoder::new(reader) — src/decoder.rs:134-154.  The bytes are copied. `device` is the HIP
 * device the pixel work runs on (-1: host-only object, usable for read_info / metadata /
 * jpgpu_decoder_decode_coefficients but not for decode()). */
int jpgpu_decoder_create(const uint8_t *data, size_t len, int device, jpgpu_decoder **out);
void jpgpu_decoder_destroy(jpgpu_decoder *d);
const char *jpgpu_decoder_last_error(const jpgpu_decoder *d);

/* set_color_transform / set_max_decoding_buffer_size — src/decoder.rs:158-165 */
int jpgpu_decoder_set_color_transform(jpgpu_decoder *d, int color_transform);
int jpgpu_decoder_set_max_decoding_buffer_size(jpgpu_decoder *d, size_t max_bytes);

/* read_info / info / scale — src/decoder.rs:170-197, 265-290.  info returns JPGPU_ERR_FORMAT
 * until read_info or decode succeeded (Option::None). */
int jpgpu_decoder_read_info(jpgpu_decoder *d);
int jpgpu_decoder_info(const jpgpu_decoder *d, jpgpu_image_info *info);
int jpgpu_decoder_scale(jpgpu_decoder *d, uint16_t requested_width, uint16_t requested_height,
                        uint16_t *out_width, uint16_t *out_height);

/* decode() -> Vec<u8> — src/decoder.rs:293-295.  `*len` receives the pixel byte count even when
 * `cap` is too small (then JPGPU_ERR_FORMAT and nothing is decoded twice: call again with room;
 * a decoder decodes once). */
int jpgpu_decoder_decode(jpgpu_decoder *d, uint8_t *dst, size_t cap, size_t *len);
/* Size decode() will produce, valid after read_info: width*height*bytes_per_pixel. */
size_t jpgpu_decoder_output_bytes(const jpgpu_decoder *d);

/* Metadata — src/decoder.rs:200-243. Pointers stay valid until the decoder is destroyed;
 * NULL/0 when absent.  icc_profile assembles the APP2 chunks (validity rules of :213-243). */
const uint8_t *jpgpu_decoder_exif_data(const jpgpu_decoder *d, size_t *len);
const uint8_t *jpgpu_decoder_xmp_data(const jpgpu_decoder *d, size_t *len);
const uint8_t *jpgpu_decoder_icc_profile(jpgpu_decoder *d, size_t *len);

/* Host half only (no GPU): run the marker loop and entropy decoder and return what would
 * cross the Worker boundary — the frame geometry / tables as a jpgpu_image_desc and, per
 * component, the coefficient rows that were appended (block-raster int16, natural order).
 * `coefs[c]` / `n_coefs[c]` are owned by the decoder.  This is what feeds jpgpu_batch_upload
 * when a batch of files is decoded (host parse of image i+1 overlaps the kernels of image i). */
int jpgpu_decoder_decode_coefficients(jpgpu_decoder *d, jpgpu_image_desc *desc, const int16_t **coefs,
                                      size_t *n_coefs);

/* ------------------------------------------------------------------------------------------------
 * Pipeline: many JPEG streams -> pixels, host entropy decoding on a thread pool feeding the batch
 * kernels ("multi-threaded per-image feeders").  The reference decodes one image per
 * `Decoder` on the calling thread (src/decoder.rs:134-154, 293-295); a pipeline is N such decoders:
 *   1. headers of all streams in parallel (read_info)            -> geometry -> one jpgpu_batch
 *   2. entropy decoding in parallel, one image per task, rows written to pinned staging memory;
 *      each image is sent to HBM (hipMemcpyAsync) as soon as its last scan is done, so the copy of
 *      image i overlaps the Huffman decoding of the others
 *   3. the batch kernels (fused launch groups per image kind) and the optional download, per sub-batch of ~64 images,
 *      overlapped with step 2 of the following sub-batches.
 * Per-image failures (src/error.rs) do not fail the call: query them per image.  The batch and its
 * arenas are kept and reused while the sequence of geometries stays the same (fixed-size frames).
 * ---------------------------------------------------------------------------------------------- */
typedef struct jpgpu_pipeline jpgpu_pipeline;

/* Wall-clock milliseconds of the last decode call.  The call's images are cut into sub-batches of about 64; the kernels and
 * the download of a sub-batch are enqueued as soon as its last image is uploaded and overlap the entropy decoding of the
 * following ones: entropy_and_upload_ms ends when the host threads are done, download_ms is the drain after that (the
 * last sub-batch's kernels + copy), kernels_ms is 0 (never exposed on its own). */
typedef struct jpgpu_pipeline_timings {
    double headers_ms, setup_ms, entropy_and_upload_ms, kernels_ms, download_ms, total_ms;
    uint32_t threads, images_ok;
    uint64_t jpeg_bytes, coefficient_bytes, pixel_bytes;
    uint32_t images_device_entropy;  /* entropy-decoded on the device (JPGPU_PIPELINE_DEVICE_ENTROPY) ... */
    uint32_t images_device_rejected; /* ... of which the device decoder handed this many back to the host */
    /* Kernel time of the device entropy route, by phase, from events on each sub-batch's own stream and summed over the
     * sub-batches (they overlap one another and the host: a breakdown of GPU work, not of the call's wall time).  Recorded only
     * when the environment variable JPGPU_BATCH_KERNEL_TIMES is set (the events cost a few microseconds per sub-batch):
     *   dev_fill_ms   zero fill of the coefficient planes and statistics
     *   dev_sync_ms   the chunk decoder's sync passes (with speculative emission) + block numbering
     *   dev_write_ms  expansion of the sync passes' entry lists into coefficient blocks (and, as a by-product, their range statistics)
     *                 + DC sums of scans whose components share their tables; images_entry_pixels (below): only the strip index of
     *                 their lists — the pixel walk reads the lists itself
     *   dev_pixel_ms  class finalize + pixel kernels (dequantize, IDCT, upsampling, colour conversion) */
    uint32_t dev_times_valid, _pad;
    double dev_fill_ms, dev_sync_ms, dev_write_ms, dev_pixel_ms;
    /* pipelines over several devices (jpgpu_pipeline_create_multi): wall clock until every device had decoded its share, then of the
     * gather to the first device (JPGPU_PIPELINE_GATHER; 0 without) and the bytes it moved.  The other wall-clock fields are then the
     * MAXIMUM over the devices, the counts and byte totals the SUM; total_ms is always the whole call. */
    double decode_ms, gather_ms;
    uint64_t gather_bytes;
    /* ---- appended in 0.2 (round 5) ---- */
    /* JPGPU_PIPELINE_GATHER: every device copies a sub-batch's pixels to the first device behind that sub-batch's kernels, on a stream
     * of its own: gather_ms (above) is what the call still WAITED for the copies once the last device had decoded — the exposed part;
     * gather_copy_ms is the longest device's summed copy time (events around each copy).  copy - exposed ran under the decode. */
    double gather_copy_ms;
    /* CPU time the whole process used during the call, all threads (CLOCK_PROCESS_CPUTIME_ID): cpu_ms / images = what an image costs the
     * host — the figure that decides how many GPUs a host with few cores can feed. */
    double cpu_ms;
    uint32_t images_host_light; /* of images_device_entropy: scans uploaded as the file holds them, marker check + unstuffing on the device */
    uint32_t input_pinned;      /* JPGPU_PIPELINE_INPUT_PINNED was in force: the DMA engine read the caller's buffers, no staging copy */
    uint32_t images_device_progressive; /* of images_device_entropy: progressive frames whose scans were decoded and accumulated on the device */
    uint32_t images_entry_pixels; /* (round 6, in what was padding) of images_device_entropy: 4:2:0 images whose pixel walk read the chunk
                                   * decoder's entry lists itself — nothing of their scan went through the coefficient arena */
} jpgpu_pipeline_timings;

enum {
    JPGPU_PIPELINE_DOWNLOAD = 1u, /* also copy the pixels to pinned host memory (jpgpu_pipeline_pixels_host) */
    JPGPU_PIPELINE_DENSE = 2u,    /* send all 64 coefficients of every block over PCIe instead of the compact form
                                   * (bitmap + index + non-zero values, jpgpu.h) — A/B switch, same pixels */
    JPGPU_PIPELINE_DEVICE_ENTROPY = 4u /* 8-bit sequential Huffman streams with one all-component scan: send the entropy-coded
                                   * bytes and decode them on the device with the self-synchronising chunk decoder (one lane
                                   * per chunk of 64 bytes to 4 kB, csrc/huff_sync_core.hpp; a restart segment — src/decoder.rs:920-956:
                                   * segments are independent — is a scan in miniature with chunk slots of its own); since 0.2 also
                                   * PROGRESSIVE frames (src/decoder.rs:1086-1298: 8-bit, no restart interval, every coefficient
                                   * refined one bit at a time): one lane per scan, the scans of a band pipelined block by block,
                                   * coefficients accumulated in the arena (csrc/huff_prog_core.hpp) — for as many of a call's frames as
                                   * the dispatcher gives the device (see JPGPU_PIPELINE_PROGRESSIVE_ON_HOST); every other stream, and
                                   * any stream a device decoder flags, takes the host path */
    /* (8u: round 2-3's per-scan delta transport for progressive streams — measured 2.5 x slower than the compact planes, deleted in round 4;
     * the bit is REFUSED since 0.2, like every unknown bit: JPGPU_ERR_FORMAT) */
    , JPGPU_PIPELINE_GATHER = 16u /* pipelines over several devices: every device copies each sub-batch's pixels to the FIRST device of
                                   * the list as soon as the sub-batch is decoded (peer-to-peer on a stream of the source device: one
                                   * xGMI link per peer, overlapped with the decode of the following sub-batches; SURVEY 8e, north_star's
                                   * final gather); jpgpu_pipeline_pixels_device then points into that copy.  Ignored by a one-device pipeline. */
    /* ---- host CPU per image (round 5).  With JPGPU_PIPELINE_DEVICE_ENTROPY the host's share of a baseline file is the staging pass:
     * 0xFF00 -> 0xFF while copying the scan into pinned memory, 20-80 us per 1080p file and core — measured: 4,096 files per call reach
     * 50-54 k images/s on 1 to 8 CPUs that way.  "Host light": the scan goes up as the file holds it (one memcpy; or no copy at all with
     * JPGPU_PIPELINE_INPUT_PINNED) and three small kernels check it for markers and drop the stuffing zeros on the device (a stream with
     * anything but 0xFF00 pairs inside is handed back to the host decoder, as the staging pass does): 73-78 k images/s on the same 1 to
     * 8 CPUs.  Host light is the DEFAULT since round 6, at every thread count (host staging is 1-2 ms of 50 faster on a quiet 16-CPU box
     * and up to 2 x slower on a busy one; light took 48-53 ms per 4,096 files on every box measured); JPGPU_PIPELINE_HOST_STAGED asks
     * for the host's staging pass, JPGPU_PIPELINE_HOST_LIGHT says the default out loud.  Streams with restart markers are staged by the host in both modes
     * (their markers must be found before the segments can be laid out). */
    , JPGPU_PIPELINE_HOST_LIGHT = 32u
    , JPGPU_PIPELINE_HOST_STAGED = 64u
    , JPGPU_PIPELINE_INPUT_PINNED = 128u /* every `data[i]` lies in page-locked host memory (jpgpu_host_alloc, or the caller's own
                                   * hipHostMalloc / hipHostRegister): host-light uploads then read the caller's buffers directly —
                                   * files that follow one another in memory (gaps of LESS than one 4-kB page: only pages that hold bytes
                                   * of a file are ever read) travel in one copy.  Passing pageable
                                   * memory with this flag is an error the runtime may or may not report: do not. */
    , JPGPU_PIPELINE_PROGRESSIVE_ON_HOST = 256u /* A/B switch: progressive frames take the host entropy decoder even with
                                   * JPGPU_PIPELINE_DEVICE_ENTROPY (round 4's behaviour); default since 0.2: the device decodes as many
                                   * of a call's progressive frames as finish while the host threads decode the rest */
};
/* Page-locked host memory for JPEG input (hipHostMalloc): read files straight into it and pass JPGPU_PIPELINE_INPUT_PINNED. */
int jpgpu_host_alloc(size_t bytes, void **out);
void jpgpu_host_free(void *p);
/* flags of jpgpu_pipeline_create_multi */
enum {
    JPGPU_PIPELINE_MULTI_PIN_CPUS = 1u /* give every device's host threads their own share of the CPUs the calling thread may run on
                                   * (sched_setaffinity): the CPUs of the NUMA node the device's PCIe root hangs off, split among the
                                   * devices of that node; contiguous slices of the allowed list when the topology is unknown
                                   * (jpgpu_plan_cpu_shares) */
};
/* The dealing rule itself, for deployments (and tests) that want to look at it: device k = PCI bus id pci_bus_ids[k]
 * ("0000:c1:00.0", jpgpu_device_pci_bus_id), its NUMA node from <sysfs>/bus/pci/devices/<id>/numa_node, the node's CPUs from
 * <sysfs>/devices/system/node/node<N>/cpulist (sysfs NULL: "/sys").  device_of_cpu[i] receives the device that gets allowed_cpus[i]
 * (-1: none), numa_nodes[k] (optional) the node found for device k (-1: unknown -> contiguous slices for everybody). */
int jpgpu_plan_cpu_shares(const char *sysfs, const char *const *pci_bus_ids, uint32_t n_devices, const int *allowed_cpus, uint32_t n_allowed,
                          int *device_of_cpu, int *numa_nodes);
int jpgpu_device_pci_bus_id(int device, char *buf, size_t cap); /* cap >= 13 */

/* n_threads = entropy / header workers; 0 = one per physical core (half the hardware threads), capped at twice a cgroup CPU quota if
 * there is one.  Half as many again (at least two) staging threads for the device-entropy route come on top, and one uploader thread
 * per call that mostly waits: jpgpu_pipeline_timings::threads reports workers + staging threads. */
int jpgpu_pipeline_create(int device, uint32_t n_threads, jpgpu_pipeline **out);
/* The same object over SEVERAL devices (the reference has one Decoder per stream, src/decoder.rs:134-154 — images are
 * independent, so a batch shards by image and nothing crosses devices while it decodes): `devices` = n_devices HIP ordinals (an
 * ordinal may appear more than once: that many sub-pipelines share the device).  Image i of a call goes to devices[i mod n_devices];
 * every per-image accessor below takes the call's own image index.  n_threads = host threads for ALL devices together, workers AND
 * staging threads (0: what ONE pipeline's workers take by default), split evenly, two thirds workers — the host's cores are the
 * shared resource, not the devices.  Peer access between every listed device and the first is switched on here; if the runtime
 * refuses it, calls with JPGPU_PIPELINE_GATHER fail with the reason. */
int jpgpu_pipeline_create_multi(const int *devices, uint32_t n_devices, uint32_t n_threads, uint32_t flags, jpgpu_pipeline **out);
uint32_t jpgpu_pipeline_device_count(const jpgpu_pipeline *p);                    /* 1 for jpgpu_pipeline_create */
int jpgpu_pipeline_image_device(const jpgpu_pipeline *p, uint32_t image);          /* HIP ordinal of the device that decoded it; -1 */
int jpgpu_pipeline_pixels_device_ordinal(const jpgpu_pipeline *p, uint32_t image); /* ... whose memory jpgpu_pipeline_pixels_device points into */
void jpgpu_pipeline_destroy(jpgpu_pipeline *p);
const char *jpgpu_pipeline_last_error(const jpgpu_pipeline *p);
/* Decoder::scale (src/decoder.rs:278-290) for every image of the calls that follow: each image is decoded at the smallest of the
 * DCT scales 1/8, 1/4, 1/2, 1 whose output is at least requested_width x requested_height (choose_idct_size, src/idct.rs:14-28) —
 * per image, as N decoders on which scale() was called would; jpgpu_pipeline_image_info then reports the scaled size.
 * 0 x 0: full size again (the default). */
int jpgpu_pipeline_set_scale(jpgpu_pipeline *p, uint16_t requested_width, uint16_t requested_height);
/* Decoder::set_color_transform (src/decoder.rs:158-161) for every image of the calls that follow: one of the JPGPU_CT_* values of
 * jpgpu.h instead of what determine_color_transform finds per image; a negative value: per image again (the default). */
int jpgpu_pipeline_set_color_transform(jpgpu_pipeline *p, int color_transform);
/* Decoder::set_max_decoding_buffer_size (src/decoder.rs:162-165): an image whose decoded size exceeds max_bytes fails with the
 * reference's error; SIZE_MAX: no limit (the default). */
int jpgpu_pipeline_set_max_decoding_buffer_size(jpgpu_pipeline *p, size_t max_bytes);
/* The streams must stay valid during the call only.  Returns JPGPU_OK if the machinery worked, even
 * when individual images failed. */
int jpgpu_pipeline_decode(jpgpu_pipeline *p, const uint8_t *const *data, const size_t *len, uint32_t n_images,
                          uint32_t flags);
/* Results of the last decode call, valid until the next one / destroy. */
int jpgpu_pipeline_image_status(const jpgpu_pipeline *p, uint32_t image);        /* JPGPU_OK or the image's error */
const char *jpgpu_pipeline_image_error(const jpgpu_pipeline *p, uint32_t image); /* its message */
int jpgpu_pipeline_image_info(const jpgpu_pipeline *p, uint32_t image, jpgpu_image_info *info);
size_t jpgpu_pipeline_pixel_bytes(const jpgpu_pipeline *p, uint32_t image);
const void *jpgpu_pipeline_pixels_device(const jpgpu_pipeline *p, uint32_t image); /* HBM, NULL if the image failed */
const uint8_t *jpgpu_pipeline_pixels_host(const jpgpu_pipeline *p, uint32_t image); /* only with JPGPU_PIPELINE_DOWNLOAD */
/* One image's pixels of the last decode call, HBM -> `dst` (blocking): for calls that left the pixels on the device and want
 * to look at a few of them.  `*len` receives the byte count even when `cap` is too small (then JPGPU_ERR_FORMAT). */
int jpgpu_pipeline_download(jpgpu_pipeline *p, uint32_t image, uint8_t *dst, size_t cap, size_t *len);
const char *jpgpu_pipeline_kernel_path(const jpgpu_pipeline *p);                   /* "fused420", "generic", ... */
int jpgpu_pipeline_last_timings(const jpgpu_pipeline *p, jpgpu_pipeline_timings *t);

/* Decoders borrow workers (streams, pinned staging, device planes) and one-image pipelines from process-wide pools of idle
 * ones, and progressive front-ends take their accumulation planes from a pool of host buffers (JPGPU_HOST_POOL_MB): release
 * whatever is idle at the moment.  Never needed for correctness; for long-running services that want the memory back. */
void jpgpu_trim_caches(void);

#ifdef __cplusplus
}
#endif
#endif /* JPGPU_DECODER_H */
