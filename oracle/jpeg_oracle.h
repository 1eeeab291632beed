/*
 * jpeg_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the *scalar* (`platform_independent`) semantics of
 * image-rs/jpeg-decoder v0.3.2 for the pixel pipeline behind the `Worker`
 * boundary (dequantize + IDCT, plane layout, upsampling, colour conversion)
 * plus the host front-end that feeds it (marker parser, Huffman / progressive
 * entropy decoder), so that the oracle can be pinned against the reference's
 * own reftest JPEG/PNG pairs and known-answer tests.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything in this directory.  The product library (libjpgpu.so) never
 * links, includes or calls it.
 *
 * Reference files restated (paths relative to the reference crate root):
 *   src/idct.rs, src/upsampler.rs, src/worker/immediate.rs, src/worker/mod.rs,
 *   src/decoder.rs, src/parser.rs, src/huffman.rs, src/marker.rs
 *
 * Parity status: the Rust reference cannot be compiled in this environment
 * (no rustc/cargo).  The oracle is pinned by (i) the three exact/±1 IDCT KATs
 * of src/idct.rs:580-657, (ii) src/parser.rs:312-329 and src/idct.rs:30-203,
 * (iii) every enabled reftest JPEG/PNG pair under the reference's <=3 rule
 * (tests/reftest/mod.rs:93-120), and (iv) the sha256 vectors of SURVEY.md
 * Appendix B which were produced by an independent restatement.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes mirror src/error.rs:16-48 (Error::{Format,Unsupported,Io,Internal}). */
enum {
    ORC_OK = 0,
    ORC_ERR_FORMAT = 1,
    ORC_ERR_UNSUPPORTED = 2,
    ORC_ERR_IO = 3,       /* unexpected end of stream */
    ORC_ERR_INTERNAL = 4, /* includes "the reference would panic here" */
};

/* ColorTransform, src/decoder.rs:76-98 (same order). */
enum {
    ORC_CT_NONE = 0,
    ORC_CT_UNKNOWN = 1,
    ORC_CT_GRAYSCALE = 2,
    ORC_CT_RGB = 3,
    ORC_CT_YCBCR = 4,
    ORC_CT_CMYK = 5,
    ORC_CT_YCCK = 6,
    ORC_CT_JCS_BG_YCC = 7,
    ORC_CT_JCS_BG_RGB = 8,
    ORC_CT_AUTO = -1, /* no user override: determine_color_transform() */
};

/* Component, src/parser.rs:76-89. */
typedef struct {
    uint8_t identifier;
    uint8_t h; /* horizontal_sampling_factor */
    uint8_t v; /* vertical_sampling_factor */
    uint8_t tq; /* quantization_table_index */
    uint32_t dct_scale; /* 8, 4, 2 or 1 */
    uint16_t size_w, size_h;   /* size */
    uint16_t block_w, block_h; /* block_size */
} orc_component;

/* ---- pixel pipeline (the hot path) ------------------------------------ */

/* src/idct.rs:205-239 */
void orc_dequantize_and_idct_block(int scale, const int16_t coefficients[64],
                                   const uint16_t quantization_table[64],
                                   size_t output_linestride, uint8_t *output);

/* src/idct.rs:14-28 */
int orc_choose_idct_size(uint16_t full_w, uint16_t full_h, uint16_t req_w, uint16_t req_h);

/* src/parser.rs:292-310; returns 0 or ORC_ERR_FORMAT. */
int orc_update_component_sizes(uint16_t width, uint16_t height, orc_component *components,
                               int ncomp, uint16_t *mcu_w, uint16_t *mcu_h);

/* bytes of one component plane: src/worker/immediate.rs:30-37 */
size_t orc_plane_bytes(const orc_component *c);

/* src/worker/immediate.rs:39-60 applied to `n_mcu_rows` consecutive MCU rows
 * starting at MCU row `first_mcu_row`; `coefs` holds n_mcu_rows*block_w*v*64 i16. */
void orc_append_rows(const orc_component *c, const uint16_t qt[64], const int16_t *coefs,
                     size_t first_mcu_row, size_t n_mcu_rows, uint8_t *plane);

/* src/decoder.rs:1300-1336 + src/worker/mod.rs:97-128 (+ upsampler.rs, colour fns).
 * planes[i] has orc_plane_bytes(&comps[i]) bytes.  out has out_w*out_h*ncomp bytes
 * (1 component: size_w*size_h).  Returns status; msg (optional, >=128 B) gets text. */
int orc_compute_image(const orc_component *comps, int ncomp, uint8_t *const *planes,
                      uint16_t out_w, uint16_t out_h, int color_transform, uint8_t *out,
                      char *msg);

/* src/decoder.rs:1486-1508 */
void orc_ycbcr_to_rgb(uint8_t y, uint8_t cb, uint8_t cr, uint8_t rgb[3]);

/* ---- front-end: JPEG bytes -> coefficients -> pixels -------------------- */

typedef struct {
    int status;
    char message[160];
    /* ImageInfo (src/decoder.rs:62-73) */
    uint16_t width, height;     /* output_size */
    uint16_t image_w, image_h;  /* image_size */
    int ncomp;
    int coding_process;  /* 0 sequential, 1 progressive, 2 lossless */
    int is_baseline;
    int color_transform; /* the transform compute_image was called with */
    uint16_t mcu_w, mcu_h;
    orc_component components[4];
    /* decoded pixels (malloc'd; free with orc_free_result) */
    uint8_t *pixels;
    size_t pixels_len;
    /* what crossed the Worker boundary, per component (NULL if that component
     * never reached the worker): full block-raster coefficient plane exactly as
     * the concatenation of append_row calls, the q-table used, and the plane. */
    int16_t *coefs[4];
    size_t coefs_len[4];   /* in i16 units */
    uint16_t qtables[4][64];
    int have_plane[4];
    uint8_t *planes[4];
    size_t planes_len[4];
} orc_result;

/* Decoder::new + (optional) scale(req_w, req_h) + (optional) set_color_transform + decode().
 * req_w == 0 means no scale() call.  keep_intermediates != 0 fills coefs/planes. */
void orc_decode(const uint8_t *data, size_t len, uint16_t req_w, uint16_t req_h,
                int color_transform_override, int keep_intermediates, orc_result *res);
void orc_free_result(orc_result *res);

/* bench.py's cpu_baseline_e2e: orc_decode of n streams, one stream per task over nthreads host threads (oracle_batch.c) */
int orc_batch_decode(const uint8_t *const *data, const size_t *len, int n, int nthreads, int *ok, unsigned long long *pixels);

#ifdef __cplusplus
}
#endif
#endif
