/*
 * oracle_front.c — CPU ORACLE front-end (test infrastructure only).
 *
 * Restates the host side of image-rs/jpeg-decoder v0.3.2 that feeds the Worker
 * boundary: marker loop (src/decoder.rs:297-615), frame / scan / table parsers
 * (src/parser.rs), Huffman bit reader and tables (src/huffman.rs), sequential and
 * progressive block decoding (src/decoder.rs:794-1298), decode_planes (:617-696)
 * and determine_color_transform (:698-764).  The pixel work is delegated to
 * oracle_pixels.c through the same calls a Worker receives (start / append_row /
 * get_result), using the ImmediateWorker layout.
 *
 * Lossless (SOF3) is out of scope for the hot path (src/decoder/lossless.rs is a
 * different pipeline) and reports ORC_ERR_UNSUPPORTED.
 */
#include "jpeg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_COMPONENTS 4 /* src/decoder.rs:21 */

/* src/decoder.rs:27-36 */
static const uint8_t UNZIGZAG[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

/* ---- error plumbing ------------------------------------------------------ */
typedef struct {
    int code;
    char msg[160];
} err_t;
#define FAIL(e, c, ...)                                   \
    do {                                                  \
        (e)->code = (c);                                  \
        snprintf((e)->msg, sizeof((e)->msg), __VA_ARGS__); \
        return (c);                                       \
    } while (0)
#define TRY(x)             \
    do {                   \
        int _rc = (x);     \
        if (_rc) return _rc; \
    } while (0)

/* ---- byte reader (std::io::Read over a slice) ------------------------------ */
typedef struct {
    const uint8_t *p;
    size_t len, pos;
} reader_t;

/* src/lib.rs:56-60 */
static int read_u8(reader_t *r, err_t *e, uint8_t *out) {
    if (r->pos >= r->len) FAIL(e, ORC_ERR_IO, "failed to fill whole buffer");
    *out = r->p[r->pos++];
    return 0;
}
/* src/lib.rs:62-66 */
static int read_u16_be(reader_t *r, err_t *e, uint16_t *out) {
    if (r->pos + 2 > r->len) {
        r->pos = r->len;
        FAIL(e, ORC_ERR_IO, "failed to fill whole buffer");
    }
    *out = (uint16_t)((r->p[r->pos] << 8) | r->p[r->pos + 1]);
    r->pos += 2;
    return 0;
}
static int read_exact(reader_t *r, err_t *e, uint8_t *dst, size_t n) {
    if (r->pos + n > r->len) {
        r->pos = r->len;
        FAIL(e, ORC_ERR_IO, "failed to fill whole buffer");
    }
    if (dst) memcpy(dst, r->p + r->pos, n);
    r->pos += n;
    return 0;
}
/* src/parser.rs:149-158 */
static int skip_bytes(reader_t *r, err_t *e, size_t n) {
    if (r->pos + n > r->len) {
        r->pos = r->len;
        FAIL(e, ORC_ERR_IO, "unexpected EOF");
    }
    r->pos += n;
    return 0;
}

/* ---- markers, src/marker.rs ------------------------------------------------- */
enum { MK_SOF, MK_JPG, MK_DHT, MK_DAC, MK_RST, MK_SOI, MK_EOI, MK_SOS, MK_DQT, MK_DNL, MK_DRI,
       MK_DHP, MK_EXP, MK_APP, MK_JPGN, MK_COM, MK_TEM, MK_RES, MK_NONE };
typedef struct {
    int kind;
    int n;
} marker_t;

/* src/marker.rs:63-135 */
static marker_t marker_from_u8(uint8_t b) {
    marker_t m = {MK_NONE, 0};
    if (b == 0x00 || b == 0xFF) return m;
    if (b == 0x01) { m.kind = MK_TEM; return m; }
    if (b <= 0xBF) { m.kind = MK_RES; return m; }
    switch (b) {
    case 0xC4: m.kind = MK_DHT; return m;
    case 0xC8: m.kind = MK_JPG; return m;
    case 0xCC: m.kind = MK_DAC; return m;
    case 0xD8: m.kind = MK_SOI; return m;
    case 0xD9: m.kind = MK_EOI; return m;
    case 0xDA: m.kind = MK_SOS; return m;
    case 0xDB: m.kind = MK_DQT; return m;
    case 0xDC: m.kind = MK_DNL; return m;
    case 0xDD: m.kind = MK_DRI; return m;
    case 0xDE: m.kind = MK_DHP; return m;
    case 0xDF: m.kind = MK_EXP; return m;
    case 0xFE: m.kind = MK_COM; return m;
    default: break;
    }
    if (b >= 0xC0 && b <= 0xCF) { m.kind = MK_SOF; m.n = b - 0xC0; return m; }
    if (b >= 0xD0 && b <= 0xD7) { m.kind = MK_RST; m.n = b - 0xD0; return m; }
    if (b >= 0xE0 && b <= 0xEF) { m.kind = MK_APP; m.n = b - 0xE0; return m; }
    m.kind = MK_JPGN; /* 0xF0..=0xFD */
    m.n = b - 0xF0;
    return m;
}

/* ---- Huffman tables, src/huffman.rs:181-285 ------------------------------- */
typedef struct {
    int present;
    uint8_t values[256];
    int nvalues;
    int32_t delta[16], maxcode[16];
    uint8_t lut_value[256], lut_size[256];
    int has_ac_lut;
    int16_t ac_value[256];
    uint8_t ac_run_size[256];
} hufftable_t;

/* src/huffman.rs:165-173 */
static int16_t extend(uint16_t value, uint8_t count) {
    uint16_t vt = (uint16_t)(1u << (count - 1));
    if (value < vt) return (int16_t)((int16_t)value + (int16_t)((uint16_t)0xFFFF << count) + 1);
    return (int16_t)value;
}

/* src/huffman.rs:191-252 + derive_huffman_codes :256-285 */
static int hufftable_new(hufftable_t *t, const uint8_t bits[16], const uint8_t *values, int nvalues,
                         int is_ac, err_t *e) {
    uint8_t huffsize[256];
    uint16_t huffcode[256];
    int n = 0;
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < bits[i]; k++) {
            if (n >= 256) FAIL(e, ORC_ERR_FORMAT, "bad huffman table");
            huffsize[n++] = (uint8_t)(i + 1);
        }
    if (n == 0 || n != nvalues) FAIL(e, ORC_ERR_FORMAT, "bad huffman table");
    uint8_t code_size = huffsize[0];
    uint32_t code = 0;
    for (int i = 0; i < n; i++) {
        while (code_size < huffsize[i]) {
            code <<= 1;
            code_size++;
        }
        if (code >= (1u << huffsize[i])) FAIL(e, ORC_ERR_FORMAT, "bad huffman code length");
        huffcode[i] = (uint16_t)code;
        code++;
    }
    memset(t, 0, sizeof(*t));
    t->present = 1;
    memcpy(t->values, values, (size_t)nvalues);
    t->nvalues = nvalues;
    int j = 0;
    for (int i = 0; i < 16; i++) {
        t->delta[i] = 0;
        t->maxcode[i] = -1;
        if (bits[i] != 0) {
            t->delta[i] = (int32_t)j - (int32_t)huffcode[j];
            j += bits[i];
            t->maxcode[i] = (int32_t)huffcode[j - 1];
        }
    }
    for (int i = 0; i < n; i++) {
        if (huffsize[i] > 8) continue;
        int bits_remaining = 8 - huffsize[i];
        int start = huffcode[i] << bits_remaining;
        for (int b = 0; b < (1 << bits_remaining); b++) {
            t->lut_value[start + b] = values[i];
            t->lut_size[start + b] = huffsize[i];
        }
    }
    t->has_ac_lut = is_ac;
    if (is_ac) {
        for (int i = 0; i < 256; i++) {
            uint8_t value = t->lut_value[i], size = t->lut_size[i];
            uint8_t run_length = value >> 4, magnitude_category = value & 0x0f;
            if (magnitude_category > 0 && size + magnitude_category <= 8) {
                uint16_t unext = (uint16_t)((((unsigned)i << size) & 0xFF) >> (8 - magnitude_category));
                t->ac_value[i] = extend(unext, magnitude_category);
                t->ac_run_size[i] = (uint8_t)((run_length << 4) | (size + magnitude_category));
            }
        }
    }
    return 0;
}

/* ---- Huffman bit reader, src/huffman.rs:14-161 ------------------------------ */
typedef struct {
    uint64_t bits;
    uint8_t num_bits;
    int has_marker;
    marker_t marker;
} huff_t;

/* :123-160 */
static int huff_read_bits(huff_t *h, reader_t *r, err_t *e) {
    while (h->num_bits <= 56) {
        uint8_t byte = 0;
        if (!h->has_marker) TRY(read_u8(r, e, &byte));
        if (byte == 0xFF) {
            uint8_t next_byte;
            TRY(read_u8(r, e, &next_byte));
            if (next_byte != 0x00) {
                while (next_byte == 0xFF) TRY(read_u8(r, e, &next_byte));
                if (next_byte == 0x00) FAIL(e, ORC_ERR_FORMAT, "FF 00 found where marker was expected");
                h->marker = marker_from_u8(next_byte);
                h->has_marker = 1;
                continue;
            }
        }
        h->bits |= (uint64_t)byte << (56 - h->num_bits);
        h->num_bits = (uint8_t)(h->num_bits + 8);
    }
    return 0;
}
static inline uint16_t huff_peek(const huff_t *h, uint8_t count) { /* :108-113 */
    if (count == 0) return 0;
    return (uint16_t)((h->bits >> (64 - count)) & ((1u << count) - 1));
}
static inline void huff_consume(huff_t *h, uint8_t count) { /* :115-121 */
    h->bits = count >= 64 ? 0 : h->bits << count;
    h->num_bits = (uint8_t)(h->num_bits - count);
}
/* :31-58 */
static int huff_decode(huff_t *h, reader_t *r, const hufftable_t *t, err_t *e, uint8_t *out) {
    if (h->num_bits < 16) TRY(huff_read_bits(h, r, e));
    uint16_t idx = huff_peek(h, 8);
    uint8_t size = t->lut_size[idx];
    if (size > 0) {
        huff_consume(h, size);
        *out = t->lut_value[idx];
        return 0;
    }
    uint16_t bits = huff_peek(h, 16);
    for (int i = 8; i < 16; i++) {
        int32_t code = (int32_t)(bits >> (15 - i));
        if (code <= t->maxcode[i]) {
            huff_consume(h, (uint8_t)(i + 1));
            int32_t index = code + t->delta[i];
            if (index < 0 || index >= t->nvalues) FAIL(e, ORC_ERR_INTERNAL, "reference would panic: huffman value index");
            *out = t->values[index];
            return 0;
        }
    }
    FAIL(e, ORC_ERR_FORMAT, "failed to decode huffman code");
}
/* :60-78 ; returns 1 in *hit when the fused LUT matched */
static int huff_decode_fast_ac(huff_t *h, reader_t *r, const hufftable_t *t, err_t *e, int *hit,
                               int16_t *value, uint8_t *run) {
    *hit = 0;
    if (t->has_ac_lut) {
        if (h->num_bits < 8) TRY(huff_read_bits(h, r, e));
        uint16_t idx = huff_peek(h, 8);
        uint8_t run_size = t->ac_run_size[idx];
        if (run_size != 0) {
            *run = run_size >> 4;
            huff_consume(h, run_size & 0x0f);
            *value = t->ac_value[idx];
            *hit = 1;
        }
    }
    return 0;
}
/* :80-90 */
static int huff_get_bits(huff_t *h, reader_t *r, uint8_t count, err_t *e, uint16_t *out) {
    if (h->num_bits < count) TRY(huff_read_bits(h, r, e));
    *out = huff_peek(h, count);
    huff_consume(h, count);
    return 0;
}
/* :92-96 */
static int huff_receive_extend(huff_t *h, reader_t *r, uint8_t count, err_t *e, int16_t *out) {
    uint16_t v;
    TRY(huff_get_bits(h, r, count, e, &v));
    *out = extend(v, count);
    return 0;
}
/* :103-105 */
static int huff_take_marker(huff_t *h, reader_t *r, err_t *e, int *has, marker_t *m) {
    TRY(huff_read_bits(h, r, e));
    *has = h->has_marker;
    *m = h->marker;
    h->has_marker = 0;
    return 0;
}

/* ---- frame / scan info, src/parser.rs:49-73 ------------------------------- */
typedef struct {
    int is_baseline, is_differential, coding_process /*0 seq,1 prog,2 lossless*/, arithmetic;
    uint8_t precision;
    uint16_t image_w, image_h, output_w, output_h, mcu_w, mcu_h;
    int ncomp;
    orc_component components[256]; /* component_count is a u8 */
} frame_t;

typedef struct {
    int ncomp;
    int component_indices[4], dc_table_indices[4], ac_table_indices[4];
    uint8_t ss_start, ss_end; /* Range: start..end (end exclusive) */
    uint8_t ah, al;
} scan_t;

typedef struct {
    reader_t reader;
    int has_frame;
    frame_t frame;
    hufftable_t dc_tables[4], ac_tables[4];
    int has_qt[4];
    uint16_t qt[4][64]; /* unzigzagged, src/decoder.rs:490-496 */
    uint16_t restart_interval;
    int has_adobe, adobe_transform; /* 0 Unknown, 1 YCbCr, 2 YCCK */
    int color_transform_override;   /* ORC_CT_AUTO or a value */
    int is_jfif, is_mjpeg;
    int16_t *coefficients[MAX_COMPONENTS]; /* progressive */
    size_t coefficients_len[MAX_COMPONENTS];
    int has_coefficients;
    uint64_t coefficients_finished[MAX_COMPONENTS];
    /* ImmediateWorker state, src/worker/immediate.rs:12-28 (index = worker index) */
    struct {
        uint8_t *result;
        size_t result_len;
        size_t offset;
        orc_component component;
        uint16_t qt[64];
        int16_t *log; /* concatenation of every append_row (for the GPU parity tests) */
        size_t log_len, log_cap;
    } worker[MAX_COMPONENTS];
    int keep;
    orc_result *res;
} dec_t;

/* src/parser.rs:137-147 */
static int read_length(reader_t *r, err_t *e, size_t *out) {
    uint16_t l;
    TRY(read_u16_be(r, e, &l));
    if (l < 2) FAIL(e, ORC_ERR_FORMAT, "encountered marker with invalid length %u", l);
    *out = (size_t)l - 2;
    return 0;
}

/* Upsampler::new validity check used at SOF time (src/decoder.rs:375-379,
 * src/upsampler.rs:76-105): only the NonIntegerSubsamplingRatio error matters. */
static int check_upsampler(const orc_component *c, int n, uint16_t w, uint16_t h, err_t *e) {
    int h_max = 0, v_max = 0;
    for (int i = 0; i < n; i++) {
        if (c[i].h > h_max) h_max = c[i].h;
        if (c[i].v > v_max) v_max = c[i].v;
    }
    for (int i = 0; i < n; i++) {
        int h1 = c[i].h == h_max || w == 1, v1 = c[i].v == v_max || h == 1;
        int h2 = c[i].h * 2 == h_max, v2 = c[i].v * 2 == v_max;
        if ((h1 && v1) || (h2 && v1) || (h1 && v2) || (h2 && v2)) continue;
        if (h_max % c[i].h != 0 || v_max % c[i].v != 0)
            FAIL(e, ORC_ERR_UNSUPPORTED, "NonIntegerSubsamplingRatio");
    }
    return 0;
}

/* src/parser.rs:161-280 */
static int parse_sof(reader_t *r, int n, frame_t *f, err_t *e) {
    size_t length;
    TRY(read_length(r, e, &length));
    if (length <= 6) FAIL(e, ORC_ERR_FORMAT, "invalid length in SOF");
    memset(f, 0, sizeof(*f));
    f->is_baseline = n == 0;
    f->is_differential = (n >= 5 && n <= 7) || (n >= 13 && n <= 15);
    f->coding_process = (n == 0 || n == 1 || n == 5 || n == 9 || n == 13) ? 0
                        : (n == 2 || n == 6 || n == 10 || n == 14)      ? 1
                                                                         : 2;
    f->arithmetic = n >= 9;
    TRY(read_u8(r, e, &f->precision));
    if (f->precision == 8) {
    } else if (f->precision == 12) {
        if (f->is_baseline) FAIL(e, ORC_ERR_FORMAT, "12 bit sample precision is not allowed in baseline");
    } else if (f->coding_process != 2 || f->precision > 16) {
        FAIL(e, ORC_ERR_FORMAT, "invalid precision %u in frame header", f->precision);
    }
    uint16_t height, width;
    TRY(read_u16_be(r, e, &height));
    TRY(read_u16_be(r, e, &width));
    if (height == 0) FAIL(e, ORC_ERR_UNSUPPORTED, "DNL");
    if (width == 0) FAIL(e, ORC_ERR_FORMAT, "zero width in frame header");
    uint8_t component_count;
    TRY(read_u8(r, e, &component_count));
    if (component_count == 0) FAIL(e, ORC_ERR_FORMAT, "zero component count in frame header");
    if (f->coding_process == 1 && component_count > 4)
        FAIL(e, ORC_ERR_FORMAT, "progressive frame with more than 4 components");
    if (length != 6 + 3 * (size_t)component_count) FAIL(e, ORC_ERR_FORMAT, "invalid length in SOF");
    for (int i = 0; i < component_count; i++) {
        uint8_t identifier, byte, tq;
        TRY(read_u8(r, e, &identifier));
        for (int k = 0; k < i; k++)
            if (f->components[k].identifier == identifier)
                FAIL(e, ORC_ERR_FORMAT, "duplicate frame component identifier %u", identifier);
        TRY(read_u8(r, e, &byte));
        uint8_t hs = byte >> 4, vs = byte & 0x0f;
        if (hs == 0 || hs > 4) FAIL(e, ORC_ERR_FORMAT, "invalid horizontal sampling factor %u", hs);
        if (vs == 0 || vs > 4) FAIL(e, ORC_ERR_FORMAT, "invalid vertical sampling factor %u", vs);
        TRY(read_u8(r, e, &tq));
        if (tq > 3 || (f->coding_process == 2 && tq != 0))
            FAIL(e, ORC_ERR_FORMAT, "invalid quantization table index %u", tq);
        orc_component *c = &f->components[i];
        c->identifier = identifier;
        c->h = hs;
        c->v = vs;
        c->tq = tq;
        c->dct_scale = 8;
    }
    f->ncomp = component_count;
    f->image_w = f->output_w = width;
    f->image_h = f->output_h = height;
    if (orc_update_component_sizes(width, height, f->components, f->ncomp, &f->mcu_w, &f->mcu_h))
        FAIL(e, ORC_ERR_FORMAT, "invalid dimensions");
    return 0;
}

/* src/parser.rs:332-482 */
static int parse_sos(reader_t *r, const frame_t *f, scan_t *s, err_t *e) {
    size_t length;
    TRY(read_length(r, e, &length));
    if (length == 0) FAIL(e, ORC_ERR_FORMAT, "zero length in SOS");
    uint8_t component_count;
    TRY(read_u8(r, e, &component_count));
    if (component_count == 0 || component_count > 4)
        FAIL(e, ORC_ERR_FORMAT, "invalid component count %u in scan header", component_count);
    if (length != 4 + 2 * (size_t)component_count) FAIL(e, ORC_ERR_FORMAT, "invalid length in SOS");
    memset(s, 0, sizeof(*s));
    for (int i = 0; i < component_count; i++) {
        uint8_t identifier, byte;
        TRY(read_u8(r, e, &identifier));
        int component_index = -1;
        for (int k = 0; k < f->ncomp; k++)
            if (f->components[k].identifier == identifier) {
                component_index = k;
                break;
            }
        if (component_index < 0)
            FAIL(e, ORC_ERR_FORMAT, "scan component identifier %u does not match any frame component", identifier);
        int maxidx = 0;
        for (int k = 0; k < i; k++) {
            if (s->component_indices[k] == component_index)
                FAIL(e, ORC_ERR_FORMAT, "duplicate scan component identifier %u", identifier);
            if (s->component_indices[k] > maxidx) maxidx = s->component_indices[k];
        }
        if (component_index < maxidx)
            FAIL(e, ORC_ERR_FORMAT, "the scan component order does not follow the order in the frame header");
        TRY(read_u8(r, e, &byte));
        uint8_t dc = byte >> 4, ac = byte & 0x0f;
        if (dc > 3 || (f->is_baseline && dc > 1)) FAIL(e, ORC_ERR_FORMAT, "invalid dc table index %u", dc);
        if (ac > 3 || (f->is_baseline && ac > 1)) FAIL(e, ORC_ERR_FORMAT, "invalid ac table index %u", ac);
        s->component_indices[i] = component_index;
        s->dc_table_indices[i] = dc;
        s->ac_table_indices[i] = ac;
    }
    s->ncomp = component_count;
    uint32_t blocks_per_mcu = 0;
    for (int i = 0; i < s->ncomp; i++) {
        const orc_component *c = &f->components[s->component_indices[i]];
        blocks_per_mcu += (uint32_t)c->h * c->v;
    }
    if (component_count > 1 && blocks_per_mcu > 10)
        FAIL(e, ORC_ERR_FORMAT, "scan with more than one component and more than 10 blocks per MCU");
    uint8_t ss, se, byte;
    TRY(read_u8(r, e, &ss));
    TRY(read_u8(r, e, &se));
    TRY(read_u8(r, e, &byte));
    uint8_t ah = byte >> 4, al = byte & 0x0f;
    if (al >= f->precision) FAIL(e, ORC_ERR_FORMAT, "invalid point transform, must be less than the frame precision");
    if (f->coding_process == 1) {
        if (se > 63 || ss > se || (ss == 0 && se != 0))
            FAIL(e, ORC_ERR_FORMAT, "invalid spectral selection parameters: ss=%u, se=%u", ss, se);
        if (ss != 0 && component_count != 1)
            FAIL(e, ORC_ERR_FORMAT, "spectral selection scan with AC coefficients can't have more than one component");
        if (ah > 13 || al > 13)
            FAIL(e, ORC_ERR_FORMAT, "invalid successive approximation parameters: ah=%u, al=%u", ah, al);
        if (ah != 0 && ah != al + 1)
            FAIL(e, ORC_ERR_FORMAT, "successive approximation scan with more than one bit of improvement");
    } else if (f->coding_process == 2) {
        if (se != 0) FAIL(e, ORC_ERR_FORMAT, "spectral selection end shall be zero in lossless scan");
        if (ah != 0) FAIL(e, ORC_ERR_FORMAT, "successive approximation high shall be zero in lossless scan");
        if (ss > 7) FAIL(e, ORC_ERR_FORMAT, "invalid predictor selection value: %u", ss);
    } else {
        if (se == 0) se = 63;
        if (ss != 0 || se != 63) FAIL(e, ORC_ERR_FORMAT, "spectral selection is not allowed in non-progressive scan");
        if (ah != 0 || al != 0) FAIL(e, ORC_ERR_FORMAT, "successive approximation is not allowed in non-progressive scan");
    }
    s->ss_start = ss;
    s->ss_end = (uint8_t)(se + 1);
    s->ah = ah;
    s->al = al;
    return 0;
}

/* src/parser.rs:485-532 + un-zigzag of src/decoder.rs:485-498 */
static int parse_dqt(dec_t *d, err_t *e) {
    reader_t *r = &d->reader;
    size_t length;
    TRY(read_length(r, e, &length));
    uint16_t tables[4][64];
    int got[4] = {0, 0, 0, 0};
    while (length > 0) {
        uint8_t byte;
        TRY(read_u8(r, e, &byte));
        size_t precision = byte >> 4, index = byte & 0x0f;
        if (precision > 1) FAIL(e, ORC_ERR_FORMAT, "invalid precision %zu in DQT", precision);
        if (index > 3) FAIL(e, ORC_ERR_FORMAT, "invalid destination identifier %zu in DQT", index);
        if (length < 65 + 64 * precision) FAIL(e, ORC_ERR_FORMAT, "invalid length in DQT");
        uint16_t table[64];
        for (int i = 0; i < 64; i++) {
            if (precision == 0) {
                uint8_t b;
                TRY(read_u8(r, e, &b));
                table[i] = b;
            } else {
                TRY(read_u16_be(r, e, &table[i]));
            }
        }
        for (int i = 0; i < 64; i++)
            if (table[i] == 0) FAIL(e, ORC_ERR_FORMAT, "quantization table contains element with a zero value");
        memcpy(tables[index], table, sizeof(table));
        got[index] = 1;
        length -= 65 + 64 * precision;
    }
    for (int i = 0; i < 4; i++)
        if (got[i]) {
            for (int j = 0; j < 64; j++) d->qt[i][UNZIGZAG[j]] = tables[i][j];
            d->has_qt[i] = 1;
        }
    return 0;
}

/* src/parser.rs:536-589 + merge of src/decoder.rs:501-518 */
static int parse_dht(dec_t *d, err_t *e) {
    reader_t *r = &d->reader;
    size_t length;
    TRY(read_length(r, e, &length));
    int have_baseline = d->has_frame, is_baseline = d->has_frame && d->frame.is_baseline;
    /* new tables are collected first and only installed if the whole segment parses */
    hufftable_t *ndc = (hufftable_t *)calloc(4, sizeof(hufftable_t));
    hufftable_t *nac = (hufftable_t *)calloc(4, sizeof(hufftable_t));
    int rc = 0;
#define DHT_FAIL(...)                                         \
    do {                                                      \
        e->code = ORC_ERR_FORMAT;                             \
        snprintf(e->msg, sizeof(e->msg), __VA_ARGS__);        \
        rc = ORC_ERR_FORMAT;                                  \
        goto done;                                            \
    } while (0)
    while (length > 17) {
        uint8_t byte;
        if ((rc = read_u8(r, e, &byte))) goto done;
        uint8_t class = byte >> 4;
        size_t index = byte & 0x0f;
        if (class != 0 && class != 1) DHT_FAIL("invalid class %u in DHT", class);
        if (have_baseline && is_baseline && index > 1)
            DHT_FAIL("a maximum of two huffman tables per class are allowed in baseline");
        if (index > 3) DHT_FAIL("invalid destination identifier %zu in DHT", index);
        uint8_t counts[16];
        if ((rc = read_exact(r, e, counts, 16))) goto done;
        size_t size = 0;
        for (int i = 0; i < 16; i++) size += counts[i];
        if (size == 0) DHT_FAIL("encountered table with zero length in DHT");
        else if (size > 256) DHT_FAIL("encountered table with excessive length in DHT");
        else if (size > length - 17) DHT_FAIL("invalid length in DHT");
        uint8_t values[256];
        if ((rc = read_exact(r, e, values, size))) goto done;
        if ((rc = hufftable_new(class == 0 ? &ndc[index] : &nac[index], counts, values, (int)size, class == 1, e)))
            goto done;
        length -= 17 + size;
    }
    if (length != 0) DHT_FAIL("invalid length in DHT");
    for (int i = 0; i < 4; i++) {
        if (ndc[i].present) d->dc_tables[i] = ndc[i];
        if (nac[i].present) d->ac_tables[i] = nac[i];
    }
done:
    free(ndc);
    free(nac);
    return rc;
#undef DHT_FAIL
}

/* src/parser.rs:592-600 */
static int parse_dri(dec_t *d, err_t *e) {
    size_t length;
    TRY(read_length(&d->reader, e, &length));
    if (length != 2) FAIL(e, ORC_ERR_FORMAT, "DRI with invalid length");
    return read_u16_be(&d->reader, e, &d->restart_interval);
}

/* src/parser.rs:614-710 (only what influences pixels: JFIF, AVI1, Adobe) */
static int parse_app(dec_t *d, int n, err_t *e) {
    reader_t *r = &d->reader;
    size_t length, bytes_read = 0;
    TRY(read_length(r, e, &length));
    if (n == 0) {
        if (length >= 5) {
            uint8_t b[5];
            TRY(read_exact(r, e, b, 5));
            bytes_read = 5;
            if (!memcmp(b, "JFIF\0", 5)) d->is_jfif = 1;
            else if (!memcmp(b, "AVI1\0", 5)) d->is_mjpeg = 1;
        }
    } else if (n == 1) {
        TRY(read_exact(r, e, NULL, length)); /* Exif / XMP: metadata only */
        bytes_read = length;
    } else if (n == 2) {
        if (length > 14) {
            uint8_t b[14];
            TRY(read_exact(r, e, b, 14));
            bytes_read = 14;
            if (!memcmp(b, "ICC_PROFILE\0", 12)) {
                TRY(read_exact(r, e, NULL, length - bytes_read));
                bytes_read = length;
            }
        }
    } else if (n == 13) {
        if (length >= 14) {
            uint8_t b[14];
            TRY(read_exact(r, e, b, 14));
            bytes_read = 14;
            if (!memcmp(b, "Photoshop 3.0\0", 14)) {
                TRY(read_exact(r, e, NULL, length - bytes_read));
                bytes_read = length;
            }
        }
    } else if (n == 14) {
        if (length >= 12) {
            uint8_t b[12];
            TRY(read_exact(r, e, b, 12));
            bytes_read = 12;
            if (!memcmp(b, "Adobe\0", 6)) {
                if (b[11] > 2) FAIL(e, ORC_ERR_FORMAT, "invalid color transform in adobe app segment");
                d->has_adobe = 1;
                d->adobe_transform = b[11];
            }
        }
    }
    return skip_bytes(r, e, length - bytes_read);
}

/* src/decoder.rs:766-791 */
static int read_marker(dec_t *d, err_t *e, marker_t *m) {
    reader_t *r = &d->reader;
    for (;;) {
        uint8_t byte;
        do {
            TRY(read_u8(r, e, &byte));
        } while (byte != 0xFF);
        TRY(read_u8(r, e, &byte));
        while (byte == 0xFF) TRY(read_u8(r, e, &byte));
        if (byte != 0x00 && byte != 0xFF) {
            *m = marker_from_u8(byte);
            return 0;
        }
    }
}

/* ---- Worker (ImmediateWorker), src/worker/immediate.rs ---------------------- */
static int worker_start(dec_t *d, int index, const orc_component *c, const uint16_t qt[64], err_t *e) {
    if (d->worker[index].result) /* assert!(self.results[data.index].is_empty()) */
        FAIL(e, ORC_ERR_INTERNAL, "reference would panic: worker start on a live result");
    size_t n = orc_plane_bytes(c);
    d->worker[index].result = (uint8_t *)calloc(n ? n : 1, 1);
    d->worker[index].result_len = n;
    d->worker[index].offset = 0;
    d->worker[index].component = *c;
    memcpy(d->worker[index].qt, qt, 128);
    d->worker[index].log_len = 0;
    return 0;
}
static int worker_append_row(dec_t *d, int index, const int16_t *data, size_t len, err_t *e) {
    const orc_component *c = &d->worker[index].component;
    size_t block_count = (size_t)c->block_w * c->v;
    if (len != block_count * 64) FAIL(e, ORC_ERR_INTERNAL, "reference would panic: append_row length");
    size_t row_bytes = block_count * c->dct_scale * c->dct_scale;
    if (d->worker[index].offset + row_bytes > d->worker[index].result_len)
        FAIL(e, ORC_ERR_INTERNAL, "reference would panic: append_row beyond the plane");
    orc_append_rows(c, d->worker[index].qt, data, 0, 1, d->worker[index].result + d->worker[index].offset);
    d->worker[index].offset += row_bytes;
    if (d->keep) {
        if (d->worker[index].log_len + len > d->worker[index].log_cap) {
            size_t cap = d->worker[index].log_cap ? d->worker[index].log_cap * 2 : 4096;
            while (cap < d->worker[index].log_len + len) cap *= 2;
            d->worker[index].log = (int16_t *)realloc(d->worker[index].log, cap * sizeof(int16_t));
            d->worker[index].log_cap = cap;
        }
        memcpy(d->worker[index].log + d->worker[index].log_len, data, len * sizeof(int16_t));
        d->worker[index].log_len += len;
    }
    return 0;
}
/* get_result: mem::take */
static void worker_get_result(dec_t *d, int index, uint8_t **plane, size_t *len) {
    *plane = d->worker[index].result;
    *len = d->worker[index].result_len;
    d->worker[index].result = NULL;
    d->worker[index].result_len = 0;
}
/* record what crossed the boundary for frame component `ci` (test hook, not in the reference) */
static void keep_boundary(dec_t *d, int windex, int ci) {
    if (!d->keep) return;
    orc_result *res = d->res;
    free(res->coefs[ci]);
    /* exactly the rows that were appended (a non-interleaved scan may stop early, :910-918) */
    size_t n = d->worker[windex].log_len;
    res->coefs[ci] = (int16_t *)calloc(n ? n : 1, sizeof(int16_t));
    if (n) memcpy(res->coefs[ci], d->worker[windex].log, n * sizeof(int16_t));
    res->coefs_len[ci] = n;
    memcpy(res->qtables[ci], d->worker[windex].qt, 128);
}

/* ---- block decoding, src/decoder.rs:1086-1298 ------------------------------- */
static int decode_block(reader_t *r, int16_t *coefficients, huff_t *h, const hufftable_t *dc_table,
                        const hufftable_t *ac_table, uint8_t ss_start, uint8_t ss_end, uint8_t al,
                        uint16_t *eob_run, int16_t *dc_predictor, err_t *e) {
    if (ss_start == 0) {
        uint8_t value;
        TRY(huff_decode(h, r, dc_table, e, &value));
        int16_t diff = 0;
        if (value == 0) diff = 0;
        else if (value <= 11) TRY(huff_receive_extend(h, r, value, e, &diff));
        else FAIL(e, ORC_ERR_FORMAT, "invalid DC difference magnitude category");
        *dc_predictor = (int16_t)((uint16_t)*dc_predictor + (uint16_t)diff); /* wrapping_add */
        coefficients[0] = (int16_t)((uint16_t)*dc_predictor << al);
    }
    uint8_t index = ss_start > 1 ? ss_start : 1;
    if (index < ss_end && *eob_run > 0) {
        *eob_run -= 1;
        return 0;
    }
    while (index < ss_end) {
        int hit;
        int16_t value;
        uint8_t run;
        TRY(huff_decode_fast_ac(h, r, ac_table, e, &hit, &value, &run));
        if (hit) {
            index = (uint8_t)(index + run);
            if (index >= ss_end) break;
            coefficients[UNZIGZAG[index]] = (int16_t)((uint16_t)value << al);
            index++;
        } else {
            uint8_t byte;
            TRY(huff_decode(h, r, ac_table, e, &byte));
            uint8_t rr = byte >> 4, s = byte & 0x0f;
            if (s == 0) {
                if (rr == 15) {
                    index = (uint8_t)(index + 16);
                } else {
                    *eob_run = (uint16_t)((1u << rr) - 1);
                    if (rr > 0) {
                        uint16_t bits;
                        TRY(huff_get_bits(h, r, rr, e, &bits));
                        *eob_run = (uint16_t)(*eob_run + bits);
                    }
                    break;
                }
            } else {
                index = (uint8_t)(index + rr);
                if (index >= ss_end) break;
                int16_t v;
                TRY(huff_receive_extend(h, r, s, e, &v));
                coefficients[UNZIGZAG[index]] = (int16_t)((uint16_t)v << al);
                index++;
            }
        }
    }
    return 0;
}

/* :1260-1298 */
static int refine_non_zeroes(reader_t *r, int16_t *coefficients, huff_t *h, uint8_t start, uint8_t end,
                             uint8_t zrl, int16_t bit, err_t *e, uint8_t *out) {
    uint8_t last = (uint8_t)(end - 1);
    uint8_t zero_run_length = zrl;
    for (uint8_t i = start; i < end; i++) {
        int16_t *coefficient = &coefficients[UNZIGZAG[i]];
        if (*coefficient == 0) {
            if (zero_run_length == 0) {
                *out = i;
                return 0;
            }
            zero_run_length--;
        } else {
            uint16_t b;
            TRY(huff_get_bits(h, r, 1, e, &b));
            if (b == 1 && (*coefficient & bit) == 0) {
                int32_t v = *coefficient > 0 ? (int32_t)*coefficient + bit : (int32_t)*coefficient - bit;
                if (v > 32767 || v < -32768) FAIL(e, ORC_ERR_FORMAT, "Coefficient overflow");
                *coefficient = (int16_t)v;
            }
        }
    }
    *out = last;
    return 0;
}

/* :1174-1258 */
static int decode_block_successive_approximation(reader_t *r, int16_t *coefficients, huff_t *h,
                                                 const hufftable_t *ac_table, uint8_t ss_start,
                                                 uint8_t ss_end, uint8_t al, uint16_t *eob_run, err_t *e) {
    int16_t bit = (int16_t)(1 << al);
    if (ss_start == 0) {
        uint16_t b;
        TRY(huff_get_bits(h, r, 1, e, &b));
        if (b == 1) coefficients[0] |= bit;
    } else {
        if (*eob_run > 0) {
            *eob_run -= 1;
            uint8_t dummy;
            return refine_non_zeroes(r, coefficients, h, ss_start, ss_end, 64, bit, e, &dummy);
        }
        uint8_t index = ss_start;
        while (index < ss_end) {
            uint8_t byte;
            TRY(huff_decode(h, r, ac_table, e, &byte));
            uint8_t rr = byte >> 4, s = byte & 0x0f;
            uint8_t zero_run_length = rr;
            int16_t value = 0;
            if (s == 0) {
                if (rr != 15) {
                    *eob_run = (uint16_t)((1u << rr) - 1);
                    if (rr > 0) {
                        uint16_t bits;
                        TRY(huff_get_bits(h, r, rr, e, &bits));
                        *eob_run = (uint16_t)(*eob_run + bits);
                    }
                    zero_run_length = 64;
                }
            } else if (s == 1) {
                uint16_t b;
                TRY(huff_get_bits(h, r, 1, e, &b));
                value = b == 1 ? bit : (int16_t)-bit;
            } else {
                FAIL(e, ORC_ERR_FORMAT, "unexpected huffman code");
            }
            TRY(refine_non_zeroes(r, coefficients, h, index, ss_end, zero_run_length, bit, e, &index));
            if (value != 0) coefficients[UNZIGZAG[index]] = value;
            index++;
        }
    }
    return 0;
}

/* K.3-K.6 default tables, src/huffman.rs:295-346 */
static const uint8_t K3_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t K4_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t K34_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t K5_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D};
static const uint8_t K5_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1, 0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3,
    0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3,
    0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA, 0xE1, 0xE2,
    0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA};
static const uint8_t K6_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t K6_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xA1, 0xB1, 0xC1, 0x09, 0x23, 0x33, 0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1,
    0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19, 0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7A,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A,
    0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA,
    0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8, 0xD9, 0xDA,
    0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA};

static void fill_default_mjpeg_tables(dec_t *d, const scan_t *s) {
    err_t e;
    int dc0 = 0, dc1 = 0, ac0 = 0, ac1 = 0;
    for (int i = 0; i < s->ncomp; i++) {
        dc0 |= s->dc_table_indices[i] == 0;
        dc1 |= s->dc_table_indices[i] == 1;
        ac0 |= s->ac_table_indices[i] == 0;
        ac1 |= s->ac_table_indices[i] == 1;
    }
    if (!d->dc_tables[0].present && dc0) hufftable_new(&d->dc_tables[0], K3_BITS, K34_VALS, 12, 0, &e);
    if (!d->dc_tables[1].present && dc1) hufftable_new(&d->dc_tables[1], K4_BITS, K34_VALS, 12, 0, &e);
    if (!d->ac_tables[0].present && ac0) hufftable_new(&d->ac_tables[0], K5_BITS, K5_VALS, 162, 1, &e);
    if (!d->ac_tables[1].present && ac1) hufftable_new(&d->ac_tables[1], K6_BITS, K6_VALS, 162, 1, &e);
}

/* ---- decode_scan, src/decoder.rs:794-1082 ----------------------------------- */
static int decode_scan(dec_t *d, const scan_t *scan, const int finished[MAX_COMPONENTS], err_t *e,
                       int *has_marker, marker_t *marker_out, uint8_t *data[MAX_COMPONENTS],
                       size_t data_len[MAX_COMPONENTS], int *has_data) {
    const frame_t *frame = &d->frame;
    reader_t *r = &d->reader;
    orc_component components[MAX_COMPONENTS];
    int nc = scan->ncomp;
    for (int i = 0; i < nc; i++) components[i] = frame->components[scan->component_indices[i]];

    for (int i = 0; i < nc; i++)
        if (!d->has_qt[components[i].tq]) FAIL(e, ORC_ERR_FORMAT, "use of unset quantization table");
    if (d->is_mjpeg) fill_default_mjpeg_tables(d, scan);
    if (scan->ss_start == 0)
        for (int i = 0; i < nc; i++)
            if (!d->dc_tables[scan->dc_table_indices[i]].present)
                FAIL(e, ORC_ERR_FORMAT, "scan makes use of unset dc huffman table");
    if (scan->ss_end > 1)
        for (int i = 0; i < nc; i++)
            if (!d->ac_tables[scan->ac_table_indices[i]].present)
                FAIL(e, ORC_ERR_FORMAT, "scan makes use of unset ac huffman table");

    for (int i = 0; i < nc; i++)
        if (finished[i]) TRY(worker_start(d, i, &components[i], d->qt[components[i].tq], e));

    int is_progressive = frame->coding_process == 1;
    int is_interleaved = nc > 1;
    int16_t dummy_block[64];
    memset(dummy_block, 0, sizeof(dummy_block));
    huff_t huffman;
    memset(&huffman, 0, sizeof(huffman));
    int16_t dc_predictors[MAX_COMPONENTS] = {0, 0, 0, 0};
    uint16_t mcus_left_until_restart = d->restart_interval;
    int expected_rst_num = 0;
    uint16_t eob_run = 0;
    int16_t *mcu_row_coefficients[MAX_COMPONENTS] = {NULL, NULL, NULL, NULL};
    size_t per_row[MAX_COMPONENTS] = {0, 0, 0, 0};
    int rc = 0;

    for (int i = 0; i < nc; i++) per_row[i] = (size_t)components[i].block_w * components[i].v * 64;
    if (!is_progressive)
        for (int i = 0; i < nc; i++)
            if (finished[i]) mcu_row_coefficients[i] = (int16_t *)calloc(per_row[i] ? per_row[i] : 1, sizeof(int16_t));

    uint16_t mcu_h_samples[MAX_COMPONENTS], mcu_v_samples[MAX_COMPONENTS];
    for (int i = 0; i < nc; i++) {
        mcu_h_samples[i] = is_interleaved ? components[i].h : 1;
        mcu_v_samples[i] = is_interleaved ? components[i].v : 1;
    }
    uint16_t max_mcu_x = is_interleaved ? frame->mcu_w : components[0].block_w;
    uint16_t max_mcu_y = is_interleaved ? frame->mcu_h : components[0].block_h;

#define SCAN_TRY(x)     \
    do {                \
        rc = (x);       \
        if (rc) goto out; \
    } while (0)
#define SCAN_FAIL(c, ...)                             \
    do {                                              \
        e->code = (c);                                \
        snprintf(e->msg, sizeof(e->msg), __VA_ARGS__); \
        rc = (c);                                     \
        goto out;                                     \
    } while (0)

    for (uint32_t mcu_y = 0; mcu_y < max_mcu_y; mcu_y++) {
        if (mcu_y * 8 >= frame->image_h) break;
        for (uint32_t mcu_x = 0; mcu_x < max_mcu_x; mcu_x++) {
            if (mcu_x * 8 >= frame->image_w) break;
            if (d->restart_interval > 0) {
                if (mcus_left_until_restart == 0) {
                    int has;
                    marker_t m;
                    SCAN_TRY(huff_take_marker(&huffman, r, e, &has, &m));
                    if (has && m.kind == MK_RST) {
                        if (m.n != expected_rst_num)
                            SCAN_FAIL(ORC_ERR_FORMAT, "found RST%d where RST%d was expected", m.n, expected_rst_num);
                        huffman.bits = 0; /* reset() */
                        huffman.num_bits = 0;
                        memset(dc_predictors, 0, sizeof(dc_predictors));
                        eob_run = 0;
                        expected_rst_num = (expected_rst_num + 1) % 8;
                        mcus_left_until_restart = d->restart_interval;
                    } else if (has) {
                        SCAN_FAIL(ORC_ERR_FORMAT, "found marker inside scan where RST%d was expected", expected_rst_num);
                    } else {
                        SCAN_FAIL(ORC_ERR_FORMAT, "no marker found where RST%d was expected", expected_rst_num);
                    }
                }
                mcus_left_until_restart--;
            }
            for (int i = 0; i < nc; i++) {
                const orc_component *component = &components[i];
                for (uint32_t v_pos = 0; v_pos < mcu_v_samples[i]; v_pos++) {
                    for (uint32_t h_pos = 0; h_pos < mcu_h_samples[i]; h_pos++) {
                        int16_t *coefficients;
                        if (is_progressive) {
                            size_t block_y = mcu_y * mcu_v_samples[i] + v_pos;
                            size_t block_x = mcu_x * mcu_h_samples[i] + h_pos;
                            size_t block_offset = (block_y * component->block_w + block_x) * 64;
                            int ci = scan->component_indices[i];
                            if (block_offset + 64 > d->coefficients_len[ci])
                                SCAN_FAIL(ORC_ERR_INTERNAL, "reference would panic: coefficient index");
                            coefficients = d->coefficients[ci] + block_offset;
                        } else if (finished[i]) {
                            uint32_t mcu_batch_current_row = is_interleaved ? 0 : (mcu_y % component->v);
                            size_t block_y = mcu_batch_current_row * mcu_v_samples[i] + v_pos;
                            size_t block_x = mcu_x * mcu_h_samples[i] + h_pos;
                            size_t block_offset = (block_y * component->block_w + block_x) * 64;
                            if (block_offset + 64 > per_row[i])
                                SCAN_FAIL(ORC_ERR_INTERNAL, "reference would panic: row coefficient index");
                            coefficients = mcu_row_coefficients[i] + block_offset;
                        } else {
                            coefficients = dummy_block;
                        }
                        if (scan->ah == 0) {
                            SCAN_TRY(decode_block(r, coefficients, &huffman,
                                                  &d->dc_tables[scan->dc_table_indices[i]],
                                                  &d->ac_tables[scan->ac_table_indices[i]], scan->ss_start,
                                                  scan->ss_end, scan->al, &eob_run, &dc_predictors[i], e));
                        } else {
                            SCAN_TRY(decode_block_successive_approximation(
                                r, coefficients, &huffman, &d->ac_tables[scan->ac_table_indices[i]],
                                scan->ss_start, scan->ss_end, scan->al, &eob_run, e));
                        }
                    }
                }
            }
        }
        /* end of MCU row: hand the rows to the worker, :1019-1059 */
        for (int i = 0; i < nc; i++) {
            if (!finished[i]) continue;
            const orc_component *component = &components[i];
            if (!is_interleaved && (mcu_y + 1) * 8 < frame->image_h && (mcu_y + 1) % component->v > 0) continue;
            if (is_progressive) {
                uint32_t worker_mcu_y = is_interleaved ? mcu_y : mcu_y / component->v;
                size_t offset = (size_t)worker_mcu_y * per_row[i];
                int ci = scan->component_indices[i];
                if (offset + per_row[i] > d->coefficients_len[ci])
                    SCAN_FAIL(ORC_ERR_INTERNAL, "reference would panic: coefficient row slice");
                SCAN_TRY(worker_append_row(d, i, d->coefficients[ci] + offset, per_row[i], e));
            } else {
                SCAN_TRY(worker_append_row(d, i, mcu_row_coefficients[i], per_row[i], e));
                memset(mcu_row_coefficients[i], 0, per_row[i] * sizeof(int16_t));
            }
        }
    }
    {
        int has;
        marker_t m;
        SCAN_TRY(huff_take_marker(&huffman, r, e, &has, &m));
        while (has && m.kind == MK_RST) { /* :1063-1066: marker = self.read_marker().ok() */
            err_t ignore;
            has = read_marker(d, &ignore, &m) == 0;
        }
        *has_marker = has;
        *marker_out = m;
    }
    *has_data = 0;
    for (int i = 0; i < nc; i++)
        if (finished[i]) *has_data = 1;
    if (*has_data) {
        for (int i = 0; i < frame->ncomp; i++) {
            data[i] = NULL;
            data_len[i] = 0;
        }
        for (int i = 0; i < nc; i++)
            if (finished[i]) {
                int ci = scan->component_indices[i];
                keep_boundary(d, i, ci);
                worker_get_result(d, i, &data[ci], &data_len[ci]);
            }
    }
out:
    for (int i = 0; i < MAX_COMPONENTS; i++) free(mcu_row_coefficients[i]);
    return rc;
#undef SCAN_TRY
#undef SCAN_FAIL
}

/* src/decoder.rs:698-764 */
static int determine_color_transform(const dec_t *d) {
    if (d->color_transform_override != ORC_CT_AUTO) return d->color_transform_override;
    const frame_t *f = &d->frame;
    if (f->ncomp == 1) return ORC_CT_GRAYSCALE;
    if (f->ncomp == 3) {
        uint8_t a = f->components[0].identifier, b = f->components[1].identifier, c = f->components[2].identifier;
        if (a == 1 && b == 2 && c == 3) return ORC_CT_YCBCR;
        if (a == 1 && b == 34 && c == 35) return ORC_CT_JCS_BG_YCC;
        if (a == 82 && b == 71 && c == 66) return ORC_CT_RGB;
        if (a == 114 && b == 103 && c == 98) return ORC_CT_JCS_BG_RGB;
        if (d->is_jfif) return ORC_CT_YCBCR;
    }
    if (d->has_adobe) {
        if (d->adobe_transform == 0) {
            if (f->ncomp == 3) return ORC_CT_RGB;
            if (f->ncomp == 4) return ORC_CT_CMYK;
        } else if (d->adobe_transform == 1) {
            return ORC_CT_YCBCR;
        } else {
            return ORC_CT_YCCK;
        }
    } else if (f->ncomp == 4) {
        return ORC_CT_CMYK;
    }
    if (f->ncomp == 4) return ORC_CT_YCCK;
    if (f->ncomp == 3) return ORC_CT_YCBCR;
    return ORC_CT_UNKNOWN;
}

/* FrameInfo::update_idct_size, src/parser.rs:119-134 */
static int update_idct_size(frame_t *f, int idct_size, err_t *e) {
    for (int i = 0; i < f->ncomp; i++) f->components[i].dct_scale = (uint32_t)idct_size;
    if (orc_update_component_sizes(f->image_w, f->image_h, f->components, f->ncomp, &f->mcu_w, &f->mcu_h))
        FAIL(e, ORC_ERR_FORMAT, "invalid dimensions");
    f->output_w = (uint16_t)ceilf((float)f->image_w * (float)idct_size / 8.0f);
    f->output_h = (uint16_t)ceilf((float)f->image_h * (float)idct_size / 8.0f);
    return 0;
}

/* decode_internal, src/decoder.rs:297-615.  `stop_after_metadata` as in read_info(). */
static int decode_internal(dec_t *d, int stop_after_metadata, err_t *e, uint8_t *planes[MAX_COMPONENTS],
                           size_t planes_len[MAX_COMPONENTS]) {
    reader_t *r = &d->reader;
    if (stop_after_metadata && d->has_frame) return 0;
    if (!d->has_frame) {
        uint8_t a, b;
        TRY(read_u8(r, e, &a));
        if (a != 0xFF) FAIL(e, ORC_ERR_FORMAT, "first two bytes are not an SOI marker");
        TRY(read_u8(r, e, &b));
        if (marker_from_u8(b).kind != MK_SOI) FAIL(e, ORC_ERR_FORMAT, "first two bytes are not an SOI marker");
    }
    marker_t previous_marker = {MK_SOI, 0};
    int has_pending = 0;
    marker_t pending = {MK_NONE, 0};
    int scans_processed = 0;

    for (;;) {
        marker_t marker;
        if (has_pending) {
            marker = pending;
            has_pending = 0;
        } else {
            TRY(read_marker(d, e, &marker));
        }
        switch (marker.kind) {
        case MK_SOF: {
            if (d->has_frame) FAIL(e, ORC_ERR_UNSUPPORTED, "Hierarchical");
            frame_t *f = &d->frame;
            TRY(parse_sof(r, marker.n, f, e));
            if (f->is_differential) FAIL(e, ORC_ERR_UNSUPPORTED, "Hierarchical");
            if (f->arithmetic) FAIL(e, ORC_ERR_UNSUPPORTED, "ArithmeticEntropyCoding");
            if (f->precision != 8 && f->coding_process != 2) FAIL(e, ORC_ERR_UNSUPPORTED, "SamplePrecision(%u)", f->precision);
            if (f->precision < 2 || f->precision > 16) FAIL(e, ORC_ERR_UNSUPPORTED, "SamplePrecision(%u)", f->precision);
            if (f->ncomp != 1 && f->ncomp != 3 && f->ncomp != 4) FAIL(e, ORC_ERR_UNSUPPORTED, "ComponentCount(%d)", f->ncomp);
            TRY(check_upsampler(f->components, f->ncomp, f->image_w, f->image_h, e));
            d->has_frame = 1;
            if (stop_after_metadata) return 0;
            break;
        }
        case MK_SOS: {
            if (!d->has_frame) FAIL(e, ORC_ERR_FORMAT, "scan encountered before frame");
            frame_t *frame = &d->frame;
            scan_t scan;
            TRY(parse_sos(r, frame, &scan, e));
            if (frame->coding_process == 1 && !d->has_coefficients) {
                for (int i = 0; i < frame->ncomp; i++) {
                    size_t n = (size_t)frame->components[i].block_w * frame->components[i].block_h * 64;
                    d->coefficients[i] = (int16_t *)calloc(n ? n : 1, sizeof(int16_t));
                    d->coefficients_len[i] = n;
                }
                d->has_coefficients = 1;
            }
            if (frame->coding_process == 2) FAIL(e, ORC_ERR_UNSUPPORTED, "lossless is outside the oracle's scope");
            int finished[MAX_COMPONENTS] = {0, 0, 0, 0};
            if (scan.al == 0) {
                for (int k = 0; k < scan.ncomp; k++) {
                    int i = scan.component_indices[k];
                    if (d->coefficients_finished[i] == ~(uint64_t)0) continue;
                    for (int j = scan.ss_start; j < scan.ss_end; j++) d->coefficients_finished[i] |= (uint64_t)1 << j;
                    if (d->coefficients_finished[i] == ~(uint64_t)0) finished[k] = 1;
                }
            }
            int has_marker = 0, has_data = 0;
            marker_t m = {MK_NONE, 0};
            uint8_t *data[MAX_COMPONENTS] = {NULL, NULL, NULL, NULL};
            size_t data_len[MAX_COMPONENTS] = {0, 0, 0, 0};
            TRY(decode_scan(d, &scan, finished, e, &has_marker, &m, data, data_len, &has_data));
            if (has_data) {
                for (int i = 0; i < frame->ncomp; i++) {
                    if (!data[i] || data_len[i] == 0) {
                        free(data[i]);
                        continue;
                    }
                    if (d->coefficients_finished[i] == ~(uint64_t)0) {
                        free(planes[i]);
                        planes[i] = data[i];
                        planes_len[i] = data_len[i];
                    } else {
                        free(data[i]);
                    }
                }
            }
            has_pending = has_marker;
            pending = m;
            scans_processed++;
            break;
        }
        case MK_DQT: TRY(parse_dqt(d, e)); break;
        case MK_DHT: TRY(parse_dht(d, e)); break;
        case MK_DAC: FAIL(e, ORC_ERR_UNSUPPORTED, "ArithmeticEntropyCoding");
        case MK_DRI: TRY(parse_dri(d, e)); break;
        case MK_COM: {
            size_t length;
            TRY(read_length(r, e, &length));
            TRY(read_exact(r, e, NULL, length));
            break;
        }
        case MK_APP: TRY(parse_app(d, marker.n, e)); break;
        case MK_RST:
            if (previous_marker.kind != MK_SOS) FAIL(e, ORC_ERR_FORMAT, "RST found outside of entropy-coded data");
            break;
        case MK_DNL:
            if (previous_marker.kind != MK_SOS || scans_processed != 1)
                FAIL(e, ORC_ERR_FORMAT, "DNL is only allowed immediately after the first scan");
            FAIL(e, ORC_ERR_UNSUPPORTED, "DNL");
        case MK_DHP:
        case MK_EXP: FAIL(e, ORC_ERR_UNSUPPORTED, "Hierarchical");
        case MK_EOI: goto eoi;
        default: FAIL(e, ORC_ERR_FORMAT, "marker found where not allowed");
        }
        previous_marker = marker;
    }
eoi:
    if (!d->has_frame) FAIL(e, ORC_ERR_FORMAT, "end of image encountered before frame");
    return 0;
}

/* decode_planes, src/decoder.rs:617-696 (without the size limit knob) */
static int decode_planes(dec_t *d, err_t *e, uint8_t *planes[MAX_COMPONENTS], size_t planes_len[MAX_COMPONENTS]) {
    frame_t *frame = &d->frame;
    orc_result *res = d->res;
    if (frame->coding_process == 1 && d->has_coefficients) {
        for (int i = 0; i < frame->ncomp; i++) {
            if (d->coefficients_finished[i] == ~(uint64_t)0) continue;
            const orc_component *component = &frame->components[i];
            if (!d->has_qt[component->tq]) continue;
            TRY(worker_start(d, i, component, d->qt[component->tq], e));
            size_t per_row = (size_t)component->block_w * component->v * 64;
            for (uint32_t mcu_y = 0; mcu_y < frame->mcu_h; mcu_y++) {
                size_t offset = (size_t)mcu_y * per_row;
                if (offset + per_row > d->coefficients_len[i])
                    FAIL(e, ORC_ERR_INTERNAL, "reference would panic: coefficient row slice");
                TRY(worker_append_row(d, i, d->coefficients[i] + offset, per_row, e));
            }
            keep_boundary(d, i, i);
            free(planes[i]);
            worker_get_result(d, i, &planes[i], &planes_len[i]);
        }
    }
    /* compute_image, :1300-1336 */
    for (int i = 0; i < frame->ncomp; i++)
        if (!planes[i] || planes_len[i] == 0) FAIL(e, ORC_ERR_FORMAT, "not all components have data");
    int ct = determine_color_transform(d);
    res->color_transform = ct;
    size_t out_len = frame->ncomp == 1 ? (size_t)frame->components[0].size_w * frame->components[0].size_h
                                       : (size_t)frame->output_w * frame->output_h * (size_t)frame->ncomp;
    res->pixels = (uint8_t *)calloc(out_len ? out_len : 1, 1);
    res->pixels_len = out_len;
    char msg[128];
    int rc = orc_compute_image(frame->components, frame->ncomp, planes, frame->output_w, frame->output_h, ct,
                               res->pixels, msg);
    if (rc) FAIL(e, rc, "%s", msg);
    return 0;
}

void orc_decode(const uint8_t *data, size_t len, uint16_t req_w, uint16_t req_h,
                int color_transform_override, int keep_intermediates, orc_result *res) {
    memset(res, 0, sizeof(*res));
    dec_t *d = (dec_t *)calloc(1, sizeof(dec_t));
    err_t e;
    memset(&e, 0, sizeof(e));
    uint8_t *planes[MAX_COMPONENTS] = {NULL, NULL, NULL, NULL};
    size_t planes_len[MAX_COMPONENTS] = {0, 0, 0, 0};
    d->reader.p = data;
    d->reader.len = len;
    d->color_transform_override = color_transform_override;
    d->keep = keep_intermediates;
    d->res = res;
    int rc = 0;
    if (req_w != 0) {
        /* Decoder::scale, src/decoder.rs:278-290 */
        rc = decode_internal(d, 1, &e, planes, planes_len);
        if (!rc) {
            int idct_size = orc_choose_idct_size(d->frame.image_w, d->frame.image_h, req_w, req_h);
            rc = update_idct_size(&d->frame, idct_size, &e);
        }
    }
    if (!rc) rc = decode_internal(d, 0, &e, planes, planes_len);
    if (!rc) rc = decode_planes(d, &e, planes, planes_len);
    res->status = rc;
    snprintf(res->message, sizeof(res->message), "%s", e.msg);
    if (d->has_frame) {
        frame_t *f = &d->frame;
        res->width = f->output_w;
        res->height = f->output_h;
        res->image_w = f->image_w;
        res->image_h = f->image_h;
        res->ncomp = f->ncomp <= 4 ? f->ncomp : 0;
        res->coding_process = f->coding_process;
        res->is_baseline = f->is_baseline;
        res->mcu_w = f->mcu_w;
        res->mcu_h = f->mcu_h;
        for (int i = 0; i < res->ncomp; i++) res->components[i] = f->components[i];
    }
    for (int i = 0; i < MAX_COMPONENTS; i++) {
        if (keep_intermediates && planes[i]) {
            res->planes[i] = planes[i];
            res->planes_len[i] = planes_len[i];
            res->have_plane[i] = 1;
        } else {
            free(planes[i]);
        }
        free(d->coefficients[i]);
        free(d->worker[i].result);
        free(d->worker[i].log);
    }
    free(d);
}

void orc_free_result(orc_result *res) {
    free(res->pixels);
    for (int i = 0; i < 4; i++) {
        free(res->coefs[i]);
        free(res->planes[i]);
    }
    memset(res, 0, sizeof(*res));
}
