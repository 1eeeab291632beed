/*
 * oracle_batch.c — CPU ORACLE, batch driver used only as bench.py's `cpu_baseline`
 * ("port" of the reference scalar path) and by tests.
 *
 * Arrangement mirrors how a rayon user of the reference would decode a batch: one
 * image per task over all host threads; inside an image the pixel pipeline is the
 * serial one of src/worker/immediate.rs + src/worker/mod.rs:97-128 (IDCT of every MCU
 * row, then upsample + colour-convert of every output row).  Only the pixel pipeline
 * (coefficients -> pixels, the path the GPU kernels replace) is run here.
 */
#include "jpeg_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const orc_component *comps;
    int ncomp;
    const uint16_t *qts; /* ncomp * 64 */
    const int16_t *const *coefs; /* n_images * ncomp pointers */
    int n_images;
    uint16_t out_w, out_h;
    int ct;
    uint8_t *const *outs;
    int next; /* work counter */
    int status;
    pthread_mutex_t mu;
} batch_t;

static void *batch_worker(void *arg) {
    batch_t *b = (batch_t *)arg;
    uint8_t *planes[4] = {NULL, NULL, NULL, NULL};
    for (int c = 0; c < b->ncomp; c++) planes[c] = (uint8_t *)malloc(orc_plane_bytes(&b->comps[c]) + 1);
    /* outs == NULL: timing mode, every thread decodes into one private, reused buffer */
    size_t out_len = b->ncomp == 1 ? (size_t)b->comps[0].size_w * b->comps[0].size_h
                                   : (size_t)b->out_w * b->out_h * (size_t)b->ncomp;
    uint8_t *priv = b->outs ? NULL : (uint8_t *)malloc(out_len + 1);
    for (;;) {
        pthread_mutex_lock(&b->mu);
        int i = b->next++;
        pthread_mutex_unlock(&b->mu);
        if (i >= b->n_images) break;
        for (int c = 0; c < b->ncomp; c++) {
            const orc_component *cp = &b->comps[c];
            size_t mcu_rows = cp->block_h / cp->v;
            orc_append_rows(cp, b->qts + 64 * c, b->coefs[(size_t)i * b->ncomp + c], 0, mcu_rows, planes[c]);
        }
        int rc = orc_compute_image(b->comps, b->ncomp, planes, b->out_w, b->out_h, b->ct, priv ? priv : b->outs[i], NULL);
        if (rc) {
            pthread_mutex_lock(&b->mu);
            b->status = rc;
            pthread_mutex_unlock(&b->mu);
        }
    }
    for (int c = 0; c < b->ncomp; c++) free(planes[c]);
    free(priv);
    return NULL;
}

/* Runs the pixel pipeline for n_images images of identical geometry on `nthreads`
 * host threads.  Returns 0 or the last non-zero status. */
int orc_batch_pixels(const orc_component *comps, int ncomp, const uint16_t *qts,
                     const int16_t *const *coefs, int n_images, uint16_t out_w, uint16_t out_h,
                     int color_transform, uint8_t *const *outs, int nthreads) {
    batch_t b;
    memset(&b, 0, sizeof(b));
    b.comps = comps;
    b.ncomp = ncomp;
    b.qts = qts;
    b.coefs = coefs;
    b.n_images = n_images;
    b.out_w = out_w;
    b.out_h = out_h;
    b.ct = color_transform;
    b.outs = outs;
    pthread_mutex_init(&b.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &b);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&b.mu);
    return b.status;
}

/* ---- whole decodes (bench.py's `cpu_baseline_e2e`): Decoder::new(bytes).decode() for every stream, one stream per task over
 * `nthreads` host threads — what a rayon user of the reference gets with par_iter over files (per image a serial
 * parse + Huffman + IDCT, src/decoder.rs:297-1298, then upsampling and colour conversion).  The pixels are dropped; `ok`
 * receives the number of streams that decoded, `pixels` the pixel bytes they produced. ---- */
typedef struct {
    const uint8_t *const *data;
    const size_t *len;
    int n, next, ok;
    unsigned long long pixels;
    pthread_mutex_t mu;
} decode_batch_t;

static void *decode_worker(void *arg) {
    decode_batch_t *b = (decode_batch_t *)arg;
    int ok = 0;
    unsigned long long pixels = 0;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        int i = b->next++;
        pthread_mutex_unlock(&b->mu);
        if (i >= b->n) break;
        orc_result res;
        orc_decode(b->data[i], b->len[i], 0, 0, -1 /* determine_color_transform */, 0, &res);
        if (res.status == 0) {
            ok++;
            pixels += res.pixels_len;
        }
        orc_free_result(&res);
    }
    pthread_mutex_lock(&b->mu);
    b->ok += ok;
    b->pixels += pixels;
    pthread_mutex_unlock(&b->mu);
    return NULL;
}

int orc_batch_decode(const uint8_t *const *data, const size_t *len, int n, int nthreads, int *ok, unsigned long long *pixels) {
    decode_batch_t b;
    memset(&b, 0, sizeof(b));
    b.data = data;
    b.len = len;
    b.n = n;
    pthread_mutex_init(&b.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, decode_worker, &b);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&b.mu);
    if (ok) *ok = b.ok;
    if (pixels) *pixels = b.pixels;
    return 0;
}
