/* worker_roundtrip.c — TEST: the call sequence of the Rust shim (rust/src/worker/hip.rs), from C.
 *
 * Built by tests/test_gpu_capi.py with gcc against libjpgpu.so (no Python, no ctypes in the data path) and run on the
 * GPU box:   worker_roundtrip <case file> <pixel file>
 *   case file:  u32 ncomp, u32 out_w, u32 out_h, i32 color_transform, u32 device_resident (1: finish_plane + NULL planes,
 *               0: get_result + host planes = INTEGRATION.md §4 compat mode), then per component: jpgpu_component,
 *               u16 q[64], u32 n_rows (MCU rows to append; may be short of the plane), i16 coefficients[n_rows * per_row]
 *   pixel file: i32 status of jpgpu_compute_image, then the pixels (status 0) or the library's message
 * Exit code 0 whenever the protocol itself worked (statuses are data). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpgpu.h"

static int read_exact(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
    if (argc != 3) return 64;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 65;
    uint32_t ncomp, out_w, out_h, resident;
    int32_t ct;
    if (read_exact(in, &ncomp, 4) || read_exact(in, &out_w, 4) || read_exact(in, &out_h, 4) || read_exact(in, &ct, 4) || read_exact(in, &resident, 4)) return 66;
    if (ncomp == 0 || ncomp > JPGPU_MAX_COMPONENTS) return 66;
    int n_dev = 0;
    if (jpgpu_device_count(&n_dev) != JPGPU_OK || n_dev < 1) return 70; /* no MI355X: the test is skipped by its marker, not here */
    jpgpu_worker *w = NULL;
    if (jpgpu_worker_create(0, &w) != JPGPU_OK) return 71; /* HipWorker::new */
    jpgpu_component comps[JPGPU_MAX_COMPONENTS];
    uint8_t *planes[JPGPU_MAX_COMPONENTS] = {NULL, NULL, NULL, NULL};
    int status = JPGPU_OK;
    for (uint32_t i = 0; i < ncomp && status == JPGPU_OK; i++) {
        uint16_t q[64];
        uint32_t n_rows;
        if (read_exact(in, &comps[i], sizeof(comps[i])) || read_exact(in, q, sizeof(q)) || read_exact(in, &n_rows, 4)) return 66;
        const size_t per_row = (size_t)comps[i].block_width * comps[i].vertical_sampling_factor * 64;
        int16_t *row = (int16_t *)malloc(per_row * sizeof(int16_t) + 2);
        status = jpgpu_worker_start(w, i, &comps[i], q); /* Worker::start(RowData) */
        for (uint32_t r = 0; r < n_rows && status == JPGPU_OK; r++) {
            if (read_exact(in, row, per_row * sizeof(int16_t))) return 66;
            status = jpgpu_worker_append_row(w, i, row, per_row); /* Worker::append_row: the Vec is dropped after the call */
            memset(row, 0x5a, per_row * sizeof(int16_t));          /* ... so scribble over it */
        }
        free(row);
        if (status != JPGPU_OK) break;
        if (resident) {
            status = jpgpu_worker_finish_plane(w, i, i); /* Worker::get_result -> placeholder, plane stays in HBM */
        } else {
            const size_t n = (size_t)comps[i].block_width * comps[i].block_height * comps[i].dct_scale * comps[i].dct_scale;
            size_t got = 0;
            planes[i] = (uint8_t *)malloc(n + 1);
            status = jpgpu_worker_get_result(w, i, planes[i], n, &got); /* compat mode: the plane comes back */
            if (status == JPGPU_OK && got != n) return 72;
        }
    }
    fclose(in);
    const size_t cap = ncomp == 1 ? (size_t)comps[0].size_width * comps[0].size_height : (size_t)out_w * out_h * ncomp;
    uint8_t *pixels = (uint8_t *)malloc(cap + 1);
    size_t len = 0;
    if (status == JPGPU_OK)
        status = jpgpu_compute_image(w, comps, ncomp, resident ? NULL : (const uint8_t *const *)planes, (uint16_t)out_w, (uint16_t)out_h, ct, pixels, cap, &len);
    FILE *out = fopen(argv[2], "wb");
    if (!out) return 67;
    const int32_t st = status;
    fwrite(&st, 4, 1, out);
    if (status == JPGPU_OK) fwrite(pixels, 1, len, out);
    else fputs(jpgpu_worker_last_error(w), out);
    fclose(out);
    jpgpu_worker_destroy(w); /* Drop */
    for (uint32_t i = 0; i < ncomp; i++) free(planes[i]);
    free(pixels);
    return 0;
}
