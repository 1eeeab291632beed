#!/usr/bin/env python3
"""Differential fuzzing of the Worker boundary (jpgpu_worker_*, jpgpu_compute_image) over call sequences: random frame
geometry (1..4 components, fused and generic samplings, dct_scale 8/4/2/1), per component start + a random PREFIX of its MCU
rows (a scan that ended early leaves the tail of the plane zero) + either get_result (plane comes back and stays on the device)
or finish_plane (device resident), rows one by one or several at once, components in random order, the worker reused for the
next frame; then compute_image from the retained planes.  Planes and pixels byte for byte against the oracle.
    python tools/fuzz_gpu_worker.py <seed> <frames>      (GPU box; prints "bad 0")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import oracle as O, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
import test_gpu_parity as T

KINDS = [([(2, 2), (1, 1), (1, 1)], "YCbCr"), ([(2, 1), (1, 1), (1, 1)], "YCbCr"), ([(1, 1)] * 3, "YCbCr"), ([(1, 2), (1, 1), (1, 1)], "YCbCr"),
         ([(1, 1)], "Grayscale"), ([(1, 1)] * 3, "RGB"), ([(1, 1)] * 4, "CMYK"), ([(1, 1)] * 4, "YCCK"), ([(4, 1), (1, 1), (1, 1)], "YCbCr"),
         ([(2, 2), (2, 1), (1, 1)], "YCbCr"), ([(1, 1)] * 3, "None")]


def run(seed, frames, verbose=True):
    T.J = J
    rng = np.random.default_rng(seed)
    bad = 0
    paths = {}
    w = J.HipWorker()
    for f in range(frames):
        samp, ct = KINDS[int(rng.integers(0, len(KINDS)))]
        scale = [8, 8, 8, 8, 8, 8, 8, 4, 2, 1][int(rng.integers(0, 10))]
        wd, ht = (int(rng.integers(1, 700)), int(rng.integers(1, 300))) if rng.integers(0, 3) else (int(rng.integers(1, 60)), int(rng.integers(1, 60)))
        ocomps, _ = O.make_components(wd, ht, samp, dct_scale=scale)
        comps = T.to_j(ocomps)
        n = len(ocomps)
        qts = [rng.integers(1, 120, 64).astype(np.uint16) for _ in range(n)]
        kind = ["sparse", "sparse", "full"][int(rng.integers(0, 3))]
        coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h) if kind == "sparse" else
                 rng.integers(-32768, 32768, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
        truncate = rng.integers(0, 5) == 0
        planes_want, resident = [None] * n, rng.integers(0, 10) < 7
        for ci in rng.permutation(n):
            oc = ocomps[ci]
            per_row = oc.block_w * oc.v * 64
            rows_total = len(coefs[ci]) // per_row
            rows = int(rng.integers(0, rows_total + 1)) if truncate else rows_total
            w.start(J.RowData(int(ci), comps[ci], qts[ci]))
            r = 0
            while r < rows:
                k = int(rng.integers(1, 5)) if rng.integers(0, 2) else 1
                k = min(k, rows - r)
                if k == 1:
                    w.append_row((int(ci), coefs[ci][r * per_row:(r + 1) * per_row]))
                else:
                    w.append_rows_contiguous(int(ci), coefs[ci][r * per_row:(r + k) * per_row], k)
                r += k
            planes_want[ci] = O.idct_plane(oc, qts[ci], coefs[ci], n_mcu_rows=rows)
            if resident:
                w.finish_plane(int(ci), int(ci))
            else:
                got = w.get_result(int(ci))
                if not np.array_equal(np.asarray(got).ravel(), np.asarray(planes_want[ci]).ravel()):
                    bad += 1
                    if verbose: print("PLANE MISMATCH frame", f, (wd, ht), samp, "scale", scale, "comp", int(ci), "rows", rows, "of", rows_total, flush=True)
        out_w, out_h = -(-wd * scale // 8), -(-ht * scale // 8)
        try:
            want = O.compute_image(ocomps, planes_want, out_w, out_h, ct.upper())
        except O.OracleError as e:
            want = e
        try:
            got = w.compute_image(list(comps), None, (out_w, out_h), ct)
        except Exception as e:  # noqa: BLE001 (compared by kind below)
            got = e
        paths[w.last_path] = paths.get(w.last_path, 0) + 1
        if isinstance(want, Exception) or isinstance(got, Exception):
            if not (isinstance(want, Exception) and isinstance(got, Exception)):
                bad += 1
                if verbose: print("ERROR MISMATCH frame", f, (wd, ht), samp, ct, "scale", scale, repr(want)[:80], repr(got)[:80], flush=True)
        elif not np.array_equal(np.asarray(got).ravel(), np.asarray(want).ravel()):
            bad += 1
            if verbose: print("PIXEL MISMATCH frame", f, (wd, ht), samp, ct, "scale", scale, "truncated" if truncate else "full", "resident" if resident else "downloaded", w.last_path, flush=True)
    w.close()
    if verbose: print("seed", seed, "frames", frames, "paths", paths, "bad", bad, flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
