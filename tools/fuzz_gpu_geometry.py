#!/usr/bin/env python3
"""Differential fuzzing of the batch kernels over geometry: random sizes x sampling kinds x colour transforms x coefficient
classes x (every fifth round) reduced scales 4 / 2 / 1, as batches of one kind (the fused kernels, with per-class launch groups; the
band kernel of csrc/fused_scaled.hpp for reduced sizes) and as mixed batches (several launch groups, the generic path);
every image must equal the oracle's pixel pipeline byte for byte.
    python tools/fuzz_gpu_geometry.py <seed> <rounds>      (on the GPU box; prints "bad 0")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import oracle as O, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
import test_gpu_parity as T

KINDS = [  # sampling factors, colour transform
    ([(2, 2), (1, 1), (1, 1)], "YCbCr"), ([(2, 1), (1, 1), (1, 1)], "YCbCr"), ([(1, 1), (1, 1), (1, 1)], "YCbCr"),
    ([(1, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 1)], "Grayscale"), ([(1, 1), (1, 1), (1, 1)], "RGB"),
    ([(1, 1), (1, 1), (1, 1), (1, 1)], "CMYK"), ([(1, 1), (1, 1), (1, 1), (1, 1)], "YCCK"),
    ([(4, 1), (1, 1), (1, 1)], "YCbCr"), ([(4, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 4), (1, 1), (1, 1)], "YCbCr"), ([(2, 4), (1, 1), (1, 1)], "YCbCr"), ([(4, 4), (1, 1), (1, 1)], "YCbCr"),  # UpsamplerGeneric layouts: fusedgen
    ([(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), ([(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"), ([(2, 2), (1, 1), (1, 1), (1, 1)], "YCCK"), ([(2, 2), (1, 1), (1, 1), (2, 2)], "CMYK"),  # four components with half-size ones: fused420x4 (round 3)
    ([(3, 1), (1, 1), (1, 1)], "YCbCr"), ([(2, 2), (2, 1), (1, 1)], "YCbCr"),  # no fused kernel: the generic path
]
COEF = ["sparse", "tight", "sane", "full"]

def run(seed, rounds, verbose=True):
    rng = np.random.default_rng(seed)
    T.J = J
    bad = total = 0
    paths = {}
    for r in range(rounds):
        mixed = r % 4 == 3
        n = int(rng.integers(2, 7))
        samp, ct = KINDS[int(rng.integers(0, len(KINDS)))]
        scale = 8 if r % 5 else int(rng.choice([4, 2, 1]))  # every fifth round: Decoder::scale
        cases = []
        # sizes: mostly small, around the tile / strip / MCU boundaries now and then
        def size():
            pick = int(rng.integers(0, 4))
            if int(rng.integers(0, 12)) == 0: return int(rng.integers(1300, 2100)), int(rng.integers(200, 1100))  # several strips and segments
            if pick == 0: return int(rng.integers(1, 40)), int(rng.integers(1, 40))
            if pick == 1: return int(rng.integers(600, 720)), int(rng.integers(1, 70))       # around one strip (42 MCUs = 672 px)
            if pick == 2: return int(rng.integers(1, 70)), int(rng.integers(300, 420))       # many MCU rows: several segments
            return int(rng.integers(40, 400)), int(rng.integers(40, 300))
        w_, h_ = size()
        for i in range(n):
            if mixed:
                samp, ct = KINDS[int(rng.integers(0, len(KINDS)))]
                w_, h_ = size()
            if scale == 8:
                cases.append(T._batch_case(rng, w_, h_, samp, ct, kind=COEF[int(rng.integers(0, 4))]))
            else:  # (mixed rounds: every image at its own reduced scale)
                cases.append(T._scaled_case(rng, w_, h_, samp, ct, int(rng.choice([4, 2, 1])) if mixed else scale, kind=("sparse", "full")[int(rng.integers(0, 2))]))
        outs, path = T._run_batch(cases)
        paths[path] = paths.get(path, 0) + 1
        for i, (oc, qts, coefs, ct_, cw, ch) in enumerate(cases):
            want = O.pixels_from_coefficients(oc, qts, coefs, cw, ch, ct_.upper())
            total += 1
            if not np.array_equal(np.asarray(outs[i]).ravel(), np.asarray(want).ravel()):
                bad += 1
                if verbose: print("MISMATCH round", r, "image", i, cw, ch, [(c.h, c.v) if hasattr(c, "h") else c for c in oc], ct_, path, flush=True)
    if verbose:
        print("seed", seed, "rounds", rounds, "images", total, "paths", paths, "bad", bad, flush=True)
    return bad, total, paths


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2]))[0] else 0)
