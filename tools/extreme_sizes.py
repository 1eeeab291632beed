"""Directed check of extreme geometries through the batch kernels against the oracle (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import oracle as O, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
import test_gpu_parity as T
T.J = J
KINDS = {"420": ([(2, 2), (1, 1), (1, 1)], "YCbCr"), "422": ([(2, 1), (1, 1), (1, 1)], "YCbCr"), "444": ([(1, 1)] * 3, "YCbCr"),
         "440": ([(1, 2), (1, 1), (1, 1)], "YCbCr"), "411": ([(4, 1), (1, 1), (1, 1)], "YCbCr"), "44x": ([(4, 4), (1, 1), (1, 1)], "YCbCr"), "gray": ([(1, 1)], "Grayscale"), "cmyk": ([(1, 1)] * 4, "CMYK")}
SIZES = [(65535, 17), (17, 65535), (65500, 40), (8191, 4097), (1, 65535), (65535, 1), (2, 2), (65535, 2), (3, 65534)]


def run(verbose=True):
    rng = np.random.default_rng(9)
    bad = []
    for w, h in SIZES:
        for name, (samp, ct) in KINDS.items():
            case = T._batch_case(rng, w, h, samp, ct, kind="sparse")
            outs, path = T._run_batch([case])
            oc, qts, coefs, ct_, cw, ch = case
            want = O.pixels_from_coefficients(oc, qts, coefs, cw, ch, ct_.upper())
            ok = np.array_equal(np.asarray(outs[0]).ravel(), np.asarray(want).ravel())
            if not ok:
                bad.append((w, h, name, path))
            if verbose:
                print(f"{w}x{h} {name}: path {path} {'ok' if ok else 'MISMATCH'}", flush=True)
    # reduced sizes (Decoder::scale -> dct_scale 4 / 2 / 1): the band kernel's longest grids — 8,192 MCU rows, 4,096 MCU columns
    for w, h in [(65535, 17), (17, 65535), (8191, 4097), (1, 65535), (65535, 1)]:
        for name in ("420", "444", "440", "gray", "cmyk"):
            samp, ct = KINDS[name]
            for scale in (4, 2, 1):
                case = T._scaled_case(rng, w, h, samp, ct, scale)
                outs, path = T._run_batch([case])
                oc, qts, coefs, ct_, cw, ch = case
                want = O.pixels_from_coefficients(oc, qts, coefs, cw, ch, ct_.upper())
                ok = np.array_equal(np.asarray(outs[0]).ravel(), np.asarray(want).ravel()) and "-s%d" % scale in path
                if not ok:
                    bad.append((w, h, name, scale, path))
                if verbose:
                    print(f"{w}x{h} {name} scale {scale}: path {path} {'ok' if ok else 'MISMATCH'}", flush=True)
    return bad


if __name__ == "__main__":
    b = run()
    print("bad", len(b))
    sys.exit(1 if b else 0)
