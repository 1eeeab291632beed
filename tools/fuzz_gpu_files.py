#!/usr/bin/env python3
"""Differential fuzzing with VALID streams over geometry: PIL-encoded JPEGs of random size, sampling (4:4:4 / 4:2:2 / 4:2:0 /
gray / CMYK), quality, progressive or not, restart intervals, optimised tables — through `Decoder.decode()` (the Worker route:
planes kept as coefficients, fused kernels per image), `Decoder.scale()` + decode (reduced IDCTs) and `Pipeline.decode()` in
batches (host and device entropy decoding); every result byte for byte against the oracle's decode of the same bytes.
    python tools/fuzz_gpu_files.py <seed> <files>      (on the GPU box; prints "bad 0")"""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import oracle as O, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
from PIL import Image


def make(rng):
    pick = int(rng.integers(0, 5))
    if pick == 0: w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    elif pick == 1: w, h = int(rng.integers(600, 720)), int(rng.integers(1, 70))
    elif pick == 2: w, h = int(rng.integers(1, 70)), int(rng.integers(300, 420))
    elif pick == 3: w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
    else: w, h = int(rng.integers(700, 1500)), int(rng.integers(200, 700))
    rgb = synth.synthetic_rgb(w, h, seed=int(rng.integers(0, 1 << 30)))
    mode = int(rng.integers(0, 6))
    kw = {"quality": int(rng.integers(30, 100))}
    if mode == 4:
        img = Image.fromarray(rgb[..., 0])
    elif mode == 5:
        img = Image.fromarray(np.concatenate([rgb, rgb[..., :1]], axis=2), mode="CMYK")
    else:
        img = Image.fromarray(rgb)
        kw["subsampling"] = ["4:4:4", "4:2:2", "4:2:0", "4:2:0"][mode % 4]
    if int(rng.integers(0, 4)) == 0: kw["progressive"] = True
    if int(rng.integers(0, 4)) == 0: kw["optimize"] = True
    r = int(rng.integers(0, 5))
    if r == 0: kw["restart_marker_rows"] = int(rng.integers(1, 4))
    elif r == 1: kw["restart_marker_blocks"] = int(rng.integers(1, 40))
    buf = io.BytesIO()
    img.save(buf, format="JPEG", **kw)
    return buf.getvalue(), (w, h, mode, kw)


def run(seed, n_files, verbose=True):
    rng = np.random.default_rng(seed)
    files, meta = zip(*[make(rng) for _ in range(n_files)])
    want = [O.decode(f).pixels for f in files]
    bad = 0
    def check(tag, i, got):
        nonlocal bad
        if isinstance(got, Exception) or not np.array_equal(np.asarray(got).ravel(), want[i]):
            bad += 1
            if verbose: print("MISMATCH", tag, i, meta[i], type(got).__name__, flush=True)
    for i, f in enumerate(files):  # one image per Decoder: the Worker route
        d = J.Decoder(f)
        check("decoder", i, d.decode())
        d.close()
    scaled = 0
    for i, f in enumerate(files):  # reduced IDCTs
        if i % 3: continue
        w, h = meta[i][0], meta[i][1]
        req = (max(1, w // int(rng.integers(2, 9))), max(1, h // int(rng.integers(2, 9))))
        d = J.Decoder(f); d.read_info(); d.scale(*req)
        got = d.decode(); d.close()
        ref = O.decode(f, scale_to=req).pixels
        scaled += 1
        if not np.array_equal(np.asarray(got).ravel(), ref):
            bad += 1
            if verbose: print("MISMATCH scaled", i, meta[i], req, flush=True)
    p = J.Pipeline(threads=8)
    on_device = entry_walk = handed_back = 0  # (valid streams: what the device decoders took, how many through the walk that reads the entry lists, how many came back)
    for flags in ({"device_entropy": False}, {"device_entropy": True}, {"device_entropy": True, "dense": True}):
        for a in range(0, n_files, 24):
            out = p.decode(list(files[a:a + 24]), **flags)
            for k, got in enumerate(out): check("pipeline" + str(flags), a + k, got)
            t = p.timings()
            on_device += t["images_device_entropy"]; entry_walk += t["images_entry_pixels"]; handed_back += t["images_device_rejected"]
    p.close()
    if verbose: print("seed", seed, "files", n_files, "scaled decodes", scaled, "on the device", on_device, "entry-list walk", entry_walk, "handed back", handed_back, "bad", bad, flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
