#!/usr/bin/env python3
"""Latency of ONE image through the drop-in surfaces: Decoder(data).decode() (the reference's API over the Worker ABI) and a
one-image Pipeline call.  python tools/decoder_latency.py [file ...]"""
import io
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)


def med(f, n=30):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    files = sys.argv[1:]
    datas = [(os.path.basename(f), open(f, "rb").read()) for f in files]
    if not datas:
        import synth
        from PIL import Image
        sizes = [tuple(int(v) for v in t.split('x')) for t in os.environ['SIZES'].split(',')] if os.environ.get('SIZES') else [(512, 512), (1920, 1080), (3840, 2160)]
        for (w, h) in sizes:
            buf = io.BytesIO()
            Image.fromarray(synth.synthetic_rgb(w, h, seed=1)).save(buf, format="JPEG", quality=85, subsampling="4:2:0")
            datas.append((f"synthetic {w}x{h} 4:2:0 q85", buf.getvalue()))
    p = J.Pipeline(threads=1)
    for name, d in datas:
        a = med(lambda: J.Decoder(d).decode())
        b = med(lambda: p.decode([d], device_entropy=False))
        c = med(lambda: p.decode([d], device_entropy=True))
        print(f"{name}: Decoder.decode {a:.2f} ms | Pipeline (1 image, host entropy) {b:.2f} ms | Pipeline (1 image, device entropy) {c:.2f} ms")


if __name__ == "__main__":
    main()
