"""One image through Pipeline with device entropy decoding: where the milliseconds go (jpgpu_pipeline_last_timings)."""
import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
import synth
from PIL import Image
p = J.Pipeline(threads=int(os.environ.get("THREADS", "1")))
for (w, h) in [(512, 512), (1920, 1080), (3840, 2160)]:
    buf = io.BytesIO()
    Image.fromarray(synth.synthetic_rgb(w, h, seed=1)).save(buf, format="JPEG", quality=85, subsampling="4:2:0")
    d = buf.getvalue()
    for dev in (True, False):
        for _ in range(5): p.decode([d], device_entropy=dev)
        acc = {}
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            p.decode([d], device_entropy=dev)
            t = p.timings()
            for k in ("headers_ms", "setup_ms", "entropy_and_upload_ms", "kernels_ms", "download_ms", "total_ms"):
                acc[k] = acc.get(k, 0.0) + getattr(t, k) if not isinstance(t, dict) else acc.get(k, 0.0) + t[k]
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{w}x{h} device_entropy={dev}: wall {wall:.2f} ms | " + " ".join(f"{k[:-3]} {v / n:.2f}" for k, v in acc.items()), flush=True)
