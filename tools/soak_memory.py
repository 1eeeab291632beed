"""Soak: many Decoder / Pipeline / Batch life cycles; device memory in use (hipMemGetInfo through torch) and host RSS must level off.
python tools/soak_memory.py   (GPU box)"""
import io, os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np, torch
import jpeg_decoder_amd as J, synth
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
from PIL import Image

def jpeg(w, h, sub="4:2:0", **kw):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(w, h, seed=w)).save(buf, format="JPEG", quality=85, subsampling=sub, **kw); return buf.getvalue()
files = [jpeg(1920, 1080), jpeg(640, 480, "4:2:2"), jpeg(333, 211, "4:4:4"), jpeg(512, 512, progressive=True), jpeg(1280, 720), jpeg(64, 64)]
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
def phase(name, fn, rounds, per):
    marks = []
    for r in range(rounds):
        for _ in range(per): fn()
        marks.append(used())
    d = [f"{m[0]:.0f}/{m[1]:.0f}" for m in marks]
    grow_dev = marks[-1][0] - marks[len(marks) // 2][0]
    grow_host = marks[-1][1] - marks[len(marks) // 2][1]
    print(f"{name}: device MiB / host RSS MiB after each round: {' '.join(d)}  -> growth over the second half: {grow_dev:+.0f} MiB device, {grow_host:+.0f} MiB host", flush=True)
    return grow_dev, grow_host
torch.zeros(1, device="cuda")
print("start", used())
bad = 0
g = phase("Decoder(data).decode() x 6 files", lambda: [J.Decoder(f).decode() for f in files], 6, 40); bad += g[0] > 64 or g[1] > 256
p = J.Pipeline(threads=8)
g = phase("Pipeline.decode(60 mixed files)", lambda: p.decode(files * 10, device_entropy=True), 6, 15); bad += g[0] > 64 or g[1] > 256
def newpipe():
    q = J.Pipeline(threads=4); q.decode(files * 4, device_entropy=True); q.close()
g = phase("Pipeline create / decode / close", newpipe, 6, 10); bad += g[0] > 64 or g[1] > 256
import test_gpu_parity as T
T.J = J
rng = np.random.default_rng(1)
cases = [T._batch_case(rng, 320, 200, [(2, 2), (1, 1), (1, 1)], "YCbCr") for _ in range(8)]
g = phase("Batch create / upload / decode / download / close", lambda: T._run_batch(cases), 6, 25); bad += g[0] > 64 or g[1] > 256
print("leaks suspected" if bad else "levels off")
sys.exit(1 if bad else 0)
