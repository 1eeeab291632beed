import io, os, sys, time
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import jpeg_decoder_amd as J, synth
from PIL import Image
files = []
for i in range(4):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=0x5EED + i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
files = [files[i % 4] for i in range(256)]
p = J.Pipeline()
ts = []
for _ in range(14):
    p.decode(files, device_entropy=True, download=False); ts.append(round(p.timings()["total_ms"], 2))
print(os.environ.get("JPGPU_SYNC_TAIL", "default"), ts)
