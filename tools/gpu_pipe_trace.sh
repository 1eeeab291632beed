JPGPU_PIPE_TRACE=1 python - 2>&1 <<'PY' | tail -40
import io, sys, time
sys.path[:0] = ['.', 'tests']
import synth
from PIL import Image
import jpeg_decoder_amd as J
buf = io.BytesIO()
Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=1)).save(buf, format="JPEG", quality=85, subsampling="4:2:0")
d = buf.getvalue()
p = J.Pipeline(threads=2)
for i in range(5):
    t0 = time.perf_counter(); p.decode([d], device_entropy=True); print("call ms", (time.perf_counter() - t0) * 1e3, flush=True)
for i in range(3):
    t0 = time.perf_counter(); p.decode([d], device_entropy=True, download=False); print("call (no download) ms", (time.perf_counter() - t0) * 1e3, flush=True)
PY
