O=gpurun_out/${1:-lat}; mkdir -p $O
JPGPU_DECODER_NO_DEVICE_ENTROPY=1 python tools/decoder_latency.py > $O/latency_worker_route.txt 2>&1
python tools/decoder_latency.py tests/golden/benches/tower.jpg tests/golden/benches/large_image.jpg > $O/latency_default.txt 2>&1
python tools/decoder_latency.py >> $O/latency_default.txt 2>&1
JPGPU_DECODER_NO_DEVICE_ENTROPY=1 JPGPU_DECODER_TRACE=1 python - > $O/trace.txt 2>&1 <<'PY'
import io, sys
sys.path[:0] = ['.', 'tests']
import synth
from PIL import Image
import jpeg_decoder_amd as J
buf = io.BytesIO()
Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=1)).save(buf, format="JPEG", quality=85, subsampling="4:2:0")
d = buf.getvalue()
for _ in range(4):
    J.Decoder(d).decode()
PY
cat $O/latency_worker_route.txt $O/latency_default.txt; tail -12 $O/trace.txt
