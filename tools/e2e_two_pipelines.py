"""Two (or N) Pipeline objects driven from as many threads, each decoding batches of F x 1080p back to back with device
entropy decoding: the upload of one call overlaps the kernels of the other.  python tools/e2e_two_pipelines.py [n_pipelines [files per call]]"""
import io, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import jpeg_decoder_amd as J, synth
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
from PIL import Image
files = []
for i in range(8):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
files = [files[i % 8] for i in range(NF)]
for npipes in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2, 3]):
    pipes = [J.Pipeline(threads=max(4, 32 // npipes)) for _ in range(npipes)]
    for p in pipes: p.decode(files, device_entropy=True, download=False)
    calls = 8 if NF <= 1024 else 4
    def work(p):
        for _ in range(calls): p.decode(files, device_entropy=True, download=False)
    ts = [threading.Thread(target=work, args=(p,)) for p in pipes]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"{npipes} pipeline(s): {npipes * calls * NF / dt:,.0f} images/s  ({dt / calls * 1e3:.2f} ms per round of {npipes} x {NF})", flush=True)
    for p in pipes: p.close()
