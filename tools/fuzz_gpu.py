#!/usr/bin/env python3
"""Differential fuzzing on the GPU: damaged variants of twelve base streams (with / without restart markers, gray, RGB,
optimised tables) in ONE pipeline call with the entropy decoding forced onto the device; every result — pixels or the kind of
error — must equal the oracle's.  python tools/fuzz_gpu.py <seed> <variants per base>  (run on the GPU box; prints "bad 0")."""
import sys, os, io
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import oracle as O, refimages as R, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
from PIL import Image
def pil(w,h,sub,gray=False,q=85,**kw):
    buf=io.BytesIO(); rgb=synth.synthetic_rgb(w,h,seed=w+h); Image.fromarray(rgb[...,0] if gray else rgb).save(buf,format="JPEG",quality=q,subsampling=sub,**kw); return buf.getvalue()
bases=[open(os.path.join(R.GOLDEN,"benches/tower.jpg"),"rb").read(), pil(96,64,"4:2:0"), pil(200,120,"4:4:4",q=95), pil(150,90,"4:2:0",gray=True), open(os.path.join(R.GOLDEN,"reftest/rgb.jpg"),"rb").read(), pil(96,64,"4:2:0",restart_marker_blocks=5), pil(320,240,"4:2:2",restart_marker_rows=1), pil(640,480,"4:2:0",optimize=True),
       # restart streams whose segments span several chunks of the chunk decoder (round 3: every segment in chunk slots of its own), gray ones too
       pil(640,480,"4:2:0",restart_marker_rows=1), pil(640,480,"4:2:0",restart_marker_rows=4), pil(400,300,"4:4:4",gray=True,restart_marker_rows=1), pil(1280,720,"4:2:0",restart_marker_blocks=20)]
rng=np.random.default_rng(int(sys.argv[1])); per=int(sys.argv[2])
files=[]
for base in bases:
    sos=base.rfind(b"\xff\xda")
    for t in range(per):
        d=bytearray(base)
        for _ in range(int(rng.integers(1,4))):
            pos=int(rng.integers(sos+12,len(d)-2)); mode=int(rng.integers(0,5))
            if mode==0: d[pos]^=1<<int(rng.integers(0,8))
            elif mode==1: del d[pos]
            elif mode==2: d[pos]=0xFF
            elif mode==3: d.insert(pos,int(rng.integers(0,256)))
            else: del d[pos:pos+int(rng.integers(1,40))]
        files.append(bytes(d))
os.environ["JPGPU_PIPE_FORCE_DEVICE"]="1"
p=J.Pipeline(threads=16)
out=p.decode(files, device_entropy=True)
t=p.timings(); print({k:t[k] for k in ("images_ok","images_device_entropy","images_device_rejected","total_ms")})
bad=0; ok=err=0
for i,(f,got) in enumerate(zip(files,out)):
    try: want=O.decode(f).pixels
    except O.OracleError as e: want=e
    if isinstance(want,O.OracleError):
        err+=1
        if not (isinstance(got,J.Error) and got.kind==want.kind): bad+=1; print("MISMATCH kind",i,type(got),getattr(got,'kind',None),want.kind)
    else:
        ok+=1
        if isinstance(got,Exception) or not np.array_equal(got,want): bad+=1; print("MISMATCH pixels",i,type(got))
print("files",len(files),"ok",ok,"err",err,"bad",bad)
