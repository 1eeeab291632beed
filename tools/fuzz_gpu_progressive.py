#!/usr/bin/env python3
"""Differential fuzzing on the GPU, progressive streams: damaged variants of five base streams (four progressive, one sequential) through the pipeline, with and without
device entropy decoding for the sequential ones;
every result — pixels or the kind of error — must equal the oracle's.  python tools/fuzz_gpu_progressive.py <seed> <variants>"""
import sys, os, io
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import oracle as O, refimages as R, synth
import jpeg_decoder_amd as J
J.process_init()  # GPU_MAX_HW_QUEUES before the HIP runtime starts (opt-in since round 4)
from PIL import Image
def pil(w,h,sub,q=85,**kw):
    buf=io.BytesIO(); rgb=synth.synthetic_rgb(w,h,seed=w+h); Image.fromarray(rgb).save(buf,format="JPEG",quality=q,subsampling=sub,**kw); return buf.getvalue()
bases=[open(os.path.join(R.GOLDEN,"benches/tower_progressive.jpg"),"rb").read(), open(os.path.join(R.GOLDEN,"reftest/progressive3.jpg"),"rb").read(), pil(200,120,"4:2:0",progressive=True), pil(96,64,"4:4:4",progressive=True), pil(96,64,"4:2:0")]
rng=np.random.default_rng(int(sys.argv[1])); per=int(sys.argv[2])
files=[]
for base in bases:
    for t in range(per):
        d=bytearray(base)
        lo=max(2,len(d)//4)
        for _ in range(int(rng.integers(0,3))):
            pos=int(rng.integers(lo,len(d)-2)); mode=int(rng.integers(0,4))
            if mode==0: d[pos]^=1<<int(rng.integers(0,8))
            elif mode==1: del d[pos]
            elif mode==2: d[pos]=0xFF
            else: del d[pos:pos+int(rng.integers(1,40))]
        files.append(bytes(d))
wants=[]
for f in files:
    try: wants.append(O.decode(f).pixels)
    except O.OracleError as e: wants.append(e)
p=J.Pipeline(threads=16)
for flags in ({"device_entropy":False},{"device_entropy":True}):
    out=p.decode(files, **flags)
    bad=ok=err=0
    for i,(want,got) in enumerate(zip(wants,out)):
        if isinstance(want,O.OracleError):
            err+=1
            if not (isinstance(got,J.Error) and got.kind==want.kind): bad+=1; print("MISMATCH kind",i,getattr(got,'kind',None),want.kind)
        else:
            ok+=1
            if isinstance(got,Exception) or not np.array_equal(got,want): bad+=1; print("MISMATCH pixels",i,type(got))
    print(flags,"files",len(files),"ok",ok,"err",err,"bad",bad)
