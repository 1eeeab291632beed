import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import oracle as O
import jpeg_decoder_amd as J
for name in ("tools/tmp/fuzz_5.jpg", "tools/tmp/fuzz_53.jpg"):
    data=open(name,'rb').read()
    want=O.decode(data)
    p=J.Pipeline(threads=4)
    out=p.decode([data]*3, device_entropy=True)
    t=p.timings()
    for k,got in enumerate(out):
        if isinstance(got, Exception): print(name, k, "error", got); continue
        g=np.asarray(got).reshape(want.height, want.width, -1); w=want.pixels.reshape(want.height, want.width, -1)
        d=np.argwhere((g!=w).any(axis=2))
        print(name, k, "on device", t["images_device_progressive"], "diff pixels", len(d), "first", d[:3].tolist(), "last", d[-2:].tolist() if len(d) else None, "blocks", sorted({(int(y)//8,int(x)//8) for y,x in d})[:6])
    p.close()
