import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import jpeg_decoder_amd as J
os.environ["JPGPU_PIPE_DUMP_COEFS"]="/tmp/dump"
name=sys.argv[1]
data=open(name,'rb').read()
desc, host = J.Decoder(data, device=-1).decode_coefficients()
p=J.Pipeline(threads=4)
out=p.decode([data], device_entropy=True)
print("on device", p.timings()["images_device_progressive"])
zz=[0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63]
for c in range(desc.ncomp):
    d=np.fromfile(f"/tmp/dump.c{c}.bin", np.int16)
    h=np.asarray(host[c], np.int16)
    n=min(len(d),len(h))
    bad=np.nonzero(d[:n]!=h[:n])[0]
    print("comp",c,"differing coefficients",len(bad))
    for i in bad[:12]:
        blk, nat = divmod(int(i),64)
        print("   block",blk,"natural",nat,"zigzag",zz.index(nat),"device",int(d[i]),"host",int(h[i]))
