import ctypes as C, os, sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import test_gpu_progw_asm as T
import jpeg_decoder_amd as J
dev = C.CDLL(J._native.LIB_PATH)
dev.jpgpu_selftest_refine_fast_ms.argtypes=[C.c_void_p, C.c_uint32, C.c_uint32]; dev.jpgpu_selftest_refine_fast_ms.restype=C.c_float
dev.jpgpu_selftest_refine_fast.argtypes=[C.c_void_p, C.c_uint32]
def case(nz, run=0, extra_kind=None):
    c=T.Case()
    c.pos=63; c.win=0xAAAAAAAAAAAAAAAA & ~1; c.nx=0x55555555; c.dp=1; c.end=64; c.k=1; c.nz=nz; c.neg=nz & 0x0f0f0f0f0f0f0f0f; c.al=0; c.eob=0
    for i in range(64):
        c.lut6[i]=T._entry(1,1,run,0,1); c.w[i]=0x5a5a5a5a; c.acc[i]=0
    return c
reps=20000
for name, nz, run in (("no corrections, run 0 (63 symbols per call)", 0, 0),
                      ("every other coefficient non-zero, run 0: one correction per symbol (32 symbols)", 0xAAAAAAAAAAAAAAAA, 0),
                      ("every other coefficient non-zero, run 1: rank select + 2 corrections (16 symbols)", 0xAAAAAAAAAAAAAAAA, 1),
                      ("3 of 4 non-zero, run 0: three corrections per symbol (16 symbols)", 0xEEEEEEEEEEEEEEEE, 0)):
    c=case(nz, run)
    chk=T.Case(); C.memmove(C.byref(chk), C.byref(c), C.sizeof(c))
    arr=(T.Case*1)(chk); dev.jpgpu_selftest_refine_fast(C.byref(arr),1)
    # how many symbols does one call take?  count via new_nz bits
    syms=bin(arr[0].new_nz).count("1"); code=arr[0].code
    for waves in (1, 1024):
        ms=dev.jpgpu_selftest_refine_fast_ms(C.byref(c), reps, waves)
        print(f"{name}: code {code}, {syms} symbols/call; {waves} wave(s): {ms:.3f} ms -> {ms*1e-3*2.4e9/reps:.0f} cycles per call, {ms*1e-3*2.4e9/reps/max(syms,1):.0f} per symbol")
