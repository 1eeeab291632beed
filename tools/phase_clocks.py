"""Per-phase wave time of the 4:2:0 strip walk.  Needs a diagnostic build of the library (-DJPGPU_PHASE_CLOCKS) loaded through
JPGPU_LIBRARY; runs the default bench workload in-process and prints, per phase, the share of the waves' wall time spent on
their own work and waiting for the barrier that closes the phase.  tools/gpu_phase_clocks.sh drives it."""
import contextlib, ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-classes"] + sys.argv[1:])
lib = ctypes.CDLL(os.environ["JPGPU_LIBRARY"])
out = (ctypes.c_ulonglong * 16)()
assert lib.jpgpu_debug_phase_clocks(out, 0) == 0
v = list(out)
names = ["read_block (LDS -> regs, dequantise)", "  barrier after read_block", "transform (IDCT, samples -> LDS)", "  barrier after transform",
         "colour (upsample, convert, store)", "  barrier after colour", "stage (global loads -> LDS)", "  barrier after stage", "prologue (set-up, seam round, first stage)", "epilogue (closing row)"]
tot = float(sum(v[:10]))
print(buf.getvalue().strip()[:400])
print(f"waves {v[15]}, mean wall time per wave {tot / max(1, v[15]):.0f} shader cycles")
for n, x in zip(names, v[:10]):
    print(f"{n:44s} {100.0 * x / tot:6.2f} %")
