#!/bin/bash
# Round 3, second half: refresh of the artefacts under profiles/round3 after the device entropy route changed (emission + expansion, 16 streams / 24 hardware
# queues, host-side changes): bench lines, rocprofv3 kernel-trace stats of the bench command (K region), the device-entropy pipeline's kernels with
# their HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), fuzz runs, full GPU test run.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof2; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -n 2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -n 2 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --force-dist --no-e2e --no-k4096 --no-cpu-baseline --no-classes > $O/bench_force_dist.json 2> $O/bench_force_dist.err
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-e2e --no-k4096 --min-seconds 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o f -- $CMD > $O/trace.log 2>&1
cd $R
python tools/prof_summary.py $O/trace > $O/kernel_trace_stats.json 2> $O/summary.err
# the device-entropy pipeline, 256 x 1080p as ONE sub-batch (kernels one after the other): durations, then HBM traffic per kernel
cat > /tmp/many.py <<PY
import io, os, sys, time
os.environ["JPGPU_PIPE_DEV_SUB"], os.environ["JPGPU_PIPE_MAX_DEV_SUBS"] = "256", "1"
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import jpeg_decoder_amd as J, synth
from PIL import Image
files = []
for i in range(4):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=0x5EED + i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
files = [files[i % 4] for i in range(256)]
p = J.Pipeline()
for _ in range(6):
    t0 = time.perf_counter(); p.decode(files, device_entropy=True, download=False); print("call ms", (time.perf_counter() - t0) * 1e3, flush=True)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/pipe256 -o k -- python /tmp/many.py > $O/pipe256.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pipe256_fetch -o p -- python /tmp/many.py > $O/pipe256_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pipe256_write -o p -- python /tmp/many.py > $O/pipe256_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pipe256_pmc1 -o p -- python /tmp/many.py > $O/pipe256_pmc1.log 2>&1
cd $R
python tools/prof_summary.py $O/pipe256_pmc1 $O/pipe256_fetch $O/pipe256_write $O/pipe256 > $O/pipe256_kernel_stats.json 2>> $O/summary.err
# the same call as the pipeline splits it by default (two sub-batches of 128) and 4096 files: per-sub-batch device timeline
JPGPU_PIPE_TRACE=1 JPGPU_BATCH_KERNEL_TIMES=1 timeout 300 python tools/e2e_bench.py --images 4096 --device-entropy --no-download --rounds 3 > $O/trace4096.txt 2>&1
# other sampling kinds through the same route
for r in 1 4; do timeout 300 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --restart-rows $r 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl; done
for s in 4:4:4 4:2:2; do timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --subsampling $s 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl; done
timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --file tests/golden/benches/tower_grayscale.jpg 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --file tests/golden/reftest/rgb.jpg 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
timeout 300 python tools/e2e_bench.py --images 1024 --width 3840 --height 2160 --device-entropy --no-download --rounds 3 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
# fuzz campaigns (differential, vs the oracle)
timeout 600 python tools/fuzz_gpu_geometry.py 51001 250 2>&1 | tail -4 > $O/fuzz.txt
timeout 600 python tools/fuzz_gpu_files.py 53001 200 2>&1 | tail -4 >> $O/fuzz.txt
timeout 600 python tools/fuzz_gpu.py 55001 300 2>&1 | tail -3 >> $O/fuzz.txt
timeout 300 python tools/fuzz_gpu_worker.py 54001 150 2>&1 | tail -3 >> $O/fuzz.txt
cat $O/fuzz.txt
python - <<PY
import json
d = json.loads(open("$O/bench_driver_command.json").read().strip().splitlines()[-1])
e = d["e2e"]
print("driver cmd: K", d["value"], d["ms_per_step"], d["roofline"]["frac"], "k4096", d["k_4096"]["roofline_frac"], "| E", {k: (e[k]["total_ms"], e[k]["images_per_s"]) for k in ("256", "1024", "4096") if k in e}, "| alone", e.get("kernels_256_one_sub_batch", {}).get("kernel_ms"), "| cpu e2e", d["cpu_baseline_e2e"]["images_per_s"])
PY
python - <<PY
import json
d = json.load(open("$O/pipe256_kernel_stats.json"))
for k, v in d.items():
    print("%-60s calls %4d avg %9.1f us  pmc %s" % (k[:60], v["calls"], v["avg_us"], {a: b for a, b in v.get("pmc", {}).items() if a in ("FETCH_SIZE", "WRITE_SIZE")}))
PY
