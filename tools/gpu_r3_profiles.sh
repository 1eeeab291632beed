#!/bin/bash
# Round 3: the artefacts summarised under profiles/round3 — bench lines, rocprofv3 kernel-trace stats of the bench command, PMC passes
# (each --pmc set in its own run with --kernel-trace only), HBM traffic, the kernel trace of the device-entropy pipeline, fuzz runs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for wl in 1080p-444 1080p-422 1080p-440 1080p-411 1080p-gray 1080p-cmyk 1080p-cmyk-2211 1080p-ycck-2212 1080p-444+gray 1080p-420-scale4 1080p-420-scale2 1080p-420-scale1 1080p-444-scale4; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-classes --min-seconds 0 >> $O/bench_other.jsonl 2>> $O/bench_other.err
done
timeout 200 python bench.py --workload 2160p-420 --no-cpu-baseline --no-classes --min-seconds 0 >> $O/bench_other.jsonl 2>> $O/bench_other.err
timeout 200 python bench.py --workload 2160p-420 --batch 64 --sub-batches 8 --no-cpu-baseline --no-classes --min-seconds 0 >> $O/bench_other.jsonl 2>> $O/bench_other.err
timeout 300 python bench.py --force-dist --no-e2e --no-k4096 --no-cpu-baseline --no-classes > $O/bench_force_dist.json 2> $O/bench_force_dist.err
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-e2e --no-k4096 --min-seconds 0"       # the bench's K region: 500 steps after 50 warm-up, + class legs
PCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-k4096 --no-classes --min-seconds 0"  # counter passes serialise dispatches
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o f -- $CMD > $O/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc1 -o p -- $PCMD > $O/pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc2 -o p -- $PCMD > $O/pmc2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3 -o p -- $PCMD > $O/pmc3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc4 -o p -- $PCMD > $O/pmc4.log 2>&1
cd $R
python tools/prof_summary.py $O/trace $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/kernel_trace_stats_and_pmc.json 2> $O/summary.err
cp profiles/round2/pmc_traffic.json $O/pmc_traffic.json
(cd tools && python make_pmc_traffic.py 1080p-420:fused420 $O/pmc3 $O/pmc4 $O/pmc_traffic.json s420_)
for spec in "1080p-cmyk-2211:fused420x4-2211:r4_" "1080p-ycck-2212:fused420x4-2212:r4_" "1080p-420-scale4:generic:idct_planes upsample_color"; do
  IFS=: read wl path pat <<< "$spec"
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/tf_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-classes --min-seconds 0 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/tw_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-classes --min-seconds 0 > /dev/null 2>&1
  (cd $R/tools && python make_pmc_traffic.py $wl:$path $O/tf_$wl $O/tw_$wl $O/pmc_traffic.json $pat)
done
cd $R
# the device-entropy pipeline, 256 x 1080p: which kernels the call's GPU time is made of
cat > /tmp/many.py <<PY
import io, os, sys, time
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
cd $R
python tools/prof_summary.py $O/pipe256 > $O/pipe256_kernel_stats.json 2>> $O/summary.err
# fuzz campaigns (differential, vs the oracle)
timeout 600 python tools/fuzz_gpu_geometry.py 31001 250 2>&1 | tail -4 > $O/fuzz.txt
JPGPU_420_STRIP=0 timeout 300 python tools/fuzz_gpu_geometry.py 32001 60 2>&1 | tail -3 >> $O/fuzz.txt
timeout 600 python tools/fuzz_gpu_files.py 33001 150 2>&1 | tail -4 >> $O/fuzz.txt
timeout 300 python tools/fuzz_gpu_worker.py 34001 150 2>&1 | tail -3 >> $O/fuzz.txt
cat $O/fuzz.txt
cut -c1-400 $O/bench_other.jsonl | python -c "
import sys, json
for l in sys.stdin:
    pass
" 2>/dev/null
python - <<PY
import json
for l in open("$O/bench_other.jsonl"):
    try:
        d = json.loads(l); print(f"{d['config']['name']:18s} {d['config']['images_per_gpu']:5d} x{d['config']['sub_batches']} {d['config']['kernel_path']:16s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} {d['verified_vs_oracle']}")
    except Exception as e: print("bad line", e)
d = json.loads(open("$O/bench_driver_command.json").read().strip().splitlines()[-1])
print("driver cmd", d["value"], d["roofline"]["frac"], {k: v.get("frac") for k, v in d["roofline_by_class"].items() if isinstance(v, dict)}, d["k_4096"]["roofline_frac"], d["e2e"]["256"]["images_per_s"], d["e2e"]["256"].get("kernels_only_images_per_s"), d["e2e"]["4096"]["images_per_s"])
print(json.dumps(json.load(open("$O/pmc_traffic.json")), indent=0)[:100])
PY
tail -c 800 $O/kernel_trace_stats_and_pmc.json
