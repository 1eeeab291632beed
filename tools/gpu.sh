#!/bin/bash
# tools/gpu.sh — the command of ONE gpurun call, assembled from steps (rounds 1-3 kept ~70 one-off scripts, one per call: git history).
#
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh <outdir> <step> [<step> ...]'
#
# Every step writes under gpurun_out/<outdir>/ and prints a one-line summary.  Steps (arguments after a colon, comma-separated
# where there are several; environment assignments are written ENV=VAL and separated by '+'):
#   tests                         all `-m gpu` tests + smoke()
#   tests:<pytest -k expression>  a subset
#   bench[:name[:args]]           one bench.py line (default: the driver's command `--gpus 1 --steps 20 --warmup 5`); args with '+' for spaces
#   k:<name>[:ENV=VAL+...]        the K figure only (500 steps, no extras), for A/B of builds / knobs; JPGPU_LIBRARY=<alt .so> selects a build
#   e:<name>[:ENV=VAL+...]        the E figures (256 / 1,024 / 4,096 files) + kernel phases of one sub-batch alone
#   ab:<reps>                     K of libjpgpu.so against every jpeg-decoder_amd/libjpgpu_alt*.so, interleaved <reps> times
#   abe:<reps>                    the same for the E figures
#   wl:<workload>[:batch[:subs[:ENV=VAL+...]]]  K of another workload (bench.py --workload)
#   trace:<name>:<cmd with + for spaces>     rocprofv3 --kernel-trace --stats of a command -> <name>_kernel_stats.json
#   tl:<name>:<cmd>[:min us]      kernels AND copies of the command's last burst of device activity on one time axis (tools/prof_timeline.py) -> <name>_device_timeline.txt
#   pmc:<name>:<cmd>              instruction / wait / LDS counters of a command (two passes of eight counters) -> <name>_pmc.json
#   disp:<name>:<kernel substr>:<last N>[:ENV]   three counter passes over the one-sub-batch pipeline calls, every dispatch of the
#                                 matching kernels on its own line (launches of very different work: the sync passes) -> <name>_dispatches.jsonl
#   lanes:<name>:<cmd>            SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU per kernel -> <name>_lanes.json
#   traffic:<workload>:<path>:<kernel substr>[:extra bench args]   FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic.json entry
#   pipe256[:ENV=VAL+...]         kernel trace + traffic + instruction counters of a 256-file pipeline call as ONE sub-batch
#   timeline:<n files>[:<e2e_bench flags, + for spaces>[:ENV=VAL+...]]   JPGPU_PIPE_TRACE host + device timeline of an <n>-file call (tools/e2e_bench.py)
#   kinds                         other sampling kinds / restart markers / 2160p through E (tools/e2e_bench.py) -> e2e_other_kinds.jsonl
#   fuzz[:n]                      the differential fuzzers, n cases each (default 200)
#   latency                       one image through Decoder / a one-image pipeline (tools/decoder_latency.py)
#   forcedist                     bench.py --force-dist (the N > 1 code path with one rank on RCCL), e2e included
#   hostbench:<file>[:reps]       tools/host_bench.cpp on the box's CPU: host entropy decoding of one file, scan by scan for progressive ones
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
OUT=$1; shift
O=$R/gpurun_out/$OUT; mkdir -p "$O"
PIPE_CMD="python $R/tools/pipe_calls.py --images 256 --calls 6 --one-sub-batch"
KARGS="--steps 500 --warmup 50 --no-cpu-baseline --no-classes --no-k4096 --no-scale-anchor --no-cpu-budget --no-e2e --min-seconds 0"

envs() { echo "${1//+/ }"; }   # "A=1+B=2" -> "A=1 B=2"
cmdline() { local c="${1//+/ }"; c="${c// bench.py/ $R/bench.py}"; echo "${c// tools\// $R/tools/}"; }  # the same for a command run from /tmp (rocprofv3): repo paths made absolute

summary_line() {  # <file> <label>
python - "$1" "$2" <<'PY'
import json, sys
try:
    lines = open(sys.argv[1]).read().strip().splitlines()
    d = json.loads(lines[-1])  # the contract line (<= 4 KB); the detail document is the `bench_detail:` line in front of it
    assert len(lines[-1]) <= 4096, "contract line of %d bytes" % len(lines[-1])
    for l in lines[:-1]:
        if l.startswith("bench_detail: "):
            d = json.loads(l[len("bench_detail: "):])
except Exception as e:  # noqa: BLE001
    print(sys.argv[2], "NO LINE:", e); sys.exit(0)
r = d.get("roofline", {})
out = [sys.argv[2], d.get("config", {}).get("kernel_path"), "K ms", d.get("ms_per_step"), "kernel", r.get("kernel_ms_per_launch"), "frac", r.get("frac"), "ok", d.get("verified_vs_oracle")]
k4 = d.get("k_4096") or {}
if k4: out += ["| k4096", k4.get("ms_per_step"), k4.get("roofline_frac")]
bc = d.get("roofline_by_class") or {}
if "classes_on_device" in bc: out += ["| dev classes", bc["classes_on_device"]["kernel_ms_per_launch"]]
e = d.get("e2e") or {}
for k in ("256", "1024", "4096"):
    if k in e: out += ["| E%s" % k, e[k]["total_ms"], "ms", e[k]["images_per_s"], "img/s", e[k].get("frac_of_floor"), e[k]["verified_vs_oracle"]]
for k in ("256_pinned_input", "4096_pinned_input"):
    if k in e: out += ["| E" + k, e[k].get("total_ms"), e[k].get("verified_vs_oracle", e[k].get("error"))]
if "kernels_256_one_sub_batch" in e: out += ["| alone", e["kernels_256_one_sub_batch"]["kernel_ms"]]
if "tower_progressive_256" in e: out += ["| prog", e["tower_progressive_256"].get("images_per_s"), e["tower_progressive_256"].get("images_device_entropy")]
if "error" in e: out += ["| E ERROR", e["error"]]
print(*out)
PY
}

kernel_table() {  # <rocprof dir>
python - "$1" <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/*.db"):
    c = sqlite3.connect(f)
    for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc"):
        if "jpgpu" in r[0]: print("%-64s calls %5d avg %9.1f us min %9.1f max %9.1f vgpr %s lds %s scratch %s" % (r[0].split("(")[0][:64], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
PY
}

for step in "$@"; do
  IFS=: read -r what a1 a2 a3 a4 <<< "$step"
  echo "=== $step"
  case $what in
    tests)
      if [ -n "$a1" ]; then timeout 2400 python -m pytest tests -m gpu -q -x -k "$(envs "$a1")" > $O/pytest_gpu.log 2>&1
      else timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; fi
      echo "pytest exit $?" >> $O/pytest_gpu.log; tail -n 6 $O/pytest_gpu.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $O/smoke.txt ;;
    bench)
      name=${a1:-bench_driver_command}; args=$(envs "${a2:---gpus+1+--steps+20+--warmup+5}")
      timeout 1500 python bench.py $args > $O/$name.json 2> $O/$name.err; summary_line $O/$name.json $name ;;
    k)
      env $(envs "$a2") timeout 600 python bench.py $KARGS > $O/k_$a1.json 2> $O/k_$a1.err; summary_line $O/k_$a1.json "k_$a1" ;;
    e)
      env JPGPU_BATCH_KERNEL_TIMES=1 $(envs "$a2") timeout 900 python bench.py --no-cpu-baseline --no-classes --no-k4096 --no-scale-anchor --no-cpu-budget --steps 30 --min-seconds 0 > $O/e_$a1.json 2> $O/e_$a1.err
      summary_line $O/e_$a1.json "e_$a1" ;;
    ab|abe)
      for rep in $(seq 1 ${a1:-2}); do
        for lib in main $(ls jpeg-decoder_amd/libjpgpu_alt*.so 2>/dev/null); do
          n=$(basename $lib .so); L=""; [ $lib != main ] && L="JPGPU_LIBRARY=$R/$lib"
          if [ $what = ab ]; then env $L timeout 600 python bench.py $KARGS > $O/ab_${n}_$rep.json 2>> $O/ab.err
          else env JPGPU_BATCH_KERNEL_TIMES=1 $L timeout 900 python bench.py --no-cpu-baseline --no-classes --no-k4096 --no-scale-anchor --no-cpu-budget --steps 30 --min-seconds 0 > $O/ab_${n}_$rep.json 2>> $O/ab.err; fi
          summary_line $O/ab_${n}_$rep.json "${n}_$rep"
        done
      done ;;
    wl)
      extra=""; [ -n "$a2" ] && extra="--batch $a2"; [ -n "$a3" ] && extra="$extra --sub-batches $a3"
      tag=$a1${a2:+_$a2}${a3:+_$a3}${a4:+_${a4//[^A-Za-z0-9]/_}}
      env $(envs "$a4") timeout 600 python bench.py --workload $a1 $extra --steps 300 --warmup 50 --no-cpu-baseline --no-classes --min-seconds 0 > $O/wl_$tag.json 2>> $O/wl.err
      summary_line $O/wl_$tag.json "wl_$tag" ;;
    trace)
      cmd=$(cmdline "$a2"); rm -rf $O/trace_$a1
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_$a1 -o t -- $cmd > $O/trace_$a1.log 2>&1)
      python tools/prof_summary.py $O/trace_$a1 > $O/${a1}_kernel_stats.json 2>> $O/summary.err; kernel_table $O/trace_$a1 ;;
    tl)
      cmd=$(cmdline "$a2"); rm -rf $O/tl_$a1
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl_$a1 -o t -- $cmd > $O/tl_$a1.log 2>&1)
      python tools/prof_timeline.py $O/tl_$a1 --min-us ${a3:-30} > $O/${a1}_device_timeline.txt 2>> $O/summary.err; head -n 3 $O/${a1}_device_timeline.txt; rm -rf $O/tl_$a1 ;;
    pmc)
      cmd=$(cmdline "$a2"); rm -rf $O/pmc1_$a1 $O/pmc2_$a1
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d $O/pmc1_$a1 -o p -- $cmd > $O/pmc1_$a1.log 2>&1)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY -d $O/pmc2_$a1 -o p -- $cmd > $O/pmc2_$a1.log 2>&1)
      python tools/prof_summary.py $O/pmc1_$a1 $O/pmc2_$a1 > $O/${a1}_pmc.json 2>> $O/summary.err; head -c 3000 $O/${a1}_pmc.json ;;
    disp)
      cmd=$PIPE_CMD; E=$(envs "$a4"); rm -rf $O/d1_$a1 $O/d2_$a1 $O/d3_$a1
      (cd /tmp && env $E timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/d1_$a1 -o p -- $cmd > $O/d1_$a1.log 2>&1)
      (cd /tmp && env $E timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/d2_$a1 -o p -- $cmd > $O/d2_$a1.log 2>&1)
      (cd /tmp && env $E timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_FLAT -d $O/d3_$a1 -o p -- $cmd > $O/d3_$a1.log 2>&1)
      python tools/prof_summary.py --dispatches "$a2" "$a3" $O/d1_$a1 $O/d2_$a1 $O/d3_$a1 > $O/${a1}_dispatches.jsonl 2>> $O/summary.err; head -c 6000 $O/${a1}_dispatches.jsonl ;;
    lanes)
      cmd=$(cmdline "$a2"); rm -rf $O/lanes_$a1
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/lanes_$a1 -o p -- $cmd > $O/lanes_$a1.log 2>&1)
      python tools/prof_summary.py $O/lanes_$a1 > $O/${a1}_lanes.json 2>> $O/summary.err; head -c 3000 $O/${a1}_lanes.json ;;
    traffic)
      extra=$(envs "$a4"); rm -rf $O/tf_$a1 $O/tw_$a1
      B="python $R/bench.py --workload $a1 --steps 40 --warmup 10 --no-cpu-baseline --no-classes --no-k4096 --no-scale-anchor --no-cpu-budget --no-e2e --min-seconds 0 $extra"
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/tf_$a1 -o p -- $B > /dev/null 2>&1)
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/tw_$a1 -o p -- $B > /dev/null 2>&1)
      (cd tools && python make_pmc_traffic.py "$a1:$a2" $O/tf_$a1 $O/tw_$a1 $O/pmc_traffic.json $a3)
      # the same entry into THIS copy's profiles/ (where bench.py reads it), so that a `bench` step later in the same call prints the
      # traffic taken on the code it runs; the file comes back under $O and is committed from there
      [ -n "$TRAFFIC_INSTALL" ] && (cd tools && python make_pmc_traffic.py "$a1:$a2" $O/tf_$a1 $O/tw_$a1 $R/$TRAFFIC_INSTALL $a3 > /dev/null) ;;
    pipe256)
      E=$(envs "$a1"); rm -rf $O/pipe256 $O/pipe256_fetch $O/pipe256_write $O/pipe256_pmc1
      (cd /tmp && env $E timeout 400 rocprofv3 --kernel-trace --stats -d $O/pipe256 -o k -- $PIPE_CMD > $O/pipe256.log 2>&1)
      (cd /tmp && env $E timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pipe256_fetch -o p -- $PIPE_CMD > $O/pipe256_fetch.log 2>&1)
      (cd /tmp && env $E timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pipe256_write -o p -- $PIPE_CMD > $O/pipe256_write.log 2>&1)
      (cd /tmp && env $E timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $O/pipe256_pmc1 -o p -- $PIPE_CMD > $O/pipe256_pmc1.log 2>&1)
      python tools/prof_summary.py $O/pipe256_pmc1 $O/pipe256_fetch $O/pipe256_write $O/pipe256 > $O/pipe256_kernel_stats.json 2>> $O/summary.err
      grep "call ms" $O/pipe256.log | tail -3; kernel_table $O/pipe256 ;;
    timeline)
      env $(envs "$a3") JPGPU_PIPE_TRACE=1 JPGPU_BATCH_KERNEL_TIMES=1 timeout 400 python tools/e2e_bench.py --images ${a1:-4096} --device-entropy --no-download --rounds 3 $(envs "$a2") > $O/timeline_${a1:-4096}${a2//+/}.txt 2>&1; tail -n 3 $O/timeline_${a1:-4096}${a2//+/}.txt ;;
    kinds)
      : > $O/e2e_other_kinds.jsonl
      for r in 1 4; do timeout 300 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --restart-rows $r 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl; done
      for s in 4:4:4 4:2:2; do timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --subsampling $s 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl; done
      timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --file tests/golden/benches/tower_grayscale.jpg 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
      timeout 200 python tools/e2e_bench.py --images 1024 --device-entropy --no-download --rounds 4 --file tests/golden/reftest/rgb.jpg 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
      timeout 300 python tools/e2e_bench.py --images 1024 --width 3840 --height 2160 --device-entropy --no-download --rounds 3 2>/dev/null | tail -1 >> $O/e2e_other_kinds.jsonl
      cat $O/e2e_other_kinds.jsonl | cut -c 1-300 ;;
    fuzz)
      n=${a1:-200}
      timeout 900 python tools/fuzz_gpu_geometry.py 61001 $n 2>&1 | tail -4 > $O/fuzz.txt
      timeout 900 python tools/fuzz_gpu_files.py 63001 $n 2>&1 | tail -4 >> $O/fuzz.txt
      timeout 900 python tools/fuzz_gpu.py 65001 $n 2>&1 | tail -3 >> $O/fuzz.txt
      timeout 600 python tools/fuzz_gpu_worker.py 64001 $((n / 2 + 1)) 2>&1 | tail -3 >> $O/fuzz.txt
      timeout 600 python tools/fuzz_gpu_progressive.py 66001 $((n / 2 + 1)) 2>&1 | tail -3 >> $O/fuzz.txt
      cat $O/fuzz.txt ;;
    latency)
      timeout 600 python tools/decoder_latency.py > $O/decoder_latency.txt 2>&1; tail -n 12 $O/decoder_latency.txt ;;
    forcedist)
      timeout 900 python bench.py --force-dist --no-k4096 --no-scale-anchor --no-cpu-budget --no-cpu-baseline --no-classes --e2e-images 256,1024 > $O/bench_force_dist.json 2> $O/bench_force_dist.err
      summary_line $O/bench_force_dist.json force_dist ;;
    hostbench)
      g++ -O3 -march=native -std=c++17 -I. tools/host_bench.cpp jpeg-decoder_amd/csrc/host/frontend.cpp jpeg-decoder_amd/csrc/image_job.cpp -o /tmp/host_bench 2> $O/host_bench_build.err
      (grep -m1 "model name" /proc/cpuinfo; /tmp/host_bench $a1 ${a2:-60}) > $O/host_bench_$(basename $a1).txt 2>&1; cat $O/host_bench_$(basename $a1).txt ;;
    *) echo "unknown step $what" ;;
  esac
done
