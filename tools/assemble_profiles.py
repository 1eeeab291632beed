#!/usr/bin/env python3
"""profiles/round6 from a closing gpurun call's output directory (tools/_gpu_call.sh: tests -> traffic -> kernel trace -> bench -> ...):
copies / composes the numbered files and `_provenance.json`, then regenerates the README (tools/make_profiles_readme.py).
    python tools/assemble_profiles.py gpurun_out/r6w
The commit recorded is tools/.head_commit — what the tree was at when the call was launched."""
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(R, sys.argv[1]) if not os.path.isabs(sys.argv[1]) else sys.argv[1]
    dst = os.path.join(R, "profiles", "round6")
    head = open(os.path.join(R, "tools", ".head_commit")).read().strip()

    def w(name, text):
        open(os.path.join(dst, name), "w").write(text)

    def rd(name):
        return open(os.path.join(src, name)).read()

    w("01_scalar_chain_ubench.txt", "tools/ubench_scalar_chain.hip on an MI355X (closing run, commit %s): what ONE wave pays per instruction of a dependent chain.\n\n" % head[:12]
      + rd("ubench_scalar_chain.txt"))
    pmc, ks = json.loads(rd("prog256_pmc.json")), json.loads(rd("prog256_kernel_stats.json"))
    t = ["Progressive frames on the device, round 6: one wave per scan (csrc/huff_prog_wave.hpp) — closing run, commit %s, one MI355X" % head[:12], "",
         "1. tools/prog_calls.py: N copies of tests/golden/benches/tower_progressive.jpg (512x512 4:4:4, 60 kB, 10 scans) through jpgpu_pipeline_decode,",
         "   JPEG bytes in host memory -> RGB in HBM, device percent 100 = every frame's scans on the device, 0 = the host route (round 4's); last two of four calls;",
         "   `walk+scan ms` = the walk kernels' + range scans' time summed over the call's (overlapping) launches.", "", rd("progressive_calls.txt"), "",
         "2. The hand-scheduled refinement loop in isolation (tools/progw_asm_bench.py: one wave / a wave per SIMD walks a synthetic block; cycles at 2.4 GHz):", "",
         rd("progw_asm_bench.txt"), "", "3. rocprofv3 --kernel-trace --stats of `tools/prog_calls.py --images 256 --calls 4` (the walk and what follows it):", ""]
    for k, v in ks.items():
        t.append("   %-70s %s" % (k[:70], json.dumps({a: v[a] for a in ("calls", "avg_us", "min_us", "vgpr", "sgpr", "lds", "scratch") if a in v})))
    t += ["", "4. Counters of the walk kernel, same command (two passes of eight; per dispatch averages; 2,560 waves = 256 frames x 10 scans):", ""]
    for k, v in pmc.items():
        if "progw" in k:
            t.append("   " + json.dumps(v))
    t += ["", "   SQ_INSTS_SALU / 256 frames = 6.1 M scalar instructions per frame, SQ_INSTS_VALU 1.5 M: the walk is scalar code.  At 4,096 frames a sub-batch of",
          "   1,024 frames takes 16.3 ms alone = 15.9 us per frame = 9.8 M CU-cycles per frame on 256 CUs: 62 % of the scalar units' one instruction per cycle and CU.",
          "   (7 / 8 waves per SIMD instead of 6 — amdgpu_waves_per_eu, 72 / 64 registers — change nothing: docs/history/round6.md.)", "",
          "5. How the walk kernel of 256 frames got here (same input, walk kernel only; docs/history/round6.md has the list with reasons):",
          "   31.3 ms round 5 (a lane per scan) -> 19.7 first wave-per-scan build -> 14.2 / 12.3 (without fences) restructured C++ -> 11.8 branch-light -> 10.5 refinement loop",
          "   hand-scheduled -> 9.2 second-level look-up in the loop, one atomic per coefficient, no pending load across the block loop -> 9.1 AC first / DC first loops",
          "   hand-scheduled -> 7.8-7.9 launch order: the longest chain's track first."]
    w("02_progressive_wave_per_scan.txt", "\n".join(t) + "\n")
    calls = rd("progressive_calls.txt")
    disp = calls.split("== the dispatcher's own choice")[1].split("== per scan")[0].strip()

    def ms_of(n, pc, distinct=False):  # the two `call ms` figures of a block of progressive_calls.txt
        key = "== 4096 distinct frames, device" if distinct else "== %d frames, device percent %d" % (n, pc)
        if key not in calls:
            return "-"
        vals = [l.split()[2] for l in calls.split(key)[1].split("==")[0].strip().splitlines() if l.startswith("call ms")]
        return "-".join(sorted(set(vals), key=float)) if vals else "-"

    rows = ["  N      device (ms)      host (ms)"]
    for n in (64, 128, 192, 256, 1024, 4096):
        rows.append("  %-6d %-16s %s" % (n, ms_of(n, 100), ms_of(n, 0)))
    rows.append("  4096 distinct (268 MB of scans)  %s" % ms_of(0, 0, True))
    w("03_progressive_cost_model.txt", "\n".join([
        "The progressive dispatcher's cost model (csrc/pipeline.cpp, progressive_share_for_the_device) against measurements — closing run, commit %s" % head[:12], "",
        "  device = 1.1 ms + max(longest scan's bytes x 470 ns, all scans' bytes x 0.19 ns) + 6 us per frame",
        "  host   = 1.5 ms + all scans' bytes x 17 ns / min(worker threads, CPUs granted)          the device must be ahead by a tenth", "",
        "Measured (tools/prog_calls.py, the tables of 02_*): tower_progressive.jpg x N on 16 granted CPUs / 32 threads, `call ms` of the last two of four calls"] + rows + [
        "", "The model's figures and the route it takes, from the dispatcher's own trace lines of this run:", disp, "",
        "Re-measure: tools/prog_calls.py --images N --percent 100 | 0; the constants are kDev* / kHost* in csrc/pipeline.cpp."]) + "\n")
    shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, "05_pytest_gpu.txt"))
    open(os.path.join(dst, "05_pytest_gpu.txt"), "a").write(rd("smoke.txt"))
    for a, b in (("bench_K_kernel_stats.json", "06_kernel_trace_stats_bench_K.json"), ("fuzz.txt", "07_fuzz.txt"), ("bench_force_dist.json", "08_bench_force_dist.json"),
                 ("other_workloads_bench.jsonl", "09_other_workloads_bench.jsonl"), ("bench_driver_command.json", "10_bench_driver_command.json"),
                 ("pmc_traffic_installed.json", "pmc_traffic.json"), ("pipe256_kernel_stats.json", "12_pipe256_one_sub_batch_kernel_stats_and_pmc.json")):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
    json.dump({"closing_run_commit": head, "gpurun_out": os.path.basename(src),
               "order": "tests -> traffic (installed for the bench step) -> kernel trace -> bench (driver's command) -> force-dist -> pipe256 -> progressive profiles -> "
                        "micro-benchmarks -> fuzzers (300 cases) -> other workloads"}, open(os.path.join(dst, "_provenance.json"), "w"), indent=1)
    subprocess.run([sys.executable, os.path.join(R, "tools", "make_profiles_readme.py"), "round6"], stdout=subprocess.DEVNULL, check=True)
    d = json.loads([l for l in rd("bench_driver_command.json").splitlines() if l.startswith("{")][-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "e2e_summary", "bench_seconds")}, indent=1))


if __name__ == "__main__":
    main()
