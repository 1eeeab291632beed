#!/usr/bin/env python3
"""profiles/roundN/README.md from the files themselves: every row's commit and figures are READ from the file it describes (its own
provenance fields, or the round's `_provenance.json` written by the closing gpurun call) — only the `what` texts are typed here.
    python tools/make_profiles_readme.py [round6]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    with open(path) as f:
        return json.load(f)


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def facts_bench(path):
    d = last_json_line(path)
    r, s = d.get("roofline", {}), d.get("e2e_summary", {})
    out = [f"{len(json.dumps(d))} B line", f"K {d.get('ms_per_step')} ms/step, kernel {r.get('kernel_ms_per_launch')} ms = {r.get('frac')} of 8 TB/s",
           f"traffic {r.get('traffic')} B (commit {str(r.get('traffic_commit'))[:7]}, on these kernel sources: {r.get('traffic_taken_on_these_kernel_sources')})"]
    for k in ("e2e_256_ms", "e2e_4096_ms", "progressive_256_images_per_s", "progressive_256_on_device", "progressive_distinct_4096_images_per_s",
              "cpu_budget_default_vs_best_forced_worst", "sharded_images_per_s"):
        if s.get(k) is not None:
            out.append(f"{k} {s[k]}")
    if d.get("bench_seconds"):
        out.append(f"{d['bench_seconds']} s in the process")
    return "; ".join(out)


def facts_trace(path):
    d = load(path)
    out = []
    for k, v in d.items():
        if "s420_kernel<" in k:
            out.append(f"`{k.split('::')[-1]}` {v['calls']} calls avg {v['avg_us']} µs (min {v['min_us']}), {v['vgpr']} VGPR, scratch {v['scratch']}")
    return "; ".join(out)


def facts_traffic(path):
    out = []
    for k, v in load(path).items():
        out.append(f"`{k}` {v['hbm_bytes_per_decode']:,} B per launch, commit `{v['commit'][:7]}`, sources sha `{v['pixel_kernel_sources_sha256'][:12]}`")
    return "; ".join(out)


def facts_pytest(path):
    t = open(path).read()
    m = re.search(r"(\d+ passed[^\n]*)", t)
    s = re.search(r"(smoke ok[^\n]*)", t)
    return "; ".join(x.group(1) for x in (m, s) if x)


def facts_fuzz(path):
    t = open(path).read()
    bad = re.findall(r"bad (\d+)", t)
    return f"{len(bad)} fuzzer summaries, bad counts {sorted(set(bad))}"


def facts_jsonl(path):
    out = []
    for l in open(path):
        if l.startswith("{"):
            d = json.loads(l)
            out.append(f"{d['config']['name']} {d['roofline']['frac']}")
    return ", ".join(out)


ROWS = [  # file, what (typed), how the facts are read
    ("10_bench_driver_command.json", "the driver's command, `python bench.py --gpus 1 --steps 20 --warmup 5`: the `bench_detail:` line, then the contract line (last)", facts_bench),
    ("08_bench_force_dist.json", "`bench.py --force-dist`: the N > 1 code path with one rank on RCCL", facts_bench),
    ("05_pytest_gpu.txt", "`pytest -m gpu` + `smoke()`", facts_pytest),
    ("pmc_traffic.json", "HBM traffic of the headline kernel by the counters (FETCH_SIZE x 2 + WRITE_SIZE, separate `--pmc` passes); `tests/test_profiles.py` fails when the pixel-kernel sources change", facts_traffic),
    ("06_kernel_trace_stats_bench_K.json", "`rocprofv3 --kernel-trace --stats` of the bench command's K region (tracer attached)", facts_trace),
    ("07_fuzz.txt", "the differential fuzzers, 300 cases each (the entry-list walk as the default)", facts_fuzz),
    ("09_other_workloads_bench.jsonl", "K of every other workload of `bench.py` (fraction of 8 TB/s)", facts_jsonl),
    ("01_scalar_chain_ubench.txt", "`tools/ubench_scalar_chain.hip`: cycles per instruction of a dependent chain, by instruction kind — what the wave-per-scan design was sized with", None),
    ("02_progressive_wave_per_scan.txt", "row n3 this round: calls by frame count on both routes, the hand-scheduled loop alone, kernel trace and counters of the walk, the build history", None),
    ("03_progressive_cost_model.txt", "the dispatcher's cost model against the measured calls", None),
    ("11_entry_list_walk.txt", "VERDICT r5 #4: the 4:2:0 walk that reads the entry lists against the expansion kernel — four builds, same-box A/Bs, segment lengths, kernel trace and counters", None),
    ("12_pipe256_one_sub_batch_kernel_stats_and_pmc.json", "the device-entropy pipeline's kernels, 256 files as ONE sub-batch: durations, registers, LDS, traffic, instruction counters", None),
    ("13_fuzz_long.txt", "the differential fuzzers at 5,000 cases each on the final code (valid streams: 7,726 device decodes, 2,556 through the entry-list walk, 54 handed back — every one with status 0x41: a stream of a few chunks whose sync passes had not settled within the call's launches)", facts_fuzz),
    ("14_soak_memory.txt", "`tools/soak_memory.py` on the final code: Decoder / Pipeline / Batch life cycles, device memory and host RSS after each round", None),
    ("04_launch_shape.txt", "headline kernel: segment cuts, work-table forms and workgroup counts on three boxes (VERDICT r5 #5)", None),
]


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "round6"
    d = os.path.join(ROOT, "profiles", rnd)
    prov = load(os.path.join(d, "_provenance.json")) if os.path.exists(os.path.join(d, "_provenance.json")) else {}
    lines = [f"# profiles/{rnd} — what each file is", "",
             "Generated by `tools/make_profiles_readme.py` from the files' own fields; do not edit.  One MI355X per `gpurun` call, 16 host CPUs granted.",
             f"Closing call: commit `{str(prov.get('closing_run_commit'))[:12]}` (`{prov.get('order', '')}`); files without a commit of their own in the third column were written by it.",
             "", "| file | what | read from the file |", "|---|---|---|"]
    for name, what, facts in ROWS:
        p = os.path.join(d, name)
        if not os.path.exists(p):
            continue
        try:
            f = facts(p) if facts else open(p).readline().strip()
        except Exception as e:  # a malformed file must show up in the README, not stop it
            f = f"UNREADABLE: {e!r}"
        lines.append(f"| `{name}` | {what} | {f} |")
    listed = {r[0] for r in ROWS} | {"README.md", "_provenance.json"}
    for name in sorted(os.listdir(d)):
        if name not in listed:
            lines.append(f"| `{name}` | (not described) | {open(os.path.join(d, name), errors='replace').readline().strip()[:160]} |")
    open(os.path.join(d, "README.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
