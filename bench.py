#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configs: the headline driver.

  metric : megapixels/s decoded (batch, whole node); % of HBM roofline for the pixel kernels
  A "step" = one pass of the hot path (dequantize + IDCT + upsample + YCbCr->RGB) over the rank's whole shard,
  coefficients already resident in HBM, RGB left resident in HBM.

  N = 1 (configs[1]): 1920x1080 baseline 4:2:0 YCbCr, batch of 256 images, one launch group per step.
  N > 1 (configs[2]): 3840x2160 baseline 4:2:0, 4096 images in total, sharded by image with
          jpeg_decoder_amd.distributed.shard() (512 per GPU at N = 8; STRONG scaling: the total stays 4096), decoded
          in launch groups; `value` has no collective in it (there is no data-path collective), `value_with_gather`
          is the same job with north_star's final RCCL gather of the pixels to rank 0 inside the timed region,
          issued per launch group so that it overlaps the decode of the next one.

  Launch: `python bench.py --gpus N` spawns the N ranks itself (torch.distributed.run, 127.0.0.1) when it is not
  already running under a launcher; under torchrun it is one rank.  A mismatch between --gpus and WORLD_SIZE, or fewer
  visible GPUs than N, is an error — never a silent 1-GPU run.  `--dry-run` walks launcher / sharding / gather / JSON on
  CPU (gloo) and measures nothing.

  OUTPUT.  Rank 0 prints TWO lines: first `bench_detail: {...}` — everything measured, also written to
  gpurun_out/bench_detail.json — and LAST the contract line: one JSON object of at most 4 KB (contract_line(): the
  headline, `roofline`, `cpu_baseline`, and ten scalars of the side legs as `e2e_summary`).  The side legs themselves
  (kernel-only figure at 4,096 images, the N > 1 job on one GPU, arithmetic classes, JPEG bytes -> RGB through
  jpgpu_pipeline_decode incl. BASELINE configs[3], the CPU-budget points, the oracle's whole decode) are
  tools/bench_e2e.py; each is checked against the oracle and none of them is inside `value`."""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a
# queue run one after the other: jpgpu_pipeline_decode keeps up to 12 sub-batches in flight on 12 compute + 4 copy streams.
# Read once when the runtime initialises, so it is set before anything touches HIP (jpeg_decoder_amd._native does the same).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# (the library reads this once, at its first device-entropy launch: events around the phases — the e2e legs report kernel time per phase)
os.environ.setdefault("JPGPU_BATCH_KERNEL_TIMES", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
sys.modules.setdefault("bench", sys.modules[__name__])  # (tools/bench_e2e.py does `import bench`: the same module when this file is the script)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
CONTRACT_LINE_MAX = 4096  # bytes of the last stdout line (the driver keeps the tail of stdout; round 5's 21 KB line was cut)
DETAIL_PATH = os.path.join("gpurun_out", "bench_detail.json")

from bench_shard import CONFIG3_IMAGES_TOTAL, CONFIG3_WORKLOAD, E2E_SHARDED_TOTAL, WORKLOADS, PixelGather, Shard, build_variants  # noqa: E402 (tools/bench_shard.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 500 at N = 1, 40 at N > 1)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: 50 at N = 1, 5 at N > 1)")
    ap.add_argument("--workload", default="", choices=[""] + sorted(WORKLOADS),
                    help="default: 1080p-420 at N = 1 (configs[1]), 2160p-420 at N > 1 (configs[2])")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (weak scaling; default: the workload's at N = 1)")
    ap.add_argument("--images-total", type=int, default=0,
                    help="images in the whole job, sharded over the ranks (strong scaling; default at N > 1: 4096)")
    ap.add_argument("--sub-batches", type=int, default=0, help="launch groups per step (default: 1 at N = 1; at N > 1 up to 8, none smaller than 64 images)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU work for the cpu_baseline sample (the whole-decode comparator takes half)")
    ap.add_argument("--generic", action="store_true", help="force the two-kernel generic path")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-steps", type=int, default=3, help="timed steps of the decode + gather region (N > 1)")
    ap.add_argument("--no-classes", action="store_true", help="skip the per-arithmetic-class timings (N = 1)")
    ap.add_argument("--class-steps", type=int, default=60)
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo walk through launcher, sharding, gather and the JSON line")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="after the K timed steps: repeat the step for at least this long and report it as `sustained` (0 = off)")
    ap.add_argument("--no-k4096", action="store_true", help="skip the 4096-image kernel-only figure (N = 1, default workload)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the JPEG-bytes -> RGB legs and their CPU comparator (N = 1, default workload)")
    ap.add_argument("--no-scale-anchor", action="store_true", help="skip the N > 1 job (2160p x 4096) on this one GPU (N = 1, default workload)")
    ap.add_argument("--no-cpu-budget", action="store_true", help="skip the E-against-host-CPUs points (N = 1, default workload)")
    ap.add_argument("--cpu-budget-matrix", action="store_true", help="all five CPU counts x four inputs (default: 2 / 8 / 16 CPUs x the library's mode and forced host staging)")
    ap.add_argument("--e2e-images", default="256,1024,4096", help="files per jpgpu_pipeline_decode call of the e2e block")
    ap.add_argument("--e2e-total", type=int, default=E2E_SHARDED_TOTAL, help="files of the e2e leg at N > 1 / --force-dist, sharded over the ranks")
    ap.add_argument("--e2e-encoder", default="auto", choices=["auto", "pillow", "builtin"],
                    help="who writes the e2e block's JPEG files: Pillow (libjpeg-turbo) or tools/baseline_encoder.py; auto = Pillow if importable")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: initialise torch.distributed (nccl = RCCL) with one rank anyway and run the collectives of the N > 1 path on the device")
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed launches BEFORE the warm-up steps (the GPU leaves its clock transient after an idle period); "
                         "default: enough to make settle + warm-up = 50 launches at N = 1")
    ap.add_argument("--detail", default=DETAIL_PATH, help="where the detail document goes (also printed as the `bench_detail:` line)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n, argv, port=None):
    """The command `python bench.py --gpus n ...` re-executes itself as (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def launch_ranks(args, argv):
    if not args.dry_run:
        have = visible_gpus()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible — refusing to run a smaller job under that label")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launcher_command(args.gpus, argv), env=env)


# ---------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------
def algorithmic_bytes_per_image(comps, out_bytes):
    """coefficient bytes in (int16) + pixel bytes out; q-tables ignored."""
    return sum(c.block_width * c.block_height * 64 * 2 for c in comps) + out_bytes


def effective_cpus():
    """CPUs this process can really use: affinity mask and the cgroup CPU quota (the GPU box runs the container with
    cpu.max = 16 CPUs although 256 hardware threads are visible)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, -(-q // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n



def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(O, ocomps, qts, coefs, w, h, ct, target_seconds):
    """Oracle ("port" of the reference's scalar path) on the CPUs the process may use, bounded sample.  The comparator is
    built on this box with -O3 -march=native and every task in flight has its own coefficient buffers."""
    cores = effective_cpus()
    flags = O.use_native_build()  # compiles oracle/liboracle_native.so here; falls back to the portable -O2 build
    distinct = max(4 * cores, 16)
    pool = [[np.array(c, copy=True) for c in coefs] for _ in range(distinct)]
    n0 = max(cores, 4)
    t0 = time.perf_counter()
    O.batch_pixels(ocomps, qts, [pool[i % distinct] for i in range(n0)], w, h, ct.upper(), cores, keep_outputs=False)
    dt = time.perf_counter() - t0
    n = int(max(n0, min(65536, n0 * target_seconds / max(dt, 1e-3))))
    n = (n // cores) * cores or cores
    t0 = time.perf_counter()
    O.batch_pixels(ocomps, qts, [pool[i % distinct] for i in range(n)], w, h, ct.upper(), cores, keep_outputs=False)
    dt = time.perf_counter() - t0
    return {"value": round(n * w * h / 1e6 / dt, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"{n} images {w}x{h} of the same workload ({distinct} distinct coefficient sets, none shared by tasks in flight), "
                      f"pixel pipeline only (coefficients -> pixels), {cores} threads one image per task, {dt:.1f} s; "
                      f"gcc {flags}; {cpu_model()}; the crate's own x86 build would add SSSE3 IDCT / colour kernels (not bit-compatible with its scalar path)"}


def measured_traffic(workload, path):
    """HBM bytes per decode from the committed rocprofv3 PMC passes (profiles/roundN/pmc_traffic.json,
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate --pmc runs) and where they come from; (None, why, None) if
    not measured.  The counters need rocprofv3 around the process: they are NOT taken in this run, and the line says so — with the
    commit the passes were taken on and whether the pixel kernels' sources have changed since (tools/kernel_sources.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_sources
    for rnd in ("round6", "round5", "round4", "round3", "round2", "round1"):
        rel = os.path.join("profiles", rnd, "pmc_traffic.json")
        try:
            t = json.load(open(os.path.join(ROOT, rel)))
            e = t.get(f"{workload}:{path}")
            if e:
                same = e.get("pixel_kernel_sources_sha256") == kernel_sources.sha256() if e.get("pixel_kernel_sources_sha256") else None
                return (e["hbm_bytes_per_decode"], f"{rel} (rocprofv3 --pmc passes of this workload committed with {rnd}; read from the file, not measured in this run)",
                        {"traffic_commit": e.get("commit"), "traffic_taken_on_these_kernel_sources": same})
        except (OSError, ValueError, KeyError):
            continue
    return None, "not measured for this workload / kernel path", None


def time_steps(torch, dev, dist, stream, steps, body):
    """`steps` calls of body() between barrier + synchronize on both sides; (wall seconds, GPU ms per step from events
    on the stream the kernels are launched on)."""
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream the kernels are launched on
    for _ in range(steps):
        body()
    ev1.record()
    while not ev1.query():  # (spin until the last step has run, THEN synchronise: a blocking wait's wake-up took a millisecond on one box of
        pass                #  the pool — 7 % of the driver's twenty steps of 0.65 ms; the contract's synchronise on both sides stays)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps


# ---------------------------------------------------------------------------------------------------------------
# the contract line (the LAST stdout line; everything else is the detail document)
# ---------------------------------------------------------------------------------------------------------------
def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _num(d, *path):
    """d[path...] if it is there and a number / bool / None-free scalar, else None."""
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d if isinstance(d, (int, float, bool)) else None


def e2e_summary(detail):
    """Ten scalars of the side legs for the contract line; None where a leg did not run."""
    e = detail.get("e2e") or {}
    pts = (e.get("cpu_budget") or {}).get("points") or []
    at2 = next((p for p in pts if p.get("cpus") == 2), None)
    worst = None  # the library's own choice against the better of the two forced modes, worst point (1.0 = the default is the better one)
    for p in pts:
        mine = _num(p, "pageable_input", "images_per_s")
        forced = [x for x in (_num(p, "pageable_input_host_staging", "images_per_s"), _num(p, "pageable_input_host_light", "images_per_s")) if x]
        if mine and forced:
            worst = min(worst, mine / max(forced)) if worst is not None else mine / max(forced)
    out = {
        "e2e_256_ms": _num(e, "256", "total_ms"), "e2e_4096_ms": _num(e, "4096", "total_ms"),
        "e2e_4096_images_per_s": _num(e, "4096", "images_per_s"),
        "to_host_4096_frac_of_floor": _num(e, "to_host_4096", "frac_of_floor"),
        "progressive_256_images_per_s": _num(e, "tower_progressive_256", "images_per_s"),
        "progressive_256_on_device": _num(e, "tower_progressive_256", "images_device_progressive"),
        "progressive_4096_images_per_s": _num(e, "tower_progressive_4096", "images_per_s"),
        "progressive_distinct_4096_images_per_s": _num(e, "progressive_distinct_4096", "images_per_s"),
        "cpu_budget_2cpus_images_per_s": _num(at2, "pageable_input", "images_per_s"),
        "cpu_budget_default_vs_best_forced_worst": round(worst, 4) if worst is not None else None,
    }
    if isinstance(e.get("sharded"), dict):  # N > 1 / --force-dist: E per rank, the job's rate
        out = {"sharded_images_per_s": _num(e, "sharded", "images_per_s"), "sharded_total_ms": _num(e, "sharded", "total_ms"),
               "sharded_mode": e["sharded"].get("mode"), **{k: v for k, v in out.items() if v is not None}}
    return out


def contract_line(detail, detail_path=None):
    """The driver's record: bench.py's contract keys + `roofline` + `cpu_baseline` + `e2e_summary`, at most CONTRACT_LINE_MAX
    bytes as JSON (free text is cut, never a number).  `detail` is the full document of this run."""
    line = _pick(detail, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                          "dtype", "data", "verified_vs_oracle", "n_ranks_seen", "value_with_gather", "ms_per_step_with_gather", "gather_ms",
                          "gather_verified", "gather_error", "dry_run", "gather_checked"))
    cfg = detail.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "name", "images_total", "images_per_gpu", "sub_batches", "kernel_path", "range_class", "parallelism",
                                  "collective_backend", "images_total_requested", "shrunk_to_fit_hbm"))
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:240]
    if "roofline" in detail:
        line["roofline"] = _pick(detail["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_commit",
                                                      "traffic_taken_on_these_kernel_sources", "algorithmic_bytes_per_launch", "kernel_ms_per_launch"))
        ref = detail["roofline"].get("stream_reference")
        if ref:
            line["roofline"]["kernel_vs_torch_add"] = ref.get("kernel_vs_reference")
    if "cpu_baseline" in detail:
        line["cpu_baseline"] = _pick(detail["cpu_baseline"], ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(detail["cpu_baseline"].get("sample", ""))[:200]
    if "sustained" in detail:
        line["sustained_value"] = _num(detail, "sustained", "value")
    for k, sub in (("k_4096_frac", ("k_4096", "roofline_frac")), ("scale_anchor_frac", ("scale_anchor", "roofline_frac")),
                   ("scale_anchor_value", ("scale_anchor", "value")), ("class0_frac", ("roofline_by_class", "class0", "frac")),
                   ("classes_on_device_frac", ("roofline_by_class", "classes_on_device", "frac")),
                   ("cpu_baseline_e2e_images_per_s", ("cpu_baseline_e2e", "images_per_s"))):
        v = _num(detail, *sub)
        if v is not None:
            line[k] = v
    if detail.get("e2e"):
        line["e2e_summary"] = e2e_summary(detail)
        if isinstance(detail["e2e"], dict) and detail["e2e"].get("error"):
            line["e2e_summary"]["error"] = str(detail["e2e"]["error"])[:160]
    if "bench_seconds" in detail:
        line["bench_seconds"] = detail["bench_seconds"]
    if detail_path:
        line["detail"] = detail_path
    txt = json.dumps(line)
    if len(txt) > CONTRACT_LINE_MAX:  # (cannot happen with the keys above; the cap is the contract, so enforce it anyway)
        for k in ("e2e_summary", "sustained_value", "detail", "bench_seconds"):
            line.pop(k, None)
        line["config"].pop("parallelism", None)
        line["config"]["workload"] = line["config"].get("workload", "")[:80]
        txt = json.dumps(line)
    assert len(txt) <= CONTRACT_LINE_MAX, len(txt)
    return txt


def emit(detail, detail_path):
    """Rank 0: the detail document (file + an earlier stdout line), then — LAST — the contract line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # (the collective library's banner sits in the C library's stdout buffer: out with it first)
    except OSError:
        pass
    where = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1)
            where = detail_path
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    print("bench_detail: " + json.dumps(detail), flush=True)
    print(contract_line(detail, where), flush=True)


# ---------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    t_start = time.perf_counter()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    workload = args.workload or ("1080p-420" if world == 1 else CONFIG3_WORKLOAD)
    w, h, sampling, mode, ct, default_batch = WORKLOADS[workload][:6]
    dct_scale = WORKLOADS[workload][6] if len(WORKLOADS[workload]) > 6 else 8
    images_total = args.images_total or (CONFIG3_IMAGES_TOTAL if world > 1 and not args.batch else 0)
    import jpeg_decoder_amd.distributed as D
    n_img = len(D.shard(images_total, rank, world)) if images_total else (args.batch or default_batch)
    # N > 1: launch groups exist so that the gather of one group's pixels runs behind the decode of the next; a group below ~64 images does
    # not fill the device (from the smallest shard, so that every rank makes the same number of groups: the gather pairs them up)
    n_sub = args.sub_batches or (1 if world == 1 else min(8, max(1, ((images_total // world) if images_total else n_img) // 64)))
    steps = args.steps or (500 if world == 1 else 40)
    warmup = args.warmup if args.warmup >= 0 else (50 if world == 1 else 5)
    import bench_e2e as E
    if args.dry_run:
        sys.exit(E.dry_run(args, rank, world, workload, images_total, n_img, n_sub))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    import jpeg_decoder_amd as J
    import synth

    legs = {}  # seconds per leg (detail document): what the default run spends where

    def leg(name, t0):
        legs[name] = round(legs.get(name, 0.0) + time.perf_counter() - t0, 2)

    dist = D.init(backend="nccl", force=args.force_dist) if (world > 1 or args.force_dist) else None
    t0 = time.perf_counter()
    variants = build_variants(J, synth, w, h, sampling, mode, ct, dct_scale)
    full_w, full_h = w, h
    if dct_scale != 8:  # a reduced-size decode produces ceil(W * scale / 8) x ceil(H * scale / 8) pixels (src/parser.rs:127-130)
        w, h = J.scaled_output_size(full_w, full_h, dct_scale)
    sampling, ct = variants[0]["sampling"], variants[0]["ct"]
    comps, qts, coefs = (variants[0][k] for k in ("comps", "qts", "coefs"))
    nv = len(variants)
    dev = torch.device("cuda", local_rank)
    # Does the job fit?  Arenas of this rank's shard, plus on the root the receive buffers of the final gather (every
    # peer's pixels).  A job that does not fit is SHRUNK, loudly (the line says by how much), never silently.
    shrunk_from = None
    if images_total:
        per_image = sum(algorithmic_bytes_per_image(v["comps"], w * h * (1 if v["ct"] == "Grayscale" else (4 if len(v["comps"]) == 4 else 3))) for v in variants) / nv
        free_b, _total_b = torch.cuda.mem_get_info(dev)
        budget = 0.92 * free_b

        def need(total):
            mine = len(D.shard(total, rank, world))
            recv = (total - mine) * (w * h * 3) if (rank == 0 and world > 1 and not args.no_gather) else 0
            return mine * per_image + recv
        fit = images_total
        while fit > world and need(fit) > budget:
            fit = max(world, int(fit * 0.9))
        if dist:
            fit = int(D.min_over_ranks(float(fit), device=dev))
        if fit < images_total:
            shrunk_from, images_total = images_total, fit
            n_img = len(D.shard(images_total, rank, world))
            if rank == 0:
                print(f"bench.py: {shrunk_from} images do not fit the free HBM of every rank (arenas + gather buffers): running {images_total}", file=sys.stderr)
    shard = Shard(J, torch, variants, n_img, n_sub, local_rank, generic=args.generic)
    stream = torch.cuda.current_stream(dev).cuda_stream
    leg("setup", t0)

    t0 = time.perf_counter()
    settle = (max(0, 50 - warmup) if world == 1 else 0) if args.settle < 0 else args.settle
    for _ in range(settle + warmup):
        shard.decode(stream)
    elapsed, gpu_ms_per_step = time_steps(torch, dev, dist, stream, steps, lambda: shard.decode(stream))
    if dist:
        elapsed, gpu_ms_per_step = D.max_over_ranks([elapsed, gpu_ms_per_step], device=dev)
    n_ranks_seen = None
    if dist:  # the record proves how many ranks the collective library saw (a sum of ones over all ranks)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
    sustained = None
    if args.min_seconds > 0:
        # The same step, repeated for at least --min-seconds (in chunks of the K steps timed above): an independent look at
        # `value` that lasts long enough for an outside sampler of GPU activity to see the device busy.
        s_steps, s_elapsed = 0, 0.0
        chunk = max(steps, 1)
        while s_elapsed < args.min_seconds and s_steps < 1000000:
            e, _ms = time_steps(torch, dev, dist, stream, chunk, lambda: shard.decode(stream))
            if dist:
                e, = D.max_over_ranks([e], device=dev)
            s_steps += chunk
            s_elapsed += e
            chunk = min(chunk * 2, max(steps, int(chunk * args.min_seconds / max(e, 1e-6)) + 1))
        sustained = (s_steps, s_elapsed)
    leg("headline", t0)

    # parity spot check inside the bench (oracle = checker only, never the thing measured)
    verified, digest, O = None, None, None
    if rank == 0:
        t0 = time.perf_counter()
        import oracle as O
        verified = True
        for k, v in enumerate(variants):
            ocomps, _ = O.make_components(full_w, full_h, v["sampling"], dct_scale=dct_scale)
            want = O.pixels_from_coefficients(ocomps, v["qts"], v["coefs"], w, h, v["ct"].upper())
            digest = hashlib.sha256(want.tobytes()).hexdigest()
            last = ((n_img - 1 - k) // nv) * nv + k
            for i in sorted({k, ((n_img // 2) // nv) * nv + k, last}):
                if i >= n_img:
                    continue
                got = shard.image_pixels(i).cpu().numpy()
                same = hashlib.sha256(got.tobytes()).hexdigest() == digest
                if not same:  # (say where: a line on stderr, the JSON line keeps its one boolean)
                    bad = np.nonzero(got != want)[0]
                    print(f"bench.py: image {i} differs from the oracle in {bad.size} bytes, first at {bad[:8].tolist()}: got {got[bad[:8]].tolist()} want {want[bad[:8]].tolist()}", file=sys.stderr)
                verified = verified and same
        ocomps, _ = O.make_components(full_w, full_h, sampling, dct_scale=dct_scale)
        leg("verify", t0)

    total_images = images_total or world * n_img
    mp_per_step = total_images * w * h / 1e6
    line = None
    if rank == 0:
        value = mp_per_step * steps / elapsed
        alg_bytes = sum(algorithmic_bytes_per_image(v["comps"], shard.image_pixels(k).numel()) * len(range(k, n_img, nv))
                        for k, v in enumerate(variants) if k < n_img)  # per step of this rank's shard
        achieved = alg_bytes / (gpu_ms_per_step * 1e-3) / 1e9
        traffic, traffic_source, traffic_prov = (measured_traffic(workload, shard.path) if (n_img == default_batch and world == 1)
                                                 else (None, "only measured for the default batch on one GPU", None))
        line = {
            "metric": "megapixels/s decoded (batch, whole node)", "value": round(value, 1), "unit": "MP/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "settle_launches_before_warmup": settle,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if images_total else "weak",
            "vs_baseline": None, "dtype": "i32", "dtype_note": "i32 fixed point: i16 coefficients -> u8 pixels", "data": "synthetic",
            "config": {"workload": f"{full_w}x{full_h} baseline " + (f"decoded at {dct_scale}/8 ({w}x{h}) " if dct_scale != 8 else "") +
                                   " + ".join('x'.join(str(hh) + str(vv) for hh, vv in v["sampling"]) + " " + v["ct"] for v in variants) +
                                   (" interleaved" if nv > 1 else "") +
                                   (f", {total_images} images in the job, {n_img} per GPU" if images_total else f", batch of {n_img} images per GPU") +
                                   " (coefficients resident in HBM -> RGB in HBM)",
                       "name": workload, "images_total": total_images, "images_per_gpu": n_img, "sub_batches": len(shard.batches),
                       "kernel_path": shard.path, "range_class": min(v["sane"] for v in variants),
                       "range_class_source": "host scan at staging time, outside the timed region (roofline_by_class.classes_on_device: the same launch with the classes decided on the device, inside the timed region)",
                       "parallelism": f"images sharded {n_img}/GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source, **(traffic_prov or {}),
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": round(gpu_ms_per_step, 4),
                         "launch": "one step of this rank's shard" + (f" = {len(shard.batches)} launch groups" if len(shard.batches) > 1 else "")},
            "verified_vs_oracle": verified,
        }
        if sustained:
            line["sustained"] = {"steps": sustained[0], "seconds": round(sustained[1], 3),
                                 "value": round(mp_per_step * sustained[0] / sustained[1], 1), "unit": "MP/s",
                                 "what": f"the timed step repeated until >= {args.min_seconds} s had passed (barrier + synchronize around every chunk of steps)"}
        if n_ranks_seen is not None:
            line["n_ranks_seen"] = n_ranks_seen
            line["config"]["collective_backend"] = "nccl (RCCL)" + (" — one rank, --force-dist" if world == 1 else "")
        if shrunk_from:
            line["config"]["images_total_requested"] = shrunk_from
            line["config"]["shrunk_to_fit_hbm"] = True
        line["config"]["arena_fill_copies"] = shard.fill_copies

    # ---- N > 1: the same job with the final gather inside the timed region, overlapped per launch group ----
    if dist and not args.no_gather:
        # (never lose the bench line to the collective: whatever goes wrong in here is reported in the line instead)
        t0 = time.perf_counter()
        gather_error = None
        try:
            gather = PixelGather(dist, torch, rank, world, [shard.pixel_slice(s) for s in range(len(shard.batches))], dev)

            def step_with_gather():
                for s, b in enumerate(shard.batches):
                    b.decode(stream)
                    gather.post(s)  # behind launch group s on the stream; group s + 1 decodes while it moves
                gather.wait()

            def gather_only():
                for s in range(len(shard.batches)):
                    gather.post(s)
                gather.wait()

            step_with_gather()  # warm-up (communicator set-up)
            g_elapsed, _ = time_steps(torch, dev, dist, stream, args.gather_steps, step_with_gather)
            o_elapsed, _ = time_steps(torch, dev, dist, stream, 1, gather_only)
            g_elapsed, o_elapsed = D.max_over_ranks([g_elapsed, o_elapsed], device=dev)
            if rank == 0:
                # the root holds every peer's pixels now: spot-check one image of the last rank
                if world > 1:
                    chk = gather.recv[world - 2][0][: shard.image_pixels(0).numel()].cpu().numpy()
                    line["gather_verified"] = bool(hashlib.sha256(chk.tobytes()).hexdigest() == hashlib.sha256(shard.image_pixels(0).cpu().numpy().tobytes()).hexdigest()) if nv == 1 else None
                else:
                    line["gather_verified"] = None  # one rank: the size exchange (all_gather) ran, there is nobody to receive from
                line["value_with_gather"] = round(mp_per_step * args.gather_steps / g_elapsed, 1)
                line["ms_per_step_with_gather"] = round(g_elapsed / args.gather_steps * 1e3, 3)
                line["gather_ms"] = round(o_elapsed * 1e3, 3)
                line["gather"] = {"bytes_into_root": int(sum(sum(t) for t in gather.sizes[1:])),
                                  "form": "per launch group isend/irecv peer -> rank 0 (RCCL), overlapped with the next group's decode",
                                  "steps": args.gather_steps}
            del gather
        except Exception as e:  # noqa: BLE001
            gather_error = f"{type(e).__name__}: {e}"[:300]
        if rank == 0 and gather_error:
            line["value_with_gather"] = None
            line["gather_ms"] = None
            line["gather_error"] = gather_error
        leg("gather", t0)

    # ---- N = 1 side legs (tools/bench_e2e.py); none of them is inside `value` ----
    if rank == 0 and world == 1 and not args.no_classes:
        t0 = time.perf_counter()
        if not args.generic:
            line["roofline_by_class"] = E.class_sweep(torch, shard, variants, dev, stream, alg_bytes, digest, n_img, args.class_steps)
        line["roofline"]["stream_reference"] = E.stream_reference(torch, shard, dev, stream, line["roofline"]["achieved"])
        leg("classes", t0)

    def drop_shard():
        nonlocal shard
        if shard is not None:
            shard.close()
            shard = None
        torch.cuda.empty_cache()

    def guarded(key, fn, into=None):
        """Never lose the line to a side leg: its error goes where its result would have gone."""
        t0 = time.perf_counter()
        into = line if into is None else into
        try:
            into[key] = fn()
        except Exception as e:  # noqa: BLE001
            into[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        leg(key, t0)
        torch.cuda.empty_cache()

    default_run = world == 1 and workload == "1080p-420" and not args.generic
    if rank == 0 and default_run and not args.no_k4096:
        drop_shard()  # north_star's literal batch: 4096 images on one GPU (51 GB of arenas), same kernels, same verification
        guarded("k_4096", lambda: E.k_4096(J, torch, O, variants, local_rank, dev, stream, w, h, digest))
    if rank == 0 and default_run and not args.no_scale_anchor and not args.force_dist:
        drop_shard()
        guarded("scale_anchor", lambda: E.scale_anchor(J, torch, O, synth, D, local_rank, dev, stream))
    if dist and not args.no_e2e and not args.generic:
        # ---- N > 1 (and --force-dist): E per rank on its shard of the file list, with a per-rank host budget ----
        drop_shard()
        t0 = time.perf_counter()
        e2e_error, sharded = None, None
        try:
            E.note_unpinned_cpus()
            share, threads = E.rank_cpu_share(rank, world, E.device_bdfs(J, world))
            pinned = E.pin_to(share)
            import oracle as O_all  # (every rank checks two of its own images: the oracle as checker)
            w1, h1 = WORKLOADS["1080p-420"][0], WORKLOADS["1080p-420"][1]
            sharded = E.e2e_sharded(J, O_all, synth, D, dist, torch, dev, rank, local_rank, world, w1, h1, args.e2e_total, args.e2e_encoder, share, threads, pinned)
        except Exception as e:  # noqa: BLE001 (e2e_sharded raises only after its last collective, so the ranks stay in step)
            e2e_error = f"{type(e).__name__}: {e}"[:300]
        if rank == 0:
            line.setdefault("e2e", {})["sharded"] = sharded if sharded else {"error": e2e_error}
        leg("e2e_sharded", t0)
    if rank == 0 and default_run and not args.no_e2e:
        drop_shard()
        t0 = time.perf_counter()
        try:
            e2e, files = E.e2e_block(J, O, synth, w, h, [int(x) for x in args.e2e_images.split(",") if x], args.e2e_encoder,
                                     E.h2d_rate_gbps(torch, dev), E.d2h_rate_gbps(torch, dev))
            line.setdefault("e2e", {}).update(e2e)
            leg("e2e_block", t0)
            if not args.no_cpu_budget:
                guarded("cpu_budget", lambda: E.e2e_cpu_budget(J, O, synth, w, h, args.e2e_encoder, matrix=args.cpu_budget_matrix), into=line["e2e"])
            if not args.no_cpu_baseline:
                guarded("cpu_baseline_e2e", lambda: E.cpu_baseline_e2e(O, files, w, h, args.cpu_seconds / 2))
        except Exception as e:  # noqa: BLE001
            line.setdefault("e2e", {})["error"] = f"{type(e).__name__}: {e}"[:300]
    if rank == 0 and not args.no_cpu_baseline and world == 1 and nv == 1:
        t0 = time.perf_counter()
        line["cpu_baseline"] = cpu_baseline(O, ocomps, qts, coefs, w, h, ct, args.cpu_seconds)
        leg("cpu_baseline", t0)
    if shard is not None:
        shard.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line["leg_seconds"] = legs
        line["bench_seconds"] = round(time.perf_counter() - t_start, 1)
        emit(line, args.detail)


from bench_e2e import e2e_files, rank_cpu_share  # noqa: E402,F401 (tests reach them through this module)

if __name__ == "__main__":
    main()
