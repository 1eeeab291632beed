#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

  metric : megapixels/s decoded (batch, whole node); % of HBM roofline for the pixel kernels
  N=1 workload (configs[1]): 1920x1080 baseline 4:2:0 YCbCr, batch of 256 images, 1 MI355X.
  A "step" = one pass of the hot path (dequantize + IDCT + upsample + YCbCr->RGB) over the whole
  batch, coefficients already resident in HBM, RGB left resident in HBM.
  N>1: one process per GPU (torch.distributed / RCCL), the batch shards one-image-per-task with
  NO data-path collective (weak scaling: 256 images per GPU); the final RCCL gather of the
  pixels to rank 0 that north_star mentions runs after the timed region and is reported apart.

Prints ONE JSON line on rank 0."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

WORKLOADS = {
    # name: (width, height, sampling, mode, colour transform, default batch)
    "1080p-420": (1920, 1080, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "2160p-420": (3840, 2160, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 64),
    "1080p-444": (1920, 1080, [(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "1080p-422": (1920, 1080, [(2, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "1080p-gray": (1920, 1080, [(1, 1)], "gray", "Grayscale", 256),
    # 4:4:4 and grayscale images interleaved in one batch (two fused launch groups, path "mixed")
    "1080p-444+gray": (1920, 1080, None, "mixed", None, 512),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="1080p-420", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the baseline sample")
    ap.add_argument("--generic", action="store_true", help="force the two-kernel generic path")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed launches BEFORE the warm-up steps that let the GPU leave its clock transient after an idle "
                         "period (DESIGN.md §5: the first ~25 launches run up to 50 %% slower); default: enough to make "
                         "settle + warm-up = 50 launches, i.e. none for the default warm-up; 0 switches it off")
    return ap.parse_args()


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


def cpu_baseline(O, ocomps, qts, coefs, w, h, ct, target_seconds):
    """Oracle ("port" of the reference's scalar path) on the CPUs the process may use, bounded sample."""
    cores = effective_cpus()
    n0 = max(cores, 4)
    t0 = time.perf_counter()
    O.batch_pixels(ocomps, qts, [coefs] * n0, w, h, ct.upper(), cores, keep_outputs=False)
    dt = time.perf_counter() - t0
    n = int(max(n0, min(65536, n0 * target_seconds / max(dt, 1e-3))))
    n = (n // cores) * cores or cores
    t0 = time.perf_counter()
    O.batch_pixels(ocomps, qts, [coefs] * n, w, h, ct.upper(), cores, keep_outputs=False)
    dt = time.perf_counter() - t0
    return {"value": round(n * w * h / 1e6 / dt, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"{n} images {w}x{h} of the same workload, pixel pipeline only (coefficients -> pixels), "
                      f"{cores} threads one image per task, {dt:.1f} s"}


def measured_traffic(workload, path):
    """HBM bytes per decode from the committed rocprofv3 PMC passes (profiles/round1/pmc_traffic.json,
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate --pmc runs); None if not measured."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "round1", "pmc_traffic.json")))
        e = t.get(f"{workload}:{path}")
        return e["hbm_bytes_per_decode"] if e else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    import jpeg_decoder_amd as J
    import jpeg_decoder_amd.distributed as D
    import synth

    dist = D.init(backend="nccl") if world > 1 else None

    w, h, sampling, mode, ct, default_batch = WORKLOADS[args.workload]
    n_img = args.batch or default_batch
    lum, chr_ = synth.quality_tables(85)
    rgb = synth.synthetic_rgb(w, h)
    # image i of the batch is variant i % len(variants); one variant except for the interleaved workload
    specs = [([(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr"), ([(1, 1)], "gray", "Grayscale")] if mode == "mixed" else [(sampling, mode, ct)]
    variants = []
    for v_sampling, v_mode, v_ct in specs:
        comps, _mcu = J.make_components(w, h, v_sampling)
        qts = [lum, chr_, chr_][: len(v_sampling)]
        coefs = synth.coefficients_from_rgb(rgb, comps, v_mode, qts)
        # range class of the dequantized coefficients (what jpgpu_batch_upload computes when it stages data itself)
        prod = [np.abs(c.astype(np.int64).reshape(-1, 8, 8) * q.astype(np.int64).reshape(8, 8)) for c, q in zip(coefs, qts)]
        sane = 0
        if all((p < (1 << 15)).all() for p in prod):
            sane = 3 if all((p.sum(axis=1) <= 5900).all() for p in prod) else 1
        variants.append({"sampling": v_sampling, "ct": v_ct, "comps": comps, "qts": qts, "coefs": coefs, "sane": sane,
                         "desc": J.image_desc(list(comps), qts, w, h, v_ct)})
    sampling, ct = variants[0]["sampling"], variants[0]["ct"]
    comps, qts, coefs, sane = (variants[0][k] for k in ("comps", "qts", "coefs", "sane"))
    nv = len(variants)

    flags = J._native.BATCH_EXTERNAL_BUFFERS | (J._native.BATCH_FORCE_GENERIC if args.generic else 0)
    batch = J.Batch([variants[i % nv]["desc"] for i in range(n_img)], device=local_rank, flags=flags)
    dev = torch.device("cuda", local_rank)
    coef_arena = torch.zeros(batch.coef_arena_bytes(), dtype=torch.uint8, device=dev)
    out_arena = torch.zeros(batch.out_arena_bytes(), dtype=torch.uint8, device=dev)
    # N distinct coefficient buffers in HBM (no aliasing): upload each variant once, replicate on device
    for k, v in enumerate(variants):
        for c in range(len(v["comps"])):
            src = torch.from_numpy(v["coefs"][c].view(np.uint8)).to(dev)
            for i in range(k, n_img, nv):
                off = batch.coef_offset(i, c)
                coef_arena[off: off + src.numel()] = src
    batch.bind(coef_arena.data_ptr(), out_arena.data_ptr())
    for i in range(n_img):
        batch.set_range_hint(i, variants[i % nv]["sane"])
    stream = torch.cuda.current_stream(dev).cuda_stream

    settle = max(0, 50 - max(args.warmup, 0)) if args.settle < 0 else args.settle
    for _ in range(settle + max(args.warmup, 0)):
        batch.decode(stream)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream the kernels are launched on
    for _ in range(args.steps):
        batch.decode(stream)
    ev1.record()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    gpu_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    if dist:
        elapsed, gpu_ms_per_step = D.max_over_ranks([elapsed, gpu_ms_per_step], device=dev)

    # parity spot check inside the bench (oracle = checker only, never the thing measured)
    verified = None
    gather_ms = None
    out_bytes = batch.out_bytes(0)
    if rank == 0:
        import oracle as O
        verified = True
        for k, v in enumerate(variants):
            ocomps, _ = O.make_components(w, h, v["sampling"])
            want = O.pixels_from_coefficients(ocomps, v["qts"], v["coefs"], w, h, v["ct"].upper())
            digest = hashlib.sha256(want.tobytes()).hexdigest()
            last = ((n_img - 1 - k) // nv) * nv + k
            for i in sorted({k, ((n_img // 2) // nv) * nv + k, last}):
                if i >= n_img:
                    continue
                off = batch.out_offset(i)
                got = out_arena[off: off + batch.out_bytes(i)].cpu().numpy()
                verified = verified and hashlib.sha256(got.tobytes()).hexdigest() == digest
        ocomps, _ = O.make_components(w, h, sampling)
    if dist and not args.no_gather:
        try:  # north_star's "RCCL over xGMI only for the final gather", outside the timed region
            pix = out_arena[: batch.out_offset(n_img - 1) + batch.out_bytes(n_img - 1)]
            torch.cuda.synchronize(dev)
            dist.barrier()
            g0 = time.perf_counter()
            gl = D.gather_pixels(pix, dst=0)
            torch.cuda.synchronize(dev)
            dist.barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            del gl
        except Exception as e:  # the gather is informational; never lose the bench line to it
            gather_ms = None
            if rank == 0:
                print(f"# gather skipped: {e}", file=sys.stderr)

    if rank == 0:
        mp_per_step = world * n_img * w * h / 1e6
        value = mp_per_step * args.steps / elapsed
        alg_bytes = sum(algorithmic_bytes_per_image(variants[i % nv]["comps"], batch.out_bytes(i)) for i in range(n_img))  # per launch (one GPU's batch)
        achieved = alg_bytes / (gpu_ms_per_step * 1e-3) / 1e9
        line = {
            "metric": "megapixels/s decoded (batch, whole node)", "value": round(value, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_launches_before_warmup": settle,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32 fixed-point (i16 coefficients -> u8 pixels)", "data": "synthetic",
            "config": {"workload": f"{w}x{h} baseline " + " + ".join('x'.join(str(hh) + str(vv) for hh, vv in v["sampling"]) + " " + v["ct"]
                                                                     for v in variants) +
                                   (" interleaved" if nv > 1 else "") +
                                   f", batch of {n_img} images per GPU (coefficients resident in HBM -> RGB in HBM)",
                       "name": args.workload, "images_per_gpu": n_img, "kernel_path": batch.path, "range_class": min(v["sane"] for v in variants),
                       "parallelism": f"images sharded {n_img}/GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": measured_traffic(args.workload, batch.path) if n_img == default_batch else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": round(gpu_ms_per_step, 4)},
            "verified_vs_oracle": verified,
        }
        if gather_ms is not None:
            line["gather_ms_after_timed_region"] = round(gather_ms, 2)
        if not args.no_cpu_baseline and world == 1 and nv == 1:
            line["cpu_baseline"] = cpu_baseline(O, ocomps, qts, coefs, w, h, ct, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    batch.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
