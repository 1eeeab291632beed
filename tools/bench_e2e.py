#!/usr/bin/env python3
"""bench.py's side legs (the headline driver is bench.py; everything it reports OUTSIDE `value` lives here).

  k_4096 / scale_anchor   the kernel-only figure on north_star's 4,096-image batch and on the N > 1 job (2160p x 4,096) on one GPU
  class_sweep             the headline launch per arithmetic class, and with the classes decided on the device
  e2e_block               JPEG bytes in host memory -> RGB in HBM / in host memory through jpgpu_pipeline_decode (entropy decoding on
                          the device), restart markers, BASELINE configs[3] (progressive frames), every entry checked against the oracle
  e2e_cpu_budget          the 4,096-file call against 2 / 8 / 16 host CPUs (the whole 5 x 4 matrix behind --cpu-budget-matrix)
  cpu_baseline_e2e        the oracle's whole Decoder::decode() on the same files
  e2e_sharded             E per rank at N > 1
  dry_run                 the N > 1 bookkeeping on CPU tensors over gloo (tests/test_bench_launcher.py)

Every function writes into the DETAIL document (gpurun_out/bench_detail.json + a `bench_detail:` line on stdout); bench.py's LAST
stdout line is the small contract line, which takes ten scalars from here (`summary`).  Run alone: `python tools/bench_e2e.py
[--legs e2e,progressive,cpu_budget,...]` prints the detail document of those legs only."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import bench as B  # noqa: E402 (the headline driver: Shard, time_steps, WORKLOADS ...; bench.py aliases itself as `bench` when run as a script)

def rank_cpu_share(rank, world, bdfs=None, sysfs="/sys"):
    """Host CPUs rank `rank` of `world` keeps its feeder threads on, and the thread budget that goes with it.  The CPUs: those of the
    NUMA node the rank's GPU hangs off (its PCI bus id -> <sysfs>/bus/pci/devices/<id>/numa_node), split among the ranks whose GPUs
    share the node; contiguous slices of the allowed list when the topology is unknown (jpeg_decoder_amd.distributed.cpu_shares, the
    rule jpgpu_pipeline_create_multi applies in C).  The threads: the CPUs the cgroup GRANTS, divided by the ranks (one pipeline keeps
    ~8 CPUs busy; `world` ranks that each started the default of one thread per physical core would oversubscribe the host
    `world`-fold: VERDICT r3).  -> (cpu list, threads for jpgpu_pipeline_create)."""
    import jpeg_decoder_amd.distributed as D
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    shares, _nodes = D.cpu_shares(allowed, bdfs if bdfs else [""] * world, sysfs)
    share = shares[rank] or allowed
    # (the library's own default for ONE pipeline is twice the granted CPUs — its threads wait for the device and the link a good
    # part of a call; 16 threads on 16 granted CPUs measured 71.9 ms per 4,096 files against 53.3 with 32)
    return share, max(2, 2 * B.effective_cpus() // world)


def device_bdfs(J, world):
    """PCI bus ids of devices 0 .. world-1 as this process sees them (rank r drives device r); [] if they cannot be read."""
    import ctypes
    out = []
    for k in range(world):
        buf = ctypes.create_string_buffer(64)
        if J._native.lib().jpgpu_device_pci_bus_id(k, buf, 64) != 0:
            return []
        out.append(buf.value.decode())
    return out


def pin_to(share):
    try:
        os.sched_setaffinity(0, share)
        return True
    except (AttributeError, OSError):
        return False


def h2d_rate_gbps(torch, dev, nbytes=1 << 30):
    """What the host link gives ONE pinned copy of 1 GB, in this run (the floor of an E call is its entropy-coded bytes at this rate)."""
    src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del src, dst
    return nbytes / (best * 1e-3) / 1e9


def huffman_symbols(J, data):
    """Huffman symbols of a baseline file's scan(s), from the coefficients the host front-end decodes: per block one DC symbol,
    one symbol per non-zero AC coefficient, a ZRL per 16 zeros inside a run, an EOB unless the last coefficient is non-zero."""
    _desc, planes = J.Decoder(data, device=-1).decode_coefficients()
    zz = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
                   57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
    total = 0
    for pl in planes:
        b = np.asarray(pl, np.int16).reshape(-1, 64)[:, zz]  # zig-zag order
        nz = b[:, 1:] != 0
        total += b.shape[0] + int(nz.sum())                   # DC symbols + coefficient symbols
        last = np.where(nz.any(axis=1), 62 - np.argmax(nz[:, ::-1], axis=1), -1)  # index (0..62) of the last non-zero AC coefficient
        total += int((last < 62).sum())                        # EOB
        # ZRL: one per 16 zeros in front of a non-zero coefficient
        idx = np.arange(63)
        pos_idx = np.where(nz, idx[None, :], -1)
        prev = np.maximum.accumulate(np.concatenate([np.full((b.shape[0], 1), -1), pos_idx[:, :-1]], axis=1), axis=1)
        total += int((np.where(nz, idx[None, :] - prev - 1, 0) // 16).sum())
    return total


def k_4096(J, torch, O, variants, device_index, dev, stream, w, h, digest, n_img=4096, steps=30):
    """The kernel-only figure (coefficients resident in HBM -> RGB in HBM) on a 4096-image batch of the default workload."""
    sh = B.Shard(J, torch, variants, n_img, 1, device_index)
    try:
        for _ in range(5):
            sh.decode(stream)
        elapsed, ms = B.time_steps(torch, dev, None, stream, steps, lambda: sh.decode(stream))
        ok = all(hashlib.sha256(sh.image_pixels(i).cpu().numpy().tobytes()).hexdigest() == digest for i in (0, n_img // 2 + 1, n_img - 1))
        alg = B.algorithmic_bytes_per_image(variants[0]["comps"], sh.image_pixels(0).numel()) * n_img
        return {"images": n_img, "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4), "kernel_ms_per_launch": round(ms, 4),
                "value": round(n_img * w * h / 1e6 * steps / elapsed, 1), "unit": "MP/s",
                "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": alg,
                "arena_bytes": int(sh.coef_arena.numel() + sh.out_arena.numel()), "kernel_path": sh.path, "arena_fill_copies": sh.fill_copies,
                "verified_vs_oracle": bool(ok),
                "what": f"{n_img} x {w}x{h} 4:2:0 in ONE launch on one GPU, coefficients resident in HBM -> RGB in HBM (north_star's batch)"}
    finally:
        sh.close()


def scale_anchor(J, torch, O, synth, D, device_index, dev, stream, steps=10):
    """The N > 1 job on ONE GPU (VERDICT r4 #2c): configs[2] — 3840x2160 4:2:0 x 4,096, coefficients resident in HBM -> RGB in HBM, the
    same launch groups the ranks of an N-GPU run make (8 groups) — so that the 1 -> 8 curve has a first point on its own workload
    (`value` at N = 1 is configs[1]: 1080p x 256).  204 GB of arenas; shrunk, loudly, if this GPU has less free."""
    w, h, sampling, mode, ct = B.WORKLOADS[B.CONFIG3_WORKLOAD][:5]
    variants = B.build_variants(J, synth, w, h, sampling, mode, ct)
    per_image = B.algorithmic_bytes_per_image(variants[0]["comps"], w * h * 3)
    want_n = B.CONFIG3_IMAGES_TOTAL
    free_b, _t = torch.cuda.mem_get_info(dev)
    n_img = want_n
    while n_img > 64 and n_img * per_image > 0.92 * free_b:
        n_img = int(n_img * 0.9)
    n_sub = min(8, max(1, n_img // 64))
    sh = B.Shard(J, torch, variants, n_img, n_sub, device_index)
    try:
        for _ in range(3):
            sh.decode(stream)
        elapsed, ms = B.time_steps(torch, dev, None, stream, steps, lambda: sh.decode(stream))
        ocomps, _ = O.make_components(w, h, sampling)
        digest = hashlib.sha256(O.pixels_from_coefficients(ocomps, variants[0]["qts"], variants[0]["coefs"], w, h, ct.upper()).tobytes()).hexdigest()
        ok = all(hashlib.sha256(sh.image_pixels(i).cpu().numpy().tobytes()).hexdigest() == digest for i in (0, n_img // 2 + 1, n_img - 1))
        alg = per_image * n_img
        return {"workload": f"{w}x{h} 4:2:0 x {n_img} on one GPU (BASELINE configs[2], the job bench.py --gpus N shards)", "name": B.CONFIG3_WORKLOAD,
                "images": n_img, "images_requested": want_n, "shrunk_to_fit_hbm": n_img < want_n, "sub_batches": n_sub, "steps": steps,
                "ms_per_step": round(elapsed / steps * 1e3, 3), "kernel_ms_per_step": round(ms, 3),
                "value": round(n_img * w * h / 1e6 * steps / elapsed, 1), "unit": "MP/s",
                "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBPS, 4), "arena_bytes": int(sh.coef_arena.numel() + sh.out_arena.numel()),
                "kernel_path": sh.path, "verified_vs_oracle": bool(ok),
                "what": "the N > 1 job's kernel-only figure at N = 1: divide the N-GPU `value` by this for the scaling efficiency on one workload"}
    finally:
        sh.close()


def class_sweep(torch, shard, variants, dev, stream, alg_bytes, digest, n_img, steps):
    """N = 1: what the arithmetic class is worth (the same launch with every image's class capped at 0 / 1 / 3), and what
    deciding the classes ON THE DEVICE costs: statistics resident on the device (the library's own writers — device entropy decoder,
    compact expansion — raise them as a by-product; here one jpgpu_batch_classify_on_device pass outside the timed region stands in
    for them), a finalize kernel + ONE `_dyn` pixel launch that branches per workgroup inside every timed step; nothing is read
    back, nothing synchronises.  (Round 2's `with_device_range_scan` — a host synchronisation per step — is gone: nothing uses it.)"""
    nv = len(variants)
    by_class = {}
    top = min(v["sane"] for v in variants)
    for cap in (0, 1, 3):
        if cap > top:
            continue
        shard.set_classes(cap)
        for _ in range(10):
            shard.decode(stream)
        _, ms = B.time_steps(torch, dev, None, stream, steps, lambda: shard.decode(stream))
        by_class[f"class{cap}"] = {"kernel_ms_per_launch": round(ms, 4), "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBPS, 4)}
    shard.set_classes(None)
    for b in shard.batches:
        b.classify_on_device(stream)
    for _ in range(10):
        shard.decode(stream)
    _, ms = B.time_steps(torch, dev, None, stream, steps, lambda: shard.decode(stream))
    counts = [sum(x) for x in zip(*[b.class_counts() for b in shard.batches])]
    got_dyn = shard.image_pixels(n_img - 1).cpu().numpy()
    by_class["classes_on_device"] = {
        "kernel_ms_per_launch": round(ms, 4), "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBPS, 4),
        "images_per_class_0_1_3": counts,
        "verified_vs_oracle": bool(hashlib.sha256(got_dyn.tobytes()).hexdigest() == digest) if nv == 1 else None,
        "note": "class statistics resident on the device, class_finalize kernel + the _dyn pixel kernel inside every timed step; no "
                "range scan, no read-back, no host synchronisation"}
    shard.set_classes(None)
    by_class["note"] = ("class 3 = every |c*q| < 2^15 and every block column sum <= 5900 (legal 8-bit JPEG data), class 1 = the first only, "
                        "class 0 = arbitrary i16 coefficients (wrap-exact kernels)")
    return by_class


def stream_reference(torch, shard, dev, stream, achieved_gbps):
    """What the memory system gives an address-ordered stream of the same size and read/write mix, on this box, in this run: an
    elementwise add from the coefficient arena into the pixel arena (the pixels are verified by now).  Context for roofline.frac,
    which is quoted against the 8 TB/s peak."""
    n4 = min(shard.coef_arena.numel(), shard.out_arena.numel()) // 4 * 4
    src, dst = shard.coef_arena[:n4].view(torch.int32), shard.out_arena[:n4].view(torch.int32)
    for _ in range(3):
        torch.add(src, 1, out=dst)
    _, ms = B.time_steps(torch, dev, None, stream, 20, lambda: torch.add(src, 1, out=dst))
    ref = 2.0 * n4 / (ms * 1e-3) / 1e9
    return {"achieved": round(ref, 1), "unit": "GB/s", "ms": round(ms, 4), "frac_of_peak": round(ref / B.HBM_PEAK_GBPS, 4),
            "kernel_vs_reference": round(achieved_gbps / ref, 4),
            "what": "torch.add over this workload's own arenas (as many bytes read as written, address order), same box, same run"}


def e2e_files(synth, w, h, encoder, distinct=4, restart_rows=0):
    """`distinct` baseline 4:2:0 q85 files of the bench's synthetic image (seeds 0x5EED + k); -> (files, who wrote them).
    restart_rows: a restart marker every that many MCU rows (DRI)."""
    rgbs = [synth.synthetic_rgb(w, h, seed=0x5EED + k) for k in range(distinct)]
    rst = f"a restart marker every {restart_rows} MCU row(s)" if restart_rows else "no restart markers"
    if encoder in ("auto", "pillow"):
        try:
            import io
            import PIL
            from PIL import Image
            out = []
            for rgb in rgbs:
                buf = io.BytesIO()
                Image.fromarray(rgb).save(buf, format="JPEG", quality=85, subsampling="4:2:0",
                                          **({"restart_marker_rows": restart_rows} if restart_rows else {}))
                out.append(buf.getvalue())
            if not restart_rows or all(b"\xff\xdd\x00\x04" in d[:1024] for d in out):  # (an older Pillow ignores the keyword)
                return out, f"Pillow {PIL.__version__} (libjpeg-turbo), quality 85, 4:2:0, default (Annex K) Huffman tables, {rst}"
        except ImportError:
            if encoder == "pillow":
                raise
    sys.path.insert(0, os.path.join(B.ROOT, "tools"))
    import baseline_encoder as E
    ri = restart_rows * ((w + 15) // 16)
    return [E.encode_rgb(rgb, 85, "420", ri) for rgb in rgbs], f"tools/baseline_encoder.py (this repo), quality 85, 4:2:0, Annex K Huffman tables, {rst}"


VALU_CYCLES = 4.6  # issue cycles per wave64 vector instruction of the decoder's kind (measured: see sync_pass_instructions)


def sync_pass_instructions(symbols_per_call):
    """Vector instructions of the chunk decoder's sync passes per Huffman symbol, from the committed counter passes of a 256-file
    call as one sub-batch (profiles/roundN/*pipeline_256*pmc*.json: SQ_INSTS_VALU per dispatch x dispatches, all sync launches of
    the call): wave-instructions per symbol, and x 64 = lane slots per symbol (a scalar decoder's step is ~100 instructions)."""
    import glob
    for rnd in ("round6", "round5", "round4", "round3"):
        for f in sorted(glob.glob(os.path.join(B.ROOT, "profiles", rnd, "*pipeline_256*pmc*.json")) + glob.glob(os.path.join(B.ROOT, "profiles", rnd, "*pipe256*stats*.json"))):
            try:
                doc = json.load(open(f))
            except (OSError, ValueError):
                continue
            wave_instr = every = 0.0
            for name, e in doc.items():
                if not isinstance(e, dict) or "pmc" not in e or "SQ_INSTS_VALU" not in e["pmc"]:
                    continue
                every += e["pmc"]["SQ_INSTS_VALU"] * e["calls"]
                if "huff_sync_pass_kernel" in name or "huff_sync_late_kernel" in name:
                    wave_instr += e["pmc"]["SQ_INSTS_VALU"] * e["calls"]
            calls = doc.get("_calls_of_the_pipeline") or 6  # (tools/pipe_calls.py / round 3's script: six calls per profiled process)
            if wave_instr:
                per_call = wave_instr / calls
                # What the vector units alone need for a call's kernels: every wave-instruction occupies its SIMD for VALU_CYCLES cycles
                # (profiles/round4/02_ubench_valu64.txt, round2/00_valu_issue_cost_ubench.txt: 4.5-4.9 for the shifts, selects and
                # logic the decoder is made of), 1,024 SIMDs at 2.4 GHz — a floor no overlap of sub-batches gets under.
                issue_ms_per_image = every / calls / 256.0 * VALU_CYCLES / (1024 * 2.4e9) * 1e3
                return {"wave_instructions_per_symbol": round(per_call / symbols_per_call, 2), "lane_slots_per_symbol": round(64 * per_call / symbols_per_call, 1),
                        "all_kernels_wave_instructions_per_image": int(every / calls / 256.0), "vector_issue_floor_ms_per_image": round(issue_ms_per_image, 5),
                        "vector_issue_floor_what": f"SQ_INSTS_VALU of every kernel of the call x {VALU_CYCLES} cycles / (1,024 SIMDs x 2.4 GHz)",
                        "source": os.path.relpath(f, B.ROOT) + " (SQ_INSTS_VALU per dispatch of one 256-file call as one sub-batch; read from the file, not measured in this run)"}
    return None


def e2e_floor_fields(e, best, h2d_gbps, alone_ms_per_image):
    """An E entry on its own roofline: the link floor (entropy-coded bytes that cross PCIe at the H2D rate one pinned 1-GB copy
    reached in this run), the device-work floor (the kernels' time per image with the device to itself — sync passes, expansion,
    pixel kernels of one sub-batch of 256 alone — times the images), and how close the call's wall clock is to the larger one."""
    link = best["coefficient_bytes"] / (h2d_gbps * 1e9) * 1e3 if h2d_gbps else None
    work = alone_ms_per_image * e["images"] if alone_ms_per_image else None
    floors = [x for x in (link, work) if x]
    e["pcie_bytes"] = int(best["coefficient_bytes"])
    e["link_floor_ms"] = round(link, 3) if link else None
    e["device_work_ms"] = round(work, 3) if work else None
    e["frac_of_floor"] = round(max(floors) / e["total_ms"], 4) if floors else None
    e["bound"] = None if not floors else ("link" if link and link >= (work or 0) else "device work")


def d2h_rate_gbps(torch, dev, nbytes=1 << 30):
    """The other direction of the link: one pinned 1-GB device-to-host copy, best of 3, this run (the floor of an E call that hands its
    pixels to a host consumer — Decoder::decode()'s Vec<u8>, src/decoder.rs:293-295 — is its pixel bytes at this rate)."""
    dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del src, dst
    return nbytes / (best * 1e-3) / 1e9


def host_memory_available():
    """Bytes of host memory this process may still take: MemAvailable, capped by what a cgroup limit leaves (None if unknown)."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
                break
    except (OSError, ValueError):
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            left = int(mx) - int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail


def warm_calls(p, files, calls, cold=1, **kw):
    """`cold` uncounted calls (the first one allocates arenas and staging), then `calls` counted ones -> their timings, in call order.
    Raises the first per-image error."""
    out = []
    for r in range(cold + calls):
        res = p.decode(files, **kw)
        bad = [x for x in res if isinstance(x, Exception)]
        if bad:
            raise bad[0]
        if r >= cold:
            out.append(p.timings())
    return out


def call_stats(ts, key="total_ms"):
    """median + min of the counted calls.  -> (the timings of the median call, median ms, min ms)"""
    order = sorted(ts, key=lambda t: t[key])
    med = order[(len(order) - 1) // 2]  # (lower median: a call that really happened, whose other fields go with it)
    return med, med[key], order[0][key]


def e2e_entry(n, ts, w, h, p, ok, extra=None):
    med, med_ms, min_ms = call_stats(ts)
    e = {"images": n, "calls": len(ts), "total_ms": round(med_ms, 3), "min_ms": round(min_ms, 3),
         "images_per_s": round(n / med_ms * 1e3, 1), "images_per_s_best": round(n / min_ms * 1e3, 1),
         "value": round(n * w * h / 1e6 / med_ms * 1e3, 1), "unit": "MP/s",
         "wall_ms": {k[:-3]: round(med[k], 3) for k in ("headers_ms", "setup_ms", "entropy_and_upload_ms", "download_ms")},
         "cpu_ms_per_image": round(med["cpu_ms"] / max(n, 1), 5),
         "images_device_entropy": int(med["images_device_entropy"]), "images_device_rejected": int(med["images_device_rejected"]),
         "images_host_light": int(med["images_host_light"]), "images_entry_pixels": int(med.get("images_entry_pixels", 0)),
         "threads": int(med["threads"]), "kernel_path": p.kernel_path, "verified_vs_oracle": bool(ok)}
    if extra:
        e.update(extra)
    return e, med


def e2e_block(J, O, synth, w, h, sizes, encoder, h2d_gbps=None, d2h_gbps=None):
    """What the metric's words say: JPEG bytes in host memory -> RGB in HBM, through jpgpu_pipeline_decode with the entropy
    decoding on the device (no host Huffman decoding, no range scan, no host synchronisation in front of the pixel kernels).
    Per batch size: median and min of 5 warm calls (wall clock of the call, everything included: header parsing, staging, H2D, kernels)
    and a check of first / middle / last image against the oracle; the kernel time per phase from one sub-batch run alone."""
    os.environ["JPGPU_BATCH_KERNEL_TIMES"] = "1"  # (read once by the library: events around the phases, microseconds per sub-batch)
    distinct, who = e2e_files(synth, w, h, encoder)
    want = [hashlib.sha256(O.decode(d).pixels.tobytes()).hexdigest() for d in distinct]
    out = {"input": f"{len(distinct)} distinct {w}x{h} files, repeated; written by {who}",
           "jpeg_bytes_per_image": int(sum(len(d) for d in distinct) / len(distinct)),
           "what": "jpgpu_pipeline_decode: JPEG bytes in host memory -> RGB resident in HBM; entropy decoding ON THE DEVICE "
                   "(self-synchronising chunk decoder with speculative emission), classes from its statistics, pixel kernels right behind it; "
                   "per entry 5 warm calls after 2 uncounted ones: total_ms / images_per_s / value = their MEDIAN, min_ms / images_per_s_best = the "
                   "fastest (SURVEY 8d); wall clock of the whole call; cpu_ms_per_image = process CPU time of the median call / images"}
    p = J.Pipeline()
    bests = {}
    try:
        for n in sizes:
            files = [distinct[i % len(distinct)] for i in range(n)]
            # (the first call allocates arenas and staging: not counted — and calls 2-3 still run ~20 % slower than the steady state)
            ts = warm_calls(p, files, 5, cold=2, download=False, device_entropy=True)
            ok = all(hashlib.sha256(p.download(i).tobytes()).hexdigest() == want[i % len(distinct)] for i in sorted({0, 1, n // 2, n - 1}))
            out[str(n)], bests[str(n)] = e2e_entry(n, ts, w, h, p, ok)
        # The same calls with the files in PINNED host memory (JPGPU_PIPELINE_INPUT_PINNED, PinnedFiles: what a loader that reads into
        # jpgpu_host_alloc memory hands over): the copy engine reads the scans where they lie — no staging copy on the host at all, and
        # the sub-batches' launches follow one another in tenths of a millisecond instead of 0.8 ms each (the host's memcpy of 25 MB)
        for n in [x for x in (256, 4096) if x in sizes]:
            key = f"{n}_pinned_input"
            try:
                arena = J.PinnedFiles([distinct[i % len(distinct)] for i in range(n)])
                try:
                    ts = warm_calls(p, arena, 5, cold=1, download=False, device_entropy=True, input_pinned=True)
                    ok = all(hashlib.sha256(p.download(i).tobytes()).hexdigest() == want[i % len(distinct)] for i in sorted({0, 1, n // 2, n - 1}))
                    out[key], _ = e2e_entry(n, ts, w, h, p, ok, {"input": "the same files in one pinned arena (jpgpu_host_alloc), JPGPU_PIPELINE_INPUT_PINNED"})
                finally:
                    arena.close()
            except Exception as e:  # noqa: BLE001 (this entry only)
                out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # The kernels alone: the same 256 files as ONE sub-batch with the device to itself (the pipeline's default splits a call into
        # sub-batches of 128 that run side by side on their own streams: their phase times overlap and do not add up to anything).
        os.environ["JPGPU_PIPE_DEV_SUB"], os.environ["JPGPU_PIPE_MAX_DEV_SUBS"] = "256", "1"
        try:
            files = [distinct[i % len(distinct)] for i in range(256)]
            best = None
            for r in range(4):
                res = p.decode(files, download=False, device_entropy=True)
                t = p.timings()
                if r > 0 and t["dev_times_valid"] and (best is None or t["dev_sync_ms"] + t["dev_write_ms"] + t["dev_pixel_ms"] < best["dev_sync_ms"] + best["dev_write_ms"] + best["dev_pixel_ms"]):
                    best = t
            if best is not None:
                km = {"sync_ms": best["dev_sync_ms"], "expand_ms": best["dev_write_ms"], "pixel_ms": best["dev_pixel_ms"]}
                out["kernels_256_one_sub_batch"] = {
                    "what": "256 files as one sub-batch, nothing else on the device: events between the phases on its stream — sync passes "
                            "(with speculative emission) + block numbering | `expand_ms`: the strip index of the entry lists (round 6: the 4:2:0 "
                            "walk reads the lists itself, csrc/fused_entries.hpp; JPGPU_PIPE_ENTRY_PIXELS=0: their expansion into whole blocks, "
                            "0.55 ms) | class finalize + pixel kernels.  No zero fill, no range scan, no write pass.",
                    "images_entry_pixels": int(best.get("images_entry_pixels", 0)),
                    "kernel_ms": {**{k: round(v, 3) for k, v in km.items()}, "sum": round(sum(km.values()), 3)},
                    "kernels_only_images_per_s": round(256 / sum(km.values()) * 1e3, 1), "total_ms": round(best["total_ms"], 3)}
                # every E entry against its own floors (VERDICT r3 next #1a)
                symbols = sum(huffman_symbols(J, d) for d in distinct) / len(distinct)
                spi = sync_pass_instructions(256 * symbols)
                for key, b in bests.items():
                    e2e_floor_fields(out[key], b, h2d_gbps, sum(km.values()) / 256.0)
                    if spi:
                        out[key]["vector_issue_floor_ms"] = round(spi["vector_issue_floor_ms_per_image"] * out[key]["images"], 3)
                        out[key]["frac_of_hard_floor"] = round(max(out[key]["vector_issue_floor_ms"], out[key]["link_floor_ms"] or 0.0) / out[key]["total_ms"], 4)
                out["roofline"] = {
                    "h2d_gbps": round(h2d_gbps, 2) if h2d_gbps else None,
                    "h2d_what": "one pinned 1-GB host-to-device copy, best of 3, this run",
                    "device_work_ms_per_image": round(sum(km.values()) / 256.0, 5),
                    "device_work_what": "kernels_256_one_sub_batch.kernel_ms.sum / 256: sync passes + block numbering + expansion + pixel kernels with the device to themselves "
                                        "(the late sync passes' chains included: an upper estimate of the work, a lower one of a lone sub-batch's latency)",
                    "huffman_symbols_per_image": int(symbols), "bits_per_symbol": round(out["jpeg_bytes_per_image"] * 8 / symbols, 2),
                    "sync_ns_per_symbol": round(km["sync_ms"] * 1e6 / (256 * symbols), 4),
                    "sync_pass_vector_instructions": spi,
                    "frac_of_floor": "max(link_floor_ms, device_work_ms) / total_ms per entry: 1.0 = the call takes what its larger floor takes "
                                     "(device_work_ms is measured, not a bound: overlapping sub-batches get under it)",
                    "frac_of_hard_floor": "max(link_floor_ms, vector_issue_floor_ms) / total_ms: the two floors nothing gets under — the PCIe link and the "
                                          "vector instructions the kernels issue"}
        finally:
            del os.environ["JPGPU_PIPE_DEV_SUB"], os.environ["JPGPU_PIPE_MAX_DEV_SUBS"]
        files_for_cpu = [distinct[i % len(distinct)] for i in range(256)]
        # The same images written with a restart marker after every MCU row (DRI): the chunk decoder takes each restart interval as
        # its own run of chunks (csrc/huff_job.hpp huff_chunk_span) — same kernels, no host entropy decoding either.
        if sizes and max(sizes) >= 1024:
            try:
                rfiles, rwho = e2e_files(synth, w, h, encoder, restart_rows=1)
                rwant = [hashlib.sha256(O.decode(d).pixels.tobytes()).hexdigest() for d in rfiles]
                n = 1024
                files = [rfiles[i % len(rfiles)] for i in range(n)]
                ts = warm_calls(p, files, 3, cold=1, download=False, device_entropy=True)
                okr = all(hashlib.sha256(p.download(i).tobytes()).hexdigest() == rwant[i % len(rfiles)] for i in (0, 1, n // 2, n - 1))
                out["restart_every_mcu_row_1024"], _ = e2e_entry(n, ts, w, h, p, okr, {"input": f"{len(rfiles)} distinct {w}x{h} files, repeated; written by {rwho}"})
            except Exception as e:  # noqa: BLE001 (this entry only)
                out["restart_every_mcu_row_1024"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # E as SURVEY 8(d) words it for one GPU — JPEG bytes in host memory -> RGB in HOST memory, what Decoder::decode() returns
        # (a Vec<u8>, src/decoder.rs:293-295): JPGPU_PIPELINE_DOWNLOAD, every sub-batch's pixels copied to pinned host memory behind its
        # kernels on a download stream of its own.  Floor: the pixel bytes at the D2H rate one pinned 1-GB copy reached in this run.
        for n in [x for x in (256, 4096) if sizes and x <= max(sizes)]:
            key = f"to_host_{n}"
            try:
                # (the pixels of the call stay in pinned host memory: 25.5 GB for 4,096 files — not on a box that cannot spare twice that)
                need, avail = n * w * h * 3, host_memory_available()
                if avail is not None and avail < 2 * need + (8 << 30):
                    out[key] = {"skipped": f"{need >> 20} MB of pinned host memory needed, {avail >> 20} MB available to this process"}
                    continue
                files = [distinct[i % len(distinct)] for i in range(n)]
                ts = warm_calls(p, files, 3, cold=1, download="pinned", device_entropy=True)
                ok = all(hashlib.sha256(p.pixels_host(i).tobytes()).hexdigest() == want[i % len(distinct)] for i in sorted({0, 1, n // 2, n - 1}))
                e, med = e2e_entry(n, ts, w, h, p, ok)
                e["pixel_bytes"] = int(med["pixel_bytes"])
                if d2h_gbps:
                    e["link_floor_ms"] = round(med["pixel_bytes"] / (d2h_gbps * 1e9) * 1e3, 3)
                    e["frac_of_floor"] = round(e["link_floor_ms"] / e["total_ms"], 4)
                    e["frac_of_floor_best"] = round(e["link_floor_ms"] / e["min_ms"], 4)
                    e["bound"] = "link (device to host)"
                    e["d2h_gbps"] = round(d2h_gbps, 2)
                e["what"] = "JPEG bytes in host memory -> RGB in pinned HOST memory (JPGPU_PIPELINE_DOWNLOAD): the D2H copy of a sub-batch runs behind its kernels on a download stream, next to the decode of the following sub-batches"
                out[key] = e
            except Exception as e:  # noqa: BLE001 (this entry only)
                out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # BASELINE configs[3]: benches/tower_progressive.jpg (512 x 512 progressive, 10 scans) x 256 and x 4,096: the scans of a
        # progressive frame are decoded ON THE DEVICE — round 6: one WAVE per scan, coefficients accumulated in the arena
        # (csrc/huff_prog_wave.hpp) — when the dispatcher's cost model says the device is ahead (csrc/pipeline.cpp).
        tp = os.path.join(B.ROOT, "tests", "golden", "benches", "tower_progressive.jpg")
        if os.path.exists(tp):
            data = open(tp, "rb").read()
            od = O.decode(data)
            for n in [x for x in (256, 4096) if sizes and x <= max(sizes)]:
                key = f"tower_progressive_{n}"
                try:
                    files = [data] * n
                    # (the route is the dispatcher's cost model — the same from the first call on; one uncounted call allocates)
                    ts = warm_calls(p, files, 5, cold=1, download=False, device_entropy=True)
                    okp = all(np.array_equal(p.download(i), od.pixels) for i in sorted({0, 1, n // 2, n - 1}))
                    e, med = e2e_entry(n, ts, od.width, od.height, p, okp)
                    e["file"] = "tests/golden/benches/tower_progressive.jpg (the reference's benches/tower_progressive.jpg: 512x512, 10 scans)"
                    e["images_device_progressive"] = int(med["images_device_progressive"])
                    if n <= 256:  # (the host route beside it where the two are close; at 4,096 frames it takes 0.4 s a call: tools/prog_calls.py --percent 0)
                        ts_h = warm_calls(p, files, 3, cold=1, download=False, device_entropy=True, progressive_on_host=True)
                        _m, med_ms, min_ms = call_stats(ts_h)
                        e["all_on_host_entropy_decoder"] = {"total_ms": round(med_ms, 3), "min_ms": round(min_ms, 3), "images_per_s": round(n / med_ms * 1e3, 1),
                                                            "what": "the same call with JPGPU_PIPELINE_PROGRESSIVE_ON_HOST (round 4's path: host entropy decoding, compact planes uploaded)"}
                    out[key] = e
                except Exception as e:  # noqa: BLE001 (this entry only)
                    out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            files_for_cpu = (files_for_cpu, [data] * 64, (od.width, od.height))
            # The same size and script, 64 DIFFERENT frames (Pillow: 512 x 512 4:4:4 progressive, quality 85, seeds 0x700 + k): the lanes of a
            # wave then walk different streams — every divergent step costs the wave — which copies of one file hide
            if sizes and max(sizes) >= 4096:
                try:
                    import io
                    from PIL import Image
                    frames = []
                    for k in range(64):
                        buf = io.BytesIO()
                        Image.fromarray(synth.synthetic_rgb(512, 512, seed=0x700 + k)).save(buf, format="JPEG", quality=85, subsampling="4:4:4", progressive=True)
                        frames.append(buf.getvalue())
                    n = 4096
                    files = [frames[i % 64] for i in range(n)]
                    ts = warm_calls(p, files, 5, cold=1, download=False, device_entropy=True)
                    okd = all(np.array_equal(p.download(i), O.decode(frames[i % 64]).pixels) for i in (0, 1, 63, n // 2 + 7, n - 1))
                    e, med = e2e_entry(n, ts, 512, 512, p, okd)
                    e["input"] = "64 distinct 512x512 4:4:4 progressive frames (Pillow / libjpeg-turbo, quality 85, default script: 10 scans), repeated"
                    e["jpeg_bytes_per_image"] = int(sum(len(f) for f in frames) / 64)
                    e["images_device_progressive"] = int(med["images_device_progressive"])
                    out["progressive_distinct_4096"] = e
                except Exception as e:  # noqa: BLE001 (this entry only)
                    out["progressive_distinct_4096"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        p.close()
        J._native.lib().jpgpu_trim_caches()
    return out, files_for_cpu


def e2e_cpu_budget(J, O, synth, w, h, encoder, n=4096, points=None, matrix=False):
    """E against the host CPUs a GPU's feeder gets: on an 8-GPU node with 16 granted CPUs every rank has two.  The
    4,096-file call with the calling thread's affinity — and with it the pipeline's pools, created under it — narrowed to the first
    1 / 2 / 4 / 8 / 16 allowed CPUs and threads = 2 x CPUs (rank_cpu_share's rule).  Per point: the files in ordinary (pageable) memory
    with the library's default (host light — raw scans copied, marker check + unstuffing on the device — at every thread count since
    round 6), the same with either mode forced (A/B), and the files in a pinned arena (PinnedFiles: what a loader
    that reads into jpgpu_host_alloc memory holds) with JPGPU_PIPELINE_INPUT_PINNED, where the copy engine reads the arena itself."""
    # Default run (the whole bench under 45 s): 2 / 8 / 16 CPUs x {the library's own mode, host staging forced}, 3 warm
    # calls — enough to see whether the default is the better mode at every point.  --cpu-budget-matrix: 1 / 2 / 4 / 8 / 16 CPUs x
    # {default, staging forced, light forced, pinned input}, 5 warm calls (round 5's table).
    points = points or ((1, 2, 4, 8, 16) if matrix else (2, 8, 16))
    calls = 5 if matrix else 3
    distinct, who = e2e_files(synth, w, h, encoder)
    want = [hashlib.sha256(O.decode(d).pixels.tobytes()).hexdigest() for d in distinct]
    files = [distinct[i % len(distinct)] for i in range(n)]
    allowed = sorted(os.sched_getaffinity(0))
    granted = B.effective_cpus()
    out = {"images": n, "input": f"{len(distinct)} distinct {w}x{h} files, repeated; written by {who}", "host_cpus_granted": granted,
           "what": "jpgpu_pipeline_decode of the same 4,096 files with affinity (sched_setaffinity before the pipeline's threads are created) and thread "
                   f"count limited: JPEG bytes in host memory -> RGB in HBM; per point and input median / min of {calls} warm calls; "
                   "cpu_ms_per_image = process CPU time of the median call / images",
           "points": []}
    arena = J.PinnedFiles(files) if matrix else None
    inputs = [("pageable_input", files, {}), ("pageable_input_host_staging", files, {"host_light": False})]
    if matrix:
        inputs += [("pageable_input_host_light", files, {"host_light": True}), ("pinned_input", arena, {"input_pinned": True})]
    try:
        for c in points:
            if c > min(granted, len(allowed)):
                continue
            row = {"cpus": c, "threads": max(2, 2 * c)}
            os.sched_setaffinity(0, allowed[:c])
            try:
                p = J.Pipeline(threads=row["threads"])
                try:
                    for name, src, kw in inputs:
                        try:
                            ts = warm_calls(p, src, calls, cold=2 if matrix else 1, download=False, device_entropy=True, **kw)
                            ok = all(hashlib.sha256(p.download(i).tobytes()).hexdigest() == want[i % len(distinct)] for i in sorted({0, n // 2, n - 1}))
                            med, med_ms, min_ms = call_stats(ts)
                            row[name] = {"total_ms": round(med_ms, 3), "min_ms": round(min_ms, 3), "images_per_s": round(n / med_ms * 1e3, 1),
                                         "cpu_ms_per_image": round(med["cpu_ms"] / n, 5), "cpus_busy": round(med["cpu_ms"] / med_ms, 2),
                                         "mode": "host-light" if med["images_host_light"] else "host staging",
                                         "images_host_light": int(med["images_host_light"]), "images_device_entropy": int(med["images_device_entropy"]),
                                         "verified_vs_oracle": bool(ok)}
                        except Exception as e:  # noqa: BLE001 (this point only)
                            row[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                finally:
                    p.close()
            finally:
                os.sched_setaffinity(0, allowed)
            out["points"].append(row)
    finally:
        if arena is not None:
            arena.close()
        J._native.lib().jpgpu_trim_caches()
    return out


def cpu_baseline_e2e(O, files, w, h, target_seconds):
    """The oracle's whole Decoder::decode() (marker parsing, Huffman decoding, IDCT, upsampling, colour conversion) on the e2e
    block's files, one file per task on every core the process may use; bounded sample."""
    progressive = None
    if isinstance(files, tuple):
        files, pfiles, (pw, ph) = files
        progressive = cpu_baseline_e2e(O, pfiles, pw, ph, max(2.0, target_seconds / 3))
    out = _cpu_e2e_sample(O, files, w, h, target_seconds)
    if progressive:
        out["tower_progressive"] = progressive
    return out


def _cpu_e2e_sample(O, files, w, h, target_seconds):
    cores = B.effective_cpus()
    flags = O.use_native_build()
    n0 = max(2 * cores, 8)
    t0 = time.perf_counter()
    ok, _px = O.batch_decode([files[i % len(files)] for i in range(n0)], cores)
    dt = time.perf_counter() - t0
    n = int(max(n0, min(65536, n0 * target_seconds / max(dt, 1e-3))))
    n = (n // cores) * cores or cores
    t0 = time.perf_counter()
    ok, px = O.batch_decode([files[i % len(files)] for i in range(n)], cores)
    dt = time.perf_counter() - t0
    return {"value": round(n * w * h / 1e6 / dt, 2), "unit": "MP/s", "images_per_s": round(n / dt, 1), "cores": cores, "kind": "port",
            "sample": f"{n} decodes of the e2e block's {w}x{h} files ({ok} ok, {px} pixel bytes), JPEG bytes -> RGB in host memory, whole decode "
                      f"(parse + Huffman + IDCT + upsampling + colour), {cores} threads one file per task, {dt:.1f} s; gcc {flags}; {B.cpu_model()}; "
                      f"the crate's own x86 build would add SSSE3 IDCT / colour kernels (not bit-compatible with its scalar path)"}


def e2e_sharded(J, O, synth, D, dist, torch, dev, rank, local_rank, world, w, h, total, encoder, share, threads, pinned):
    """E at N ranks (north_star: "throughput on synthetic 4:2:0 baseline JPEGs is reported at 1, 2, 4 and 8 GPUs"): `total` files, rank r
    decodes D.shard(total, r, world) of them through a pipeline of its own on ITS GPU with ITS share of the host (CPU affinity set,
    threads = 2 x granted CPUs / ranks); every call starts behind a barrier, so the ranks contend for the host at the same moment; the
    job's time is the slowest rank's (MAX over ranks) median — and min — of 5 warm calls.  No collective on the data path; pixels
    stay in each rank's HBM.
    Collectives and failures: every rank runs the SAME sequence of collectives whatever happens to it locally — a local
    error is caught, kept, and an all-reduced flag after every call lets all ranks leave the loop together; the error is re-raised only
    after the last collective."""
    err, p, files, ts, distinct, who = None, None, [], [], [], ""
    mine = D.shard(total, rank, world)
    try:
        distinct, who = e2e_files(synth, w, h, encoder)
        files = [distinct[i % len(distinct)] for i in mine]
        p = J.Pipeline(device=local_rank, threads=threads)
    except Exception as e:  # noqa: BLE001 (kept: see above)
        err = e
    for r in range(6):  # the first call allocates arenas and staging: not counted
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        if err is None:
            try:
                res = p.decode(files, download=False, device_entropy=True)
                bad = [x for x in res if isinstance(x, Exception)]
                if bad:
                    raise bad[0]
                if r > 0:
                    ts.append(p.timings())
            except Exception as e:  # noqa: BLE001
                err = e
        if D.min_over_ranks(0.0 if err else 1.0, device=dev) < 1.0:
            break  # (every rank sees the same flag: all leave here together)
    ok, mine_ms, mine_min, med, kernel_path = 0.0, 0.0, 0.0, None, ""
    if err is None:
        try:
            ok = 1.0
            if files:
                want = {k: hashlib.sha256(O.decode(distinct[k]).pixels.tobytes()).hexdigest() for k in {mine[0] % len(distinct), mine[len(mine) - 1] % len(distinct)}}
                for j in (0, len(files) - 1):
                    ok = min(ok, 1.0 if hashlib.sha256(p.download(j).tobytes()).hexdigest() == want[mine[j] % len(distinct)] else 0.0)
            if ts:
                med, mine_ms, mine_min = call_stats(ts)
            kernel_path = p.kernel_path
        except Exception as e:  # noqa: BLE001
            err, ok = e, 0.0
    slowest, slowest_min, cpu_ms = D.max_over_ranks([mine_ms, mine_min, med["cpu_ms"] if med else 0.0], device=dev)
    all_ok = D.min_over_ranks(ok if err is None else 0.0, device=dev)
    if p is not None:
        p.close()
    if err is not None:
        raise err  # (after the last collective of this leg)
    return {"images": total, "images_per_rank": len(files), "ranks": world, "calls": len(ts), "total_ms": round(slowest, 3), "min_ms": round(slowest_min, 3),
            "images_per_s": round(total / slowest * 1e3, 1) if slowest else None,
            "images_per_s_best": round(total / slowest_min * 1e3, 1) if slowest_min else None,
            "value": round(total * w * h / 1e6 / slowest * 1e3, 1) if slowest else None, "unit": "MP/s",
            "rank0_ms": round(mine_ms, 3), "threads_per_rank": threads, "cpus_per_rank": len(share), "cpu_affinity_set": bool(pinned),
            "cpu_ms_per_image_slowest_rank": round(cpu_ms / max(len(files), 1), 5),
            "mode": ("host-light" if med and med["images_host_light"] else "host staging"),
            "host_cpus_granted": effective_cpus_unpinned(), "kernel_path": kernel_path, "verified_vs_oracle": bool(all_ok >= 1.0), "every_rank_ok": bool(all_ok >= 1.0),
            "input": f"{len(distinct)} distinct {w}x{h} files, repeated; written by {who}",
            "what": "jpgpu_pipeline_decode per rank on its shard of the file list (entropy decoding on the device), every call behind a barrier; "
                    "MAX over ranks of each rank's median (total_ms) and min (min_ms) of 5 warm calls; pixels stay in each rank's HBM"}


_UNPINNED_CPUS = None


def note_unpinned_cpus():
    """Remember effective_cpus() before this process narrows its own affinity mask."""
    global _UNPINNED_CPUS
    _UNPINNED_CPUS = B.effective_cpus()


def effective_cpus_unpinned():
    """B.effective_cpus() as it was before this process narrowed its own affinity mask (rank_cpu_share / pin_to)."""
    return _UNPINNED_CPUS if _UNPINNED_CPUS is not None else B.effective_cpus()


# ---------------------------------------------------------------------------------------------------------------
def dry_run(args, rank, world, workload, images_total, n_img, n_sub):
    """No GPU: the N>1 bookkeeping on CPU tensors over gloo — shard sizes, per-sub-batch gather to rank 0, max over
    ranks, the per-rank host budget of the e2e leg (CPU share, thread count, its shard of the file list) and the JSON contract.
    Measures nothing (value is null)."""
    import torch
    import jpeg_decoder_amd.distributed as D

    dist = D.init(backend="gloo") if world > 1 else None
    # the e2e leg's host budget, exactly as the GPU run sets it up (rank_cpu_share + pin_to), checked across ranks below
    global _UNPINNED_CPUS
    _UNPINNED_CPUS = B.effective_cpus()
    share, threads = rank_cpu_share(rank, world)
    pinned = pin_to(share)
    e2e_mine = D.shard(B.E2E_SHARDED_TOTAL, rank, world)
    budget = torch.tensor([float(threads), float(len(share)), float(min(share)), float(max(share)), float(len(e2e_mine)), 1.0 if pinned else 0.0], dtype=torch.float64)
    budgets = [torch.zeros_like(budget) for _ in range(world)]
    if dist:
        dist.all_gather(budgets, budget)
    else:
        budgets = [budget]
    mine = D.shard(images_total, rank, world) if images_total else range(rank * n_img, (rank + 1) * n_img)
    n_sub = max(1, min(n_sub, len(mine)))
    bounds = [(len(mine) * s // n_sub, len(mine) * (s + 1) // n_sub) for s in range(n_sub)]
    tag = 5  # bytes standing in for one image's pixels: the image's global index
    slices = []
    for a, b in bounds:
        t = torch.zeros((b - a) * tag, dtype=torch.uint8)
        for k in range(a, b):
            t[(k - a) * tag:(k - a + 1) * tag] = mine[k] % 251
        slices.append(t)
    ok = True
    if dist:
        g = B.PixelGather(dist, torch, rank, world, slices, "cpu")
        for s in range(n_sub):
            g.post(s)
        g.wait()
        dist.barrier()
        if rank == 0:
            for r in range(1, world):
                theirs = D.shard(images_total, r, world) if images_total else range(r * n_img, (r + 1) * n_img)
                tb = [(len(theirs) * s // n_sub, len(theirs) * (s + 1) // n_sub) for s in range(n_sub)]
                for s, (a, b) in enumerate(tb):
                    ok = ok and g.recv[r - 1][s].numel() == (b - a) * tag
                    for k in range(a, b):
                        ok = ok and bool((g.recv[r - 1][s][(k - a) * tag:(k - a + 1) * tag] == theirs[k] % 251).all())
        t = D.max_over_ranks([float(rank)])
        ok = ok and t == [float(world - 1)]
    if rank == 0:
        w, h = B.WORKLOADS[workload][0], B.WORKLOADS[workload][1]
        rows = [[float(x) for x in b] for b in budgets]
        disjoint = all(rows[i][3] < rows[i + 1][2] for i in range(len(rows) - 1)) or len(sorted(os.sched_getaffinity(0))) < world
        print(json.dumps({"metric": "megapixels/s decoded (batch, whole node)", "value": None, "unit": "MP/s", "n_gpus": world,
                          "dry_run": True, "scaling": "strong" if images_total else "weak",
                          "config": {"workload": f"{w}x{h}", "name": workload, "images_total": images_total or world * n_img,
                                     "images_per_gpu": len(mine), "sub_batches": n_sub},
                          "gather_checked": ok, "value_with_gather": None, "gather_ms": None,
                          "e2e": {"sharded": {"images": B.E2E_SHARDED_TOTAL, "ranks": world, "images_per_rank": [int(r[4]) for r in rows],
                                              "threads_per_rank": [int(r[0]) for r in rows], "cpus_per_rank": [int(r[1]) for r in rows],
                                              "cpu_shares_disjoint": bool(disjoint), "cpu_affinity_set": [bool(r[5]) for r in rows],
                                              "host_cpus_granted": _UNPINNED_CPUS, "total_ms": None}}}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1



def main(argv=None):
    """The side legs alone (one GPU): `python tools/bench_e2e.py --legs e2e,cpu_budget [--e2e-images 256,4096]` prints the detail
    document of those legs as one JSON line (and writes it to --out)."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="e2e", help="comma list of: e2e, cpu_budget, cpu_e2e")
    ap.add_argument("--e2e-images", default="256,1024,4096")
    ap.add_argument("--e2e-encoder", default="auto")
    ap.add_argument("--cpu-budget-matrix", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    ap.add_argument("--out", default=os.path.join("gpurun_out", "bench_e2e_detail.json"))
    args = ap.parse_args(argv)
    import torch
    import jpeg_decoder_amd as J
    import oracle as O
    import synth
    if not torch.cuda.is_available():
        raise SystemExit("tools/bench_e2e.py needs an MI355X; there is no CPU fallback")
    dev = torch.device("cuda", 0)
    w, h = B.WORKLOADS["1080p-420"][:2]
    legs, doc, files = set(args.legs.split(",")), {}, None
    if "e2e" in legs:
        doc["e2e"], files = e2e_block(J, O, synth, w, h, [int(x) for x in args.e2e_images.split(",") if x], args.e2e_encoder,
                                      h2d_rate_gbps(torch, dev), d2h_rate_gbps(torch, dev))
    if "cpu_budget" in legs:
        doc.setdefault("e2e", {})["cpu_budget"] = e2e_cpu_budget(J, O, synth, w, h, args.e2e_encoder, matrix=args.cpu_budget_matrix)
    if "cpu_e2e" in legs:
        if files is None:
            d, _who = e2e_files(synth, w, h, args.e2e_encoder)
            files = [d[i % len(d)] for i in range(256)]
        doc["cpu_baseline_e2e"] = cpu_baseline_e2e(O, files, w, h, args.cpu_seconds)
    doc["e2e_summary"] = B.e2e_summary(doc)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    os.environ.setdefault("JPGPU_BATCH_KERNEL_TIMES", "1")
    main()
