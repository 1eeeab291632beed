"""bench.py's multi-GPU entry point, exercised without a GPU: `python bench.py --gpus N` must either run N ranks or
fail loudly — never report a 1-GPU run under an N-GPU label.  `--dry-run` walks the launcher, the
sharding of BASELINE configs[2] (4096 images of 3840x2160), the per-sub-batch gather to rank 0 (gloo here, RCCL on
the GPUs) and the JSON contract on CPU tensors."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_n_without_n_gpus_is_an_error():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    r = _run(["--gpus", "8"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 8" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # and no bench line under a false label


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_launcher_command_is_one_rank_per_gpu_on_loopback():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], port=12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "3"]


@pytest.mark.timeout(300)
def test_two_ranks_dry_run_shards_config3_and_gathers():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["scaling"] == "strong"
    assert line["config"]["name"] == "2160p-420" and line["config"]["images_total"] == 4096
    assert line["config"]["images_per_gpu"] == 2048 and line["config"]["sub_batches"] == 8
    assert line["gather_checked"] is True
    for key in ("value", "value_with_gather", "gather_ms"):
        assert key in line
    # the e2e leg's per-rank host budget (VERDICT r3 next #2a/c): the file list sharded, the granted CPUs DIVIDED between the ranks
    # (never the one-pipeline default in every rank), every rank's feeder threads on a share of its own
    sys.path.insert(0, ROOT)
    import bench
    sh = line["e2e"]["sharded"]
    assert sh["images"] == 4096 and sh["ranks"] == 2 and sh["images_per_rank"] == [2048, 2048]
    granted = sh["host_cpus_granted"]
    assert granted == bench.effective_cpus()
    assert sh["threads_per_rank"] == [max(2, 2 * granted // 2)] * 2   # one pipeline's default (twice the granted CPUs) DIVIDED by the ranks
    assert sum(sh["threads_per_rank"]) <= max(4, 2 * granted)
    assert sh["cpu_shares_disjoint"] is True and sum(sh["cpus_per_rank"]) <= len(os.sched_getaffinity(0))


@pytest.mark.timeout(300)
def test_three_ranks_dry_run_uneven_total_and_weak_mode():
    r = _run(["--gpus", "3", "--dry-run", "--images-total", "100", "--sub-batches", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 3 and line["config"]["images_total"] == 100 and line["config"]["images_per_gpu"] == 34
    assert line["gather_checked"] is True
    assert line["e2e"]["sharded"]["images_per_rank"] == [1366, 1365, 1365] and len(set(line["e2e"]["sharded"]["threads_per_rank"])) == 1
    r = _run(["--gpus", "2", "--dry-run", "--batch", "16", "--workload", "1080p-420"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["scaling"] == "weak" and line["config"]["images_per_gpu"] == 16 and line["config"]["images_total"] == 32


@pytest.mark.parametrize("encoder", ["auto", "baseline"])
def test_e2e_input_files_with_and_without_restart_markers(encoder):
    """bench.py's e2e inputs (the 256 / 1,024 / 4,096-file lines and the restart-marker line): written by Pillow where it is installed,
    by tools/baseline_encoder.py otherwise; the restart variant really carries a DRI segment and RSTn markers, and the oracle decodes
    all of them (same picture with and without markers when the encoder is the same)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import bench
    import oracle as O
    import synth
    if encoder == "auto":
        pytest.importorskip("PIL")
    plain, who = bench.e2e_files(synth, 160, 96, encoder, distinct=2)
    rst, who_r = bench.e2e_files(synth, 160, 96, encoder, distinct=2, restart_rows=1)
    assert "no restart markers" in who and "restart marker every 1 MCU row" in who_r
    for a, b in zip(plain, rst):
        assert b"\xff\xdd\x00\x04" not in a and b"\xff\xdd\x00\x04" in b and b.count(b"\xff\xd0") >= 1
        pa, pb = O.decode(a).pixels, O.decode(b).pixels
        assert pa.size == 96 * 160 * 3 and np.array_equal(pa, pb)


# ---- the contract line (round 5's one line had grown to 21 KB and the driver's record lost its headline) ----
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "verified_vs_oracle")


def _canned_detail():
    """A whole detail document as a default run produces it: round 5's 21 KB line, kept under profiles/."""
    return json.load(open(os.path.join(ROOT, "profiles", "round5", "07_bench_driver_command_final.json")))


def test_contract_line_is_small_and_complete():
    sys.path.insert(0, ROOT)
    import bench
    detail = _canned_detail()
    assert len(json.dumps(detail)) > 20000  # (the thing that did not fit)
    txt = bench.contract_line(detail, "gpurun_out/bench_detail.json")
    assert "\n" not in txt and len(txt) < 4096 == bench.CONTRACT_LINE_MAX
    line = json.loads(txt)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["value"] == detail["value"] and line["ms_per_step"] == detail["ms_per_step"]
    assert set(line["config"]) >= {"workload", "name", "images_total", "images_per_gpu", "kernel_path", "range_class"}
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_commit", "algorithmic_bytes_per_launch", "kernel_ms_per_launch"}
    assert line["roofline"]["frac"] == detail["roofline"]["frac"]
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and len(line["cpu_baseline"]["sample"]) <= 200
    s = line["e2e_summary"]
    assert len(s) <= 10 and all(v is None or isinstance(v, (int, float, bool)) for v in s.values())
    assert s["e2e_4096_ms"] == detail["e2e"]["4096"]["total_ms"] and s["progressive_256_on_device"] == 0
    assert s["cpu_budget_2cpus_images_per_s"] == next(p for p in detail["e2e"]["cpu_budget"]["points"] if p["cpus"] == 2)["pageable_input"]["images_per_s"]


def test_contract_line_stays_small_whatever_the_side_legs_hold():
    """Errors, long texts, the N > 1 keys and a sharded E leg: the line keeps its cap and its required keys."""
    sys.path.insert(0, ROOT)
    import bench
    detail = _canned_detail()
    detail["config"]["workload"] = "w" * 5000
    detail["cpu_baseline"]["sample"] = "s" * 5000
    detail["e2e"] = {"error": "x" * 3000, "sharded": {"images_per_s": 1.0, "total_ms": 2.0, "mode": "host-light", "what": "y" * 4000}}
    detail.update({"n_ranks_seen": 8, "value_with_gather": 1.0, "ms_per_step_with_gather": 2.0, "gather_ms": 3.0, "gather_verified": True,
                   "gather": {"form": "z" * 3000}, "k_4096": {"error": "e" * 300}, "scale_anchor": {"error": "e" * 300}})
    txt = bench.contract_line(detail, "gpurun_out/bench_detail.json")
    assert len(txt) < 4096
    line = json.loads(txt)
    for k in CONTRACT_KEYS + ("value_with_gather", "gather_ms", "n_ranks_seen"):
        assert k in line, k
    assert line["e2e_summary"]["sharded_images_per_s"] == 1.0


def test_emit_prints_the_detail_first_and_the_contract_line_last(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import bench
    detail = _canned_detail()
    path = str(tmp_path / "d" / "bench_detail.json")
    bench.emit(detail, path)
    out = capsys.readouterr().out.splitlines()
    assert out[-2].startswith("bench_detail: {") and json.loads(out[-2][len("bench_detail: "):]) == detail
    last = json.loads(out[-1])
    assert len(out[-1]) < 4096 and last["value"] == detail["value"] and last["detail"] == path
    assert json.load(open(path)) == detail
