#!/bin/bash
# round 3, call A: GPU tests, smoke, the driver's bench command, the one-rank RCCL run
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 25 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench exit $?"
tail -c 1500 $O/bench_driver.err
timeout 600 python bench.py --force-dist --no-e2e --no-k4096 --no-cpu-baseline > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "force-dist exit $?"
tail -c 600 $O/bench_forcedist.err
python - <<'PY'
import json
for f in ("bench_driver", "bench_forcedist"):
    try:
        d = json.loads(open(f"gpurun_out/r3a/{f}.json").read().strip().splitlines()[-1])
        keep = {k: d.get(k) for k in ("value", "ms_per_step", "sustained", "n_ranks_seen", "k_4096", "e2e", "cpu_baseline_e2e")}
        keep["roofline"] = {k: d["roofline"].get(k) for k in ("frac", "kernel_ms_per_launch")}
        keep["by_class"] = {k: (v.get("kernel_ms_per_launch") if isinstance(v, dict) else None) for k, v in d.get("roofline_by_class", {}).items()}
        print(f, json.dumps(keep)[:3000])
    except Exception as e:
        print(f, "unreadable", e)
PY
