#!/bin/bash
# full GPU check + the bench lines of the other kinds
O=gpurun_out/${1:-full2}
mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt
python __graft_entry__.py smoke > $O/smoke.txt 2>&1
: > $O/bench.jsonl
for wl in 1080p-420 1080p-440 1080p-422 1080p-444; do python bench.py --workload $wl --no-cpu-baseline --no-classes >> $O/bench.jsonl 2>> $O/bench.err; done
cat $O/tests.txt; tail -1 $O/smoke.txt
python - <<PY
import json
for l in open("$O/bench.jsonl"):
    l = json.loads(l); print(l["config"]["name"], l["config"]["kernel_path"], l["ms_per_step"], l["roofline"]["frac"], l["verified_vs_oracle"])
PY
