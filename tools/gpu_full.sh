#!/bin/bash
# full GPU check: every -m gpu test, smoke, default bench line (to gpurun_out/<dir>)
O=gpurun_out/${1:-full}
mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/tests.txt
python __graft_entry__.py smoke > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.txt; tail -2 $O/smoke.txt; cat $O/bench.json; tail -3 $O/bench.err
