#!/bin/bash
# shader / memory clocks and power while the default bench workload runs (and while the pure-VALU micro-benchmark does)
O=gpurun_out/clocks; mkdir -p $O
rocm-smi --showclocks --showpower > $O/idle.txt 2>&1
python bench.py --steps 12000 --warmup 50 > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 14
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|power" ; sleep 1; done > $O/under_load.txt
wait $BP
cat $O/idle.txt | grep -E "sclk|mclk|Power|power"; echo ---; cat $O/under_load.txt; tail -c 300 $O/bench.json
