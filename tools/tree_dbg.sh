#!/bin/bash
O=gpurun_out/tree_dbg
mkdir -p $O; : > $O/summary.txt
for m in "$@"; do
  timeout 300 python bench.py --workload tree --tile-mode $m --steps 50 --blocks 6 --no-cpu-baseline --no-other-workloads --no-end-to-end > $O/bench_mode$m.json 2> $O/bench_mode$m.err
  python - <<P >> $O/summary.txt
import json
d=json.load(open("$O/bench_mode$m.json"))
print("mode $m", d["ms_per_step"], d["kernels"], d["config"].get("tile_plan"))
P
done
cat $O/summary.txt
