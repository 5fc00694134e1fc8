#!/bin/bash
# GPU box: tree parity tests under the given tile modes (1 = level by level, 2 = subtree tiles), then the tree bench per mode.   bash tools/tree_modes.sh <tag> "<test modes>" "<bench modes>"
TAG=${1:-x}
O=gpurun_out/tree_$TAG
mkdir -p $O; : > $O/summary.txt
K="tree or hierarch or forest or chain or inherit or propagat or static or shard or tile"
for m in $2; do
  MI_TEST_TILE_MODE=$m timeout 600 python -m pytest tests -m gpu -x -q -k "$K" > $O/test_mode$m.log 2>&1; echo "test mode $m rc=$? $(tail -1 $O/test_mode$m.log)" >> $O/summary.txt
done
for m in $3; do
  timeout 300 python bench.py --workload tree --tile-mode $m --steps 50 --blocks 8 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic > $O/bench_mode$m.json 2> $O/bench_mode$m.err
  python - <<P >> $O/summary.txt
import json
d=json.load(open("bench_full.json"))
print("bench mode $m", d["ms_per_step"], d["kernels"], d["config"].get("tile_plan"))
P
done
cat $O/summary.txt
