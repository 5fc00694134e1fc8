#!/bin/bash
# GPU box: full GPU test suite + short benches of the main workloads.   bash tools/quick_bench.sh <tag>
TAG=${1:-x}
O=gpurun_out/quick_$TAG
mkdir -p $O; : > $O/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gputest.log)" >> $O/summary.txt
for wl in frame flat flat_static tree lights batching; do
  timeout 300 python bench.py --workload $wl --steps 50 --blocks 8 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic > $O/$wl.json 2> $O/$wl.err
  cp bench_full.json $O/$wl.full.json 2>/dev/null
  python - <<P >> $O/summary.txt
import json
try:
    d=json.load(open("$O/$wl.full.json")); print("$wl", d["ms_per_step"], d["kernels"], d["roofline"]["frac"])
except Exception as e: print("$wl FAILED", e)
P
done
cat $O/summary.txt
