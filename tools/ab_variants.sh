#!/bin/bash
# GPU box: the same bench under several library builds (bevy_amd.build.build(variant=...)), interleaved so that clock and box
# differences spread evenly.   bash tools/ab_variants.sh <tag> "<variants: - = the product build>" <rounds> -- <bench.py arguments>
TAG=$1; VARS=$2; ROUNDS=$3; shift 4
O=gpurun_out/ab_$TAG
mkdir -p $O; : > $O/summary.txt
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    if [ "$v" = "-" ]; then unset MI_LIB_VARIANT; else export MI_LIB_VARIANT=$v; fi
    timeout 300 python bench.py "$@" --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic > $O/${v}_$r.json 2> $O/${v}_$r.err
    python - <<P >> $O/summary.txt
import json
try:
    d = json.load(open("bench_full.json")); print("round $r variant $v", d["ms_per_step"], d["kernels"])
except Exception as e: print("round $r variant $v FAILED", e)
P
  done
done
cat $O/summary.txt
