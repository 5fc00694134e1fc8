#!/bin/bash
# GPU box, round 3 call B: which of the three changes moved the metric frame (SLP flag, walk epilogue)
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
for v in "" slp oldwalk slp_oldwalk; do
  for wl in frame lights flat; do
    MI_LIB_VARIANT=$v timeout 200 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/${wl}_${v:-cur}.json 2> $O/${wl}_${v:-cur}.err
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), "us", d["kernels"])
    except Exception as e:
        print(f, "ERR", e)
P
