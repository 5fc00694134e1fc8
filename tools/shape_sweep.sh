#!/bin/bash
# bash tools/shape_sweep.sh <outdir>: one bench line per hierarchy stress shape and frame kind (bench.py --workload tree --tree-shape ...)
OUT=${1:-gpurun_out/shapes}
mkdir -p $OUT
for shape in ${SHAPES:-large_tree wide_tree deep_tree chain ropes bundle update_leaves update_shallow humanoids_active humanoids_inactive humanoids_mixed tree_4ary_depth11 tree_4ary_depth12}; do
  for kind in all movers; do
    case $shape in tree_4ary*) [ $kind = movers ] && continue ;; esac
    timeout 300 python bench.py --workload tree --tree-shape $shape --tree-shape-frame $kind --steps 20 --warmup 5 --blocks 8 \
        --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic --full-line 2> $OUT/${shape}_$kind.err | tail -n 2 | head -n 1 > $OUT/${shape}_$kind.json
  done
done
python - "$OUT" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
print("| shape | frame | nodes | levels | launches/frame | kernels us/frame | frac (141 B/node over the frame's kernels) | ms/step |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print("|", os.path.basename(f), "| unreadable:", e, "|")
        continue
    rf, c = d.get("roofline_frame") or {}, d["config"]
    print(f"| {c['shape']} | {'movers' if f.endswith('_movers.json') else 'all dirty'} | {c['nodes']} | {c['levels']} | {rf.get('launches_per_frame')} | {rf.get('kernels_us_per_frame')} | {rf.get('frac')} | {d['ms_per_step']} |")
PY
