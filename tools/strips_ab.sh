#!/bin/bash
# bash tools/strips_ab.sh <outdir>: the hierarchy stress shapes through the default plan, the workgroup tiles (--tile-mode 4) and strips forced
# (--tile-mode 5, at MI_STRIP_W = 32 / 64 / 128): all-dirty and movers frames, us per step and kernel us per frame.
OUT=${1:-gpurun_out/strips_ab}
mkdir -p $OUT
run() {  # name, shape, kind, tile mode
  timeout 300 python bench.py --workload tree --tree-shape $2 --tree-shape-frame $3 --tile-mode $4 --steps 50 --warmup 10 --blocks 8 \
      --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic --full-line 2> $OUT/$1.err | tail -n 2 | head -n 1 > $OUT/$1.json
}
for shape in ${SHAPES:-large_tree deep_tree update_leaves update_shallow humanoids_active humanoids_mixed wide_tree tree_4ary_depth11}; do
  for kind in all movers; do
    case $shape in tree_4ary*) [ $kind = movers ] && continue ;; esac
    run ${shape}_${kind}_default $shape $kind 0
    run ${shape}_${kind}_wgtiles $shape $kind 4
    for w in ${WIDTHS:-32 64 128}; do MI_STRIP_W=$w run ${shape}_${kind}_strips$w $shape $kind 5; done
  done
done
python - "$OUT" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
print("| shape | frame | plan | launches/frame | tiles | kernels us/frame | us/step |")
print("|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(f)[:-5]
    try:
        d = json.load(open(f))
    except Exception as e:
        print("|", name, "| unreadable:", e, "|")
        continue
    rf, c = d.get("roofline_frame") or {}, d["config"]
    print(f"| {c['shape']} | {'movers' if '_movers_' in name else 'all dirty'} | {name.rsplit('_', 1)[1]} | {rf.get('launches_per_frame')} | {c.get('tile_plan')} | {rf.get('kernels_us_per_frame')} | {round(1e3 * d['ms_per_step'], 2)} |")
PY
