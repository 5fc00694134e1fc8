#!/bin/bash
# GPU box, round 3 call E: whole suite (default, tile pre-test forced, sphere path forced), whole bench line, packed-planes A/B
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" >> $O/summary.txt
MI_TEST_TILE_PRETEST=2 MI_TEST_SPHERE_PATH=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_tile_kernels.py tests/test_gpu_sphere_path.py tests/test_gpu_cluster.py -m gpu -x -q > $O/pytest_forced.log 2>&1; echo "pytest forced rc=$?" >> $O/summary.txt
timeout 900 python bench.py --steps 100 --warmup 20 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" >> $O/summary.txt
for v in "" packed; do
  MI_LIB_VARIANT=$v timeout 200 python bench.py --workload flat_static --entities 10000000 --views 4 --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/static_10m4_${v:-cur}.json 2> $O/static_10m4_${v:-cur}.err
  MI_LIB_VARIANT=$v timeout 200 python bench.py --workload flat --entities 10000000 --views 4 --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/flat_10m4_${v:-cur}.json 2> $O/flat_10m4_${v:-cur}.err
  MI_LIB_VARIANT=$v timeout 200 python bench.py --workload frame --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/frame_${v:-cur}.json 2> $O/frame_${v:-cur}.err
done
for pt in 1 2; do
  for mv in subtree leaves; do
    MI_TEST_TILE_PRETEST=$pt timeout 200 python bench.py --workload tree --tree-moved $mv --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/tree_${mv}_pt$pt.json 2> $O/tree_${mv}_pt$pt.err
  done
done
tail -n 4 $O/pytest_default.log $O/pytest_forced.log | cat
cat $O/summary.txt
python - <<'P'
import json,glob
d=json.loads(open("gpurun_out/r03e/bench_full.json").read().strip().splitlines()[-1])
print("frame", d["ms_per_step"]*1e3, "us", d["kernels"], d["roofline"]["frac"])
e=d["end_to_end"]
for k in ("1pct_dirty","10pct_dirty","100pct_dirty"):
    print(k, e[k]["us_per_frame"], e[k]["pcie_frac"], e[k]["stage_us"])
h=d["end_to_end_host_layer"]
for form in ("three_systems","fused_frame"):
    print(form, {k: {kk: vv for kk,vv in v.items() if kk.endswith("us") or kk=="us_per_frame" or kk=="device_waits"} for k,v in h.get(form,{}).items()})
for k,v in d.get("other_workloads",{}).items():
    print("   ", k, round(v["ms_per_step"]*1e3,2), "us", v["kernels"], (v.get("roofline") or {}).get("frac"))
for f in sorted(glob.glob("gpurun_out/r03e/*_cur.json")+glob.glob("gpurun_out/r03e/*_packed.json")+glob.glob("gpurun_out/r03e/tree_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), "us", d["kernels"], (d.get("roofline") or {}).get("frac"))
    except Exception as ex:
        print(f, "ERR", ex)
P
