#!/bin/bash
# GPU box, round 3 call F: whole suite (default; tile pre-test + sphere path forced; tiled sorted phases forced), benches
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" >> $O/summary.txt
MI_TEST_TILE_PRETEST=2 MI_TEST_SPHERE_PATH=2 MI_TEST_SORTED_TILED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_tile_kernels.py tests/test_gpu_sphere_path.py tests/test_gpu_cluster.py tests/test_gpu_batching.py -m gpu -q > $O/pytest_forced.log 2>&1; echo "pytest forced rc=$?" >> $O/summary.txt
timeout 900 python bench.py --steps 100 --warmup 20 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_default.log; tail -n 6 $O/pytest_forced.log
cat $O/summary.txt
python - <<'P'
import json,glob
d=json.loads(open("gpurun_out/r03g/bench_full.json").read().strip().splitlines()[-1])
print("frame", d["ms_per_step"]*1e3, "us", d["kernels"], d["roofline"]["frac"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["thread_sweep_ms_per_frame"])
e=d["end_to_end"]
for k in ("1pct_dirty","10pct_dirty","100pct_dirty"):
    print(k, e[k]["us_per_frame"], e[k]["pcie_frac"], e[k]["stage_us"])
h=d["end_to_end_host_layer"]
for form in ("three_systems","fused_frame"):
    print(form, {k: {kk: vv for kk,vv in v.items() if kk.endswith("us") or kk=="us_per_frame" or kk=="device_waits"} for k,v in h.get(form,{}).items()})
for k,v in d.get("other_workloads",{}).items():
    print("   ", k, round(v["ms_per_step"]*1e3,2), "us", v["kernels"], (v.get("roofline") or {}).get("frac"))
P
