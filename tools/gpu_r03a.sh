#!/bin/bash
# GPU box, round 3 call A: parity suite (default + world-sphere path forced), A/B of -fno-slp-vectorize, static workloads.
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" >> $O/summary.txt
MI_TEST_SPHERE_PATH=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_visibility_ext.py tests/test_gpu_cluster.py tests/test_gpu_sphere_path.py -m gpu -x -q > $O/pytest_sphere2.log 2>&1; echo "pytest sphere2 rc=$?" >> $O/summary.txt
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/bench_noslp.json 2> $O/bench_noslp.err; echo "bench noslp rc=$?" >> $O/summary.txt
MI_LIB_VARIANT=slp timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/bench_slp.json 2> $O/bench_slp.err; echo "bench slp rc=$?" >> $O/summary.txt
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/bench_noslp2.json 2> $O/bench_noslp2.err; echo "bench noslp2 rc=$?" >> $O/summary.txt
tail -3 $O/pytest_default.log $O/pytest_sphere2.log
cat $O/summary.txt
python - <<'P'
import json
for f in ("bench_noslp","bench_slp","bench_noslp2"):
    try:
        d=json.loads(open(f"gpurun_out/r03a/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "frame", d["ms_per_step"]*1e3, "us", d["kernels"])
    for k,v in d.get("other_workloads",{}).items():
        print("   ", k, round(v["ms_per_step"]*1e3,2), "us", v["kernels"], (v.get("roofline") or {}).get("frac"))
P
