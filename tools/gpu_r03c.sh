#!/bin/bash
# GPU box, round 3 call C: the fused host layer (tests + bench through it), the new entry points, static 10M x 4 with / without the sphere column
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 300 ./tests/cpp/host_systems_test > $O/host_tests.log 2>&1; echo "host tests rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_sphere_path.py tests/test_gpu_cluster.py tests/test_cpp_host.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $O/summary.txt
timeout 600 ./tests/cpp/host_systems_test --bench 1000000 8 > $O/host_bench.json 2> $O/host_bench.err; echo "host bench rc=$?" >> $O/summary.txt
for sp in 0 1; do
  timeout 200 python bench.py --workload flat_static --entities 10000000 --views 4 --sphere-path $sp --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/static_10m4_sp$sp.json 2> $O/static_10m4_sp$sp.err
  MI_LIB_VARIANT=slp timeout 200 python bench.py --workload flat_static --entities 10000000 --views 4 --sphere-path $sp --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/static_10m4_sp${sp}_slp.json 2> $O/static_10m4_sp${sp}_slp.err
done
MI_LIB_VARIANT=slp timeout 200 python bench.py --workload flat_static --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/static_1m_slp.json 2> $O/static_1m_slp.err
timeout 200 python bench.py --workload flat_static --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end --no-other-workloads > $O/static_1m.json 2> $O/static_1m.err
tail -n 40 $O/host_tests.log
tail -n 15 $O/pytest_new.log
cat $O/summary.txt $O/host_bench.json
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03c/static*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), "us", d["kernels"], (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
P
