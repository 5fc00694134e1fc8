#!/bin/bash
# GPU box, round 3 call D: host layer + new entry points, whole bench line, SQ counters of the sphere kernel at 10M x 4 views
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
timeout 300 ./tests/cpp/host_systems_test > $O/host_tests.log 2>&1; echo "host tests rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_cpp_host.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $O/summary.txt
timeout 900 python bench.py --steps 100 --warmup 20 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" >> $O/summary.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -o t -- python bench.py --workload flat_static --entities 10000000 --views 4 --steps 10 --warmup 2 --blocks 2 --no-cpu-baseline --no-other-workloads --no-end-to-end > $O/pmc_$i.log 2>&1
done
python - <<'P'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/r03d/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_frame" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d), {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
P
tail -n 5 $O/host_tests.log
tail -n 8 $O/pytest_new.log
cat $O/summary.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03d/bench_full.json").read().strip().splitlines()[-1])
print("frame", d["ms_per_step"]*1e3, "us", d["kernels"], d["roofline"]["frac"])
print("cpu_baseline", d["cpu_baseline"])
print("e2e", json.dumps(d["end_to_end"], indent=0)[:3000])
print("host", json.dumps(d["end_to_end_host_layer"], indent=0)[:3000])
for k,v in d.get("other_workloads",{}).items():
    print("   ", k, round(v["ms_per_step"]*1e3,2), "us", v["kernels"], (v.get("roofline") or {}).get("frac"))
P
