#!/bin/bash
# GPU box: SQ / TCC counters of the tree kernel per tile mode.   bash tools/tree_pmc.sh "<modes>" [extra bench.py arguments, e.g. --tree-cull] [SQ = the two SQ sets only]
export TMPDIR=/tmp
O=gpurun_out/tree_pmc
mkdir -p $O
COMMON="--workload tree --steps 10 --warmup 2 --blocks 2 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic"
EXTRA=$2
for m in $1; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    if [ "$3" = "SQ" ] && [ $i -gt 2 ]; then continue; fi
    timeout -k 5 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/m${m}_$i -o t -- python bench.py $COMMON --tile-mode $m $EXTRA > $O/m${m}_$i.log 2>&1
  done
done
python - <<'P'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/tree_pmc/m*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "propagate" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d), {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
P
