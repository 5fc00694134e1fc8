#!/bin/bash
# GPU box: SQ counters of the flat frame kernel (k_frame) under each MI_MULTI_VIEW setting.   bash tools/flat_pmc.sh <outdir> "<bench.py arguments>" "<MI_MULTI_VIEW values>"
export TMPDIR=/tmp
O=${1:-gpurun_out/flat_pmc}
ARGS=${2:---workload flat --entities 10000000 --views 4}
mkdir -p $O
COMMON="--steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic"
for mv in ${3:-0 1}; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    MI_MULTI_VIEW=$mv timeout -k 5 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/mv${mv}_$i -o t -- python bench.py $ARGS $COMMON > $O/mv${mv}_$i.log 2>&1
  done
done
python - "$O" <<'P'
import csv, glob, collections, os, sys
for d in sorted(glob.glob(sys.argv[1] + "/mv*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_frame" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d), {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
P
