#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:   bash tools/profile.sh <tag>
#   - kernel-trace + stats of the bench command for each workload (CSV)
#   - separate PMC passes (FETCH_SIZE, WRITE_SIZE) for the dominant kernels, as MI355X_MICROARCH.md prescribes
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); tools/summarize_profiles.py turns it into profiles/<tag>/.
TAG=${1:-r01}
export TMPDIR=/tmp
P=gpurun_out/prof_$TAG
mkdir -p $P
for wl in flat tree lights flat_static batching; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$wl -o $wl -- \
      python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads > $P/$wl.log 2>&1
  if [ $wl = flat_static ] || [ $wl = batching ]; then continue; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $P/${wl}_$ctr -o $wl -- \
        python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > $P/${wl}_$ctr.log 2>&1
  done
done
python tools/summarize_profiles.py $TAG
