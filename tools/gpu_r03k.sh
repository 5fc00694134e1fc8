#!/bin/bash
# GPU box, round 3 call K: what the riding walk costs the rows -- occupancy (rows capped at 5 waves per SIMD through LDS), bigger
# walk chunks (31 KB arena), the 63-VGPR walk (7 waves)
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
for v in "" occ5 nohoist; do
  export MI_LIB_VARIANT=$v
  timeout 120 $B --workload frame > $O/frame_$v.json 2> $O/frame_$v.err
  timeout 120 $B --workload flat --entities 1110000 > $O/flat1110k_$v.json 2> $O/flat1110k_$v.err
  timeout 120 $B --workload flat > $O/flat_$v.json 2> $O/flat_$v.err
  timeout 120 $B --workload flat --entities 10000000 --views 4 > $O/flat10m4_$v.json 2> $O/flat10m4_$v.err
done
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03k/*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
