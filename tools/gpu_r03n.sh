#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03r
mkdir -p $O
MI_LIB_VARIANT=timeline timeout 200 python tools/exp_timeline.py > $O/timeline.json 2> $O/timeline.err
timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
timeout 120 $B --workload frame > $O/frame.json 2> $O/frame.err
timeout 120 $B --workload lights > $O/lights.json 2> $O/lights.err
timeout 120 $B --workload flat --entities 1110000 > $O/flat1110k.json 2> $O/flat1110k.err
cat $O/timeline.json $O/summary.txt; tail -n 2 $O/pytest.log
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03r/[fl]*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
