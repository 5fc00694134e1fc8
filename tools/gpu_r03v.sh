#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_sphere_path.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
MI_TEST_WALK_INROW=1 timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > $O/pytest_riders.log 2>&1; echo "pytest riders rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
timeout 120 $B --workload frame > $O/frame.json 2> $O/frame.err
timeout 120 $B --workload frame --walk-inrow 1 > $O/frame_riders.json 2> $O/frame_riders.err
timeout 120 $B --workload flat --entities 1110000 > $O/flat1110k.json 2> $O/flat1110k.err
MI_LIB_VARIANT=timeline timeout 200 python tools/exp_timeline.py > $O/timeline.json 2> $O/timeline.err
cat $O/summary.txt; tail -n 4 $O/pytest.log; tail -n 4 $O/pytest_riders.log; cat $O/timeline.json | cut -c1-900
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03v/f*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
