#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03t
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tree_frame.py tests/test_gpu_round3.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
timeout 120 $B --workload tree > $O/tree.json 2> $O/tree.err
timeout 120 $B --workload tree --tree-cull > $O/tree_frame.json 2> $O/tree_frame.err
timeout 120 $B --workload tree --tree-cull --tree-cull-launches 2 > $O/tree_frame_2l.json 2> $O/tree_frame_2l.err
timeout 120 $B --workload tree --tree-cull --views 4 > $O/tree_frame_4v.json 2> $O/tree_frame_4v.err
cat $O/summary.txt; tail -n 15 $O/pytest.log
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03t/*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
