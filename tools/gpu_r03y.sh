#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tree_frame.py tests/test_gpu_differential.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_sphere_path.py tests/test_gpu_row_summary.py tests/test_gpu_visibility_ext.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
timeout 120 $B --workload tree --tree-cull --tree-cull-launches 2 > $O/tree_frame_2l.json 2> $O/tree_frame_2l.err
timeout 120 $B --workload tree --tree-cull --tree-cull-launches 1 > $O/tree_frame_fused.json 2> $O/tree_frame_fused.err
timeout 120 $B --workload tree --tree-cull --tree-cull-launches 1 --views 4 > $O/tree_frame_fused_4v.json 2> $O/tree_frame_fused_4v.err
timeout 120 $B --workload tree --tree-cull --views 4 --tree-cull-launches 2 > $O/tree_frame_2l_4v.json 2> $O/tree_frame_2l_4v.err
cat $O/summary.txt; tail -n 6 $O/pytest.log | cut -c1-300
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03y/*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
