#!/bin/bash
# GPU box, round 3 call I: the row summary (RowSummary) -- parity with it on and off, A/B of the frames that use it; the dense
# OBB stage of k_frame_sph
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
rocminfo | grep -m1 gfx > $O/device.txt
timeout 600 python -m pytest tests/test_gpu_row_summary.py tests/test_gpu_sphere_path.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
for rs in 0 1; do
  timeout 120 $B --workload frame --row-summary $rs > $O/frame_rs$rs.json 2> $O/frame_rs$rs.err
  timeout 120 $B --workload flat --row-summary $rs > $O/flat_rs$rs.json 2> $O/flat_rs$rs.err
  timeout 120 $B --workload flat --entities 10000000 --views 4 --row-summary $rs > $O/flat_10m4_rs$rs.json 2> $O/flat_10m4_rs$rs.err
  timeout 120 $B --workload flat_static --row-summary $rs > $O/static_rs$rs.json 2> $O/static_rs$rs.err
  timeout 120 $B --workload flat_static --entities 10000000 --views 4 --row-summary $rs > $O/static_10m4_rs$rs.json 2> $O/static_10m4_rs$rs.err
done
echo "bench rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" >> $O/summary.txt
MI_TEST_ROW_SUMMARY=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > $O/pytest_plain.log 2>&1; echo "pytest plain columns rc=$?" >> $O/summary.txt
cat $O/summary.txt
tail -n 5 $O/pytest_new.log $O/pytest_default.log $O/pytest_plain.log
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03i/*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], r['frac'], r.get('frac_of_layout_bytes'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
