#!/bin/bash
# quick GPU check of a change: the suites named in $1 (default: all) + the bench workloads named in the rest
export TMPDIR=/tmp
O=gpurun_out/quick
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest ${QUICK_TESTS:-tests} -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic"
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 150 $B $line > $O/b$i.json 2> $O/b$i.err
  echo "$line" > $O/b$i.cmd
done <<< "$QUICK_BENCH"
cat $O/summary.txt; grep -E 'passed|failed' $O/pytest.log | tail -n 2; grep -m5 'Error\|assert' $O/pytest.log | cut -c1-250
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/quick/b*.json')):
    cmd=open(p.replace('.json','.cmd')).read().strip()
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(f"{cmd:70s} {d['ms_per_step']*1e3:8.2f} us  kernel {r['avg_kernel_us']}")
    except Exception as e: print(cmd, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
