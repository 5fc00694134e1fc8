#!/bin/bash
# GPU box, round 3 call J: where the metric frame's time goes now that the rows read 29 B less -- the cluster walk riding in the
# launch, on a stream of its own, as calls of its own; rows alone
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end"
timeout 120 $B --workload frame > $O/frame_riding.json 2> $O/frame_riding.err
timeout 120 $B --workload frame --concurrent-clusters > $O/frame_concurrent.json 2> $O/frame_concurrent.err
timeout 120 $B --workload frame --separate-cluster-calls > $O/frame_separate.json 2> $O/frame_separate.err
timeout 120 $B --workload flat --entities 1110000 > $O/flat_1110k.json 2> $O/flat_1110k.err
timeout 120 $B --workload lights > $O/lights.json 2> $O/lights.err
timeout 120 $B --workload frame --profile-all > $O/frame_profile_all.json 2> $O/frame_profile_all.err
python - <<'PY'
import json,glob,os
for p in sorted(glob.glob('gpurun_out/r03j/*.json')):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(p), d['ms_per_step'], r['avg_kernel_us'], d.get('kernels'))
    except Exception as e: print(p, 'ERR', e, open(p.replace('.json','.err')).read()[-300:])
PY
