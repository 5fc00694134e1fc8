#!/bin/bash
# GPU box, round 3 call H: rocprofv3 evidence of the round's final state (kernel traces + FETCH_SIZE / WRITE_SIZE passes), the
# single-process multi-GPU driver, the bench line
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 200 ./tests/cpp/multi_gpu_single_process 1000000 6 > $O/multi_gpu.json 2> $O/multi_gpu.err; echo "multi gpu rc=$?" >> $O/summary.txt
timeout 900 python bench.py --steps 100 --warmup 20 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" >> $O/summary.txt
bash tools/profile.sh r03h > $O/profile.log 2>&1; echo "profile rc=$?" >> $O/summary.txt
cp -r profiles/r03h $O/profiles_r03h 2>/dev/null
cp profiles/rocprof_summary.json $O/rocprof_summary.json 2>/dev/null
cat $O/summary.txt $O/multi_gpu.json
tail -n 60 $O/profile.log
