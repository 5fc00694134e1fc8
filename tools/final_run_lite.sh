#!/bin/bash
# A short form of tools/final_run.sh for a late change that touches few workloads:   bash tools/final_run_lite.sh <tag> <workload ...>
#   GPU tests, smoke, the rocprofv3 evidence of the named workloads only, the default bench line.
TAG=$1; shift
export TMPDIR=/tmp
P=gpurun_out/prof_$TAG
mkdir -p $P
rocminfo | grep -m1 gfx > $P/device.txt
timeout 900 python -m pytest tests -m gpu -q > $P/gputest.log 2>&1; echo "gpu tests rc=$?" >> $P/gputest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.log 2>&1; echo "smoke rc=$?" >> $P/smoke.log
bash tools/profile.sh $TAG "$@" > $P/profile.log 2>&1
timeout 1200 python bench.py > $P/bench_stdout.txt 2> $P/bench.err; echo "bench rc=$?" >> $P/bench.err
tail -n 1 $P/bench_stdout.txt > $P/bench_line.json; cp bench_full.json $P/bench_full.json 2>/dev/null
cp -r profiles/$TAG $P/profiles_$TAG 2>/dev/null
cp profiles/rocprof_summary.json $P/rocprof_summary.json 2>/dev/null
grep -E "passed|failed|rc=" $P/gputest.log $P/smoke.log $P/bench.err | tail -6
