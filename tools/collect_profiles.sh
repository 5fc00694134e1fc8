#!/bin/bash
# bash tools/collect_profiles.sh <tag>: what tools/final_run.sh <tag> left under gpurun_out/prof_<tag>/ (scratch) -> profiles/<tag>/ (committed):
# the summarised rocprofv3 evidence, the test logs, the bench line, the tables -- and profiles/rocprof_summary.json (what bench.py replays).
TAG=$1
P=gpurun_out/prof_$TAG
D=profiles/$TAG
mkdir -p $D
cp -r $P/profiles_$TAG/. $D/
for f in device.txt gputest.log gputest_other_paths.log gputest_strips_forced.log strip_traces.txt strips_ab.md smoke.log host_systems_test.log host_visibility_test.log multi_gpu_single_process.json shapes_table.md \
         sharded_1rank.md bench_line.json bench_full.json bench_stdout.txt tree_frame_sq_counters.txt tree_sq_counters.txt; do
  [ -f $P/$f ] && cp $P/$f $D/
done
[ -f $P/sharded_1rank/default_calibrated.json ] && cp $P/sharded_1rank/default_calibrated.json $D/sharded_1rank.json
[ -f $P/rocprof_summary.json ] && cp $P/rocprof_summary.json profiles/rocprof_summary.json
python tools/kernel_resources.py $D/kernel_resources.md > /dev/null
du -sh $D; ls $D | wc -l
