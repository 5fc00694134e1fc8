#!/bin/bash
# One call on the GPU box for a state of the code:   bash tools/final_run.sh <tag>
#   GPU tests (+ the same with the internal fast paths forced the other way), smoke(), the rocprofv3 evidence (tools/profile.sh),
#   the single-process multi-GPU driver, the default bench line.  Everything lands under gpurun_out/prof_<tag>/ (merged back).
TAG=$1
export TMPDIR=/tmp
P=gpurun_out/prof_$TAG
mkdir -p $P
rocminfo | grep -m1 gfx > $P/device.txt
timeout 900 python -m pytest tests -m gpu -q > $P/gputest.log 2>&1; echo "gpu tests rc=$?" >> $P/gputest.log
MI_TEST_ROW_SUMMARY=1 MI_TEST_WALK_INROW=1 MI_TEST_TREE_CULL=2 MI_TEST_SPHERE_PATH=2 MI_TEST_TILE_PRETEST=2 MI_TEST_CHUNKED_FRAMES=2 MI_TEST_STATIC_CULL_ORDER=2 MI_TEST_TILE_MODE=1 timeout 900 python -m pytest tests -m gpu -q \
    --deselect tests/test_gpu_differential.py > $P/gputest_other_paths.log 2>&1; echo "gpu tests (other paths) rc=$?" >> $P/gputest_other_paths.log
MI_TEST_TILE_MODE=5 MI_TEST_TILE_PRETEST=2 timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_differential.py > $P/gputest_strips_forced.log 2>&1; echo "gpu tests (strips wherever they can be planned) rc=$?" >> $P/gputest_strips_forced.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.log 2>&1; echo "smoke rc=$?" >> $P/smoke.log
timeout 200 ./tests/cpp/multi_gpu_single_process 1000000 6 > $P/multi_gpu_single_process.json 2> $P/multi_gpu.err; echo "multi gpu rc=$?" >> $P/multi_gpu.err
timeout 600 ./tests/cpp/host_systems_test > $P/host_systems_test.log 2>&1; echo "host systems rc=$?" >> $P/host_systems_test.log
timeout 600 ./tests/cpp/host_visibility_test > $P/host_visibility_test.log 2>&1; echo "host visibility rc=$?" >> $P/host_visibility_test.log
bash tools/shape_sweep.sh $P/shapes > $P/shapes_table.md 2>&1
for s in large_tree deep_tree update_leaves update_shallow; do python tools/strip_trace.py $s 0; done > $P/strip_traces.txt 2>&1
SHAPES="large_tree deep_tree update_leaves update_shallow humanoids_active wide_tree" WIDTHS="64 128" bash tools/strips_ab.sh $P/strips_ab > $P/strips_ab.md 2>&1
bash tools/sharded_1rank.sh $P/sharded_1rank > $P/sharded_1rank.md 2>&1
bash tools/profile.sh $TAG > $P/profile.log 2>&1
timeout 1200 python bench.py > $P/bench_stdout.txt 2> $P/bench.err; echo "bench rc=$?" >> $P/bench.err
tail -n 1 $P/bench_stdout.txt > $P/bench_line.json; cp bench_full.json $P/bench_full.json 2>/dev/null
cp -r profiles/$TAG $P/profiles_$TAG 2>/dev/null
cp profiles/rocprof_summary.json $P/rocprof_summary.json 2>/dev/null
grep -E "passed|failed|rc=" $P/gputest.log $P/gputest_other_paths.log $P/gputest_strips_forced.log $P/smoke.log $P/multi_gpu.err $P/host_systems_test.log $P/host_visibility_test.log $P/bench.err | tail -12
