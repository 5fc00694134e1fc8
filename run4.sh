set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6/pytest.log 2>&1; tail -15 gpurun_out/r6/pytest.log

timeout 300 python bench.py --no-cpu-baseline --workload tree --steps 100 > gpurun_out/r6/bench_tree.json 2> gpurun_out/r6/bench_tree.err; cat gpurun_out/r6/bench_tree.json; tail -3 gpurun_out/r6/bench_tree.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/tree -o tree -- python bench.py --workload tree --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r6/rocprof_tree.log 2>&1
cat gpurun_out/r6/tree/tree_kernel_stats.csv
