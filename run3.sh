set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest.log 2>&1; tail -15 gpurun_out/r3/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_flat.json 2> gpurun_out/r3/bench_flat.err; cat gpurun_out/r3/bench_flat.json; tail -3 gpurun_out/r3/bench_flat.err
timeout 300 python bench.py --no-cpu-baseline --unfused > gpurun_out/r3/bench_unfused.json 2> gpurun_out/r3/bench_unfused.err; cat gpurun_out/r3/bench_unfused.json
timeout 300 python bench.py --no-cpu-baseline --views 4 > gpurun_out/r3/bench_flat4.json 2> gpurun_out/r3/bench_flat4.err; cat gpurun_out/r3/bench_flat4.json
timeout 300 python bench.py --no-cpu-baseline --entities 10000000 --views 4 --steps 50 > gpurun_out/r3/bench_flat10m.json 2> gpurun_out/r3/bench_flat10m.err; cat gpurun_out/r3/bench_flat10m.json; tail -3 gpurun_out/r3/bench_flat10m.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/flat -o flat -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3/rocprof_flat.log 2>&1
cat gpurun_out/r3/flat/flat_kernel_stats.csv
