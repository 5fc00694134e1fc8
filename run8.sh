export TMPDIR=/tmp
mkdir -p gpurun_out/r12
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r12/pytest.log 2>&1; tail -15 gpurun_out/r12/pytest.log
timeout 300 python bench.py --no-cpu-baseline --workload lights --steps 100 --profile-all > gpurun_out/r12/bench_lights.json 2> gpurun_out/r12/bench_lights.err; cat gpurun_out/r12/bench_lights.json; tail -3 gpurun_out/r12/bench_lights.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r12/lights -o lights -- python bench.py --workload lights --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r12/rocprof.log 2>&1
cat gpurun_out/r12/lights/lights_kernel_stats.csv
