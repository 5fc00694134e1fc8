set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r01_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r01_pytest_gpu.log
tail -30 gpurun_out/r01_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r01_smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/r01_smoke.log
timeout 300 python bench.py > gpurun_out/r01_bench_flat.json 2> gpurun_out/r01_bench_flat.err; echo "bench rc=$?"; cat gpurun_out/r01_bench_flat.json; tail -5 gpurun_out/r01_bench_flat.err
timeout 300 python bench.py --workload tree --steps 100 > gpurun_out/r01_bench_tree.json 2> gpurun_out/r01_bench_tree.err; cat gpurun_out/r01_bench_tree.json; tail -5 gpurun_out/r01_bench_tree.err
timeout 300 python bench.py --workload lights --steps 100 > gpurun_out/r01_bench_lights.json 2> gpurun_out/r01_bench_lights.err; cat gpurun_out/r01_bench_lights.json; tail -5 gpurun_out/r01_bench_lights.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_flat -o flat -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r01_rocprof_flat.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof_flat | head -30
