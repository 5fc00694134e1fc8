mkdir -p gpurun_out; : > gpurun_out/tree.txt
for rep in 1 2; do
for e in 1000000 1398101 87381; do
python bench.py --workload tree --entities $e --steps 100 --blocks 8 --no-cpu-baseline --no-other-workloads --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($e, d['ms_per_step'], d['kernels'], d['roofline']['frac'])" >> gpurun_out/tree.txt
done; done
timeout 300 python -m pytest tests/test_gpu_tile_kernels.py tests/test_gpu_round2.py -x -q 2>&1 | grep -E "passed|failed|Error" >> gpurun_out/tree.txt
