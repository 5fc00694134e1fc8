mkdir -p gpurun_out; : > gpurun_out/tree.txt
for rep in 1 2; do for w in "" 1; do
for e in 1000000 1398101; do
MI_EXP_W8=$w; if [ -z "$w" ]; then unset MI_EXP_W8; else export MI_EXP_W8; fi
python bench.py --workload tree --entities $e --steps 100 --blocks 8 --no-cpu-baseline --no-other-workloads --no-end-to-end 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('w8=$w', $e, d['ms_per_step'], d['kernels'], d['roofline']['frac'])" >> gpurun_out/tree.txt
done; done; done
unset MI_EXP_W8
timeout 300 python -m pytest tests/test_gpu_tile_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error" >> gpurun_out/tree.txt
MI_EXP_W8=1 timeout 300 python -m pytest tests/test_gpu_tile_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error" >> gpurun_out/tree.txt
