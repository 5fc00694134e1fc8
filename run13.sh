export TMPDIR=/tmp
mkdir -p gpurun_out/r13
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r13/pytest.log 2>&1; tail -15 gpurun_out/r13/pytest.log
timeout 600 python bench.py > gpurun_out/r13/bench.json 2> gpurun_out/r13/bench.err; cat gpurun_out/r13/bench.json; tail -3 gpurun_out/r13/bench.err
