#!/bin/bash
# Are the tests sensitive?  A build with a deliberate one-row-in-4096 error in the row-summary path and in the GlobalTransforms that
# travel ahead of the frame (k_globals_ahead, the scatter launch's write-back) must fail them.
export TMPDIR=/tmp
O=gpurun_out/mutant
mkdir -p $O
MI_LIB_VARIANT=mutant timeout 600 python -m pytest tests/test_gpu_differential.py tests/test_gpu_row_summary.py tests/test_gpu_chunked_frames.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest on the mutant rc=$? (must be non-zero)" > $O/summary.txt
cat $O/summary.txt; grep -E 'passed|failed' $O/pytest.log | tail -n 3; grep -c '^FAILED tests/test_gpu_chunked_frames' $O/pytest.log; grep -m3 'differs\|mismatches' $O/pytest.log | cut -c1-300
