#!/bin/bash
# Is the differential test sensitive?  A build with a deliberate one-row-in-4096 error in the row-summary path must fail it.
export TMPDIR=/tmp
O=gpurun_out/mutant
mkdir -p $O
MI_LIB_VARIANT=mutant timeout 600 python -m pytest tests/test_gpu_differential.py tests/test_gpu_row_summary.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest on the mutant rc=$? (must be non-zero)" > $O/summary.txt
cat $O/summary.txt; grep -E 'passed|failed' $O/pytest.log | tail -n 3; grep -m3 'differs\|mismatches' $O/pytest.log | cut -c1-300
