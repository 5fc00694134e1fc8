#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
MI_LIB_VARIANT=timeline timeout 200 python tools/exp_timeline.py > $O/timeline.json 2> $O/timeline.err
MI_LIB_VARIANT=timeline timeout 200 python tools/exp_timeline.py --row-summary 1 > $O/timeline_plain.json 2> $O/timeline_plain.err
cat $O/timeline.json $O/timeline_plain.json; tail -n 5 $O/timeline.err
