#!/bin/bash
# Collects the rocprofv3 evidence for one state of the code on the GPU box:   bash tools/profile.sh <tag> [workload ...]
#   - kernel-trace + stats of the bench command for each workload (CSV)
#   - separate PMC passes (FETCH_SIZE, WRITE_SIZE) for the streaming kernels, as MI355X_MICROARCH.md prescribes
#     (never combined with the trace domains gpurun refuses)
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); tools/summarize_profiles.py turns it into profiles/<tag>/ and
# profiles/rocprof_summary.json (what bench.py quotes as rocprof_avg_kernel_us / traffic).
TAG=${1:-r03}
shift
WLS=${@:-frame frame_plain_columns flat flat_plain_columns flat_10m_1view flat_10m_4views tree tree_subtree tree_leaves tree_frame tree_frame_two_launches lights flat_static flat_static_no_sphere flat_static_10m_4views flat_static_10m_4views_no_cull_order flat_static_4m_4views tree_by_levels batching batching_sorted_1k batching_sorted_4k batching_sorted_64k batching_sorted_1m flat_1250k_4views tree_shape_chain tree_shape_humanoids_mixed tree_shape_humanoids_active tree_shape_deep_tree tree_shape_large_tree tree_shape_update_leaves tree_shape_tree_4ary_depth12}
export TMPDIR=/tmp
P=gpurun_out/prof_$TAG
mkdir -p $P
COMMON="--no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic"
for wl in $WLS; do
  case $wl in
    frame_plain_columns) ARGS="--workload frame --row-summary 1" ;;
    flat_plain_columns)  ARGS="--workload flat --row-summary 1" ;;
    flat_10m_1view)  ARGS="--workload flat --entities 10000000 --views 1" ;;
    flat_10m_4views) ARGS="--workload flat --entities 10000000 --views 4" ;;
    tree_subtree)    ARGS="--workload tree --tree-moved subtree" ;;
    tree_leaves)     ARGS="--workload tree --tree-moved leaves" ;;
    tree_frame)      ARGS="--workload tree --tree-cull" ;;
    tree_frame_two_launches) ARGS="--workload tree --tree-cull --tree-cull-launches 2" ;;
    flat_static_no_sphere)  ARGS="--workload flat_static --sphere-path 1" ;;
    flat_static_10m_4views) ARGS="--workload flat_static --entities 10000000 --views 4" ;;
    flat_static_10m_4views_no_cull_order) ARGS="--workload flat_static --entities 10000000 --views 4 --static-cull-order 1" ;;
    flat_static_4m_4views)  ARGS="--workload flat_static --entities 4000000 --views 4" ;;
    tree_by_levels)         ARGS="--workload tree --tile-mode 1" ;;
    flat_1250k_4views)      ARGS="--workload flat --entities 1250000 --views 4" ;;  # the N = 8 shard of configs[3]
    tree_shape_*)           ARGS="--workload tree --tree-shape ${wl#tree_shape_}" ;;  # transform_hierarchy.rs's shapes
    batching_sorted_1k)     ARGS="--workload batching_sorted --sorted-items 1024" ;;
    batching_sorted_4k)     ARGS="--workload batching_sorted --sorted-items 4096" ;;
    batching_sorted_64k)    ARGS="--workload batching_sorted --sorted-items 65536" ;;
    batching_sorted_1m)     ARGS="--workload batching_sorted --sorted-items 1000000" ;;
    *)               ARGS="--workload $wl" ;;
  esac
  echo "python bench.py $ARGS --steps 50 --warmup 10 --blocks 4 $COMMON" > $P/$wl.cmd
  timeout -k 5 180 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$wl -o $wl -- \
      python bench.py $ARGS --steps 50 --warmup 10 --blocks 4 $COMMON > $P/$wl.log 2>&1
  case $wl in batching|batching_sorted_1k|batching_sorted_4k|batching_sorted_64k|lights|tree_shape_chain|tree_shape_humanoids*|tree_shape_tree_4ary*) continue ;; esac  # (PMC passes for the streaming kernels and the strips)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 180 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $P/${wl}_$ctr -o $wl -- \
        python bench.py $ARGS --steps 10 --warmup 2 --blocks 2 $COMMON > $P/${wl}_$ctr.log 2>&1
  done
done
# the fused hierarchy frame is bound by vector-instruction issue: its SQ counters next to the plain tile launch's (tools/tree_pmc.sh)
case " $WLS " in *" tree_frame "*)
bash tools/tree_pmc.sh 0 "--tree-cull" SQ > $P/tree_frame_sq_counters.txt 2>&1; rm -rf $P/tree_frame_sq; mv gpurun_out/tree_pmc $P/tree_frame_sq
bash tools/tree_pmc.sh 0 "" SQ > $P/tree_sq_counters.txt 2>&1; rm -rf $P/tree_sq; mv gpurun_out/tree_pmc $P/tree_sq ;;
esac
python tools/summarize_profiles.py $TAG $WLS
