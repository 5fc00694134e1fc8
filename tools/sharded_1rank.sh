#!/bin/bash
# bash tools/sharded_1rank.sh <outdir>: the sharded frame of configs[3] at the N = 8 shard size (1.25 M rows x 4 views) on ONE GPU with a 1-rank
# RCCL communicator (MI_FORCE_DIST=1): everything of the N > 1 path but the wire.  One bench line per exchange mode.
OUT=${1:-gpurun_out/sharded_1rank}
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MI_FORCE_DIST=1
COMMON="--workload sharded --entities 1250000 --views 4 --steps 50 --warmup 10 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic"
python bench.py $COMMON 2> $OUT/default_calibrated.err | tail -n 1 > $OUT/default_calibrated.json
cp bench_full.json $OUT/default_calibrated_full.json
MI_XCH_MODE=simple python bench.py $COMMON 2> $OUT/simple_async.err | tail -n 1 > $OUT/simple_async.json
MI_XCH_MODE=simple MI_XCH_SYNC_ENQUEUE=1 python bench.py $COMMON 2> $OUT/simple_sync.err | tail -n 1 > $OUT/simple_sync.json
MI_XCH_MODE=pipelined python bench.py $COMMON 2> $OUT/pipelined.err | tail -n 1 > $OUT/pipelined.json
unset MI_FORCE_DIST
python bench.py $COMMON 2> $OUT/no_exchange.err | tail -n 1 > $OUT/no_exchange.json
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
print("| mode | us / frame | host enqueue us / frame | of which waiting for the device (back-pressure) | host busy us / frame | kernel us | gathered masks == single-GPU masks | all-gather alone us | exchange_mode |")
print("|---|---|---|---|---|---|---|---|---|")
for name in ("no_exchange", "simple_sync", "simple_async", "pipelined", "default_calibrated"):
    try:
        d = json.load(open(os.path.join(out, name + ".json")))
    except Exception as e:
        print("|", name, "| unreadable:", e, "|")
        continue
    bp, busy = d.get('host_backpressure_ms_per_step'), d.get('host_busy_ms_per_step')
    print(f"| {name} | {1e3 * d['ms_per_step']:.2f} | {1e3 * d.get('host_enqueue_ms_per_step', float('nan')):.2f} | {'-' if bp is None else round(1e3 * bp, 2)} | "
          f"{'-' if busy is None else round(1e3 * busy, 2)} | {d['roofline']['avg_kernel_us']} | {d.get('gathered_masks_match_single_gpu')} | {d.get('all_gather_us')} | "
          f"{d['config'].get('exchange_mode')} ranks={d['config'].get('rccl_ranks')} |")
PY
python -c "
import json,sys
d=json.load(open('$OUT/default_calibrated_full.json'))
print('calibration:', json.dumps(d['config'].get('exchange_calibration')))"
