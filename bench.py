#!/usr/bin/env python
"""bench.py -- throughput of the render-prep hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path over device-resident columns with every Transform dirty.

N = 1 (default workload `frame`) is BASELINE.json's metric as worded -- "entities/sec through propagate+cull+cluster at 1M
entities" -- in ONE context on ONE stream: configs[1]'s 1 000 000 many_cubes entities plus configs[2]'s many_lights set
(10 000 meshes + 100 000 point lights, which are rows like everything else), 1 camera:
    mi_cluster_upload_view     this frame's camera (host constants; the view-space planes are cached on the device)
    mi_propagate_and_cull      sync_simple_transforms + reset_view_visibility + check_visibility_cpu_culling (meshes through
      (MI_CULL_WITH_CLUSTERS)  their Aabb, lights through their bounding Sphere) + mark_newly_hidden + VisibleEntities lists,
                               then the gather of the visible lights + assign_objects_to_clusters on 16x9x24 clusters
`value` counts the 1 000 000 entities of the metric's name only (the 110 000 rows of configs[2] are processed, not counted).

N > 1 (default workload `sharded`) is configs[3]: 10 000 000 entities x 4 camera frusta, STRONG scaling -- the row range is
split over the N ranks, every rank runs the fused frame kernel on its rows and the packed ViewVisibility bitmasks are
exchanged with ONE in-place RCCL all-gather per frame (--scaling weak: --entities rows per rank instead).  Rank 0 also times
the whole scene alone on its GPU (outside the timed region) and reports it as `single_gpu_same_workload`.

Timing: W warm-up steps, then B blocks of exactly K steps, each bracketed by barrier + synchronize pairs; per block the MAX
over ranks; `ms_per_step` / `value` come from the MEDIAN block (p10 / p90 / min / max under `blocks`).  B is chosen so that the
timed blocks cover ~0.4 s (at least 15).  Three further blocks run with per-dispatch HIP events on the kernels (`kernels`,
`roofline`) and are kept out of the statistics -- binding events to a dispatch fences it off from its neighbours.
Rank 0 prints ONE compact JSON line (< 4 KB: the contract's keys, benchlib/line.py) as the LAST line of stdout and writes everything
else it measured -- the other workloads, the CPU thread sweep, the end-to-end stages -- to bench_full.json (repo root and gpurun_out/).
DESIGN.md section 5 explains the byte accounting behind `roofline`.  The parts live in benchlib/: one module per workload
(wl_*.py), the timing loop (measure.py), the CPU baselines (cpu_baseline.py -- the only importer of the oracle), the PCIe-inclusive
block (end_to_end.py), the live PMC traffic passes (traffic.py) and the result line (line.py).
"""
import argparse
import os
import sys

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from benchlib import cpu_baseline as cpu, end_to_end as e2e, line as result_line, traffic  # noqa: E402
from benchlib.common import PROFILED_BLOCKS, with_args  # noqa: E402
from benchlib.measure import block_stats, measure, roofline_of  # noqa: E402
from benchlib.wl_batching import build_batching, build_batching_sorted  # noqa: E402
from benchlib.wl_flat import build_flat, build_flat_static  # noqa: E402
from benchlib.wl_frame import build_frame  # noqa: E402
from benchlib.wl_lights import build_lights  # noqa: E402
from benchlib.wl_tree import build_tree, build_tree_shape, reprice_per_frame, roofline_frame  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["auto", "frame", "sharded", "flat", "tree", "lights", "flat_static", "batching", "batching_sorted"], default="auto",
                    help="auto = frame at N=1 (the BASELINE metric), sharded (configs[3]) at N>1")
    ap.add_argument("--walk-inrow", type=int, default=0, choices=[0, 1], help="frame: 0 = the lights' row workgroups walk them (default), 1 = extra workgroups re-derive their visibility")
    ap.add_argument("--row-summary", type=int, default=0, choices=[0, 1],
                    help="0 = waves whose 64 rows agree in Aabb / flags / RenderLayers read the 32-byte summary (default), 1 = off")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="sharded workload: total rows fixed / rows per GPU fixed")
    ap.add_argument("--entities", type=int, default=0, help="frame/flat/tree: entities (default 1M); sharded: total (strong, default 10M) or per GPU (weak, default 1M)")
    ap.add_argument("--views", type=int, default=0, help="camera frusta (default 1; sharded: 4)")
    ap.add_argument("--lights", type=int, default=100_000)
    ap.add_argument("--meshes", type=int, default=10_000)
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks of --steps steps (0 = enough for ~0.4 s, at least 15)")
    ap.add_argument("--unfused", action="store_true", help="flat: mi_propagate + mi_cull instead of the fused kernel")
    ap.add_argument("--inline-compaction", action="store_true",
                    help="launch the VisibleEntities compaction as its own kernel every frame (default: MI_CULL_MORE_FRAMES, the "
                         "compaction of frame f rides in frame f+1's launch)")
    ap.add_argument("--separate-cluster-calls", action="store_true",
                    help="frame: mi_propagate_and_cull, then mi_cluster_assign_resident behind it (default: ONE call with "
                         "MI_CULL_WITH_CLUSTERS, the assignment concurrent with the frame kernel on the cluster stream)")
    ap.add_argument("--concurrent-clusters", action="store_true", help="frame: add MI_CULL_CLUSTERS_CONCURRENT")
    ap.add_argument("--sorted-items", type=int, default=0, help="batching_sorted: items of the phase (default 65 536)")
    ap.add_argument("--sorted-one-wg-limit", type=int, default=None, help="batching_sorted: phases up to this long take the single-workgroup kernel (default 4096; 4294967295 = always)")
    ap.add_argument("--tree-moved", choices=["all", "subtree", "leaves"], default="all", help="tree: the root moves and every Transform counts as changed (default) / change-driven frames: one level-5 node moves / 10 000 leaves move")
    ap.add_argument("--tree-shape", default="", help="tree: one of the reference's hierarchy stress shapes (bevy_amd.workloads.HIERARCHY_SHAPES: large_tree, wide_tree, deep_tree, chain, update_leaves, update_shallow, humanoids_active / _inactive / _mixed, tree_4ary_depth11 / _depth12) instead of configs[4]'s tree")
    ap.add_argument("--tree-shape-frame", choices=["all", "movers"], default="all", help="--tree-shape: every Transform counts as changed / the example's own frame (its update system's movers, StaticTransformOptimizations)")
    ap.add_argument("--tree-cull-launches", type=int, default=0, choices=[0, 1, 2], help="tree --tree-cull: 0 = the library's choice (one view: the tiles cull their own rows), 1 = always that, 2 = tile launch + cull launch")
    ap.add_argument("--tree-cull", action="store_true", help="tree: the hierarchy FRAME -- mi_propagate_and_cull on a context with a hierarchy (tile launch + cull launch, one call)")
    ap.add_argument("--sphere-path", type=int, default=0, help="flat_static / frame: 0 = world-sphere cull path from the second quiet frame (default), 1 = never (k_frame<0> over GlobalTransform + Aabb), 2 = at once")
    ap.add_argument("--static-cull-order", type=int, default=0, choices=[0, 1, 2], help="flat_static: 0 = frames of a static scene of >= 3 000 000 rows run over the cell order (default), 1 = never, 2 = at once, any size")
    ap.add_argument("--tile-mode", type=int, default=0, help="tree: 0 = subtree tiles where they fit, 1 = level by level, 3 = tiles with the streamed-level thresholds at their test values")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget")
    ap.add_argument("--profile-all", action="store_true", help="profiled blocks time every kernel, not only the workload's own")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this run (N = 1)")
    ap.add_argument("--full-line", action="store_true", help="also print the full object (bench_full.json's content) on the line BEFORE the compact one")
    return ap.parse_args()



# ---------------------------------------------------------------------------------------------------------------------
OTHER_WORKLOADS = [  # the other BASELINE configs, measured briefly on fresh contexts: they go to bench_full.json
    ("frame_plain_columns", lambda c, a: build_frame(c, with_args(a, row_summary=1))),  # the metric frame, every row reading its own Aabb / flags / layers
    ("flat", lambda c, a: build_flat(c, a, 0, 1, [], a.entities or 1_000_000, 1, "flat")),
    ("flat_plain_columns", lambda c, a: build_flat(c, with_args(a, row_summary=1), 0, 1, [], a.entities or 1_000_000, 1, "flat")),
    ("flat_10m_4views", lambda c, a: build_flat(c, a, 0, 1, [], 10_000_000, 4, "sharded")),
    ("flat_10m_1view", lambda c, a: build_flat(c, a, 0, 1, [], 10_000_000, 1, "flat")),
    ("tree", lambda c, a: build_tree(c, a)),
    ("tree_one_subtree_moves", lambda c, a: build_tree(c, with_args(a, tree_moved="subtree"))),
    ("tree_10k_leaves_move", lambda c, a: build_tree(c, with_args(a, tree_moved="leaves"))),
    ("tree_frame", lambda c, a: build_tree(c, with_args(a, tree_cull=True))),
    ("tree_frame_two_launches", lambda c, a: build_tree(c, with_args(a, tree_cull=True, tree_cull_launches=2))),
    ("lights", lambda c, a: build_lights(c, a)),
] + [  # the reference's own hierarchy stress shapes (examples/stress_tests/transform_hierarchy.rs:29-160) + config 5's full-size trees
    (f"tree_shape_{shape}" + ("" if kind == "all" else "_movers"), (lambda c, a, shape=shape, kind=kind: build_tree_shape(c, with_args(a, tree_shape=shape, tree_shape_frame=kind))))
    for shape in ("large_tree", "wide_tree", "deep_tree", "chain", "update_leaves", "update_shallow", "humanoids_active", "humanoids_inactive", "humanoids_mixed",
                  "tree_4ary_depth11", "tree_4ary_depth12")
    for kind in (("all", "movers") if not shape.startswith("tree_4ary") else ("all",))
] + [
    ("flat_static", lambda c, a: build_flat_static(c, a)),
    ("flat_static_no_sphere_column", lambda c, a: build_flat_static(c, with_args(a, sphere_path=1))),
    ("flat_static_10m_4views", lambda c, a: build_flat_static(c, with_args(a, entities=10_000_000, views=4))),
    ("flat_static_10m_4views_no_cull_order", lambda c, a: build_flat_static(c, with_args(a, entities=10_000_000, views=4, static_cull_order=1))),
    ("flat_static_no_cull_order", lambda c, a: build_flat_static(c, with_args(a, static_cull_order=1))),
    ("batching", lambda c, a: build_batching(c, a)),
    ("batching_sorted_1k", lambda c, a: build_batching_sorted(c, with_args(a, sorted_items=1024))),
    ("batching_sorted_4k", lambda c, a: build_batching_sorted(c, with_args(a, sorted_items=4096))),
    ("batching_sorted_64k", lambda c, a: build_batching_sorted(c, with_args(a, sorted_items=65_536))),
    ("batching_sorted_1m", lambda c, a: build_batching_sorted(c, with_args(a, sorted_items=1_000_000))),
]


def traffic_args(workload, args):
    """argv tail that makes a child bench.py run this workload the same way (benchlib/traffic.py)."""
    a = ["--workload", workload, "--row-summary", str(args.row_summary)]
    for flag, val in (("--entities", args.entities), ("--views", args.views), ("--sorted-items", args.sorted_items), ("--sphere-path", args.sphere_path), ("--static-cull-order", args.static_cull_order),
                      ("--tile-mode", args.tile_mode), ("--tree-cull-launches", args.tree_cull_launches)):
        if val:
            a += [flag, str(val)]
    a += ["--lights", str(args.lights), "--meshes", str(args.meshes), "--tree-moved", args.tree_moved]
    if args.tree_shape:
        a += ["--tree-shape", args.tree_shape, "--tree-shape-frame", args.tree_shape_frame]
    for flag, on in (("--unfused", args.unfused), ("--inline-compaction", args.inline_compaction), ("--tree-cull", args.tree_cull),
                     ("--separate-cluster-calls", args.separate_cluster_calls), ("--concurrent-clusters", args.concurrent_clusters)):
        if on:
            a.append(flag)
    return a


def main():
    args = parse()
    import torch
    from bevy_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("MI_FORCE_DIST") == "1"  # MI_FORCE_DIST: exercise the N > 1 code on one GPU
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    workload = args.workload
    if workload == "auto":
        workload = "frame" if world == 1 else "sharded"
    if workload in ("frame", "lights", "flat_static", "batching", "batching_sorted") and world > 1:
        raise SystemExit(f"--workload {workload} is a single-GPU workload (replicas only); N > 1 runs `sharded` or `tree`")

    stream = torch.cuda.Stream()
    ctx = api.Context(local_rank, stream.cuda_stream)
    full_holder = []
    scaling = None  # N = 1 makes no scaling claim; N > 1: "strong" (total rows fixed) unless --scaling weak
    with torch.cuda.stream(stream):
        if workload == "frame":
            wl = build_frame(ctx, args)
        elif workload in ("sharded", "flat"):
            if workload == "sharded":
                n_views = args.views or 4
                scaling = args.scaling
                n_global = (args.entities or 10_000_000) if scaling == "strong" else (args.entities or 1_000_000) * world
            else:
                n_views = args.views or 1
                n_global = (args.entities or 1_000_000) * world
            wl = build_flat(ctx, args, rank, world, full_holder, n_global, n_views, workload)
        elif workload == "tree":
            scaling = args.scaling if world > 1 else None
            if args.tree_shape and world > 1:
                raise SystemExit("--tree-shape is a single-GPU workload")
            wl = build_tree_shape(ctx, args) if args.tree_shape else build_tree(ctx, args, rank, world)
        elif workload == "flat_static":
            wl = build_flat_static(ctx, args)
        elif workload == "batching":
            wl = build_batching(ctx, args)
        elif workload == "batching_sorted":
            wl = build_batching_sorted(ctx, args)
        else:
            wl = build_lights(ctx, args)

        def reduce_max(ts):
            tt = torch.tensor(ts, dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return tt.cpu().tolist()

        def agree(nb):
            tt = torch.tensor([nb], dtype=torch.int64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            return int(tt.item())
        times, prof, info = measure(ctx, wl, args.steps, args.warmup, args.blocks, args.profile_all,
                                    (lambda: dist.barrier()) if use_dist else None, reduce_max if use_dist else None,
                                    agree if use_dist else None)

    out = None
    if rank == 0:
        med = float(np.median(times))
        units = getattr(wl, "global_units", wl.units * world)
        live = None
        if world == 1 and not args.no_live_traffic and os.environ.get("MI_BENCH_CHILD") != "1":
            live = traffic.measure_live(traffic_args(workload, args), getattr(wl, "kernel_name", wl.dominant).split(" ")[0])
        out = {"metric": wl.metric, "value": round(units * args.steps / med, 1), "unit": wl.unit, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * med / args.steps, 5), "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": wl.config,
               "timing": f"median of {len(times)} blocks of exactly {args.steps} steps, each between barrier + synchronize pairs, MAX over ranks per block",
               "blocks": block_stats(np.array(times), args.steps), "roofline": roofline_of(wl, prof, args.steps, live), "cpu_baseline": None,
               "kernels": {k: round(v["avg_us"], 3) for k, v in prof.items() if v["launches"]}}
        if getattr(wl, "frame_level_roofline", False):
            out["roofline_frame"] = roofline_frame(wl, prof, args.steps, PROFILED_BLOCKS)
            out["roofline"] = reprice_per_frame(out["roofline"], out["roofline_frame"])
        if live:
            out["live_traffic"] = live
        if scaling is None:
            del out["scaling"]  # one GPU makes no scaling claim
        out.update(info)
        if world == 1 and not args.no_cpu_baseline:
            if workload == "frame":
                out["cpu_baseline"] = cpu.cpu_baseline_frame(wl, args.cpu_seconds)
                out["config0_cpu_plumbing"] = cpu.config0_cpu_plumbing(args.cpu_seconds)
            elif workload in ("flat", "sharded"):
                out["cpu_baseline"] = cpu.cpu_baseline_flat(wl, args.cpu_seconds, wl.n_views)
            elif workload == "tree" and args.tree_shape:
                out["cpu_baseline"] = cpu.cpu_baseline_tree_shape(wl, args.tree_shape_frame)
            else:
                out["cpu_baseline"] = cpu.cpu_baseline_other(workload, wl)
        if world == 1 and workload == "frame" and not args.no_end_to_end:
            with torch.cuda.stream(stream):
                out["end_to_end"] = e2e.end_to_end(ctx, wl, cpu_frame_ms=(out["cpu_baseline"] or {}).get("frame_ms"))
            out["end_to_end_host_layer"] = e2e.end_to_end_host_layer(wl.units)
    if use_dist and workload == "sharded" and getattr(wl, "gather", None) is not None:
        # The first N > 1 run has to certify more than speed: after the timed region every rank runs ONE more frame
        # with a known camera set, rank 0 fetches the GATHERED masks and compares them bit for bit with the masks of the same frame
        # computed over the whole scene in ONE context on its own GPU; the collective's own latency (events on the communication
        # stream) and every rank's kernel time go into the line as well.
        VERIFY_FRAME = 3
        with torch.cuda.stream(stream):
            gathered = wl.gathered_masks(VERIFY_FRAME)
        ag_us = wl.gather.all_gather_latency_us()
        dk = prof.get(wl.dominant) or {}
        mine = torch.tensor([float(dk.get("avg_us") or 0.0)], dtype=torch.float64, device="cuda")
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        # the same scene, whole, on rank 0's GPU alone (outside the timed region): what N = 1 gives for THIS workload
        single, match = None, None
        if rank == 0:
            c1 = api.Context(local_rank, stream.cuda_stream)
            with torch.cuda.stream(stream):
                w1 = build_flat(c1, args, 0, 1, [], wl.global_units, wl.n_views, "sharded", no_gather=True)
                t1, _, _ = measure(c1, w1, args.steps, args.warmup, 15)
                whole = w1.own_masks(VERIFY_FRAME)
            m1 = float(np.median(t1))
            single = {"value": round(wl.global_units * args.steps / m1, 1), "unit": wl.unit, "ms_per_step": round(1e3 * m1 / args.steps, 5),
                      "note": "the whole scene on rank 0's GPU alone, measured after the timed region while the other ranks wait"}
            match = bool(gathered.shape == whole.shape and np.array_equal(gathered, whole))
            if not match:
                single["mask_bits_differing"] = int(np.count_nonzero(gathered != whole)) if gathered.shape == whole.shape else -1
            c1.close()
        dist.barrier()
        if rank == 0:
            out["single_gpu_same_workload"] = single
            out["gathered_masks_match_single_gpu"] = match
            out["all_gather_us"] = ag_us
            out["kernel_us_per_rank"] = [round(float(t.item()), 2) for t in per_rank]
            # what the driver's curve cannot show (its N = 1 point is the metric frame, another workload): this line's value against the
            # SAME scene on one GPU, per GPU
            out["scaling_efficiency"] = round(out["value"] / (world * single["value"]), 4)
    if rank == 0 and world == 1 and workload == "frame" and not args.no_other_workloads:
        others = {}
        for name, builder in OTHER_WORKLOADS:
            c2 = api.Context(local_rank, stream.cuda_stream)
            with torch.cuda.stream(stream):
                w2 = builder(c2, args)
                w2.profile_key = name  # the committed rocprofv3 evidence is filed per command (flat at 10 M rows is not "flat")
                t2, p2, i2 = measure(c2, w2, 50, 10, 15)
            m2 = float(np.median(t2))
            others[name] = {"metric": w2.metric, "value": round(getattr(w2, "global_units", w2.units) * 50 / m2, 1), "unit": w2.unit,
                            "ms_per_step": round(1e3 * m2 / 50, 5), "blocks": block_stats(np.array(t2), 50), "config": w2.config,
                            "roofline": roofline_of(w2, p2, 50), "kernels": {k: round(v["avg_us"], 3) for k, v in p2.items() if v["launches"]}}
            if getattr(w2, "frame_level_roofline", False):
                others[name]["roofline_frame"] = roofline_frame(w2, p2, 50, PROFILED_BLOCKS)
                others[name]["roofline"] = reprice_per_frame(others[name]["roofline"], others[name]["roofline_frame"])
            if name == "batching":
                others[name]["batch_build_us_per_frame"] = round(1e3 * (others[name]["ms_per_step"] - others["flat"]["ms_per_step"]), 2)
            if not args.no_cpu_baseline and (name in ("tree", "lights", "batching") or name.startswith("batching_sorted")):
                others[name]["cpu_baseline"] = cpu.cpu_baseline_other(name.split("_sorted")[0] + ("_sorted" if "_sorted" in name else ""), w2)
            if not args.no_cpu_baseline and name.startswith("tree_shape_"):
                # the CPU path beside every hierarchy shape: where the device LOSES (a chain) the table says so
                cb = cpu.cpu_baseline_tree_shape(w2, "movers" if name.endswith("_movers") else "all", 0.5 if w2.units > 1_000_000 else 0.3)
                cb["x_device_step"] = round(cb["ms_per_frame"] / others[name]["ms_per_step"], 2)  # > 1: the device's frame (kernels + its dirty-row upload) is faster
                others[name]["cpu_baseline"] = cb
            c2.close()
        out["other_workloads"] = others
    if rank == 0:
        import json
        result_line.write_full(out)
        sys.stdout.flush()
        try:  # anything native code left in C stdio buffers (e.g. RCCL's version banner) goes out BEFORE the result line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if args.full_line:
            print(json.dumps(out), flush=True)
        print(result_line.compact(out), flush=True)
    ctx.close()          # stops the library's exchange thread before the communicator goes away
    for g in full_holder:
        g.close()
    if use_dist:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
