#!/usr/bin/env python
"""bench.py -- throughput of the render-prep hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path over device-resident columns with every Transform dirty:
  flat (default, BASELINE.json configs[1]): 1M flat entities per GPU, 1 camera frustum: ONE frame kernel
        (sync_simple_transforms + reset_view_visibility + check_visibility_cpu_culling + check_visibility_gpu_culling
        + mark_newly_hidden_entities_invisible) and ONE VisibleEntities compaction kernel (at N = 1 the compaction of frame f rides in
        the tail workgroups of frame f+1's launch, MI_CULL_MORE_FRAMES; --inline-compaction launches it on its own).
        With N > 1 GPUs every rank owns a 1M-row range of an N x 1M scene (weak scaling) and the packed
        ViewVisibility bitmasks are exchanged with ONE RCCL all-gather per frame.
  tree  (configs[4]): depth-12/branch-4 tree truncated to 1M nodes, root moved every frame, propagate only.
  lights (configs[2]): 100k point lights, 16x9x24 clusters, assign_objects_to_clusters only.
Rank 0 prints ONE JSON line (DESIGN.md section 5 explains the byte accounting behind `roofline`); at N=1 the
default run also measures tree, lights, the 0 %-dirty flat frame and the batching work-item build briefly and reports
them under `other_workloads`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["flat", "tree", "lights", "flat_static", "batching"], default="flat")
    ap.add_argument("--entities", type=int, default=1_000_000, help="rows per GPU (flat) / nodes (tree)")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--lights", type=int, default=100_000)
    ap.add_argument("--unfused", action="store_true", help="flat: mi_propagate + mi_cull instead of the fused kernel")
    ap.add_argument("--inline-compaction", action="store_true",
                    help="flat: launch the VisibleEntities compaction as its own kernel every frame (default at N=1: "
                         "MI_CULL_MORE_FRAMES, the compaction of frame f rides in frame f+1's launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel, not only the dominant one")
    ap.add_argument("--profile-every", type=int, default=0, help="time every Nth launch of the dominant kernel (0 = workload default)")
    return ap.parse_args()


def flat_bytes_per_entity(n_views, fused=True):
    # frame kernel: read Aabb 24 + flags 1 + layers 4 + vv 1 and T 40 (fused) or resident G 48 (unfused);
    # write vv 1 + (V view masks + vv change mask)/8 bits + V/64 wave counts, and G 48 + its change mask (fused)
    rd = 30.0 + (40.0 if fused else 48.0)
    wr = 1.0 + (n_views + 1) / 8.0 + n_views / 64.0 + ((48.0 + 1.0 / 8.0) if fused else 0.0)
    return rd + wr


class Workload:
    """step(f) enqueues one frame; units = work items per frame on this rank."""

    def __init__(self, name, step, units, bytes_per_unit, dominant, config, metric, unit):
        self.name, self.step, self.units, self.bytes_per_unit = name, step, units, bytes_per_unit
        self.dominant, self.config, self.metric, self.unit = dominant, config, metric, unit
        # The dominant kernel is timed on the first eighth of the timed region's launches only: binding start/stop events to
        # a dispatch fences it off from its neighbours (as rocprofv3's kernel trace does) and costs ~5 us of GPU time per
        # launch on this stack (30.5 vs 25.5 us per flat frame), which would otherwise tax `value` itself.
        self.profile_every = 1
        self.profile_fraction = 8


def build_flat(ctx, args, rank, world, total_frames, full_holder):
    import torch
    import bevy_amd as B
    from bevy_amd import api, sharding, workloads as W
    n_views = args.views
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)

    def frusta_of_frame(f):
        return np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(f, yaw=v * np.pi / 2), W.CAMERA_FAR)
                               for v in range(n_views)])

    n_local = args.entities
    if world > 1:  # shards start on a workgroup (256-row) boundary: sharding.shard_rows(n_global, world, rank) is then exactly
        n_local = -(-n_local // sharding.ROW_ALIGN) * sharding.ROW_ALIGN  # [rank * n_local, (rank + 1) * n_local)
    n_global = n_local * world
    lo, hi = sharding.shard_rows(n_global, world, rank)
    assert (lo, hi) == (rank * n_local, (rank + 1) * n_local)
    radius = 500.0 * (n_global / 1_000_000.0) ** (1.0 / 3.0)
    scene = W.many_cubes(n_global, radius=radius, start=lo, count=n_local)
    ctx.resize(n_local)
    ctx.upload_transforms(scene["translation"], scene["rotation"], scene["scale"])
    ctx.upload_bounds(scene["aabb_center"], scene["aabb_half"], scene["flags"], scene["layers"])
    frames = [api.PreparedFrusta(frusta_of_frame(f)) for f in range(total_frames)]
    gather = None
    if world > 1 or os.environ.get("MI_FORCE_GATHER") == "1" or os.environ.get("MI_FORCE_DIST") == "1":
        # frame f's all-gather overlaps frame f+1's kernels (two gathered buffers, own stream)
        gather = sharding.MaskGatherer(n_global, world, n_views, rank, device=torch.device("cuda", torch.cuda.current_device()))
        full_holder.append(gather)
        gather.attach(ctx)  # direct RCCL available: the library issues the exchange itself, one FFI call per frame
    py_exchange = gather is not None and not gather.native
    # frames follow back to back: each frame's compaction is deferred into the next frame's launch (MI_CULL_MORE_FRAMES);
    # measure()'s final mi_synchronize enqueues the last one.  With the exchange on the flag is ignored by the library.
    deferred_compaction = not args.inline_compaction and not py_exchange
    more = B.CULL_MORE_FRAMES if deferred_compaction is True else 0  # frames follow back to back; measure()'s synchronize joins

    def step(f):
        if py_exchange:
            gather.before_kernels(f)
            ctx.bind_visibility_output(*gather.bind_args(f))
        if args.unfused:
            ctx.propagate(B.PROPAGATE_ALL_DIRTY)
            ctx.cull(frames[f], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
        else:
            ctx.propagate_and_cull(frames[f], flags=B.CULL_END_FRAME | more)
        if py_exchange:
            gather.after_kernels(f)

    config = {"workload": f"many_cubes-shaped flat scene, {n_local} entities/GPU ({n_global} total), {n_views} camera "
                          f"frustum(s), all Transforms dirty, columns resident in HBM: "
                          f"{'mi_propagate + mi_cull' if args.unfused else 'fused frame kernel'} (propagate + reset + frustum "
                          "cull + mark-newly-hidden) + VisibleEntities compaction"
                          + (" (MI_CULL_MORE_FRAMES: the compaction of frame f rides in the tail workgroups of frame f+1's "
                             "kernel, the last one is enqueued by the final mi_synchronize inside the timed region)"
                             if deferred_compaction is True else "")
                          + (f" + one in-place RCCL all-gather of the visibility bitmask per frame over {world} GPUs "
                             f"({gather.mode}, pipelined one frame deep on its own stream)" if gather is not None else ""),
              "baseline_config": "BASELINE.json configs[1] (propagate + frustum-cull; the cluster stage of the metric's name is "
                                 "configs[2], reported under other_workloads.lights)",
              "entities_per_gpu": n_local, "views": n_views, "deferred_compaction": deferred_compaction, "parallelism": f"row-range shard x{world}"}
    if gather is not None and gather.fallback_reason:
        config["rccl_direct_fallback"] = gather.fallback_reason
    wl = Workload("flat", step, n_local, flat_bytes_per_entity(n_views, not args.unfused),
                  "k_cull" if args.unfused else "k_flat_propagate_cull", config, "entities/sec through propagate+cull",
                  "entities/s")
    wl.scene, wl.frusta_of_frame, wl.n_views = scene, frusta_of_frame, n_views
    return wl


def build_tree(ctx, args, rank=0, world=1):
    import bevy_amd as B
    from bevy_amd import sharding, workloads as W
    tr = W.gen_tree(12, 4, args.entities * world)
    n_global = tr["n"]
    if world > 1:
        # shard by root subtree -- the one giant tree is opened up, its top rows are replicated and the
        # subtrees below are bin-packed on the GPUs; no collective (a rank's rows never read another rank's)
        sh = sharding.shard_hierarchy(tr["parent"], tr["level_offsets"], world, rank)
        rows = sh["rows"].astype(np.int64)
        tr = dict(n=len(rows), parent=sh["parent"], level_offsets=sh["level_offsets"],
                  translation=tr["translation"].reshape(-1, 3)[rows].reshape(-1), rotation=tr["rotation"].reshape(-1, 4)[rows].reshape(-1),
                  scale=tr["scale"].reshape(-1, 3)[rows].reshape(-1), owned=int(sh["owned"].sum()))
    ctx.resize(tr["n"])
    ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
    ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
    # the root moves every frame (a 40-byte dirty-row upload), so set_if_neq really rewrites every descendant
    root_t = [tr["translation"][:3].copy(), tr["translation"][:3] + np.float32(1.0)]

    def step(f):
        ctx.upload_transforms(root_t[f & 1], tr["rotation"][:4], tr["scale"][:3], first_row=0)
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    config = {"workload": f"gen_tree(12,4) truncated to {n_global} nodes ({len(tr['level_offsets']) - 1} levels), root moved "
                          "every frame (dirty-row upload), LDS subtree-tile propagation"
                          + (f", sharded by root subtree over {world} GPUs (this rank holds {tr['n']} rows, no collective)" if world > 1 else ""),
              "nodes": n_global, "parallelism": f"root-subtree shard x{world}"}
    # T 40, parent_idx 4, old G 48 (set_if_neq), G 48, changed byte 1
    wl = Workload("tree", step, tr["n"], 141.0, "k_propagate_tiles", config, "nodes/sec through hierarchy propagate", "nodes/s")
    wl.tree = tr
    wl.global_units = n_global  # every node is owned by exactly one rank (replicated top rows are recomputed, not counted)
    return wl


def build_lights(ctx, args):
    from bevy_amd import api, workloads as W
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    lights = W.many_lights(args.lights, 50.0, 0.3)
    cam = W.many_cubes_camera(0)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    ctx.cluster_upload_objects(lights)
    ctx.cluster_upload_view(view)

    def step(f):
        ctx.cluster_assign_resident()
    config = {"workload": f"many_lights-shaped: {args.lights} point lights (range 0.3, shell R=50), 16x9x24 clusters, "
                          "assign_objects_to_clusters, objects resident (replicas per GPU)", "lights": args.lights}
    wl = Workload("lights", step, args.lights, 17.0, "k_cluster_walk", config,
                  "lights/sec through assign_objects_to_clusters", "lights/s")
    wl.keep = (view, keep, lights)
    wl.oracle_args = (cam, cfv, fr)
    return wl


def build_flat_static(ctx, args):
    """configs[1], second run: 0 % dirty -- mi_propagate finds nothing changed, mi_cull reads the resident G."""
    import bevy_amd as B
    from bevy_amd import api, workloads as W
    n = args.entities
    sc = W.many_cubes(n)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.upload_changed(np.zeros(n, np.uint8))  # the change column exists from here on: only marked rows are recomputed
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    frames = [api.PreparedFrusta(api.compute_frustum(cfv, W.many_cubes_camera(f), W.CAMERA_FAR)) for f in range(128)]
    more = 0 if args.inline_compaction else B.CULL_MORE_FRAMES  # as in the flat workload: frames back to back

    def step(f):
        ctx.propagate(0)
        ctx.cull(frames[f & 127], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
    config = {"workload": f"many_cubes-shaped flat scene, {n} entities, 1 frustum, 0 % of the Transforms dirty: mi_propagate "
                          "(no row was marked since the last one: returns without a launch) + mi_cull (G resident) + VisibleEntities "
                          "compaction" + (" deferred into the next frame's launch" if more else ""), "entities": n, "deferred_compaction": bool(more)}
    # cull with G resident: read G 48 + Aabb 24 + flags 1 + layers 4 + vv 1, write vv 1 + masks
    return Workload("flat_static", step, n, flat_bytes_per_entity(1, False), "k_cull", config,
                    "entities/sec through propagate+cull", "entities/s")


def build_batching(ctx, args):
    """the flat frame followed by the batching work-item build of the camera's list."""
    import bevy_amd as B
    from bevy_amd import api, workloads as W
    n = args.entities
    sc = W.many_cubes(n)
    bs = W.batching_scene(n, n_sets=64, max_bins=40, seed=7)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
    ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
    frames = [api.PreparedFrusta(api.compute_frustum(cfv, W.many_cubes_camera(f), W.CAMERA_FAR)) for f in range(128)]

    def step(f):
        ctx.propagate_and_cull(frames[f & 127], flags=B.CULL_END_FRAME)
        ctx.batch_build(0, 0)
    step(0)
    ctx.synchronize()
    rows = ctx.download_visible_entities(0, 0)[1]
    items = int(np.count_nonzero(bs["row_set"][rows] != 0xFFFFFFFF))
    config = {"workload": f"flat frame of {n} entities + batching work-item build of the camera's VisibleEntities list: "
                          f"{len(rows)} visible rows -> {items} PreprocessWorkItems in {len(bs['set_indexed'])} batch sets / "
                          f"{len(bs['bin_metadata'])} bins (stable partition by set, allocate_uniforms, unpack_bins)",
              "entities": n, "work_items_per_frame": items}
    wl = Workload("batching", step, n, flat_bytes_per_entity(1, True), "k_flat_propagate_cull", config,
                  "entities/sec through propagate+cull+batch build", "entities/s")
    wl.batch = (bs, rows)
    return wl


def cpu_baseline_other(name, wl):
    """The oracle's C port of the same stage on the host: the hierarchy on all cores (rows of a level in parallel, levels in
    order -- the parallelism propagate_parent_transforms gets from the task pool), assign_objects_to_clusters and the batch
    bookkeeping on ONE core (single-threaded in the reference); a few seconds' worth of frames."""
    import oracle_lib as O
    if name == "flat_static":
        return None  # the flat line's CPU baseline is the same stage with the propagate included
    if name == "batching":
        bs, rows = wl.batch
        a = (rows, bs["row_set"], bs["row_bin"], bs["row_input"], bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"],
             bs["meta_offset"], bs["bin_metadata"])
        t0 = time.perf_counter()
        O.batch_build(*a)
        one = time.perf_counter() - t0
        iters = int(max(1, min(200, 2.0 / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(iters):
            O.batch_build(*a)
        secs = time.perf_counter() - t0
        return {"value": round(len(rows) * iters / secs, 1), "unit": "visible rows/s (batch build only)", "cores": 1, "kind": "port",
                "sample": f"{iters} builds over {len(rows)} visible rows: oracle C restatement of the bin bookkeeping + "
                          f"allocate_uniforms + unpack_bins, {secs:.2f}s"}
    if name == "tree":
        tr = wl.tree
        cores = os.cpu_count() or 1
        a = (tr["parent"], tr["level_offsets"], tr["translation"], tr["rotation"], tr["scale"])
        one, _ = O.bench_tree_frame(*a, cores, 1)
        iters = int(max(1, min(2000, 3.0 / max(one, 1e-4))))
        secs, _ = O.bench_tree_frame(*a, cores, iters)
        return {"value": round(tr["n"] * iters / secs, 1), "unit": "nodes/s", "cores": cores, "kind": "port",
                "sample": f"{iters} frames of {tr['n']} nodes, every Transform changed: oracle C port of propagate_parent_transforms "
                          f"(set_if_neq), rows of a level split over a persistent pool of {cores} threads, levels in order, {secs:.2f}s"}
    cam, cfv, fr = wl.oracle_args
    view, lights = O.cluster_view_setup(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0), wl.keep[2]
    t0 = time.perf_counter()
    O.assign_objects_to_clusters(view, lights)
    one = time.perf_counter() - t0
    iters = int(max(1, min(200, 3.0 / max(one, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(iters):
        O.assign_objects_to_clusters(view, lights)
    secs = time.perf_counter() - t0
    n = len(lights) // 4
    return {"value": round(n * iters / secs, 1), "unit": "lights/s", "cores": 1, "kind": "port",
            "sample": f"{iters} frames of {n} lights: oracle C port of assign_objects_to_clusters (two passes per frame: size, then "
                      f"fill), {secs:.2f}s"}


def measure(ctx, wl, steps, warmup, profile_all, sync_extra=None):
    """warmup untimed frames, then exactly `steps` frames between barrier+synchronize pairs.  Returns
    (elapsed_s, per-kernel profile)."""
    import torch

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if sync_extra:
            sync_extra()

    # A generation-2 pass of Python's cyclic GC over everything torch imported takes ~50 ms -- a thousand frames of
    # this workload -- and fires after a fixed number of allocations, i.e. at a random frame: collect now, and keep the
    # collector off while frames are being enqueued (what timeit does).
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    for f in range(warmup):
        wl.step(f)
    sync_all()
    ctx.profile_filter(None if profile_all else [wl.dominant])
    ctx.profile_sample(getattr(wl, "profile_every", 1))
    wl.timed_launches = max(8, steps // getattr(wl, "profile_fraction", 1)) if getattr(wl, "profile_fraction", 1) > 1 else 0
    ctx.profile_burst(wl.timed_launches)
    ctx.profile_enable(True)
    sync_all()
    t0 = time.perf_counter()
    for f in range(warmup, warmup + steps):
        wl.step(f)
    wl.host_enqueue_s = time.perf_counter() - t0  # host time to enqueue the frames (GPU-bound if well below elapsed)
    sync_all()
    t1 = time.perf_counter()
    if gc_was_enabled:
        gc.enable()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    return t1 - t0, prof


def roofline_of(wl, prof, steps):
    dk = prof.get(wl.dominant)
    if not dk:
        return None
    avg_s = dk["avg_us"] * 1e-6
    launches_per_step = 1.0 if getattr(wl, "timed_launches", 0) else dk["launches"] * getattr(wl, "profile_every", 1) / steps
    alg_bytes = wl.bytes_per_unit * wl.units / launches_per_step
    achieved = alg_bytes / avg_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # HBM bytes per launch from the separate --pmc passes
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(wl.dominant, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": wl.dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "avg_kernel_us": round(dk["avg_us"], 3),
            "launches": dk["launches"], "algorithmic_bytes_per_launch": int(alg_bytes),
            "timing": "per-dispatch start/stop events (hipExtLaunchKernelGGL) on the first "
                      f"{dk['launches']} launches inside the timed region of {steps} steps"}


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

    stream = torch.cuda.Stream()
    ctx = api.Context(local_rank, stream.cuda_stream)
    total_frames = args.steps + args.warmup
    full_holder = []
    with torch.cuda.stream(stream):
        if args.workload == "flat":
            wl = build_flat(ctx, args, rank, world, total_frames, full_holder)
        elif args.workload == "tree":
            wl = build_tree(ctx, args, rank, world)
        elif args.workload == "flat_static":
            wl = build_flat_static(ctx, args)
        elif args.workload == "batching":
            wl = build_batching(ctx, args)
        else:
            wl = build_lights(ctx, args)
        if args.profile_every > 0:
            wl.profile_every, wl.profile_fraction = args.profile_every, 1
        barrier = (lambda: dist.barrier()) if use_dist else None
        elapsed, prof = measure(ctx, wl, args.steps, args.warmup, args.profile_all, barrier)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        value = getattr(wl, "global_units", wl.units * world) * args.steps / elapsed
        out = {"metric": wl.metric, "value": round(value, 1), "unit": wl.unit, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": wl.config,
               "host_enqueue_ms_per_step": round(1e3 * wl.host_enqueue_s / args.steps, 5),
               "roofline": roofline_of(wl, prof, args.steps), "cpu_baseline": None,
               "kernels": {k: round(v["avg_us"], 3) for k, v in prof.items()}}
        if not args.no_cpu_baseline and args.workload == "flat" and world == 1:  # reported at N = 1 only
            import oracle_lib as O  # the oracle doubles as the reported CPU baseline ("port"), never as the product
            from bevy_amd import workloads as W
            cores = os.cpu_count() or 1
            n_cpu = min(wl.units, 1_000_000)
            sc = wl.scene if n_cpu == wl.units else W.many_cubes(n_cpu)
            fr0 = wl.frusta_of_frame(args.warmup)
            a = (sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], fr0)
            secs, _, _, _ = O.bench_flat_frame(*a, cores, 1)
            iters = int(max(1, min(5000, args.cpu_seconds / max(secs, 1e-4))))
            secs, _, _, _ = O.bench_flat_frame(*a, cores, iters)
            out["cpu_baseline"] = {
                "value": round(n_cpu * iters / secs, 1), "unit": "entities/s", "cores": cores, "kind": "port",
                "sample": f"{iters} frames of {n_cpu} entities x {wl.n_views} view(s): oracle C port of sync_simple_transforms + "
                          "reset + check_visibility + mark_newly_hidden on a persistent thread pool, one ceil(n/threads) batch per "
                          f"thread and system (Bevy's par_iter batching), {secs:.2f}s"}
        if world == 1 and args.workload == "flat" and not args.no_other_workloads:
            # configs[4] and configs[2], measured briefly on fresh contexts so the one line carries every stage
            others = {}
            for name, builder in (("tree", build_tree), ("lights", build_lights), ("flat_static", build_flat_static),
                                  ("batching", build_batching)):
                c2 = api.Context(local_rank, stream.cuda_stream)
                with torch.cuda.stream(stream):
                    w2 = builder(c2, args)
                    e2, p2 = measure(c2, w2, 100, 10, False)
                others[name] = {"metric": w2.metric, "value": round(w2.units * 100 / e2, 1), "unit": w2.unit,
                                "ms_per_step": round(1e3 * e2 / 100, 5), "config": w2.config,
                                "roofline": roofline_of(w2, p2, 100)}
                if name == "batching":
                    others[name]["batch_build_us_per_frame"] = round(1e6 * e2 / 100 - 1e3 * out["ms_per_step"], 2)
                    with torch.cuda.stream(stream):
                        _, p3 = measure(c2, w2, 20, 2, True)  # per-kernel breakdown, every launch timed (not the rate above)
                    others[name]["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in p3.items()}
                if not args.no_cpu_baseline:
                    others[name]["cpu_baseline"] = cpu_baseline_other(name, w2)
                c2.close()
            out["other_workloads"] = others
        sys.stdout.flush()
        try:  # anything native code left in C stdio buffers (e.g. RCCL's version banner) goes out BEFORE the result line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    ctx.close()          # stops the library's exchange thread before the communicator goes away
    for g in full_holder:
        g.close()
    if use_dist:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
