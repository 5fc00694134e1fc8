#!/usr/bin/env python
"""bench.py -- throughput of the render-prep hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame of the hot path over device-resident columns with every Transform dirty:
  flat (default, BASELINE.json configs[1]): 1M flat entities per GPU, 1 camera frustum:
        fused sync_simple_transforms + reset_view_visibility + check_visibility_cpu_culling (one kernel),
        VisibleEntities compaction, check_visibility_gpu_culling + mark_newly_hidden_entities_invisible.
        With N > 1 GPUs every rank owns a 1M-row range of an N x 1M scene (weak scaling) and the packed
        ViewVisibility bitmasks are exchanged with ONE RCCL all-gather per frame.
  tree  (configs[4]): depth-12/branch-4 tree truncated to 1M nodes, propagate only.
  lights (configs[2]): 100k point lights, 16x9x24 clusters, assign_objects_to_clusters only.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the byte accounting behind `roofline`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["flat", "tree", "lights"], default="flat")
    ap.add_argument("--entities", type=int, default=1_000_000, help="rows per GPU (flat) / nodes (tree)")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--lights", type=int, default=100_000)
    ap.add_argument("--unfused", action="store_true", help="flat: mi_propagate + mi_cull instead of the fused kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel with HIP events (perturbs `value`)")
    return ap.parse_args()


def flat_bytes_per_entity(n_views):
    # fused kernel: read T 40 + Aabb 24 + flags 1 + layers 4 + vv 1; write G 48 + vv 1 + (V + 2 change masks)/8
    return 70.0 + 49.0 + (n_views + 2) / 8.0


def main():
    args = parse()
    import torch
    import bevy_amd as B
    from bevy_amd import api, sharding, workloads as W

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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    stream = torch.cuda.Stream()
    ctx = api.Context(local_rank, stream.cuda_stream)
    n_views = args.views
    total_frames = args.steps + args.warmup
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)

    def frusta_of_frame(f):
        return np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(f, yaw=v * np.pi / 2), W.CAMERA_FAR)
                               for v in range(n_views)])

    dominant = None
    units_per_rank = 0
    bytes_per_unit = 0.0
    step = None
    config = {}
    scene = None
    full = None

    if args.workload == "flat":
        n_local = args.entities
        n_global = n_local * world
        lo = rank * n_local
        radius = 500.0 * (n_global / 1_000_000.0) ** (1.0 / 3.0)
        scene = W.many_cubes(n_global, radius=radius, start=lo, count=n_local)
        ctx.resize(n_local)
        ctx.upload_transforms(scene["translation"], scene["rotation"], scene["scale"])
        ctx.upload_bounds(scene["aabb_center"], scene["aabb_half"], scene["flags"], scene["layers"])
        frames = [frusta_of_frame(f) for f in range(total_frames)]
        if world > 1:
            words = sharding.gathered_words(n_global, world, n_views)
            full = torch.zeros(words, dtype=torch.int64, device="cuda")
            wpv, woff = sharding.block_offset_words(n_global, world, n_views, rank)
            ctx.bind_visibility_output(full.data_ptr(), wpv, woff)
        units_per_rank = n_local
        bytes_per_unit = flat_bytes_per_entity(n_views) if not args.unfused else flat_bytes_per_entity(n_views) + 48.0
        dominant = "k_cull" if args.unfused else "k_flat_propagate_cull"

        def step(f):
            if args.unfused:
                ctx.propagate(B.PROPAGATE_ALL_DIRTY)
                ctx.cull(frames[f], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            else:
                ctx.propagate_and_cull(frames[f], flags=B.CULL_END_FRAME)
            if world > 1:
                sharding.all_gather_visibility(full, n_global, world, n_views, rank)
        config = {"workload": f"many_cubes-shaped flat scene, {n_local} entities/GPU ({n_global} total), {n_views} camera "
                              f"frustum(s), all Transforms dirty: {'unfused' if args.unfused else 'fused'} propagate + frustum cull "
                              "+ VisibleEntities compaction + mark-newly-hidden"
                              + (f" + RCCL all-gather of the visibility bitmask over {world} GPUs" if world > 1 else ""),
                  "entities_per_gpu": n_local, "views": n_views, "parallelism": f"row-range shard x{world}"}
    elif args.workload == "tree":
        tr = W.gen_tree(12, 4, args.entities)
        ctx.resize(tr["n"])
        ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
        ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
        units_per_rank = tr["n"]
        bytes_per_unit = 40.0 + 4.0 + 48.0 + 48.0 + 1.0  # T, parent_idx, old G (set_if_neq), G, changed byte
        dominant = "k_propagate_tiles"
        # the root moves every frame (a 40-byte upload), so set_if_neq really rewrites every descendant
        root_t = [tr["translation"][:3].copy(), tr["translation"][:3] + np.float32(1.0)]

        def step(f):
            ctx.upload_transforms(root_t[f & 1], tr["rotation"][:4], tr["scale"][:3], first_row=0)
            ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        config = {"workload": f"gen_tree(12,4) truncated to {tr['n']} nodes ({len(tr['level_offsets']) - 1} levels), "
                              "all dirty, LDS subtree-tile propagation (replicas per GPU)", "nodes": tr["n"]}
    else:
        lights = W.many_lights(args.lights, 50.0, 0.3)
        cam = W.many_cubes_camera(0)
        fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
        view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
        ctx.cluster_upload_objects(lights)
        ctx.cluster_upload_view(view)
        units_per_rank = args.lights
        bytes_per_unit = 17.0
        dominant = "k_cluster_count"

        def step(f):
            ctx.cluster_assign_resident()
        config = {"workload": f"many_lights-shaped: {args.lights} point lights (range 0.3, shell R=50), 16x9x24 clusters, "
                              "assign_objects_to_clusters (replicas per GPU)", "lights": args.lights}

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    with torch.cuda.stream(stream):
        for f in range(args.warmup):
            step(f)
        sync_all()
        ctx.profile_filter(None if args.profile_all else [dominant])
        ctx.profile_enable(True)
        sync_all()
        t0 = time.perf_counter()
        for f in range(args.warmup, total_frames):
            step(f)
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        prof = ctx.profile_read()
        ctx.profile_enable(False)

    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        units_total = units_per_rank * world
        value = units_total * args.steps / elapsed
        dk = prof.get(dominant)
        roofline = None
        if dk:
            avg_s = dk["avg_us"] * 1e-6
            launches_per_step = dk["launches"] / args.steps
            alg_bytes = bytes_per_unit * units_per_rank / launches_per_step
            achieved = alg_bytes / avg_s / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dominant, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                        "avg_kernel_us": round(dk["avg_us"], 3), "launches": dk["launches"],
                        "algorithmic_bytes_per_launch": int(alg_bytes)}
        cpu_baseline = None
        if not args.no_cpu_baseline and args.workload == "flat":
            import oracle_lib as O  # the oracle doubles as the reported CPU baseline ("port"), never as the product
            cores = os.cpu_count() or 1
            n_cpu = min(units_per_rank, 1_000_000)
            sc = scene if n_cpu == units_per_rank else W.many_cubes(n_cpu)
            fr0 = frusta_of_frame(args.warmup)
            secs, _, _, _ = O.bench_flat_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"],
                                               sc["aabb_half"], sc["flags"], sc["layers"], fr0, cores, 1)
            iters = int(max(1, min(400, args.cpu_seconds / max(secs, 1e-4))))
            secs, _, _, _ = O.bench_flat_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"],
                                               sc["aabb_half"], sc["flags"], sc["layers"], fr0, cores, iters)
            cpu_baseline = {"value": round(n_cpu * iters / secs, 1), "unit": "entities/s", "cores": cores, "kind": "port",
                            "sample": f"{iters} frames of {n_cpu} entities x {n_views} view(s): oracle C port of sync_simple_transforms + "
                                      "reset + check_visibility + mark_newly_hidden, one ceil(n/threads) batch per thread "
                                      f"(Bevy's par_iter batching), {secs:.2f}s"}
        out = {"metric": "entities/sec through propagate+cull", "value": round(value, 1), "unit": "entities/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": config, "roofline": roofline, "cpu_baseline": cpu_baseline,
               "kernels": {k: round(v["avg_us"], 3) for k, v in prof.items()}}
        if args.workload == "tree":
            out["metric"] = "nodes/sec through hierarchy propagate"
            out["unit"] = "nodes/s"
        if args.workload == "lights":
            out["metric"] = "lights/sec through assign_objects_to_clusters"
            out["unit"] = "lights/s"
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
