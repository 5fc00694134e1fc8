"""Flat rows, optionally sharded over ranks (configs[1] at N = 1, configs[3] at N > 1), and the 0 %-dirty run."""
import os

import numpy as np

from .common import N_FRAMES, ROW_SUMMARY_SAVES, Workload, camera_frusta, flat_bytes_per_entity


def build_flat(ctx, args, rank, world, full_holder, n_global, n_views, name, no_gather=False):
    import torch
    import bevy_amd as B
    from bevy_amd import api, sharding, workloads as W
    lo, hi = sharding.shard_rows(n_global, world, rank)
    n_local = hi - lo
    radius = 500.0 * (n_global / 1_000_000.0) ** (1.0 / 3.0)
    scene = W.many_cubes(n_global, radius=radius, start=lo, count=n_local)
    ctx.resize(n_local)
    ctx.upload_transforms(scene["translation"], scene["rotation"], scene["scale"])
    ctx.debug_set_row_summary(args.row_summary)
    ctx.upload_bounds(scene["aabb_center"], scene["aabb_half"], scene["flags"], scene["layers"])
    frames = [api.PreparedFrusta(camera_frusta(n_views, f)) for f in range(N_FRAMES)]
    gather = None
    if not no_gather and (world > 1 or os.environ.get("MI_FORCE_GATHER") == "1" or os.environ.get("MI_FORCE_DIST") == "1"):
        gather = sharding.MaskGatherer(n_global, world, n_views, rank, device=torch.device("cuda", torch.cuda.current_device()))
        full_holder.append(gather)
        gather.attach(ctx)  # direct RCCL available: the library issues the exchange itself, one FFI call per frame
    py_exchange = gather is not None and not gather.native
    deferred = not args.inline_compaction and not py_exchange
    more = B.CULL_MORE_FRAMES if deferred else 0
    fcount = [0]
    calibration = None
    if gather is not None and gather.native and not args.unfused:
        def run_frames(k):
            for f in range(k):
                ctx.propagate_and_cull(frames[f % N_FRAMES], flags=B.CULL_END_FRAME | more)
            ctx.exchange_last(True)
            ctx.synchronize()
        calibration = gather.calibrate(ctx, run_frames)

    def step(f):
        i = f % N_FRAMES
        k = fcount[0]
        fcount[0] += 1
        if py_exchange:
            gather.before_kernels(k)
            ctx.bind_visibility_output(*gather.bind_args(k))
        if args.unfused:
            ctx.propagate(B.PROPAGATE_ALL_DIRTY)
            ctx.cull(frames[i], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
        else:
            ctx.propagate_and_cull(frames[i], flags=B.CULL_END_FRAME | more)
        if py_exchange:
            gather.after_kernels(k)

    config = {"workload": f"propagate+cull" + (f"+all-gather, {world} ranks" if gather is not None else "") + f": many_cubes-shaped flat scene, {n_global} entities, {n_views} camera frustum(s), all Transforms dirty, columns "
                          f"resident in HBM: {'mi_propagate + mi_cull' if args.unfused else 'fused frame kernel'} (propagate + reset + "
                          "frustum cull + mark-newly-hidden) + VisibleEntities compaction"
                          + (" (deferred into the next frame's launch)" if deferred else "")
                          + (f"; rows sharded over {world} GPUs ({n_local} on this rank) + ONE in-place RCCL all-gather of the packed "
                             f"ViewVisibility bitmasks per frame ({gather.mode})" if gather is not None else ""),
              "baseline_config": "BASELINE.json configs[3]" if name == "sharded" else "BASELINE.json configs[1]",
              "entities_total": n_global, "entities_this_rank": n_local, "views": n_views, "deferred_compaction": deferred,
              "parallelism": f"row-range shard x{world}", "row_summary": args.row_summary == 0}
    if gather is not None and gather.fallback_reason:
        config["rccl_direct_fallback"] = gather.fallback_reason
    if gather is not None:
        # what the exchange is, as the code set it up: ranks RCCL itself reports for the communicator (ncclCommCount), who drives it
        config["rccl_ranks"] = gather.rccl_ranks()
        if calibration:
            config["exchange_calibration"] = calibration
        config["exchange_mode"] = gather.mode + (", ncclAllGather enqueued by the library's exchange thread" if gather.native and not gather.pipelined
                                                 and os.environ.get("MI_XCH_SYNC_ENQUEUE") is None else "")
    metric = ("entities/sec through propagate+cull (10M entities x 4 frusta, 1/2/4/8-GPU scaling)" if name == "sharded"
              else "entities/sec through propagate+cull")
    wl = Workload(name, step, n_local, flat_bytes_per_entity(n_views, not args.unfused),
                  "k_cull" if args.unfused else "k_flat_propagate_cull", config, metric, "entities/s",
                  kernels=["k_cull" if args.unfused else "k_flat_propagate_cull", "k_compact_fast"])
    wl.scene, wl.n_views, wl.global_units = scene, n_views, n_global
    wl.gather = gather

    def gathered_masks(frame):
        """One more frame with camera set `frame` (every rank calls it: the frame's all-gather is a collective) -> the gathered
        masks on the host, bool [views][n_global] (None without an exchange)."""
        if gather is None:
            return None
        step(frame)
        if gather.native:
            words = ctx.exchange_download(gather.world * gather.block * 8)
        else:
            gather.synchronize()
            ctx.synchronize()
            words = gather.buffer(fcount[0] - 1).cpu().numpy()
        return gather.unpack_gathered(words)

    def own_masks(frame):
        """The same frame's masks of a context that holds the whole scene (the single-GPU run): bool [views][n]."""
        step(frame)
        return np.stack([ctx.download_visibility(v) for v in range(n_views)]).astype(bool)
    wl.gathered_masks, wl.own_masks = gathered_masks, own_masks
    # (the timer slot's name is not the symbol's; 2 .. 4 camera views take the pair-pass kernel unless MI_MULTI_VIEW=1)
    pairs = 2 <= n_views <= 4 and os.environ.get("MI_MULTI_VIEW", "0") != "1"
    wl.kernel_name = "k_frame<0>" if args.unfused else "k_frame_pairs<1>" if pairs else "k_frame<1,true,0>"
    if args.row_summary == 0:
        wl.layout_bytes_per_row = wl.bytes_per_row - ROW_SUMMARY_SAVES
    return wl


def build_flat_static(ctx, args):
    """configs[1], second run: 0 % dirty -- mi_propagate finds nothing changed, mi_cull reads the resident G."""
    import bevy_amd as B
    from bevy_amd import api, workloads as W
    n = args.entities or 1_000_000
    sc = W.many_cubes(n, radius=500.0 * (n / 1_000_000.0) ** (1.0 / 3.0))  # configs[3]'s scaling: the density stays
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.debug_set_row_summary(args.row_summary)
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.upload_changed(np.zeros(n, np.uint8))  # the change column exists from here on: only marked rows are recomputed
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    n_views = args.views or 1
    frames = [api.PreparedFrusta(camera_frusta(n_views, f)) for f in range(N_FRAMES)]
    more = 0 if args.inline_compaction else B.CULL_MORE_FRAMES  # as in the flat workload: frames back to back
    sphere = getattr(args, "sphere_path", 0) != 1
    ctx.debug_set_sphere_path(getattr(args, "sphere_path", 0))
    order_mode = getattr(args, "static_cull_order", 0)
    ctx.debug_set_static_cull_order(order_mode)
    cells = sphere and order_mode != 1 and (n >= 3_000_000 or order_mode == 2)  # the library's rule (ctx.h, Cells::min_rows)

    def step(f):
        ctx.propagate(0)
        ctx.cull(frames[f % N_FRAMES], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
    # byte models per row.  G resident (k_frame<0>): read G 48 + Aabb 24 + flags 1 + layers 4 + vv 1, write vv 1 + masks.
    # World-sphere column (k_frame_sph): read (cw, sr) 16 + flags 1 + layers 4 + vv 1, write vv 1 + masks; GlobalTransform and half
    # extents only for the rows that pass a sphere test (a few percent: not counted -- the PMC traffic shows them).
    wr = 1.0 + (n_views + 1) / 8.0 + n_views / 64.0
    models = {"world_sphere_column": 22.0 + wr, "global_transform_resident": flat_bytes_per_entity(n_views, False)}
    config = {"workload": f"many_cubes-shaped flat scene, {n} entities, {n_views} frustum(s), 0 % of the Transforms dirty: mi_propagate "
                          "(no row was marked since the last one: returns without a launch) + mi_cull ("
                          + ("the world-sphere column: 16 B per row instead of GlobalTransform + Aabb, k_frame_sph" if sphere else "G resident, k_frame<0>")
                          + ") + VisibleEntities compaction" + (" deferred into the next frame's launch" if more else ""),
              "baseline_config": "BASELINE.json configs[1], 0 %-dirty run", "entities": n, "views": n_views, "deferred_compaction": bool(more),
              "sphere_path": sphere, "bytes_per_row_models": models}
    wl = Workload("flat_static", step, n, models["world_sphere_column" if sphere else "global_transform_resident"], "k_cull", config,
                  "entities/sec through propagate+cull", "entities/s", kernels=["k_cull", "k_compact_fast"])
    pairs = 2 <= n_views <= 4 and os.environ.get("MI_MULTI_VIEW", "0") != "1"  # (several camera views: the pair-pass kernels)
    wl.kernel_name = ("k_frame_sph_pairs<false>" if pairs else "k_frame_sph<false>") if sphere else ("k_frame_pairs<0>" if pairs else "k_frame<0>")
    if args.row_summary == 0:  # the sphere path reads flags + layers per row (5 B), the resident-G path Aabb as well
        wl.layout_bytes_per_row = wl.bytes_per_row - ((5.0 - 0.5) if sphere else ROW_SUMMARY_SAVES)
    config["row_summary"] = args.row_summary == 0
    config["static_cull_order"] = cells
    if n != 1_000_000 or n_views != 1:  # (the committed rocprofv3 evidence is filed per command: bench.py names this one in OTHER_WORKLOADS)
        wl.profile_key = f"flat_static_{n // 1_000_000}m_{n_views}views" + ("_no_cull_order" if sphere and order_mode == 1 and n >= 3_000_000 else "")
    if cells:
        # The static cull order (kernels_cells.hip): from the fourth quiet frame on a frame is four short launches over the cell-ordered
        # copy -- k_cells_test (a thread per cell of 64 slots: 36 B), k_frame_cells (a wave per cell that is left: 73 B per slot --
        # row 4, ViewVisibility 1, sphere 16, last frame's bits 4, GlobalTransform 48 -- and a handful of atomics for the bits that
        # changed), k_cells_blocks + k_cells_lists (masks -> lists and the next frame's masks).  What they move depends on what the
        # views see; warm frames are run here to count it.  Cells that straddle a frustum border make the processed slots about 1.2 x
        # the visible rows.
        for f in range(6):
            step(f)
        ctx.synchronize()
        builds, order_frames = ctx.debug_static_cull_counts()
        vis_rows = int(np.count_nonzero(ctx.download_view_visibility()[0] & 1))
        vis_total = sum(int(len(ctx.download_visible_entities(v, 0)[1])) for v in range(n_views))
        processed = min(float(n), 1.2 * vis_rows)
        mask_bytes = n / 64.0 * 8.0 * n_views
        frame_kernel = processed * 73.0 + processed / 64.0 * 16.0
        whole_frame = n / 64.0 * 36.0 + frame_kernel + 3.0 * mask_bytes + 4.0 * vis_total
        config["workload"] = config["workload"].replace("mi_cull (", "mi_cull over the static cull order (k_cells_test + k_frame_cells + k_cells_blocks + k_cells_lists: 64 "
                                                        "spatial neighbours per cell, whole cells rejected against each view first; without it: ")
        config["static_cull_order_state"] = {"orders_built": builds, "frames_over_it": order_frames, "rows_visible_in_some_view": vis_rows,
                                             "visible_list_entries": vis_total, "estimated_bytes_k_frame_cells": int(frame_kernel),
                                             "estimated_bytes_whole_frame": int(whole_frame)}
        wl.kernel_name = "k_frame_cells<true>" if n_views <= 8 else "k_frame_cells<false>"
        wl.kernels = ["k_cull", "k_compact_fast", "k_vis_begin", "k_compact_count"]  # timer slots: frame, lists, cell test, blocks
        wl.layout_bytes_per_row = min(wl.bytes_per_row, frame_kernel / n)
    return wl
