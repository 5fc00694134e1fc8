"""the batching work-item build (binned phase behind the flat frame; sorted phases)."""
import numpy as np

from .common import ROW_SUMMARY_SAVES, N_FRAMES, Workload, camera_frusta, flat_bytes_per_entity


def build_batching(ctx, args):
    """the flat frame followed by the batching work-item build of the camera's list."""
    import bevy_amd as B
    from bevy_amd import api, workloads as W
    n = args.entities or 1_000_000
    sc = W.many_cubes(n)
    bs = W.batching_scene(n, n_sets=64, max_bins=40, seed=7)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
    ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
    frames = [api.PreparedFrusta(camera_frusta(1, f)) for f in range(N_FRAMES)]

    def step(f):
        ctx.propagate_and_cull(frames[f % N_FRAMES], flags=B.CULL_END_FRAME)
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
                  "entities/sec through propagate+cull+batch build", "entities/s",
                  kernels=["k_flat_propagate_cull", "k_compact_fast", "k_batch_hist", "k_batch_emit", "k_batch_scan", "k_batch_scatter",
                           "k_batch_bounds", "k_batch_plan"])
    wl.batch = (bs, rows)
    if getattr(args, "row_summary", 0) == 0:  # (the flat frame in front of the build reads the row summary like any other: moved < algorithmic)
        wl.layout_bytes_per_row = wl.bytes_per_row - ROW_SUMMARY_SAVES
    return wl


def build_batching_sorted(ctx, args):
    """Sorted phases (Transparent3d, the 2D phases): gpu_preprocessing::batch_and_prepare_sorted_render_phase over a phase of
    --sorted-items items in their sorted order.  A step = mi_batch_sorted_build: the items are staged (16 B each: they are the CPU's
    sorted phase) and the walk runs -- up to 8 192 items one workgroup reading the pinned staging block itself, beyond that tiles of
    4 096 items over the whole chip behind one H2D copy (kernels_sorted.hip)."""
    from bevy_amd import workloads as W
    n = getattr(args, "sorted_items", 0) or 65_536
    items = W.sorted_items(n, seed=5)
    ctx.resize(1)
    limit = getattr(args, "sorted_one_wg_limit", None)
    if limit is not None:
        ctx.debug_set_sorted_one_wg_limit(limit)

    def step(f):
        ctx.batch_sorted_build(items, True, False, False, None)
    tiled = n > min(8192, 8192 if limit is None else limit)
    config = {"workload": f"sorted render phase of {n} items (runs of equal batch-set / bin keys, some without an input index): "
                          "mi_batch_sorted_build = " + (f"H2D of the items + k_sorted_walk<16, true> + k_sorted_walk<16, false> (two launches, {(n + 4095) // 4096} tiles)"
                                                        if tiled else "k_sorted_walk (one workgroup, one launch, reading the items from the pinned staging block)"),
              "items": n, "tiled": tiled}
    # per item: read 16 (item), write a work item 8 (+ 20 B of metadata per batch, 24 + 8 per batch set)
    wl = Workload("batching_sorted", step, n, 16.0 + 8.0 + 8.0, "k_batch_sorted", config, "items/sec through the sorted-phase batch build", "items/s",
                  kernels=["k_batch_sorted", "k_batch_scan"])
    wl.sorted_items = items
    wl.kernel_name = "k_sorted_walk<16u, false>" if (tiled or n <= 4096) else "k_sorted_walk<32u, false>"
    return wl
