"""configs[4]: the deep hierarchy (all dirty, change-driven, and the hierarchy frame with its cull)."""
import numpy as np

from .common import N_FRAMES, ROW_SUMMARY_SAVES, Workload, camera_frusta, flat_bytes_per_entity


def build_tree(ctx, args, rank=0, world=1):
    import bevy_amd as B
    from bevy_amd import sharding, workloads as W
    tr = W.gen_tree(12, 4, (args.entities or 1_000_000) * (world if args.scaling == "weak" else 1))
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
    if args.tile_mode:
        ctx.debug_set_tile_mode(args.tile_mode)
    ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
    plan = ctx.debug_tile_plan()
    # the root moves every frame (a 40-byte dirty-row upload), so set_if_neq really rewrites every descendant
    root_t = [tr["translation"][:3].copy(), tr["translation"][:3] + np.float32(1.0)]
    moved = getattr(args, "tree_moved", "all")
    if moved != "all":
        # change-driven frames under StaticTransformOptimizations (the frame a game mostly runs): "subtree" = one node of level 5
        # moves (1 / 1024 of a 4-ary tree follows it), "leaves" = 10 000 random leaves move
        n, lv = tr["n"], tr["level_offsets"]
        rng = np.random.default_rng(3)
        rows = (np.array([int(lv[5]) + 17], np.uint32) if moved == "subtree"
                else np.sort(rng.choice(np.arange(int(lv[-2]), n), 10_000, replace=False)).astype(np.uint32))
        t3 = tr["translation"].reshape(n, 3)
        sets = [(np.ascontiguousarray(t3[rows] + np.float32(d)).reshape(-1), np.ascontiguousarray(tr["rotation"].reshape(n, 4)[rows]).reshape(-1),
                 np.ascontiguousarray(tr["scale"].reshape(n, 3)[rows]).reshape(-1)) for d in (0.0, 1.0)]
        ctx.upload_changed(np.ones(n, np.uint8))
        ctx.propagate(B.PROPAGATE_STATIC_OPT)

        def step(f):
            ctx.upload_transforms_indexed(rows, *sets[f & 1])
            ctx.propagate(B.PROPAGATE_STATIC_OPT)
        config = {"workload": f"gen_tree(12,4) truncated to {n_global} nodes, StaticTransformOptimizations enabled, per frame "
                              + ("ONE node of level 5 moves (1 / 1024 of the tree follows)" if moved == "subtree" else "10 000 random leaves move")
                              + ": mi_upload_transforms_indexed + mi_propagate(MI_PROPAGATE_STATIC_OPT) = mark_dirty_trees + the tile launch",
                  "nodes": n_global, "moved_rows": int(len(rows)), "tile_plan": plan}
        # algorithmic bytes of a change-driven tile launch: every tile's descriptor (64 B) and flags pre-test (chain change bytes + top
        # marks, <= 136 B), plus the all-dirty 141 B for the rows that are re-evaluated (the moved rows' subtrees / the moved leaves and
        # the marked ancestors' tiles are a superset: counted as the rows below the moved ones only -- a lower bound)
        follows = (tr["n"] // 1024 if moved == "subtree" else len(rows))
        alg = (plan["tiles"] * 200.0 + follows * 141.0) / tr["n"]
        wl = Workload("tree_" + moved, step, tr["n"], alg, "k_propagate_fans", config, "nodes/sec through change-driven hierarchy propagate", "nodes/s",
                      kernels=["k_propagate_fans", "k_mark_dirty"])
        wl.tree = tr
        wl.kernel_name = "k_propagate_fans<false>"
        return wl

    if getattr(args, "tree_cull", False):
        # the hierarchy FRAME: every node carries a unit-cube Aabb; one call = the tile launch (every Transform counts as changed) + the
        # cull launch behind it (reset + check_visibility + mark-newly-hidden over the GlobalTransforms just written) + the deferred compaction
        from bevy_amd import api
        n = tr["n"]
        ctx.debug_set_row_summary(args.row_summary)
        tcl = getattr(args, "tree_cull_launches", 0)
        fused = tcl == 1 or (tcl == 0 and (args.views or 1) == 1)
        ctx.debug_set_tree_cull({0: 0, 1: 2, 2: 1}[tcl])
        ctx.upload_bounds(np.zeros(3 * n, np.float32), np.full(3 * n, 0.5, np.float32), np.full(n, 0x05, np.uint8), np.ones(n, np.uint32))
        n_views = args.views or 1
        frames = [api.PreparedFrusta(camera_frusta(n_views, f)) for f in range(N_FRAMES)]
        more = 0 if args.inline_compaction else B.CULL_MORE_FRAMES

        def step_frame(f):
            ctx.upload_transforms(root_t[f & 1], tr["rotation"][:4], tr["scale"][:3], first_row=0)
            ctx.propagate_and_cull(frames[f % N_FRAMES], flags=B.CULL_END_FRAME | more)
        config = {"workload": f"gen_tree(12,4) truncated to {n_global} nodes, every node with an Aabb, {n_views} camera frustum(s): the hierarchy frame in one "
                              "call -- mi_propagate_and_cull = subtree-tile propagation (root moved, every Transform counts as changed) "
                              + ("in which every tile also runs the visibility systems over its own rows (k_propagate_fans<true, true>)" if fused
                                 else "+ the cull launch over the GlobalTransforms it wrote") + " + VisibleEntities compaction",
                  "baseline_config": "BASELINE.json configs[4] + the cull of configs[1]", "nodes": n_global, "views": n_views, "tile_plan": plan,
                  "row_summary": args.row_summary == 0}
        # propagate 141 B per node (above) + cull with G resident: read G 48 + Aabb 24 + flags 1 + layers 4 + vv 1, write vv 1 + masks
        config["bytes_per_node"] = {"tile_launch": 141.0, "cull_launch_G_resident": flat_bytes_per_entity(n_views, False)}
        wl = Workload("tree_frame", step_frame, tr["n"], 141.0, "k_propagate_fans", config,
                      "nodes/sec through hierarchy propagate + cull", "nodes/s", kernels=["k_propagate_fans", "k_cull", "k_compact_fast"])
        wl.tree = tr
        wl.kernel_name = "k_propagate_fans<true,true>" if fused else "k_propagate_fans<true> + k_frame<0>"
        if fused:  # + read Aabb 24 + flags 1 + layers 4 (summarised: 0.5) + vv 1, write vv 1 + masks
            wl.bytes_per_row = 141.0 + 31.0 + (n_views + 1) / 8.0 + n_views / 64.0
            if args.row_summary == 0:
                wl.layout_bytes_per_row = wl.bytes_per_row - ROW_SUMMARY_SAVES
        return wl

    def step(f):
        ctx.upload_transforms(root_t[f & 1], tr["rotation"][:4], tr["scale"][:3], first_row=0)
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    config = {"workload": f"gen_tree(12,4) truncated to {n_global} nodes ({len(tr['level_offsets']) - 1} levels), root moved "
                          "every frame (dirty-row upload), subtree-tile propagation"
                          + (f", sharded by root subtree over {world} GPUs (this rank holds {tr['n']} rows, no collective)" if world > 1 else ""),
              "baseline_config": "BASELINE.json configs[4]", "nodes": n_global, "parallelism": f"root-subtree shard x{world}", "tile_plan": plan}
    # T 40, parent_idx 4, old G 48 (set_if_neq), G 48, changed byte 1
    wl = Workload("tree", step, tr["n"], 141.0, "k_propagate_fans", config, "nodes/sec through hierarchy propagate", "nodes/s",
                  kernels=["k_propagate_fans", "k_propagate_stream"])
    wl.tree = tr
    wl.kernel_name = "k_propagate_level<false>" if args.tile_mode == 1 else "k_propagate_fans<true>"
    if args.tile_mode == 1:  # (the level sweep's launches are timed in their own slot: one per level, the rows of a frame between them)
        wl.dominant = "k_propagate_stream"
        wl.rows = tr["n"] / max(1, len(tr["level_offsets"]) - 1)
    wl.global_units = n_global  # every node is owned by exactly one rank (replicated top rows are recomputed, not counted)
    return wl


def build_tree_shape(ctx, args):
    """The reference's own hierarchy stress shapes (examples/stress_tests/transform_hierarchy.rs:29-160; bevy_amd.workloads.hierarchy_shape)
    and SURVEY 8(d) config 5's other sizes.  --tree-shape-frame all: every root moves and every Transform counts as changed (the
    all-dirty frame of configs[4] on this shape); movers: the frame the example itself runs -- its `update` system rewrote the
    Transforms of the nodes that carry UpdateValue, StaticTransformOptimizations enabled (the Bevy default)."""
    import bevy_amd as B
    from bevy_amd import workloads as W
    name = args.tree_shape
    sh = W.hierarchy_shape(name)
    n = sh["n"]
    ctx.resize(n)
    ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
    if args.tile_mode:
        ctx.debug_set_tile_mode(args.tile_mode)
    ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
    plan = ctx.debug_tile_plan()
    frame_kind = getattr(args, "tree_shape_frame", "all")
    rows_of = lambda idx, col, w: np.ascontiguousarray(sh[col].reshape(n, w)[idx]).reshape(-1)
    base = {"shape": name, "nodes": n, "levels": sh["n_levels"], "movers": int(len(sh["movers"])), "tile_plan": plan,
            "reference": "examples/stress_tests/transform_hierarchy.rs:29-160" if not name.startswith("tree_4ary") else "SURVEY 8(d) config 5 (gen_tree(depth, 4) in full)"}
    if frame_kind == "movers" and len(sh["movers"]):
        mv = sh["movers"]
        rot, scl = rows_of(mv, "rotation", 4), rows_of(mv, "scale", 3)
        frames = [sh["mover_translation"](f) for f in range(1, 9)]
        ctx.propagate(B.PROPAGATE_ALL_DIRTY | B.PROPAGATE_STATIC_OPT)

        def step(f):
            ctx.upload_transforms_indexed(mv, frames[f % len(frames)], rot, scl)
            ctx.propagate(B.PROPAGATE_STATIC_OPT)
        # rows a movers frame has to touch: the movers' subtrees (closure under "child of"), 141 B each
        below = np.zeros(n, bool)
        below[mv] = True
        lv = sh["level_offsets"].astype(np.int64)
        for lo, hi in zip(lv[1:-1], lv[2:]):
            r = np.arange(lo, hi)
            below[r] |= below[sh["parent"][r]]
        touched = int(below.sum())
        config = dict(base, workload=f"{name}: {n} nodes, {sh['n_levels']} levels; per frame the `update` system's {len(mv)} movers get new translations "
                                     f"(mi_upload_transforms_indexed) + mi_propagate(MI_PROPAGATE_STATIC_OPT); {touched} rows lie in the movers' subtrees",
                      rows_in_moved_subtrees=touched)
        wl = Workload("tree_shape_" + name + "_movers", step, n, 141.0 * touched / n, "k_propagate_fans", config,
                      "nodes/sec through change-driven hierarchy propagate", "nodes/s", kernels=["k_propagate_fans", "k_mark_dirty", "k_propagate_stream", "k_level0_propagate"])
    else:
        roots = np.nonzero(sh["parent"] == W.NO_PARENT)[0].astype(np.uint32)
        rt, rr, rs = rows_of(roots, "translation", 3), rows_of(roots, "rotation", 4), rows_of(roots, "scale", 3)
        root_sets = [rt, (rt.reshape(-1, 3) + np.float32(1.0)).reshape(-1).copy()]

        def step(f):
            ctx.upload_transforms_indexed(roots, root_sets[f & 1], rr, rs)
            ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        config = dict(base, workload=f"{name}: {n} nodes, {sh['n_levels']} levels, {len(roots)} root(s) moved every frame, every Transform counts as changed "
                                     "(mi_propagate(MI_PROPAGATE_ALL_DIRTY))")
        wl = Workload("tree_shape_" + name, step, n, 141.0, "k_propagate_fans", config, "nodes/sec through hierarchy propagate", "nodes/s",
                      kernels=["k_propagate_fans", "k_propagate_stream", "k_level0_propagate", "k_mark_dirty"])
    wl.tree = sh
    wl.kernel_name = "k_propagate_fans<false>" if frame_kind == "movers" else "k_propagate_fans<true>"
    if name.startswith("humanoids") and not args.tile_mode:  # (a forest of small trees: a wave per tile)
        wl.kernel_name = "k_propagate_wave_tiles<false,true>" if frame_kind == "movers" else "k_propagate_wave_tiles<true,true>"
    if len(ctx.debug_strip_plan()[0]):  # (a deep or lopsided tree: strips)
        wl.kernel_name = "k_propagate_strips<false>" if frame_kind == "movers" else "k_propagate_strips<true>"
        config["strips"] = {"strips": int(len(ctx.debug_strip_plan()[0])), "rounds": int(ctx.debug_strip_plan()[2])}
    wl.frame_level_roofline = True  # several launches per frame on some shapes: priced per FRAME (sum of the frame's kernels), see roofline_frame
    return wl


def reprice_per_frame(roofline, frame):
    """A frame of several launches: `roofline` as measure.roofline_of made it divides the FRAME's bytes by ONE launch's average duration
    (0.61 for large_tree's four launches of 7.5 us: four times what the frame reaches).  Its achieved / frac become the frame's
    (roofline_frame), the launch's own figure stays as per_launch_*."""
    if not roofline or not frame or frame.get("launches_per_frame", 1.0) <= 1.05:
        return roofline
    r = dict(roofline)
    r["per_launch_avg_kernel_us"] = r.get("avg_kernel_us")
    for k in ("achieved", "frac", "frac_algorithmic"):
        r[k] = frame["achieved"] if k == "achieved" else frame["frac"]
    r.pop("rocprof_frac", None)
    r["avg_kernel_us"] = frame["kernels_us_per_frame"]
    r["launches_per_frame"] = frame["launches_per_frame"]
    r["note"] = "several dependent launches per frame: achieved / frac = the frame's bytes over the SUM of its kernels' time (roofline_frame)"
    return r


def roofline_frame(wl, prof, steps, profiled_blocks):
    """A hierarchy frame that is several launches (2 500 dependent levels, streamed wide levels): algorithmic bytes of the frame over
    the SUM of its kernels' device time per frame -- the figure comparable with the one-launch tree's `frac`."""
    from .common import HBM_PEAK_GBPS
    frames = steps * profiled_blocks
    total_us = sum(v["avg_us"] * v["launches"] for v in prof.values() if v["launches"])
    launches = sum(v["launches"] for v in prof.values() if v["launches"])
    if not total_us:
        return None
    us = total_us / frames
    alg = wl.bytes_per_row * wl.units
    return {"bound": "hbm", "kernels_us_per_frame": round(us, 3), "launches_per_frame": round(launches / frames, 2), "algorithmic_bytes_per_frame": int(alg),
            "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "per_kernel": {k: {"avg_us": round(v["avg_us"], 3), "launches_per_frame": round(v["launches"] / frames, 2)} for k, v in prof.items() if v["launches"]}}
