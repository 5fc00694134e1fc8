"""The BASELINE metric: propagate + cull + cluster in one frame (N = 1)."""
import numpy as np

from .common import N_FRAMES, ROW_SUMMARY_SAVES, Workload, flat_bytes_per_entity


def build_frame(ctx, args):
    import bevy_amd as B
    from bevy_amd import api, workloads as W
    n_ent = args.entities or 1_000_000
    sc, first_light, pr = W.frame_scene(n_ent, args.lights, args.meshes)
    n_rows = sc["n"]
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ctx.resize(n_rows)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    # the lights' bounding Spheres follow their rows (MI_SPHERE_AT_TRANSLATION: what the plugin uploads, so that a moved light is
    # not a bounds change): centre = the row's GlobalTransform translation = the position the scene (and the CPU baseline) holds
    c_dev, h_dev = sc["aabb_center"].reshape(-1, 3).copy(), sc["aabb_half"].reshape(-1, 3).copy()
    c_dev[first_light:] = 0.0
    h_dev[first_light:, 1] = np.frombuffer(np.uint32(0x7FC0A11D).tobytes(), np.float32)[0]
    import os
    if os.environ.get("MI_EXP_LIGHTS_HIDDEN") == "1":  # (an experiment: the lights' InheritedVisibility off -- the launch carries the walk's code, and
        sc["flags"][first_light:] &= np.uint8(0xFE)     # nothing walks: what the walk-carrying kernel variant costs by itself)
    ctx.debug_set_row_summary(args.row_summary)
    ctx.debug_set_walk_inrow(getattr(args, "walk_inrow", 0))
    ctx.upload_bounds(c_dev.reshape(-1), h_dev.reshape(-1), sc["flags"], sc["layers"])
    ctx.cluster_upload_objects(pr)
    ctx.cluster_bind_objects_to_rows(first_light, args.lights)
    frames, views, keep = [], [], []
    for f in range(N_FRAMES):
        cam = W.many_cubes_camera(f)
        fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
        frames.append(api.PreparedFrusta(fr))
        # ClusterConfig::XYZ{(16,9,24), first_slice_depth 5.0, Constant(1000.0), dynamic_resizing: false}
        v, k = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0, with_spheres=False)
        views.append(v)
        keep.append(k)
    more = 0 if args.inline_compaction else B.CULL_MORE_FRAMES

    separate = bool(getattr(args, "separate_cluster_calls", False))
    concurrent = B.CULL_CLUSTERS_CONCURRENT if getattr(args, "concurrent_clusters", False) else 0

    def step(f):
        i = f % N_FRAMES
        ctx.cluster_upload_view(views[i])
        if separate:
            ctx.propagate_and_cull(frames[i], flags=B.CULL_END_FRAME | more)
            ctx.cluster_assign_resident()
        else:
            ctx.propagate_and_cull(frames[i], flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS | concurrent | more)

    config = {"workload": f"propagate+cull+cluster 16x9x24 in ONE frame, ONE context: {n_ent} many_cubes entities (configs[1]) + {args.meshes} meshes "
                          f"and {args.lights} point lights of the many_lights shape (configs[2]; range 0.3, shell R = 50; lights are rows with a "
                          f"bounding Sphere) = {n_rows} rows, 1 camera, all Transforms dirty, columns resident in HBM: fused frame kernel "
                          "(propagate + reset + frustum cull + mark-newly-hidden) + VisibleEntities compaction"
                          + (" (deferred into the next frame's launch)" if more else "")
                          + " + device-side gather of the visible lights + assign_objects_to_clusters on 16x9x24 clusters "
                            "(ClusterConfig::XYZ, first slice 5.0, far Constant(1000))"
                          + (" -- two calls" if separate else " -- ONE call (MI_CULL_WITH_CLUSTERS), the assignment enqueued behind the cull"
                             if not concurrent else " -- ONE call (MI_CULL_WITH_CLUSTERS | MI_CULL_CLUSTERS_CONCURRENT): the assignment "
                             "re-derives the lights' ViewVisibility with the cull's rule and runs on the cluster stream next to the frame kernel"),
              "baseline_config": "BASELINE.json configs[1] + configs[2] in one frame; value counts the entities of configs[1] only",
              "entities": n_ent, "rows_per_frame": n_rows, "lights": args.lights, "meshes": args.meshes, "views": 1,
              "deferred_compaction": bool(more), "parallelism": "1 GPU", "row_summary": args.row_summary == 0}
    wl = Workload("frame", step, n_ent, flat_bytes_per_entity(1, True), "k_flat_propagate_cull", config,
                  "entities/sec through propagate+cull+cluster at 1M entities", "entities/s", rows=n_rows,
                  kernels=["k_flat_propagate_cull", "k_compact_fast", "k_cluster_walk", "k_cluster_fill"])
    wl.scene, wl.first_light, wl.pos_range, wl.frusta0, wl.keep = sc, first_light, pr, frames[0].array, (views, keep)
    # "k_flat_propagate_cull" is the library's timer slot; the symbol rocprofv3 shows is the k_frame instantiation
    # <PROPAGATE, INLINE_VIEWS, WALK>: the cluster walk rides in the launch unless it runs as calls or a stream of its own
    wl.kernel_name = "k_frame<1,true,0>" if (separate or concurrent) else "k_frame<1,true,1>"
    if args.row_summary == 0:
        wl.layout_bytes_per_row = wl.bytes_per_row - ROW_SUMMARY_SAVES  # (every wave of this scene but three is uniform)
    return wl
