"""GPU tests of the cluster stage beyond bit-parity with the oracle (tests/test_gpu_parity.py has those):

* the oracle-independent invariants of tests/cluster_invariants.py over the HIP library's own output;
* the default ClusterConfig feedback loop through mi_cluster_assign_frame (assign.rs:324-404,810-811);
* lights as rows of the frame context (mi_cluster_bind_objects_to_rows): propagate + cull + gather-visible + assign in one
  context, against the reference's sequence check_visibility -> gather -> assign restated with the oracle;
* the BASELINE.json shapes at full size: 100 k lights / range 0.3 / R = 50 with the bench's camera, and the 1 M-node tree.
"""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O
import cluster_invariants as CI
from test_gpu_parity import ctx_factory, frusta_for, upload_scene, upload_tree, assert_bits  # noqa: F401
from test_cluster_invariants import CASES, PERSP, rand_lights, two_frames
from test_abi_and_host import ortho_clip_from_view

pytestmark = pytest.mark.gpu
F = np.float32


def both_views(cam, dims=(16, 9, 24), far=1000.0, fsd=5.0, ortho=False, screen=(1920, 1080), view_mask=1):
    cfv = ortho_clip_from_view(-60.0, 60.0, -33.75, 33.75, 0.1, 1000.0) if ortho else api.perspective_clip_from_view(
        W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, screen[0], screen[1], dims, fsd, far, view_mask)
    ov = O.cluster_view_setup(cam, cfv, fr, screen[0], screen[1], dims, fsd, far, view_mask)
    return view, keep, ov


def assert_same_assignment(got, want):
    off, idx, counts, far, total = got
    eoff, eidx, ecounts, efar, etotal = want
    assert total == etotal, (total, etotal)
    assert np.array_equal(off, eoff), "cluster offsets"
    assert np.array_equal(idx, eidx), "cluster index lists (push order)"
    assert np.array_equal(counts, ecounts), "per-type counts"
    assert np.float32(far).tobytes() == np.float32(efar).tobytes(), (far, efar)


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_invariants_hold_for_the_hip_output(ctx_factory, name, case):
    case = dict(case)
    lights = case.pop("lights")
    view, keep, ov = both_views(case.pop("cam", W.many_cubes_camera(0)), **case)
    ctx = ctx_factory()
    got = ctx.cluster_assign(view, lights)
    CI.check_all(view, lights, None, None, *got)          # the HIP output on its own, against the float64 definitions
    assert_same_assignment(got, O.assign_objects_to_clusters(ov, lights))


def test_assign_frame_default_config_feedback(ctx_factory):
    """mi_cluster_assign_frame = one view of one frame of the system, default ClusterConfig: frame 0 runs on far_z = 1000,
    frame 1 on last frame's farthest_z; with big lights frame 0 overflows MAX_INDICES and frame 1 runs on a coarser grid."""
    cam = W.many_cubes_camera(0)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    for lights in (W.many_lights(100_000, 50.0, 0.3), W.many_lights(20_000, 50.0, 6.0)):
        want = two_frames(lights, cam)
        ctx = ctx_factory()
        ctx.cluster_upload_objects(lights)
        cfg, hist = api.cluster_config_default(), api.ClusterHistory()
        for f, (req, far, oview, eoff, eidx, ecounts, efar, etotal) in enumerate(want):
            view, active = ctx.cluster_assign_frame(cfg, hist, cam, cfv, fr, 1920, 1080)
            assert active and tuple(view.dims) == tuple(oview.dims), (f, tuple(view.dims), tuple(oview.dims))
            assert F(view.far_).tobytes() == F(oview.far_).tobytes() and F(view.near_).tobytes() == F(oview.near_).tobytes()
            got = ctx.cluster_download(view.n_clusters)
            assert_same_assignment(got, (eoff, eidx, ecounts, efar, etotal))
            assert hist.has_farthest_z and hist.has_total_cluster_index_count
            assert hist.total_cluster_index_count == etotal and F(hist.farthest_z).tobytes() == F(efar).tobytes()
            CI.check_all(view, lights, None, None, *got, superset=False)
    # ClusterConfig::None clears the view and leaves the statistics alone
    none = api.cluster_config_default()
    none.kind = api.CLUSTER_CONFIG_NONE
    before = (hist.farthest_z, hist.total_cluster_index_count)
    _, active = ctx.cluster_assign_frame(none, hist, cam, cfv, fr, 1920, 1080)
    assert not active and (hist.farthest_z, hist.total_cluster_index_count) == before


def reference_sequence(sc, first_light, pr, frusta, cam, types=None, layers=None, sc_sincos=None):
    """check_visibility over every row, then the gather of the visible lights (assign.rs:190-215) and the assignment of the
    gathered list -- with the oracle.  Returns the assignment in terms of LIGHT indices (row - first_light)."""
    n = sc["n"]
    g, vv, vis, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"],
                                 sc["layers"], np.zeros(n, np.uint8), frusta)
    n_l = len(pr) // 4
    visible = (vv[first_light:first_light + n_l] & 1) != 0
    keep = np.nonzero(visible)[0]
    pr_g = np.asarray(pr, F).reshape(-1, 4)[keep].copy()
    pr_g[:, :3] = g.reshape(-1, 12)[first_light + keep, 9:12]   # GlobalTransform::from_translation(transform.translation())
    spot_dir = None
    if types is not None and (types == 1).any():               # transform.back() = (matrix3 * Vec3::Z).normalize()
        z = g.reshape(-1, 12)[first_light + keep, 6:9].astype(F)
        ln = np.sqrt((z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1]) + z[:, 2] * z[:, 2]).astype(F)
        spot_dir = (z * (F(1.0) / ln)[:, None]).astype(F).reshape(-1)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ov = O.cluster_view_setup(cam, cfv, frusta[:24], 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    off, idx, counts, far, total = O.assign_objects_to_clusters(
        ov, pr_g.reshape(-1), None if types is None else types[keep].copy(), None if layers is None else layers[keep].copy(), spot_dir,
        None if sc_sincos is None else sc_sincos.reshape(-1, 2)[keep].copy().reshape(-1))
    return off, keep[idx].astype(np.uint32), counts, far, total, visible, vv, g


@pytest.mark.parametrize("mode", ["separate_calls", "with_clusters", "with_clusters_concurrent", "with_clusters_behind_the_cull"])
@pytest.mark.parametrize("spots", [False, True])
def test_lights_as_rows_of_the_frame_context(ctx_factory, spots, mode):
    """The whole metric frame in one context: rows = cubes + meshes + lights; the cull decides every row's ViewVisibility
    (lights through their bounding Sphere), the assignment gathers the visible lights ON THE DEVICE and assigns them.  Equal,
    list for list, to the reference's sequence restated with the oracle, whichever way the frame is driven:
      separate_calls                  mi_propagate_and_cull, then mi_cluster_assign_resident (reads the ViewVisibility column);
      with_clusters                   ONE call, MI_CULL_WITH_CLUSTERS: the assignment is enqueued behind the frame kernel;
      with_clusters_concurrent        ... | MI_CULL_CLUSTERS_CONCURRENT: the assignment re-derives the lights' visibility and
                                      runs on the cluster stream next to the frame kernel;
      with_clusters_behind_the_cull   MI_CULL_WITH_CLUSTERS on a call that does not close the frame (no MI_CULL_END_FRAME): the
                                      assignment runs behind the cull and reads the column.
    Lights move between the frames (dirty-row uploads on the main stream while the cluster stream may still be busy)."""
    sc, first_light, pr = W.frame_scene(60_000, 30_000, 3_000, light_range=1.5, ragged_flags=True)
    n_l = len(pr) // 4
    rng = np.random.default_rng(3)
    types = layers = sincos = None
    if spots:
        types = np.sort(rng.integers(0, 6, n_l)).astype(np.uint8)   # every kind in gather order: points, spots, rect lights, reflection probes, irradiance volumes, decals
        layers = np.where(rng.random(n_l) < 0.1, 2, 1).astype(np.uint32)
        ang = rng.uniform(0.1, 1.2, n_l).astype(F)
        sincos = np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(F).reshape(-1)
        hidden = first_light + rng.integers(0, n_l, 500)             # some lights are hidden by inheritance
        sc["flags"][hidden] &= ~np.uint8(0x01)
        if mode != "with_clusters_behind_the_cull":  # (their ViewVisibility is only set when the frame is closed)
            ncc = first_light + rng.integers(0, n_l, 300)            # ... and some are NoCpuCulling (gpu-culling rule)
            sc["flags"][ncc] |= np.uint8(0x10)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.cluster_upload_objects(pr, types, layers, None, sincos)
    ctx.cluster_bind_objects_to_rows(first_light, n_l)
    ctx.profile_filter(None)
    ctx.profile_enable(True)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    t3 = sc["translation"].reshape(-1, 3)
    c3 = sc["aabb_center"].reshape(-1, 3)
    for f in (0, 40, 41, 42):
        cam = W.many_cubes_camera(f, yaw=0.3 * f)
        frusta = frusta_for([cam])
        view, keep = api.cluster_view_build(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
        if f:  # a few hundred lights move: Transform (dirty rows) and bounding Sphere
            moved = np.unique(first_light + rng.integers(0, n_l, 400)).astype(np.uint32)
            t3[moved] = (t3[moved] * F(0.97)).astype(F)
            c3[moved] = t3[moved]
            ctx.upload_transforms_indexed(moved, t3[moved].reshape(-1), sc["rotation"].reshape(-1, 4)[moved].reshape(-1),
                                          sc["scale"].reshape(-1, 3)[moved].reshape(-1))
            lo, hi = int(moved.min()), int(moved.max()) + 1
            ctx.upload_bounds(c3[lo:hi].reshape(-1), sc["aabb_half"].reshape(-1, 3)[lo:hi].reshape(-1), sc["flags"][lo:hi], sc["layers"][lo:hi],
                              first_row=lo)
        ctx.cluster_upload_view(view)
        ctx.upload_view_visibility(np.zeros(sc["n"], np.uint8))
        if mode == "separate_calls":
            ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES)
            ctx.cluster_assign_resident()
        elif mode == "with_clusters":
            ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES | B.CULL_WITH_CLUSTERS)
        elif mode == "with_clusters_concurrent":
            ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES | B.CULL_WITH_CLUSTERS | B.CULL_CLUSTERS_CONCURRENT)
        else:
            ctx.propagate(B.PROPAGATE_ALL_DIRTY)
            ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_WITH_CLUSTERS)
            ctx.visibility_end_frame()
        got = ctx.cluster_download(view.n_clusters)
        eoff, eidx, ecounts, efar, etotal, visible, vv, g = reference_sequence(sc, first_light, pr, frusta, cam, types, layers, sincos)
        assert 0 < visible.sum() < n_l and etotal > 0
        assert_same_assignment(got, (eoff, eidx, ecounts, efar, etotal))
        assert_bits(ctx.download_view_visibility()[0], vv, "ViewVisibility of every row (cubes, meshes, lights)")
        pr_rows = np.asarray(pr, F).reshape(-1, 4).copy()
        pr_rows[:, :3] = g.reshape(-1, 12)[first_light:first_light + n_l, 9:12]
        CI.check_all(view, pr_rows.reshape(-1), types, layers, *got, visible=visible, superset=not spots)
    ctx.synchronize()
    prof = ctx.profile_read()
    if mode == "with_clusters":
        # the ONE-launch path: the walk rode in the frame kernel -- with spot lights too (its cone-test variant, round 4) -- so the walk
        # kernel of its own never ran
        assert "k_cluster_walk" not in prof and prof["k_flat_propagate_cull"]["launches"] == 4, prof
    elif mode == "separate_calls":
        assert prof["k_cluster_walk"]["launches"] == 4, prof


def test_two_clustered_cameras_through_view_slots(ctx_factory):
    """Split screen: two cameras with Clusters of their own over the same lights (assign.rs:324-486 runs per view).  Slot 0's walk rides
    in the frame kernel (MI_CULL_WITH_CLUSTERS), slot 1 is assigned behind it (mi_cluster_select_view + mi_cluster_assign_resident):
    every frame both assignments equal the reference's sequence for their camera -- lists, counts, farthest_z -- and a slot keeps its
    results while the other one is selected.  Spot lights and other RenderLayers among the objects; different grids per view."""
    sc, first_light, pr = W.frame_scene(40_000, 20_000, 2_000, light_range=1.5, ragged_flags=True)
    n_l = len(pr) // 4
    rng = np.random.default_rng(8)
    types = np.sort(rng.integers(0, 2, n_l)).astype(np.uint8)
    layers = np.where(rng.random(n_l) < 0.1, 2, 1).astype(np.uint32)
    ang = rng.uniform(0.1, 1.2, n_l).astype(F)
    sincos = np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(F).reshape(-1)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.cluster_upload_objects(pr, types, layers, None, sincos)
    ctx.cluster_bind_objects_to_rows(first_light, n_l)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    dims = [(16, 9, 24), (12, 7, 16)]
    for f in (0, 30, 31):
        cams = [W.many_cubes_camera(f, yaw=0.2 * f), W.many_cubes_camera(f, yaw=2.0 + 0.1 * f, position=(4.0, 1.0, -3.0))]
        frusta = frusta_for(cams)
        views = []
        for k in range(2):
            v, keep = api.cluster_view_build(cams[k], cfv, frusta[24 * k:24 * k + 24], 1920, 1080, dims[k], 5.0, 1000.0)
            views.append((v, keep))
        ctx.upload_view_visibility(np.zeros(sc["n"], np.uint8))
        ctx.cluster_select_view(0)
        ctx.cluster_upload_view(views[0][0])
        ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES | B.CULL_WITH_CLUSTERS)
        ctx.cluster_select_view(1)
        ctx.cluster_upload_view(views[1][0])
        ctx.cluster_assign_resident()
        got1 = ctx.cluster_download(views[1][0].n_clusters)
        ctx.cluster_select_view(0)
        got0 = ctx.cluster_download(views[0][0].n_clusters)
        # the reference: ViewVisibility is the OR over both cameras; every view gathers the same visible lights
        n = sc["n"]
        g, vv, vis, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"],
                                     np.zeros(n, np.uint8), frusta)
        visible = (vv[first_light:first_light + n_l] & 1) != 0
        keep_l = np.nonzero(visible)[0]
        pr_g = np.asarray(pr, F).reshape(-1, 4)[keep_l].copy()
        pr_g[:, :3] = g.reshape(-1, 12)[first_light + keep_l, 9:12]
        z = g.reshape(-1, 12)[first_light + keep_l, 6:9].astype(F)
        ln = np.sqrt((z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1]) + z[:, 2] * z[:, 2]).astype(F)
        spot_dir = (z * (F(1.0) / ln)[:, None]).astype(F).reshape(-1)
        for k, got in ((0, got0), (1, got1)):
            ov = O.cluster_view_setup(cams[k], cfv, frusta[24 * k:24 * k + 24], 1920, 1080, dims[k], 5.0, 1000.0)
            off, idx, counts, far, total = O.assign_objects_to_clusters(ov, pr_g.reshape(-1), types[keep_l].copy(), layers[keep_l].copy(), spot_dir,
                                                                        sincos.reshape(-1, 2)[keep_l].copy().reshape(-1))
            assert total > 0
            assert_same_assignment(got, (off, keep_l[idx].astype(np.uint32), counts, far, total))
        assert_bits(ctx.download_view_visibility()[0], vv, "ViewVisibility (the OR over both cameras)")


def test_baseline_lights_config_at_full_size(ctx_factory):
    """BASELINE.json configs[2] exactly as bench.py runs it: 100 000 point lights (range 0.3, shell R = 50) + 10 000 meshes,
    16 x 9 x 24 clusters, the bench's camera; plus the 1 M cubes of configs[1] in the same context."""
    sc, first_light, pr = W.frame_scene(1_000_000, 100_000, 10_000)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.cluster_upload_objects(pr)
    ctx.cluster_bind_objects_to_rows(first_light, 100_000)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(7)
    frusta = frusta_for([cam])
    view, keep = api.cluster_view_build(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    ctx.cluster_upload_view(view)
    for _ in range(3):  # what bench.py's frame step is: one call, the assignment concurrent with the frame kernel
        ctx.upload_view_visibility(np.zeros(sc["n"], np.uint8))
        ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES | B.CULL_WITH_CLUSTERS)
    got = ctx.cluster_download(view.n_clusters)
    eoff, eidx, ecounts, efar, etotal, visible, vv, g = reference_sequence(sc, first_light, pr, frusta, cam)
    assert_same_assignment(got, (eoff, eidx, ecounts, efar, etotal))
    assert_bits(ctx.download_view_visibility()[0], vv, "ViewVisibility")
    assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes()
    assert 3000 < visible.sum() < 8000 and etotal >= visible.sum()
    CI.check_all(view, pr, None, None, *got, visible=visible, superset=False)
    # the standalone object-list form of the same configuration (what bench.py --workload lights times)
    ctx2 = ctx_factory()
    assert_same_assignment(ctx2.cluster_assign(view, pr), O.assign_objects_to_clusters(
        O.cluster_view_setup(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0), pr))


@pytest.mark.parametrize("tile_mode", [0, 1])
def test_baseline_tree_config_at_full_size(ctx_factory, tile_mode):
    """BASELINE.json configs[4]: gen_tree(12, 4) truncated to 1 000 000 nodes, bit-exact against the oracle, then a
    partially dirty frame and a static frame.  tile_mode 0 = the path the library picks at this size (subtree tiles), 1 = the level-by-level sweep."""
    tr = W.gen_tree(12, 4, 1_000_000)
    assert tr["n"] == 1_000_000
    ctx = ctx_factory()
    ctx.debug_set_tile_mode(tile_mode)
    upload_tree(ctx, tr)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g, chg = ctx.download_global_transforms()
    rc, g0, chg0 = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert rc == 0
    bad = np.nonzero((g.view(np.uint32) != g0.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of 1 000 000 rows differ, first {bad[:5].tolist()}"
    assert_bits(chg, chg0, "change ticks")
    # the root moves (what bench.py does every frame): every descendant is rewritten
    t = tr["translation"].copy()
    t[:3] += F(1.0)
    ctx.upload_transforms(t[:3], tr["rotation"][:4], tr["scale"][:3], first_row=0)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g, chg = ctx.download_global_transforms()
    rc, g1, chg1 = O.propagate_transforms(tr["parent"], t, tr["rotation"], tr["scale"], global_in=g0)
    assert g.tobytes() == g1.tobytes() and chg1.all()
    assert_bits(chg, chg1, "change ticks after the root moved")
    # a few dirty nodes deep in the tree under the static-scene rule
    rows = np.array([5, 1400, 349_530, 999_999], np.uint32)
    t3 = t.reshape(-1, 3).copy()
    t3[rows] += F(0.5)
    ctx.upload_transforms_indexed(rows, t3[rows].reshape(-1), tr["rotation"].reshape(-1, 4)[rows].reshape(-1), tr["scale"].reshape(-1, 3)[rows].reshape(-1))
    ctx.propagate(B.PROPAGATE_STATIC_OPT)
    changed = np.zeros(tr["n"], np.uint8)
    changed[rows] = 1
    rc, g2, chg2 = O.propagate_transforms(tr["parent"], t3.reshape(-1), tr["rotation"], tr["scale"], global_in=g1, static_opt=True,
                                          tree_changed=O.mark_dirty_trees(tr["parent"], changed), transform_changed=changed)
    g, chg = ctx.download_global_transforms()
    assert g.tobytes() == g2.tobytes()
    assert_bits(chg, chg2, "change ticks of the sparse frame")


def test_sharded_assignment_over_a_one_rank_rccl_group(ctx_factory):
    """SURVEY.md 8e row 3 on the GPU box: bevy_amd.sharding.cluster_assign_sharded with the three collectives running over RCCL
    (a one-rank communicator: everything but the wire; world sizes 2 and 3 run over gloo in tests/test_sharding_clusters_gloo.py),
    against the unsharded HIP assignment and the oracle.  Unmeasured on more than one GPU."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from bevy_amd import sharding
    from test_sharding_clusters_gloo import scene
    pr, ty, layers, sd, sc = scene(20_000, 3_000, seed=4)
    view, keep, ov = both_views(W.many_cubes_camera(7))
    want = O.assign_objects_to_clusters(ov, pr, ty, layers, sd, sc)
    ctx = ctx_factory()
    assert_same_assignment(ctx.cluster_assign(view, pr, ty, layers, sd, sc), want)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = sharding.cluster_assign_sharded(ctx, view, pr, ty, layers, sd, sc, world=1, rank=0, device=torch.device("cuda", 0), always_exchange=True)
    finally:
        dist.destroy_process_group()
    assert_same_assignment(got, want)
    assert got[4] > 10_000


def test_cluster_objects_on_render_layers_32_to_63(ctx_factory):
    """RenderLayers::intersects compares the masks word by word (render_layers.rs:121-135): with mi_cluster_upload_object_layers_hi and
    mi_cluster_view.view_layer_mask_hi the first u64 word is covered for cluster objects as it is for rows and views.  Objects on layer
    0, on layer 40, on both, on layer 5 only; views on layer 0, on layer 40, on both: the lists equal the oracle's -- through
    mi_cluster_assign_resident and with the walk riding in the frame kernel (objects bound to rows)."""
    sc, first_light, pr = W.frame_scene(30_000, 10_000, 3_000, light_range=2.5)
    n_l = len(pr) // 4
    rng = np.random.default_rng(21)
    kind = rng.integers(0, 4, n_l)
    lo = np.select([kind == 0, kind == 1, kind == 2], [1, 0, 1], default=1 << 5).astype(np.uint32)
    hi = np.select([kind == 1, kind == 2], [1 << 8, 1 << 8], default=0).astype(np.uint32)  # layer 40 = bit 8 of the second word
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(3, yaw=0.4)
    frusta = frusta_for([cam])
    n = sc["n"]
    g, vv, vis, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"],
                                 np.zeros(n, np.uint8), frusta)
    keep_l = np.nonzero((vv[first_light:first_light + n_l] & 1) != 0)[0]
    pr_g = np.asarray(pr, F).reshape(-1, 4)[keep_l].copy()
    pr_g[:, :3] = g.reshape(-1, 12)[first_light + keep_l, 9:12]
    seen = set()
    for view_lo, view_hi in ((1, 0), (0, 1 << 8), (1, 1 << 8), (1 << 5, 1 << 9)):
        view, keep = api.cluster_view_build(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0, view_layer_mask=view_lo)
        view.view_layer_mask_hi = view_hi
        ov = O.cluster_view_setup(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0, view_layer_mask=view_lo)
        ov.view_layer_mask_hi = view_hi
        off, idx, counts, far, total = O.assign_objects_to_clusters(ov, pr_g.reshape(-1), None, lo[keep_l].copy(), layer_mask_hi=hi[keep_l].copy())
        seen.add(int(total))
        for ride in (False, True):
            ctx.cluster_upload_objects(pr, None, lo)
            ctx.cluster_upload_object_layers_hi(hi)
            ctx.cluster_bind_objects_to_rows(first_light, n_l)
            ctx.upload_view_visibility(np.zeros(n, np.uint8))
            ctx.cluster_upload_view(view)
            if ride:
                ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS)
            else:
                ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
                ctx.cluster_assign_resident()
            goff, gidx, gcounts, gfar, gtotal = ctx.cluster_download(view.n_clusters)
            assert gtotal == total, (view_lo, view_hi, ride, gtotal, total)
            assert np.array_equal(goff, off) and np.array_equal(gidx, keep_l[idx]), (view_lo, view_hi, ride)  # (object indices: positions in the upload)
    assert len(seen) == 4  # the four views really see different object sets
    # a new object upload clears the second word again
    ctx.cluster_upload_objects(pr, None, lo)
    view, keep = api.cluster_view_build(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0, view_layer_mask=0)
    view.view_layer_mask_hi = 1 << 8
    ctx.cluster_bind_objects_to_rows(first_light, n_l)
    ctx.cluster_upload_view(view)
    ctx.cluster_assign_resident()
    assert ctx.cluster_download(view.n_clusters)[4] == 0


def test_host_libm_is_checked_before_the_first_perspective_view(ctx_factory, monkeypatch):
    """view_z_to_z_slice's ln() (assign.rs:1057) is the one library call of the path: the device carries glibc's logf, and the library
    compares it with THIS host's logf when a context gets its first perspective view (3 297 probes).  They agree on the supported
    hosts (this box); when they do not -- forced here -- every assignment is refused with MI_ERR_DEVICE, i.e. handed to the stock
    system, while propagate and cull go on, and orthographic views (no ln) are still served."""
    sc = W.many_cubes(5_000)
    pr = rand_lights(300, 40.0, 6.0, 4)
    cam = W.many_cubes_camera(3)
    frusta = frusta_for([cam])
    view, keep, ov = both_views(cam)
    ok = ctx_factory()
    assert_same_assignment(ok.cluster_assign(view, pr), O.assign_objects_to_clusters(ov, pr))  # (the check passed on this host)
    monkeypatch.setenv("MI_DEBUG_FORCE_LIBM_MISMATCH", "1")
    bad = ctx_factory()
    upload_scene(bad, sc)
    for _ in range(2):  # the verdict sticks
        with pytest.raises(api.MiError) as e:
            bad.cluster_assign(view, pr)
        assert e.value.code == api.MI_ERR_DEVICE and "logf" in str(e.value)
    bad.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)  # the rest of the path is unaffected
    g_exp, vv_exp, vis_exp, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"],
                                             sc["layers"], np.zeros(sc["n"], np.uint8), frusta)
    assert_bits(bad.download_visibility(0), vis_exp[0], "cull on a context whose clusters fell back")
    oview, okeep, oov = both_views(cam, ortho=True)
    assert_same_assignment(bad.cluster_assign(oview, pr), O.assign_objects_to_clusters(oov, pr))
