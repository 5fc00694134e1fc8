"""GPU parity of the world-sphere frame (kernels_flat.hip, k_frame_sph): the cull-only frame and the changed-rows frame evaluated
over the 16-byte (affine * aabb.center, |M3 * half_extents|) column instead of GlobalTransform + Aabb.

The column holds exactly the values check_visibility_cpu_culling computes per entity and frame (visibility/mod.rs:824-832), so every
result must stay bit-identical to the oracle -- and to the k_frame path (mi_debug_set_sphere_path(1)) -- whatever the sequence of
propagates, culls, uploads and moved rows in between: the tests below walk the column through its three states (invalid, valid but
for the rows of the last change mask, valid), with ballot-word masks (flat rows) and byte masks (hierarchies)."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32


def assert_bits(a, b, what):
    bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first rows {bad[:8].tolist()}"


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


def oracle_cull(sc, g, vv, frusta):
    """reset + check_visibility + gpu_culling + mark_newly_hidden over the given GlobalTransforms."""
    vv1 = O.reset_view_visibility(sc["flags"], vv)
    vv2, vis, chg = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], vv1, frusta)
    vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
    vv4, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
    return vv4, vis, chg | chg2 | chg3


def check_frame(ctx, vv_exp, vis_exp, chg_exp, what):
    for v in range(len(vis_exp)):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"{what}: view {v}")
        rows = ctx.download_visible_entities(v, 0)[1]
        assert np.array_equal(rows, np.nonzero(vis_exp[v])[0].astype(np.uint32)), f"{what}: VisibleEntities of view {v}"
    vv, chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, f"{what}: ViewVisibility")
    assert_bits(chg, chg_exp, f"{what}: ViewVisibility change ticks")


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 4097, 50_003])
@pytest.mark.parametrize("mode", [0, 2])
def test_static_frames_over_the_sphere_column(n, mode):
    """A scene in which nothing moves, a camera that does: frame 0 is the all-dirty propagate + cull (k_frame), frame 1 rebuilds the
    column (every row stale; mode 2: frame 0 already does), the frames after it read 22 bytes per row."""
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        ctx.debug_set_sphere_path(mode)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.ones(n, np.uint8))
        vv = np.zeros(n, np.uint8)
        for frame in range(5):
            ctx.propagate(0)  # frame 0: every row; later: nothing was marked
            frusta = frusta_for([W.many_cubes_camera(frame * 50), W.many_cubes_camera(frame * 50, yaw=1.3, position=(5.0, -3.0, 11.0))])
            ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"n={n} mode={mode} frame {frame}")
        assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_moving_rows_keep_the_sphere_column_current(fused, mode):
    """Every frame some rows move (mi_upload_transforms_indexed): unfused = mi_propagate(0) + mi_cull (the stale rows are the
    propagate's change words), fused = mi_propagate_and_cull(MI_CULL_CHANGED_ROWS) (the rows are propagated by the same kernel).
    In between: a frame in which nothing moves, two propagates without a cull (the first change mask is lost: rebuild), a
    bounds upload (rebuild), an all-dirty frame."""
    n = 30_011
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    t = sc["translation"].reshape(n, 3).copy()
    r4, s3 = sc["rotation"].reshape(n, 4), sc["scale"].reshape(n, 3)
    rng = np.random.default_rng(3)
    with api.Context(0) as ctx:
        ctx.debug_set_sphere_path(mode)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.ones(n, np.uint8))
        ctx.propagate(0)
        g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
        vv = np.zeros(n, np.uint8)
        script = ["move", "move", "still", "move", "double", "move", "bounds", "move", "all", "move", "move"]
        for frame, what in enumerate(script):
            moved = np.zeros(0, np.uint32)
            if what in ("move", "double"):
                k = 1 + frame * 97 % 2000
                moved = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
                t[moved] += rng.normal(0.0, 4.0, (k, 3)).astype(F)
                ctx.upload_transforms_indexed(moved, t[moved].reshape(-1), r4[moved].reshape(-1), s3[moved].reshape(-1))
            if what == "double":  # a propagate whose change mask no cull consumes, then more moved rows
                ctx.propagate(0)
                more = np.sort(rng.choice(n, 333, replace=False)).astype(np.uint32)
                t[more] -= F(2.5)
                ctx.upload_transforms_indexed(more, t[more].reshape(-1), r4[more].reshape(-1), s3[more].reshape(-1))
            if what == "bounds":
                sc["aabb_half"] = (sc["aabb_half"] * F(1.25)).astype(F)
                ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            frusta = frusta_for([W.many_cubes_camera(frame * 30), W.many_cubes_camera(frame * 30, yaw=2.1)])
            if what == "all":
                ctx.upload_changed(np.ones(n, np.uint8))
            if fused:
                ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
            else:
                ctx.propagate(0)
                ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"fused={fused} mode={mode} frame {frame} ({what})")
            assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes(), f"frame {frame}: GlobalTransform"


@pytest.mark.parametrize("static_opt", [False, True])
def test_hierarchy_change_bytes_feed_the_sphere_column(static_opt):
    """With a hierarchy the propagate leaves its change mask as a byte per row: a moved inner node makes its whole subtree stale."""
    tr = W.gen_tree(8, 4)
    n = tr["n"]
    t = tr["translation"].reshape(n, 3).copy()
    r4, s3 = tr["rotation"].reshape(n, 4), tr["scale"].reshape(n, 3)
    sc = dict(aabb_center=np.zeros(3 * n, F), aabb_half=np.full(3 * n, 0.5, F), flags=np.full(n, 0x05, np.uint8), layers=np.ones(n, np.uint32))
    pflags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    with api.Context(0) as ctx:
        ctx.debug_set_sphere_path(2)
        ctx.resize(n)
        ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
        ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.ones(n, np.uint8))
        vv = np.zeros(n, np.uint8)
        for frame, node in enumerate([None, None, 5, None, 100, 0, None, 21]):
            if node is not None:
                t[node] += F(3.0)
                ctx.upload_transforms_indexed(np.array([node], np.uint32), t[node], r4[node], s3[node])
            ctx.propagate(pflags)
            frusta = frusta_for([W.many_cubes_camera(frame * 20, position=(0.0, 0.0, 150.0))])
            ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            _, g, _ = O.propagate_transforms(tr["parent"], t.reshape(-1), tr["rotation"], tr["scale"])
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"static_opt={static_opt} frame {frame}")
            assert 0 < vis[0].sum() < n


def test_ranges_classes_and_many_views_over_the_sphere_column():
    """VisibilityRange rows (model position = the sphere's centre with use_aabb, the GlobalTransform's translation without),
    VisibilityClass segments, a camera with NoCpuCulling and 11 views (the device view table) on the sphere path; a frame with a
    shadow view goes back to k_frame and leaves the column untouched."""
    n = 20_000
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    rnd = W.splitmix64(5, n)
    ranged = rnd % np.uint64(3) == 0
    sc["flags"][ranged] |= np.uint8(B.FLAG_HAS_VISIBILITY_RANGE)
    sc["flags"][ranged & (rnd % np.uint64(5) == 0)] |= np.uint8(B.FLAG_RANGE_USE_AABB)
    sc["flags"][rnd % np.uint64(4) != 1] |= np.uint8(B.FLAG_SHADOW_CASTER)
    lo = (W.uniform01(6, n) * 70.0).astype(F)
    ranges = np.stack([lo, lo + (W.uniform01(7, n) * 40.0).astype(F)], axis=1).reshape(-1).copy()
    class_mask = np.where(rnd % np.uint64(5) == 0, 0b101, np.where(rnd % np.uint64(5) == 1, 0b100, 0b001)).astype(np.uint32)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    for n_views in (3, 11):
        cams = [W.many_cubes_camera(k, yaw=0.6 * k, position=(3.0 * k, -2.0, 1.0 * k)) for k in range(n_views)]
        fr = frusta_for(cams)
        pos = np.array([c[9:12] for c in cams], F)
        vflags = [B.VIEW_FLAG_RANGES if k % 3 != 2 else 0 for k in range(n_views)]
        vflags[1] |= B.VIEW_FLAG_NO_CPU_CULLING
        masks = [1 if k % 2 == 0 else 3 for k in range(n_views)]
        with api.Context(0) as ctx:
            ctx.debug_set_sphere_path(2)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            ctx.upload_visibility_ranges(ranges)
            ctx.upload_visibility_classes(class_mask)
            ctx.upload_changed(np.ones(n, np.uint8))
            vv = np.zeros(n, np.uint8)
            for frame in range(3):
                ctx.propagate(0)
                views, oviews = api.make_views(fr, masks, vflags, pos), O.make_views(fr, masks, vflags, pos)
                ctx.cull_views(views, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
                vv1 = O.reset_view_visibility(sc["flags"], vv)
                vv2, vis, chg = O.check_visibility_views(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], ranges, vv1, oviews)
                vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
                vv, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
                for v in range(n_views):
                    assert_bits(ctx.download_visibility(v), vis[v], f"{n_views} views, frame {frame}, view {v}")
                    for cb in (0, 2):
                        k, rows = ctx.download_visible_entities(v, cb)
                        ek, er = O.visible_entities_sorted(vis[v], class_mask, cb, np.arange(n, dtype=np.uint64))
                        assert np.array_equal(rows, er), (n_views, frame, v, cb)
                got_vv, got_chg = ctx.download_view_visibility()
                assert_bits(got_vv, vv, "ViewVisibility")
                assert_bits(got_chg, chg | chg2 | chg3, "change ticks")
            # one frame with a cascade among the views: k_frame's business; the next camera-only frame is the sphere path again
            sflags = list(vflags)
            sflags[0] = B.VIEW_KIND_CASCADE
            views, oviews = api.make_views(fr, masks, sflags, pos), O.make_views(fr, masks, sflags, pos)
            for fl, ov in ((views, oviews), (api.make_views(fr, masks, vflags, pos), O.make_views(fr, masks, vflags, pos))):
                ctx.cull_views(fl, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
                vv1 = O.reset_view_visibility(sc["flags"], vv)
                vv2, vis, _ = O.check_visibility_views(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], ranges, vv1, ov)
                vv3, _ = O.check_visibility_gpu_culling(sc["flags"], vv2)
                vv, _ = O.mark_newly_hidden(sc["flags"], vv3)
                for v in range(n_views):
                    assert_bits(ctx.download_visibility(v), vis[v], f"shadow / camera frame, view {v}")


def test_metric_frame_riders_travel_with_the_sphere_kernel():
    """The changed-rows metric frame (MI_CULL_CHANGED_ROWS | MI_CULL_WITH_CLUSTERS | MI_CULL_MORE_FRAMES): the deferred compaction,
    the deferred cluster fill and this frame's cluster walk ride in k_frame_sph exactly as they ride in k_frame -- same lists."""
    sc, first_light, pr = W.frame_scene(40_000, 4_000, 400)
    n = sc["n"]
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    results = {}
    for mode in (1, 2):
        t = sc["translation"].reshape(n, 3).copy()
        rng = np.random.default_rng(11)
        with api.Context(0) as ctx:
            ctx.debug_set_sphere_path(mode)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            ctx.cluster_upload_objects(pr)
            ctx.cluster_bind_objects_to_rows(first_light, len(pr) // 4)
            ctx.upload_changed(np.ones(n, np.uint8))
            out = []
            for frame in range(5):
                cam = W.many_cubes_camera(frame * 25)
                fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
                view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0, with_spheres=False)
                ctx.cluster_upload_view(view)
                if frame:
                    moved = np.sort(rng.choice(n, 500, replace=False)).astype(np.uint32)
                    t[moved] += rng.normal(0.0, 1.0, (500, 3)).astype(F)
                    ctx.upload_transforms_indexed(moved, t[moved].reshape(-1), sc["rotation"].reshape(n, 4)[moved].reshape(-1),
                                                  sc["scale"].reshape(n, 3)[moved].reshape(-1))
                ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS | B.CULL_CHANGED_ROWS | B.CULL_MORE_FRAMES)
                if frame in (2, 4):  # the downloads join the deferred riders; the frames in between leave them to the next launch
                    off, idx, counts, far, total = ctx.cluster_download(view.n_clusters)
                    out.append((ctx.download_visible_entities(0, 0)[1].tobytes(), off.tobytes(), idx[:total].tobytes(), counts.tobytes(),
                                ctx.download_view_visibility()[0].tobytes(), ctx.download_global_transforms(want_changed=False).tobytes()))
            results[mode] = out
    assert results[1] == results[2]
    assert len(results[1][0][0]) > 0 and len(results[1][0][2]) > 0
