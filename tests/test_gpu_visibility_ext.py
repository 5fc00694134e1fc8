"""GPU parity for the rows either side of check_visibility:
  - VisibilityRange evaluated on the device (check_visibility_ranges, visibility/range.rs:225-284)
  - shadow views: directional cascades and point / spot light frusta (bevy_light/src/lib.rs:342-757)
  - InheritedVisibility propagation (visibility/mod.rs:638-729)
Bit-exact against the oracle, through the C ABI."""
import math

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


def scene_with_ranges(n, seed=5):
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    rnd = W.splitmix64(seed, n)
    ranged = rnd % np.uint64(3) == 0
    sc["flags"][ranged] |= np.uint8(B.FLAG_HAS_VISIBILITY_RANGE)
    sc["flags"][ranged & (rnd % np.uint64(5) == 0)] |= np.uint8(B.FLAG_RANGE_USE_AABB)
    sc["flags"][rnd % np.uint64(4) != 1] |= np.uint8(B.FLAG_SHADOW_CASTER)
    lo = (W.uniform01(seed + 1, n) * 70.0).astype(F)
    hi = lo + (W.uniform01(seed + 2, n) * 40.0).astype(F)
    sc["ranges"] = np.stack([lo, hi], axis=1).reshape(-1).copy()
    return sc


def upload(ctx, sc, vv0=None, ranges=True):
    ctx.resize(sc["n"])
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.upload_visibility_ranges(sc["ranges"] if ranges else None)
    if vv0 is not None:
        ctx.upload_view_visibility(vv0)


def oracle_views_frame(sc, vv0, oviews, ranges=True):
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    vv1 = O.reset_view_visibility(sc["flags"], vv0)
    vv2, vis, chg = O.check_visibility_views(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"],
                                             sc["ranges"] if ranges else None, vv1, oviews)
    vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
    vv4, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
    return g, vv4, vis, (chg | chg2 | chg3)


def camera_frustum(cam, fov=W.CAMERA_FOV, aspect=W.CAMERA_ASPECT, near=W.CAMERA_NEAR, far=W.CAMERA_FAR):
    return api.compute_frustum(api.perspective_clip_from_view(fov, aspect, near), cam, far)


@pytest.mark.parametrize("n", [700, 40_001])
def test_visibility_ranges_on_device(n):
    sc = scene_with_ranges(n)
    cams = [W.many_cubes_camera(0, position=(0.0, 0.0, 0.0)), W.many_cubes_camera(3, yaw=1.0, position=(10.0, -5.0, 20.0)),
            W.many_cubes_camera(0, yaw=2.0, position=(-30.0, 0.0, 0.0))]
    fr = np.concatenate([camera_frustum(c) for c in cams])
    pos = np.array([c[9:12] for c in cams], F)
    vflags = [B.VIEW_FLAG_RANGES, B.VIEW_FLAG_RANGES, 0]          # the third view has no index in VisibleEntityRanges
    vv0 = (W.splitmix64(1, n) % np.uint64(4)).astype(np.uint8)
    for ranges in (True, False):
        views = api.make_views(fr, [1, 3, 1], vflags, pos)
        oviews = O.make_views(fr, [1, 3, 1], vflags, pos)
        with api.Context(0) as ctx:
            upload(ctx, sc, vv0, ranges)
            ctx.propagate_and_cull_views(views, flags=B.CULL_END_FRAME)
            g, vv_exp, vis_exp, chg_exp = oracle_views_frame(sc, vv0, oviews, ranges)
            assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes()
            for v in range(3):
                assert_bits(ctx.download_visibility(v), vis_exp[v], f"ranges={ranges} view {v}")
            vv, chg = ctx.download_view_visibility()
            assert_bits(vv, vv_exp, "vv")
            assert_bits(chg, chg_exp, "vv changed")
        if ranges:  # the on-device evaluation equals check_visibility_ranges' table
            table = O.check_visibility_ranges(g, sc["aabb_center"], sc["flags"], sc["ranges"], pos)
            ranged = (sc["flags"] & B.FLAG_HAS_VISIBILITY_RANGE) != 0
            assert not (vis_exp[0][ranged] & ~table[0][ranged]).any()


def cube_face_cameras(position):
    """Six 90-degree views around a point (+x,-x,+y,-y,+z,-z), like a point light's cubemap faces."""
    px, py, pz = position
    faces = []
    for axis, angle in (("y", -math.pi / 2), ("y", math.pi / 2), ("x", math.pi / 2), ("x", -math.pi / 2), ("y", math.pi), ("y", 0.0)):
        faces.append(W.affine_from_quat_translation(W.quat_axis(axis, angle), (px, py, pz)))
    return faces


@pytest.mark.parametrize("n", [900, 60_000])
def test_shadow_views_cascades_cube_faces_and_spot(n):
    sc = scene_with_ranges(n, seed=9)
    cam = W.many_cubes_camera(2)
    frs, masks, flags, pos, sph = [], [], [], [], []
    # the camera itself
    frs.append(camera_frustum(cam)); masks.append(1); flags.append(B.VIEW_FLAG_RANGES); pos.append(cam[9:12]); sph.append([0, 0, 0, 0])
    # three cascades of a directional light: slabs of the camera frustum seen from the light (any frusta will do)
    for k, (near, far) in enumerate([(0.1, 20.0), (20.0, 60.0), (60.0, 200.0)]):
        lightcam = W.many_cubes_camera(0, yaw=0.4 + 0.1 * k, position=(0.0, 30.0, 0.0))
        frs.append(camera_frustum(lightcam, fov=1.2, aspect=1.0, near=near, far=far)); masks.append(3)
        flags.append(B.VIEW_KIND_CASCADE | B.VIEW_FLAG_RANGES); pos.append(cam[9:12]); sph.append([0, 0, 0, 0])
    # a point light with a shadow LOD origin, and one without
    for lp, lrange, lflags in (((20.0, 10.0, -40.0), 45.0, B.VIEW_FLAG_RANGES), ((-35.0, 0.0, 25.0), 30.0, B.VIEW_FLAG_RANGES_NO_ORIGIN)):
        for face in cube_face_cameras(lp):
            frs.append(camera_frustum(face, fov=math.pi / 2, aspect=1.0, near=0.1, far=lrange)); masks.append(1)
            flags.append(B.VIEW_KIND_CUBE_FACE_OR_SPOT | lflags); pos.append(cam[9:12]); sph.append([*lp, lrange])
    # a spot light
    spot = W.many_cubes_camera(0, yaw=2.2, position=(5.0, 40.0, 5.0))
    frs.append(camera_frustum(spot, fov=1.0, aspect=1.0, near=0.1, far=80.0)); masks.append(2)
    flags.append(B.VIEW_KIND_CUBE_FACE_OR_SPOT | B.VIEW_FLAG_RANGES); pos.append(cam[9:12]); sph.append([5.0, 40.0, 5.0, 80.0])
    fr = np.concatenate(frs)
    views = api.make_views(fr, masks, flags, np.array(pos, F), np.array(sph, F))
    oviews = O.make_views(fr, masks, flags, np.array(pos, F), np.array(sph, F))
    vv0 = (W.splitmix64(2, n) % np.uint64(4)).astype(np.uint8)
    g, vv_exp, vis_exp, chg_exp = oracle_views_frame(sc, vv0, oviews)
    assert vis_exp[1:].any(), "the scene should have shadow casters in some shadow view"
    with api.Context(0) as ctx:       # fused flat path (17 views: device view table)
        upload(ctx, sc, vv0)
        ctx.propagate_and_cull_views(views, flags=B.CULL_END_FRAME)
        for v in range(len(frs)):
            assert_bits(ctx.download_visibility(v), vis_exp[v], f"view {v}")
            keys, rows = ctx.download_visible_entities(v, 0)
            assert np.array_equal(rows, np.nonzero(vis_exp[v])[0].astype(np.uint32))
        vv, chg = ctx.download_view_visibility()
        assert_bits(vv, vv_exp, "vv")
        assert_bits(chg, chg_exp, "vv changed")
    with api.Context(0) as ctx:       # unfused: propagate, then cull the camera and the shadow views separately
        upload(ctx, sc, vv0)
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        ctx.cull_views(api.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)), flags=B.CULL_BEGIN_FRAME)
        ctx.cull_views(api.make_views(fr[24:], masks[1:], flags[1:], np.array(pos[1:], F), np.array(sph[1:], F)), flags=B.CULL_END_FRAME)
        for v in range(1, len(frs)):
            assert_bits(ctx.download_visibility(v - 1), vis_exp[v], f"unfused view {v}")
        vv, chg = ctx.download_view_visibility()
        assert_bits(vv, vv_exp, "vv (two cull calls)")
        assert_bits(chg, chg_exp, "vv changed (two cull calls)")
    # the plugin's sequence: the cameras' frame (propagate + cull, the frame left open), then SimulationLightSystems::CheckLightVisibility
    # as ONE call that also closes the frame -- every shadow view's VisibleMeshEntities mask and the set_visible() union in one wait
    shadow = api.make_views(fr[24:], masks[1:], flags[1:], np.array(pos[1:], F), np.array(sph[1:], F))
    with api.Context(0) as ctx:
        upload(ctx, sc, vv0)
        ctx.propagate_and_cull_views(api.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)), flags=0)
        cam_rows = ctx.download_visible_entities(0, 0)[1]
        assert np.array_equal(cam_rows, np.nonzero(vis_exp[0])[0].astype(np.uint32))
        per_view, any_ = ctx.check_light_mesh_visibility(shadow, flags=B.CULL_END_FRAME)
        for v in range(1, len(frs)):
            assert_bits(per_view[v - 1], vis_exp[v], f"light pass view {v}")
        assert_bits(any_, np.bitwise_or.reduce(vis_exp[1:], axis=0), "rows set_visible() was called on")
        vv, chg = ctx.download_view_visibility()
        assert_bits(vv, vv_exp, "vv (camera frame + light pass)")
        assert_bits(chg, chg_exp, "vv changed (camera frame + light pass)")
        keys, rows = ctx.download_visible_entities(3, 0)   # the lists now belong to the shadow views (the shadow phases batch from them)
        assert np.array_equal(rows, np.nonzero(vis_exp[4])[0].astype(np.uint32))
        # no shadow view at all: the call only closes the frame
        ctx.propagate_and_cull_views(api.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)), flags=0)
        per_view, any_ = ctx.check_light_mesh_visibility(None, flags=B.CULL_END_FRAME)
        assert per_view.shape == (0, n) and not any_.any()
        g2, vv_exp2, _, chg_exp2 = oracle_views_frame(sc, vv_exp, O.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)))
        vv, chg = ctx.download_view_visibility()
        assert_bits(vv, vv_exp2, "vv (camera frame + empty light pass)")
        assert_bits(chg, chg_exp2, "vv changed (camera frame + empty light pass)")
        # the Rust shim's sequence: both of its camera paths close the frame on the device (MI_CULL_END_FRAME) and the light pass follows
        # with flags = 0 -- the ECS components get their set_visible() from the masks; on the device only bit 0 (ViewVisibility::get(),
        # what the next frame's reset reads) has to come out right
        ctx.upload_view_visibility(vv0)
        ctx.propagate_and_cull_views(api.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)), flags=B.CULL_END_FRAME)
        per_view, any_ = ctx.check_light_mesh_visibility(shadow, flags=0)
        for v in range(1, len(frs)):
            assert_bits(per_view[v - 1], vis_exp[v], f"light pass behind a closed frame, view {v}")
        vv, _ = ctx.download_view_visibility()
        assert_bits(vv & 1, vv_exp & 1, "ViewVisibility::get() after a closed camera frame + light pass")
        with pytest.raises(api.MiError):   # a camera view is not a shadow view
            ctx.check_light_mesh_visibility(api.make_views(fr[:24], masks[:1], flags[:1], np.array(pos[:1], F)))


# ---- InheritedVisibility -------------------------------------------------------------------------

def random_forest(n, seed, flat_fraction=0.3, max_children=6):
    rng = np.random.default_rng(seed)
    parent = np.full(n, B.NO_PARENT, np.uint32)
    order = rng.permutation(n)
    nodes = order[int(n * flat_fraction):]
    for k in range(1, len(nodes)):
        if rng.random() < 0.01:
            continue
        lo = max(0, k - 1 - int(rng.integers(0, 50 * max_children)))
        parent[nodes[k]] = nodes[rng.integers(lo, k)]
    return parent


def random_visibility(n, rng):
    v = rng.choice(np.array([0, 0, 0, 1, 2, 0x80], np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.12, 0.12, 0.06])
    return v.astype(np.uint8)


def test_inherited_visibility_reference_scenarios():
    """visibility/mod.rs:950-1037 and :1189-1262 on the device."""
    parent = np.array([B.NO_PARENT, 0, 0, 1, 2, B.NO_PARENT, 5, 5, 6, 7], np.uint32)
    vis = np.array([1, 0, 1, 0, 0, 0, 0, 1, 0, 0], np.uint8)
    new_to_old, pidx, offs = api.hierarchy_sort(parent)
    with api.Context(0) as ctx:
        ctx.resize(10)
        ctx.upload_hierarchy(pidx, offs)
        ctx.upload_bounds(np.zeros(30, F), np.zeros(30, F), np.zeros(10, np.uint8), None)  # InheritedVisibility::default() = false
        ctx.upload_visibility(vis[new_to_old])
        ctx.visibility_propagate()
        inh, _ = ctx.download_inherited_visibility()
        back = np.zeros(10, np.uint8); back[new_to_old] = inh
        assert back.tolist() == [0, 0, 0, 0, 0, 1, 1, 0, 1, 0]
    parent = np.array([B.NO_PARENT, 0, 1, 2], np.uint32)   # chain, rows already in level order
    offs = np.array([0, 1, 2, 3, 4], np.uint32)
    with api.Context(0) as ctx:
        ctx.resize(4)
        ctx.upload_hierarchy(parent, offs)
        ctx.upload_bounds(np.zeros(12, F), np.zeros(12, F), np.zeros(4, np.uint8), None)
        vis = np.array([0, 0, 1, 0], np.uint8)
        steps = [(None, [1, 1, 0, 0], None), ((0, 1), None, [1, 1, 0, 0]), (None, None, [0, 0, 0, 0]), ((2, 0), None, [0, 0, 0, 0]),
                 ((1, 2), [0, 1, 1, 1], [0, 1, 1, 1]), (None, None, [0, 0, 0, 0])]
        for edit, exp_inh, exp_chg in steps:
            if edit:
                vis[edit[0]] = edit[1]
            ctx.upload_visibility(vis)
            ctx.visibility_propagate()
            inh, chg = ctx.download_inherited_visibility()
            if exp_inh is not None:
                assert inh.tolist() == exp_inh
            if exp_chg is not None:
                assert chg.tolist() == exp_chg


@pytest.mark.parametrize("shape", ["forest", "deep", "flat", "wide"])
def test_inherited_visibility_propagation_matches_oracle(shape):
    rng = np.random.default_rng(21)
    if shape == "forest":
        n = 30_000
        new_to_old, parent, offs = api.hierarchy_sort(random_forest(n, 3))
    elif shape == "deep":
        tr = W.gen_tree(9, 4)
        n, parent, offs = tr["n"], tr["parent"], tr["level_offsets"]
    elif shape == "wide":
        tr = W.gen_tree(4, 40)   # levels wider than the LDS budget of a tile
        n, parent, offs = tr["n"], tr["parent"], tr["level_offsets"]
    else:
        n, parent, offs = 5_000, None, None
    vis = random_visibility(n, rng)
    inh0 = (rng.random(n) < 0.5).astype(np.uint8)
    flags0 = (inh0 | np.uint8(B.FLAG_HAS_AABB)).astype(np.uint8)
    with api.Context(0) as ctx:
        ctx.resize(n)
        if parent is not None:
            ctx.upload_hierarchy(parent, offs)
        ctx.upload_bounds(np.zeros(3 * n, F), np.zeros(3 * n, F), flags0, None)
        inh = inh0
        for frame in range(3):
            if frame:
                edit = rng.random(n) < 0.02
                vis = np.where(edit, random_visibility(n, rng), vis).astype(np.uint8)
            ctx.upload_visibility(vis)
            ctx.visibility_propagate()
            got, got_chg = ctx.download_inherited_visibility()
            rc, inh, chg = O.visibility_propagate(parent if parent is not None else np.full(n, B.NO_PARENT, np.uint32), vis, inh)
            assert rc == 0
            assert_bits(got, inh, f"{shape} frame {frame} InheritedVisibility")
            assert_bits(got_chg, chg, f"{shape} frame {frame} change ticks")


def test_render_layers_up_to_63():
    """RenderLayers::intersects compares the masks word by word (render_layers.rs:121-135); the first u64 word = the row column
    `layer_mask` + mi_upload_render_layers_hi and mi_view.layer_mask / layer_mask_hi.  Rows and views spread over layers 0..63, on
    every frame kernel: the all-rows frame, the cull-only frame (resident GlobalTransforms and the world-sphere column), the
    changed-rows frame, the fused hierarchy frame -- against the oracle's 64-layer restatement; rows above layer 31 sit inside
    waves that are otherwise uniform (the row summary must not cover them)."""
    n = 20_000
    sc = W.many_cubes(n, radius=60.0)
    rng = np.random.default_rng(8)
    lo = np.ones(n, np.uint32)
    hi = np.zeros(n, np.uint32)
    odd = np.sort(rng.choice(n, 3000, replace=False))
    layer = rng.integers(0, 64, len(odd))
    lo[odd] = np.where(layer < 32, np.uint64(1) << np.minimum(layer, 31).astype(np.uint64), 0).astype(np.uint32)
    hi[odd] = np.where(layer >= 32, np.uint64(1) << np.maximum(layer - 32, 0).astype(np.uint64), 0).astype(np.uint32)
    both = odd[:200]
    lo[both] |= 1
    hi[both] |= np.uint32(1 << 7)
    vm_lo = np.array([1, 0, 1 << 5, 0xFFFFFFFF], np.uint32)
    vm_hi = np.array([0, 1 << 7, 1 << 20, 0xFFFFFFFF], np.uint32)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])

    def expected(vv, frusta):
        vv1 = O.reset_view_visibility(sc["flags"], vv)
        vv2, vis, chg = O.check_visibility_layers64(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], lo, hi, vv1, frusta, vm_lo, vm_hi)
        vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
        vv4, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
        return vv4, vis, chg | chg2 | chg3

    def views_of(frame):
        fr = np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(frame * 30, yaw=1.5 * v), W.CAMERA_FAR) for v in range(4)])
        return fr, api.make_views(fr, layer_masks=vm_lo, layer_masks_hi=vm_hi)

    for sphere_path in (1, 2):
        with api.Context(0) as ctx:
            ctx.debug_set_sphere_path(sphere_path)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], lo)
            ctx.upload_render_layers_hi(hi)
            ctx.upload_changed(np.ones(n, np.uint8))
            vv = np.zeros(n, np.uint8)
            for frame, kind in enumerate(["all", "cull", "cull", "changed", "cull"]):
                fr, views = views_of(frame)
                if kind == "all":
                    ctx.propagate_and_cull_views(views, flags=B.CULL_END_FRAME)
                elif kind == "changed":
                    ctx.propagate_and_cull_views(views, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
                else:
                    ctx.propagate(0)
                    ctx.cull_views(views, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
                vv, vis, chg = expected(vv, fr)
                for v in range(4):
                    got = ctx.download_visibility(v)
                    bad = np.nonzero(got != vis[v])[0]
                    assert bad.size == 0, f"sphere path {sphere_path} frame {frame} ({kind}) view {v}: rows {bad[:8].tolist()}"
                va, cva = ctx.download_view_visibility()
                assert np.array_equal(va, vv) and np.array_equal(cva, chg), f"sphere path {sphere_path} frame {frame} ({kind}): ViewVisibility"
        assert vis[1].any() and vis[2].any()  # the views that live above layer 31 see something

    # the same rows under a hierarchy (a forest of single nodes): the tile kernel's own cull
    with api.Context(0) as ctx:
        ctx.debug_set_tree_cull(2)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_hierarchy(np.full(n, 0xFFFFFFFF, np.uint32), np.array([0, n], np.uint32))
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], lo)
        ctx.upload_render_layers_hi(hi)
        fr, views = views_of(3)
        ctx.propagate_and_cull_views(views, flags=B.CULL_END_FRAME)
        vv, vis, chg = expected(np.zeros(n, np.uint8), fr)
        for v in range(4):
            assert np.array_equal(ctx.download_visibility(v), vis[v]), f"hierarchy frame, view {v}"
        assert np.array_equal(ctx.download_view_visibility()[0], vv)
