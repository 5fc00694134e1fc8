"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bit-exact everywhere: GlobalTransform matrices (stronger than the 1e-5 the north star asks),
ViewVisibility flags, per-view bitmasks, VisibleEntities lists, change-tick masks."""
import math

import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def ctx_factory():
    made = []

    def make():
        c = api.Context(0)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


def upload_scene(ctx, sc, vv0=None):
    ctx.resize(sc["n"])
    if sc["n"] == 0:
        return
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    if vv0 is not None:
        ctx.upload_view_visibility(vv0)


def oracle_frame(sc, vv, frusta, vmasks, vflags):
    return O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"],
                        sc["flags"], sc["layers"], vv, frusta, vmasks, vflags)


def assert_bits(a, b, what):
    bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first rows {bad[:8].tolist()}"


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 4096, 50_003])
def test_flat_fused_frame_matches_oracle(ctx_factory, n):
    sc = W.many_cubes(n, radius=500.0 if n > 1000 else 5.0, ragged_flags=True)
    cams = [W.many_cubes_camera(0), W.many_cubes_camera(5, yaw=math.pi / 2), W.many_cubes_camera(9, yaw=math.pi)]
    frusta = frusta_for(cams)
    vmasks = np.array([1, 3, 2], np.uint32)
    vflags = np.array([0, 0, B.VIEW_FLAG_NO_CPU_CULLING], np.uint8)
    vv0 = (W.splitmix64(3, n) % np.uint64(2)).astype(np.uint8)  # some rows visible last frame
    ctx = ctx_factory()
    upload_scene(ctx, sc, vv0)
    ctx.propagate_and_cull(frusta, vmasks, vflags)
    ctx.visibility_end_frame()
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, vv0, frusta, vmasks, vflags)
    g, g_chg = ctx.download_global_transforms()
    assert g.tobytes() == g_exp.tobytes(), "GlobalTransform not bit-exact"
    assert_bits(g_chg, np.ones(n, np.uint8), "GlobalTransform change mask")
    for v in range(3):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"view {v} visibility")
    vv, vv_chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, "ViewVisibility bytes")
    assert_bits(vv_chg, chg_exp, "ViewVisibility change mask")


@pytest.mark.parametrize("sphere_path", [1, 2])
@pytest.mark.parametrize("n_views", [2, 3, 4])
def test_several_camera_views_pair_pass(ctx_factory, n_views, sphere_path, monkeypatch):
    """k_frame's MULTI path (intersects_obb over the wave's (row, view) pairs, kernels_flat.hip): cameras that look almost the same
    way, so that a wave queues up to n_views x 64 pairs -- several passes; a ragged last wave; rows with a Sphere, without bounds,
    NoFrustumCulling, NoCpuCulling; one camera with NoCpuCulling.  Against the oracle, against the per-view rule (MI_MULTI_VIEW=1), in
    the fused frame, the changed-rows frame and the cull over resident GlobalTransforms -- the last two through k_frame_pairs
    (sphere_path 1) and through the world-sphere column's k_frame_sph_pairs (sphere_path 2)."""
    n = 20_011
    sc = W.many_cubes(n, radius=40.0, ragged_flags=True)
    cams = [W.many_cubes_camera(3 * k, yaw=math.pi + 0.05 * k) for k in range(n_views)]  # (looking at the spiral's dense start)
    frusta = frusta_for(cams)
    vmasks = np.array([1, 3, 1, 3][:n_views], np.uint32)
    vflags = np.array([0, 0, B.VIEW_FLAG_NO_CPU_CULLING, 0][:n_views], np.uint8)
    vv0 = (W.splitmix64(5, n) % np.uint64(2)).astype(np.uint8)
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, vv0, frusta, vmasks, vflags)
    assert max(int(v.sum()) for v in vis_exp) > n // 8, "the cameras see too little for several passes"
    results = []
    for mode in ("0", "1"):
        monkeypatch.setenv("MI_MULTI_VIEW", mode)  # (read when a context is created)
        ctx = ctx_factory()
        ctx.debug_set_sphere_path(sphere_path)  # 1: the resident cull below through k_frame; 2: through the world-sphere kernel, at once
        upload_scene(ctx, sc, vv0)
        ctx.propagate_and_cull(frusta, vmasks, vflags)
        ctx.visibility_end_frame()
        fused = [ctx.download_visibility(v) for v in range(n_views)]
        vv, vv_chg = ctx.download_view_visibility()
        for v in range(n_views):
            assert_bits(fused[v], vis_exp[v], f"MI_MULTI_VIEW={mode}: view {v}")
        assert_bits(vv, vv_exp, f"MI_MULTI_VIEW={mode}: ViewVisibility bytes")
        assert_bits(vv_chg, chg_exp, f"MI_MULTI_VIEW={mode}: ViewVisibility change mask")
        # the same views over the resident GlobalTransforms
        ctx.cull(frusta, vmasks, vflags)
        resident = [ctx.download_visibility(v) for v in range(n_views)]
        for v in range(n_views):
            assert_bits(resident[v], vis_exp[v], f"MI_MULTI_VIEW={mode}: resident cull, view {v}")
        # ... and the changed-rows frame: a tenth of the rows moved
        moved = np.nonzero(W.splitmix64(11, n) % np.uint64(10) == 0)[0].astype(np.uint32)
        t2 = sc["translation"].reshape(n, 3).copy()
        t2[moved] *= F(0.5)
        ctx.upload_transforms_indexed(moved, np.ascontiguousarray(t2[moved]).reshape(-1), np.ascontiguousarray(sc["rotation"].reshape(n, 4)[moved]).reshape(-1),
                                      np.ascontiguousarray(sc["scale"].reshape(n, 3)[moved]).reshape(-1))
        ctx.propagate_and_cull(frusta, vmasks, vflags, flags=B.CULL_CHANGED_ROWS | B.CULL_END_FRAME)
        results.append([ctx.download_visibility(v) for v in range(n_views)] + list(ctx.download_view_visibility()) + [ctx.download_global_transforms()[0]])
    sc2 = dict(sc, translation=np.ascontiguousarray(t2).reshape(-1))
    g2, vv2, vis2, _ = oracle_frame(sc2, vv_exp, frusta, vmasks, vflags)
    for v in range(n_views):
        assert_bits(results[0][v], vis2[v], f"changed-rows frame: view {v}")
    for a, b in zip(results[0], results[1]):
        assert a.tobytes() == b.tobytes(), "the pair pass and the per-view rule disagree"
    assert results[0][-1].tobytes() == g2.tobytes()


def test_unfused_equals_fused_and_oracle(ctx_factory):
    n = 20_011
    sc = W.many_cubes(n, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(3), W.many_cubes_camera(3, yaw=2.0)])
    vv0 = (W.splitmix64(5, n) % np.uint64(2)).astype(np.uint8)
    a, b = ctx_factory(), ctx_factory()
    upload_scene(a, sc, vv0)
    upload_scene(b, sc, vv0)
    a.propagate_and_cull(frusta)
    a.visibility_end_frame()
    b.propagate(B.PROPAGATE_ALL_DIRTY)
    b.visibility_begin_frame()
    b.cull(frusta)
    b.visibility_end_frame()
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, vv0, frusta, None, None)
    for c in (a, b):
        assert c.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes()
        for v in range(2):
            assert_bits(c.download_visibility(v), vis_exp[v], "visibility")
        vv, chg = c.download_view_visibility()
        assert_bits(vv, vv_exp, "vv")
        assert_bits(chg, chg_exp, "vv changed")


@pytest.mark.parametrize("n", [1, 64, 1000, 33_333])
def test_fused_begin_end_flags_equal_separate_passes(ctx_factory, n):
    """MI_CULL_BEGIN_FRAME / MI_CULL_END_FRAME fold reset_view_visibility and check_visibility_gpu_culling +
    mark_newly_hidden into the cull pass: same bytes and change ticks as the separate systems (and the oracle)."""
    sc = W.many_cubes(n, radius=500.0 if n > 1000 else 5.0, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(2), W.many_cubes_camera(2, yaw=1.5)])
    vv0 = (W.splitmix64(11, n) % np.uint64(4)).astype(np.uint8)  # all four 2-bit states
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, vv0, frusta, None, None)
    a, b = ctx_factory(), ctx_factory()
    upload_scene(a, sc, vv0)
    a.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    upload_scene(b, sc, vv0)
    b.propagate(B.PROPAGATE_ALL_DIRTY)
    b.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
    for c in (a, b):
        assert c.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes()
        for v in range(2):
            assert_bits(c.download_visibility(v), vis_exp[v], "visibility")
        vv, chg = c.download_view_visibility()
        assert_bits(vv, vv_exp, "vv")
        assert_bits(chg, chg_exp, "vv changed")


def test_many_views_use_the_device_view_table(ctx_factory):
    """More than MAX_INLINE_VIEWS (8) cameras: views come from a device array instead of the kernarg segment."""
    n = 9_001
    sc = W.many_cubes(n, ragged_flags=True)
    cams = [W.many_cubes_camera(k, yaw=0.6 * k) for k in range(11)]
    frusta = frusta_for(cams)
    vmasks = np.array([1, 3, 2, 1, 1, 3, 1, 2, 1, 1, 3], np.uint32)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(frusta, vmasks, flags=B.CULL_END_FRAME)
    _, vv_exp, vis_exp, chg_exp = oracle_frame(sc, np.zeros(n, np.uint8), frusta, vmasks, None)
    for v in range(11):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"view {v}")
        k, rows = ctx.download_visible_entities(v, 0)
        assert np.array_equal(rows, np.nonzero(vis_exp[v])[0].astype(np.uint32))
    vv, chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, "vv")
    assert_bits(chg, chg_exp, "vv changed")


@pytest.mark.parametrize("n", [5, 4096, 4097, 70_001])
def test_visible_entities_fast_path_with_classes(ctx_factory, n):
    """Rows already in Entity-key order (the shim's numbering): single-launch compaction from the per-wave counts
    and class-filtered segment masks, fused and unfused."""
    sc = W.many_cubes(n, radius=500.0 if n > 1000 else 5.0, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(0), W.many_cubes_camera(0, yaw=1.0), W.many_cubes_camera(7, yaw=2.5)])
    rnd = W.splitmix64(7, n)
    keys = (np.arange(n, dtype=np.uint64) * np.uint64(3)) + np.uint64(1 << 32)  # ascending with the row
    class_mask = np.where(rnd % np.uint64(4) == 0, 0b1010, np.where(rnd % np.uint64(4) == 1, 0b1000, 0b0010)).astype(np.uint32)
    class_mask[rnd % np.uint64(29) == 0] = 0
    _, _, vis_exp, _ = oracle_frame(sc, np.zeros(n, np.uint8), frusta, None, None)
    a, b = ctx_factory(), ctx_factory()
    for c in (a, b):
        upload_scene(c, sc)
        c.upload_entity_keys(keys)
        c.upload_visibility_classes(class_mask)
    a.propagate_and_cull(frusta)
    b.propagate(B.PROPAGATE_ALL_DIRTY)
    b.cull(frusta, flags=B.CULL_BEGIN_FRAME)
    for c in (a, b):
        for v in range(3):
            for cb in (1, 3, 0):
                k, rows = c.download_visible_entities(v, cb)
                ek, er = O.visible_entities_sorted(vis_exp[v], class_mask, cb, keys)
                assert np.array_equal(k, ek) and np.array_equal(rows, er), (v, cb, len(k), len(ek))


def test_view_visibility_lifecycle_over_frames(ctx_factory):
    """visibility/mod.rs:1313-1448 at scale: the 2-bit protocol and change ticks over 6 frames of a moving camera."""
    n = 30_000
    sc = W.many_cubes(n, ragged_flags=True)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    vv = np.zeros(n, np.uint8)
    for frame in range(6):
        frusta = frusta_for([W.many_cubes_camera(frame * 40)])
        ctx.propagate_and_cull(frusta)
        ctx.visibility_end_frame()
        _, vv, _, chg = oracle_frame(sc, vv, frusta, None, None)
        got_vv, got_chg = ctx.download_view_visibility()
        assert_bits(got_vv, vv, f"frame {frame} vv")
        assert_bits(got_chg, chg, f"frame {frame} change ticks")


def test_partial_dirty_flat_rows(ctx_factory):
    """sync_simple_transforms only rewrites rows whose Transform changed (systems.rs:45-50)."""
    n = 10_000
    sc = W.many_cubes(n)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g0 = ctx.download_global_transforms(want_changed=False)
    t2 = sc["translation"].copy()
    t2[::2] += F(1.0)
    changed = np.zeros(n, np.uint8)
    changed[::7] = 1
    ctx.upload_transforms(t2, sc["rotation"], sc["scale"])
    ctx.upload_changed(changed)
    ctx.propagate(0)
    g1, chg = ctx.download_global_transforms()
    exp, exp_chg = O.sync_simple_transforms(t2, sc["rotation"], sc["scale"], changed, g0)
    assert g1.tobytes() == exp.tobytes()
    assert_bits(chg, exp_chg, "changed rows")
    ctx.propagate(0)  # change flags were consumed: nothing is rewritten
    g2, chg2 = ctx.download_global_transforms()
    assert g2.tobytes() == exp.tobytes() and not chg2.any()


def test_visible_entities_sorted_lists(ctx_factory):
    n = 40_000
    sc = W.many_cubes(n, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(0), W.many_cubes_camera(0, yaw=1.0)])
    rnd = W.splitmix64(99, n)
    # Entity::to_bits: generation << 32 | !index  -> arbitrary order relative to rows
    keys = ((rnd % np.uint64(3)) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.random.default_rng(1).permutation(n).astype(np.uint64))
    class_mask = np.where(rnd % np.uint64(5) == 0, 0b101, np.where(rnd % np.uint64(5) == 1, 0b100, 0b001)).astype(np.uint32)
    class_mask[rnd % np.uint64(41) == 0] = 0  # no VisibilityClass: set_visible but in no list (mod.rs:848-857)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.upload_entity_keys(keys)
    ctx.upload_visibility_classes(class_mask)
    ctx.propagate_and_cull(frusta)
    _, _, vis_exp, _ = oracle_frame(sc, np.zeros(n, np.uint8), frusta, None, None)
    for v in range(2):
        for cb in (0, 2, 1):
            k, rows = ctx.download_visible_entities(v, cb)
            ek, er = O.visible_entities_sorted(vis_exp[v], class_mask, cb, keys)
            assert np.array_equal(k, ek) and np.array_equal(rows, er), (v, cb, len(k), len(ek))
            assert np.all(k[1:] >= k[:-1])
    # identity order fast path: no keys uploaded -> key == row
    ctx2 = ctx_factory()
    upload_scene(ctx2, sc)
    ctx2.propagate_and_cull(frusta)
    k, rows = ctx2.download_visible_entities(0, 0)
    assert np.array_equal(rows, np.nonzero(vis_exp[0])[0].astype(np.uint32)) and np.array_equal(k, rows.astype(np.uint64))


# ---- hierarchy -----------------------------------------------------------------------------------

def upload_tree(ctx, tr, g_init=None):
    ctx.resize(tr["n"])
    ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
    ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
    if g_init is not None:
        ctx.upload_global_transforms(g_init)


def test_reference_propagate_scenarios_on_gpu(ctx_factory):
    """systems.rs:888-925 did_propagate and :1048-1097 correct_transforms_when_no_children, through the GPU path."""
    ident_q = [0, 0, 0, 1]
    t = np.array([1, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3], F)  # flat root, parent, two children
    r = np.array(ident_q * 4, F)
    s = np.ones(12, F)
    parent = np.array([B.NO_PARENT, B.NO_PARENT, 1, 1], np.uint32)
    ctx = ctx_factory()
    ctx.resize(4)
    ctx.upload_transforms(t, r, s)
    ctx.upload_hierarchy(parent, np.array([0, 2, 4], np.uint32))
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | B.PROPAGATE_STATIC_OPT)
    g = ctx.download_global_transforms(want_changed=False).reshape(4, 12)
    assert g[2][9:].tolist() == [1.0, 2.0, 0.0] and g[3][9:].tolist() == [1.0, 0.0, 3.0]
    assert g[2][:9].tolist() == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    # chain of three: parent(1,0,0) <- identity <- identity
    t = np.array([1, 0, 0, 0, 0, 0, 0, 0, 0], F)
    ctx.resize(3)
    ctx.upload_transforms(t, np.array(ident_q * 3, F), np.ones(9, F))
    ctx.upload_hierarchy(np.array([B.NO_PARENT, 0, 1], np.uint32), np.array([0, 1, 2, 3], np.uint32))
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g = ctx.download_global_transforms(want_changed=False).reshape(3, 12)
    for row in g:
        assert row.tolist() == [1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0]


def test_malformed_hierarchy_is_reported(ctx_factory):
    ctx = ctx_factory()
    ctx.resize(3)
    with pytest.raises(api.MiError) as e:  # row 2's parent is not in the previous level
        ctx.upload_hierarchy(np.array([B.NO_PARENT, 0, 0], np.uint32), np.array([0, 1, 2, 3], np.uint32))
    assert e.value.code == api.MI_ERR_MALFORMED_HIERARCHY


@pytest.mark.parametrize("depth,branch,cap", [(5, 4, None), (8, 4, None), (11, 2, None), (9, 4, 60_000), (4, 40, None),
                                              (10, 4, None), (12, 4, 400_000), (9, 7, None), (20, 2, 300_000)])
def test_tree_propagation_all_dirty(ctx_factory, depth, branch, cap):
    tr = W.gen_tree(depth, branch, cap)
    ctx = ctx_factory()
    upload_tree(ctx, tr)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g, chg = ctx.download_global_transforms()
    rc, g_exp, chg_exp = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert rc == 0
    bad = np.nonzero((g.view(np.uint32) != g_exp.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5].tolist()}"
    assert_bits(chg, chg_exp, "changed")
    # independent second oracle: TransformHelper chain product on a few leaves
    for row in (tr["n"] - 1, tr["n"] // 2, 1 % tr["n"]):
        assert np.allclose(g.reshape(-1, 12)[row], O.compute_global_transform(tr["parent"], tr["translation"],
                                                                              tr["rotation"], tr["scale"], row),
                           rtol=1e-5, atol=1e-3)
    # second run with unchanged inputs: set_if_neq sees equal values everywhere except re-assigned roots
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    g2, chg2 = ctx.download_global_transforms()
    rc, g_exp2, chg_exp2 = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"], global_in=g_exp)
    assert g2.tobytes() == g_exp2.tobytes()
    assert_bits(chg2, chg_exp2, "changed (2nd run)")


@pytest.mark.parametrize("static_opt", [False, True])
def test_deep_tree_partial_dirty_through_chain_tiles(ctx_factory, static_opt):
    """A deep narrow-topped tree runs as ONE launch: the tiles below level 5 re-evaluate their ancestor chain instead
    of waiting for the tile that owns it.  Dirty nodes high in the tree (owned by the top tile), in the chain and
    inside the chain tiles must give the oracle's GlobalTransforms and change ticks, frame after frame."""
    tr = W.gen_tree(10, 4)
    n, parent = tr["n"], tr["parent"]
    t = tr["translation"].reshape(n, 3).copy()
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    upload_tree(ctx, tr)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    rc, g0, _ = O.propagate_transforms(parent, tr["translation"], tr["rotation"], tr["scale"], static_opt=static_opt)
    assert rc == 0 and ctx.download_global_transforms(want_changed=False).tobytes() == g0.tobytes()
    rng = np.random.default_rng(5)
    picks = [np.array([0]), np.array([3, 17]), np.array([300, 1200]), np.array([1365 + 7, 5461 + 100, 90_000]),
             np.zeros(0, np.int64), rng.integers(0, n, 40)]
    for frame, dirty in enumerate(picks):
        dirty = np.unique(dirty).astype(np.uint32)
        t[dirty] += F(0.5)
        ctx.upload_transforms_indexed(dirty, t[dirty].reshape(-1), tr["rotation"].reshape(n, 4)[dirty].reshape(-1),
                                      tr["scale"].reshape(n, 3)[dirty].reshape(-1))
        ctx.propagate(flags)
        changed = np.zeros(n, np.uint8); changed[dirty] = 1
        rc, g1, chg = O.propagate_transforms(parent, t.reshape(-1), tr["rotation"], tr["scale"], global_in=g0, static_opt=static_opt,
                                             tree_changed=O.mark_dirty_trees(parent, changed), transform_changed=changed)
        assert rc == 0
        g, got_chg = ctx.download_global_transforms()
        bad = np.nonzero((g.view(np.uint32) != g1.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
        assert bad.size == 0, f"frame {frame}: {bad.size} rows differ, first {bad[:5].tolist()}"
        assert_bits(got_chg, chg, f"frame {frame} change ticks")
        g0 = g1


def random_forest(n, seed, flat_fraction=0.3, max_children=6):
    rng = np.random.default_rng(seed)
    parent = np.full(n, B.NO_PARENT, np.uint32)
    n_flat = int(n * flat_fraction)
    order = rng.permutation(n)
    nodes = order[n_flat:]
    for k in range(1, len(nodes)):
        if rng.random() < 0.01:
            continue
        lo = max(0, k - 1 - int(rng.integers(0, 50 * max_children)))
        parent[nodes[k]] = nodes[rng.integers(lo, k)]
    return parent


@pytest.mark.parametrize("static_opt", [False, True])
def test_random_forest_with_static_optimisation(ctx_factory, static_opt):
    n = 30_000
    parent_old = random_forest(n, 7)
    new_to_old, parent, offs = api.hierarchy_sort(parent_old)
    rng = np.random.default_rng(11)
    t = rng.normal(size=(n, 3)).astype(F).reshape(-1)
    q = rng.normal(size=(n, 4)); q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F).reshape(-1)
    s = rng.uniform(0.8, 1.25, size=(n, 3)).astype(F).reshape(-1)
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    ctx.resize(n)
    ctx.upload_transforms(t, q, s)
    ctx.upload_hierarchy(parent, offs)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    g0 = ctx.download_global_transforms(want_changed=False)
    rc, g0_exp, _ = O.propagate_transforms(parent, t, q, s, static_opt=static_opt)
    assert rc == 0 and g0.tobytes() == g0_exp.tobytes()
    # frame 2: 1% of the transforms change
    changed = (rng.random(n) < 0.01).astype(np.uint8)
    t2 = t.copy().reshape(n, 3); t2[changed == 1] += F(0.5); t2 = t2.reshape(-1)
    ctx.upload_transforms(t2, q, s)
    ctx.upload_changed(changed)
    ctx.propagate(flags)
    g1, chg = ctx.download_global_transforms()
    tree_changed = O.mark_dirty_trees(parent, changed)
    rc, g1_exp, chg_exp = O.propagate_transforms(parent, t2, q, s, global_in=g0_exp, static_opt=static_opt,
                                                 tree_changed=tree_changed, transform_changed=changed)
    assert rc == 0
    bad = np.nonzero((g1.view(np.uint32) != g1_exp.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5].tolist()}"
    assert_bits(chg, chg_exp, "changed")


def test_cull_after_tree_propagation(ctx_factory):
    tr = W.gen_tree(8, 4)
    n = tr["n"]
    ctx = ctx_factory()
    upload_tree(ctx, tr)
    c = np.zeros(3 * n, F); h = np.full(3 * n, 0.5, F)
    ctx.upload_bounds(c, h)
    frusta = frusta_for([W.many_cubes_camera(0, position=(0.0, 0.0, 150.0))])
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    ctx.visibility_begin_frame()
    ctx.cull(frusta)
    ctx.visibility_end_frame()
    rc, g_exp, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    flags = np.full(n, 0x05, np.uint8)
    vv = O.reset_view_visibility(flags, np.zeros(n, np.uint8))
    vv, vis, _ = O.check_visibility(g_exp, c, h, flags, np.ones(n, np.uint32), vv, frusta)
    assert_bits(ctx.download_visibility(0), vis[0], "tree visibility")
    assert 0 < vis[0].sum() < n


# ---- clustering -----------------------------------------------------------------------------------

def cluster_case(ctx, cam, pos_range, obj_type=None, layers=None, spot_dir=None, spot_sin_cos=None,
                 req=(16, 9, 24), far_z=1000.0, ortho=False, view_mask=1, screen=(1920, 1080)):
    if ortho:
        from test_abi_and_host import ortho_clip_from_view
        cfv = ortho_clip_from_view(-60.0, 60.0, -33.75, 33.75, 0.1, 1000.0)
    else:
        cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, screen[0], screen[1], req, 5.0, far_z, view_mask)
    ov = O.cluster_view_setup(cam, cfv, fr, screen[0], screen[1], req, 5.0, far_z, view_mask)
    off, idx, counts, far, total = ctx.cluster_assign(view, pos_range, obj_type, layers, spot_dir, spot_sin_cos)
    eoff, eidx, ecounts, efar, etotal = O.assign_objects_to_clusters(ov, pos_range, obj_type, layers, spot_dir, spot_sin_cos)
    assert total == etotal, (total, etotal)
    assert np.array_equal(off, eoff), "cluster offsets"
    assert np.array_equal(idx, eidx), "cluster index lists (push order)"
    assert np.array_equal(counts, ecounts), "per-type counts"
    assert np.float32(far).tobytes() == np.float32(efar).tobytes(), (far, efar)
    return total


def test_cluster_many_lights_shape(ctx_factory):
    ctx = ctx_factory()
    pr = W.many_lights(20_000, 50.0, 0.3)
    total = cluster_case(ctx, W.many_cubes_camera(0), pr)
    assert total > 0
    cluster_case(ctx, W.many_cubes_camera(30, yaw=0.4), pr)
    cluster_case(ctx, W.many_cubes_camera(0), pr, ortho=True)


def test_cluster_mixed_objects(ctx_factory):
    rng = np.random.default_rng(5)
    n = 3000
    pos = rng.uniform(-60, 60, size=(n, 3)).astype(F)
    rng_r = np.where(rng.random(n) < 0.1, rng.uniform(20, 200, n), rng.uniform(0.5, 12, n)).astype(F)
    pr = np.concatenate([pos, rng_r[:, None]], axis=1).astype(F).reshape(-1)
    types = np.sort(rng.integers(0, 6, n)).astype(np.uint8)  # gather order: grouped by type
    layers = np.where(rng.random(n) < 0.1, 2, 1).astype(np.uint32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F).reshape(-1)
    ang = rng.uniform(0.1, 1.4, n).astype(F)
    sc = np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(F).reshape(-1)
    ctx = ctx_factory()
    for cam in (W.many_cubes_camera(0), W.many_cubes_camera(12, yaw=2.2, position=(5.0, 1.0, -3.0))):
        cluster_case(ctx, cam, pr, types, layers, d, sc)
        cluster_case(ctx, cam, pr, types, layers, d, sc, req=(17, 9, 24), far_z=120.0)
        cluster_case(ctx, cam, pr, types, layers, d, sc, req=(5, 3, 1))
        cluster_case(ctx, cam, pr, types, layers, d, sc, ortho=True)


def test_cluster_edge_cases(ctx_factory):
    ctx = ctx_factory()
    cam = W.many_cubes_camera(0)
    assert cluster_case(ctx, cam, np.zeros(0, F)) == 0                                  # no objects
    assert cluster_case(ctx, cam, np.array([0, 0, 500, 1.0], F)) == 0                   # behind the camera
    assert cluster_case(ctx, cam, np.array([0, 0, -20, 1e6], F)) == 16 * 9 * 24         # covers every cluster
    cluster_case(ctx, cam, np.array([0, 0, 0, 3.0, 0, 0, -5.0, 0.0, 0.2, 0.1, -0.1, 0.05], F))  # at the eye / zero range
    cluster_case(ctx, cam, np.array([0, 0, -20, 5.0], F), view_mask=2)                  # layer mismatch


def test_cluster_degenerate_grids(ctx_factory):
    """Grids whose plane tables do not fit in LDS next to the bit rows (4096 x 1 x 1), a single cluster, and an
    odd cluster count (unaligned accumulator sections)."""
    ctx = ctx_factory()
    cam = W.many_cubes_camera(0)
    pr = W.many_lights(6_000, 50.0, 3.0)
    assert cluster_case(ctx, cam, pr, req=(4096, 1, 1), screen=(4096, 8)) > 0
    assert cluster_case(ctx, cam, pr, req=(1, 1, 1)) > 0
    assert cluster_case(ctx, cam, pr, req=(3, 3, 3)) > 0
    assert cluster_case(ctx, cam, pr, req=(5, 7, 11)) > 0


def test_cluster_many_blocks_and_growth(ctx_factory):
    """More (cluster, block) pairs than fill waves, an index list that outgrows the initial device buffer, and
    repeated assignment on one context (frame-parity accumulators)."""
    ctx = ctx_factory()
    cam = W.many_cubes_camera(0)
    pr = W.many_lights(150_000, 50.0, 10.0)
    t1 = cluster_case(ctx, cam, pr)
    assert t1 > (1 << 18)  # outgrows the initial 2^18-entry device index buffer
    assert cluster_case(ctx, cam, pr[-4 * 70_001:]) > 0  # a different block count on the same context
    assert cluster_case(ctx, W.many_cubes_camera(30, yaw=1.0), pr) > 0
    assert cluster_case(ctx, cam, pr) == t1


def test_cluster_wire_format_bindings(ctx_factory):
    """The storage-buffer bindings (cluster_offsets_and_counts + clusterable_object_index_lists) built on the device
    equal the element-by-element CPU assembly of prepare_clusters_for_cpu_clustering, with and without a remap."""
    ctx = ctx_factory()
    cam = W.many_cubes_camera(0)
    rng = np.random.default_rng(4)
    n = 3_000
    pr = W.many_lights(n, 50.0, 3.0).reshape(n, 4)[::-1].copy()   # visible ones first
    types = np.sort(rng.integers(0, 6, n)).astype(np.uint8)        # gather order: grouped by type
    types[types == 1] = 0                                          # (no spot lights: they need cone inputs)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    off, idx, counts, far, total = ctx.cluster_assign(view, pr.reshape(-1), types)
    assert total > 0
    remap = rng.permutation(n).astype(np.uint32)
    remap[rng.random(n) < 0.05] = 0xFFFFFFFF                       # probes that have not loaded yet
    for rm in (None, remap):
        oc, il = ctx.cluster_download_bindings(view.n_clusters, rm)
        eoc, eil = O.cluster_bindings_storage(off, counts, idx, rm)
        assert np.array_equal(oc.reshape(-1), eoc) and np.array_equal(il, eil)
    assert np.array_equal(oc[:, 0], off[:-1]) and np.array_equal(oc[:, 1:4], counts[:, 0:3]) and np.array_equal(oc[:, 4:7], counts[:, 3:6])


def test_device_logf_matches_libm(ctx_factory):
    ctx = ctx_factory()
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 0x7F800000, 2_000_000, dtype=np.uint32)
    x = np.concatenate([bits.view(F), np.array([0.0, 1.0, np.inf, 1e-45, 1.17549435e-38, -1.0, np.nan], F)])
    got = ctx.debug_logf(x)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype = ctypes.c_float
    libm.logf.argtypes = [ctypes.c_float]
    sample = np.concatenate([np.arange(0, len(x), 37), np.arange(len(x) - 7, len(x))])
    for i in sample:
        e = F(libm.logf(float(x[i])))
        assert (np.isnan(e) and np.isnan(got[i])) or e.tobytes() == got[i].tobytes(), (x[i], e, got[i])


# ---- BASELINE-size run -------------------------------------------------------------------------

def test_big_context_fetches_transforms_past_the_caches(ctx_factory):
    """From 6 Mi rows on the fused frame reads the Transform columns with nontemporal loads (CULL_NT_LOADS, kernels.h): the same frame
    row for row against the oracle just above the threshold, ragged flags and a ragged last wave."""
    n = (6 << 20) + 77
    sc = W.many_cubes(n, radius=700.0, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(2), W.many_cubes_camera(4, yaw=2.0)])
    vmasks = np.array([1, 3], np.uint32)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(frusta, vmasks)
    ctx.visibility_end_frame()
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, np.zeros(n, np.uint8), frusta, vmasks, None)
    assert ctx.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes()
    for v in range(2):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"view {v} @6 Mi")
    vv, vv_chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, "ViewVisibility bytes @6 Mi")
    ctx.resize(0)  # (give the columns back: the module's contexts live until its end)


def test_one_million_flat_entities(ctx_factory):
    """configs[1]: 1M flat entities, 1 frustum -- compared row-for-row with the oracle (a few seconds of CPU)."""
    n = 1_000_000
    sc = W.many_cubes(n)
    frusta = frusta_for([W.many_cubes_camera(1)])
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(frusta)
    ctx.visibility_end_frame()
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, np.zeros(n, np.uint8), frusta, None, None)
    assert ctx.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes()
    vis = ctx.download_visibility(0)
    assert_bits(vis, vis_exp[0], "visibility @1M")
    frac = vis.mean()
    assert 0.01 < frac < 0.15, frac
    k, rows = ctx.download_visible_entities(0, 0)
    assert np.array_equal(rows, np.nonzero(vis_exp[0])[0].astype(np.uint32))


@pytest.mark.parametrize("n_comms,more,pipelined", [(1, 0, False), (1, 4, False), (1, 0, True), (2, 0, True), (2, 4, True), (3, 4, True)])
def test_mask_gatherer_single_rank_pipeline(ctx_factory, n_comms, more, pipelined):
    """The pipelined exchange bench.py uses for N > 1, on one rank: kernels write their masks in place into the
    gatherer's alternating buffers and the (1-rank) all-gather runs on the communication stream -- through RCCL
    directly when the library can be set up on this box, else through torch.distributed."""
    import torch
    from bevy_amd import sharding
    n, n_views = 20_000, 2
    sc = W.many_cubes(n, ragged_flags=True)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    g = sharding.MaskGatherer(n, 1, n_views, 0, device=torch.device("cuda", 0), n_comms=n_comms, pipelined=pipelined)
    assert g.mode in ("rccl-direct", "torch.distributed"), g.mode
    assert g.mode != "rccl-direct" or len(g.comms) == n_comms
    vv = np.zeros(n, np.uint8)
    outs = []
    for frame in range(4):
        frusta = frusta_for([W.many_cubes_camera(frame * 30), W.many_cubes_camera(frame * 30, yaw=1.3)])
        g.before_kernels(frame)
        ctx.bind_visibility_output(*g.bind_args(frame))
        ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
        ctx.synchronize()                      # the context runs on its own stream in this test
        buf = g.after_kernels(frame)
        _, vv, vis_exp, _ = oracle_frame(sc, vv, frusta, None, None)
        outs.append((buf, vis_exp))
        for v in range(n_views):
            assert_bits(ctx.download_visibility(v), vis_exp[v], f"frame {frame} view {v}")
    g.synchronize()
    for frame in (2, 3):                       # recent frames are still resident in the rotating buffers
        buf, vis_exp = outs[frame]
        words = buf.cpu().numpy()
        for v in range(n_views):
            assert_bits(sharding.unpack_view(words, n, 1, n_views, v), vis_exp[v], f"gathered frame {frame} view {v}")
    # the same exchange issued natively by the library (mi_exchange_configure), when RCCL could be set up
    ctx2 = ctx_factory()
    upload_scene(ctx2, sc)
    if g.attach(ctx2):
        assert g.mode == ("rccl-native-pipelined" if pipelined else "rccl-native")
        # more = MI_CULL_MORE_FRAMES: ignored while the exchange is on (the inline compaction carries its signal)
        vv = np.zeros(n, np.uint8)
        for frame in range(5):
            frusta = frusta_for([W.many_cubes_camera(frame * 30), W.many_cubes_camera(frame * 30, yaw=1.3)])
            ctx2.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | more)
            _, vv, vis_exp, _ = oracle_frame(sc, vv, frusta, None, None)
            ptr = ctx2.exchange_last(wait=True)
            assert ptr == g.buffer(frame).data_ptr()
            words = g.buffer(frame).cpu().numpy()
            for v in range(n_views):
                assert_bits(sharding.unpack_view(words, n, 1, n_views, v), vis_exp[v], f"native exchange frame {frame} view {v}")
                assert_bits(ctx2.download_visibility(v), vis_exp[v], f"native exchange download frame {frame} view {v}")
        # a burst with nothing read in between: the caller's thread is paced by the per-communicator counters
        for frame in range(5, 5 + 23):
            frusta = frusta_for([W.many_cubes_camera(frame * 30), W.many_cubes_camera(frame * 30, yaw=1.3)])
            ctx2.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | more)
            _, vv, vis_exp, _ = oracle_frame(sc, vv, frusta, None, None)
        assert ctx2.exchange_last(wait=True) == g.buffer(frame).data_ptr()
        ctx2.synchronize()
        words = g.buffer(frame).cpu().numpy()
        for v in range(n_views):
            assert_bits(sharding.unpack_view(words, n, 1, n_views, v), vis_exp[v], f"native exchange after the burst, view {v}")
            assert np.array_equal(ctx2.download_visible_entities(v, 0)[1], np.nonzero(vis_exp[v])[0].astype(np.uint32))
        ctx2.exchange_configure(None, None, None, 0, 0, 0, 0)
        ctx2.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
        assert_bits(ctx2.download_visibility(0), vis_exp[0], "after switching the exchange off")
        g.native = False
        if n_comms == 1 and g.attach(ctx2):
            # the start-up rule that picks the mode (MaskGatherer.calibrate: both modes timed over the same frames, the faster one kept):
            # whichever it keeps, the frames behind it gather the right masks
            def run_frames(k):
                for f in range(k):
                    ctx2.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | more)
                ctx2.exchange_last(True)
                ctx2.synchronize()
            rec = g.calibrate(ctx2, run_frames, frames=12)
            assert rec["chosen"] in ("simple", "pipelined") and g.native
            assert g.mode == ("rccl-native-pipelined" if rec["chosen"] == "pipelined" else "rccl-native")
            if "simple_us_per_frame" in rec:
                assert rec["simple_us_per_frame"] > 0 and rec["pipelined_us_per_frame"] > 0
            k0 = None
            for frame in range(3):
                fr2 = frusta_for([W.many_cubes_camera(7 + frame * 11), W.many_cubes_camera(frame * 3, yaw=0.6)])
                ctx2.propagate_and_cull(fr2, flags=B.CULL_END_FRAME | more)
                _, vv, vis_exp, _ = oracle_frame(sc, vv, fr2, None, None)
                ptr = ctx2.exchange_last(wait=True)
                bufs = [b for b in g.bufs if b.data_ptr() == ptr]
                assert len(bufs) == 1
                words = bufs[0].cpu().numpy()
                for v in range(n_views):
                    assert_bits(sharding.unpack_view(words, n, 1, n_views, v), vis_exp[v], f"behind the calibration, frame {frame} view {v}")
            ctx2.exchange_configure(None, None, None, 0, 0, 0, 0)
    g.close()


@pytest.mark.parametrize("static_opt", [False, True])
def test_sparse_dirty_upload_and_changed_readback(ctx_factory, static_opt):
    """The steady-state PCIe path: upload only the Changed<Transform> rows (indexed), propagate, read back only the
    rows whose GlobalTransform tick was bumped.  Same result as the dense upload + full download."""
    n = 40_000
    parent_old = random_forest(n, 17)
    new_to_old, parent, offs = api.hierarchy_sort(parent_old)
    rng = np.random.default_rng(23)
    t = rng.normal(size=(n, 3)).astype(F)
    q = rng.normal(size=(n, 4)); q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    s = rng.uniform(0.8, 1.25, size=(n, 3)).astype(F)
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    ctx.resize(n)
    ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s.reshape(-1))
    ctx.upload_hierarchy(parent, offs)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    rows0, g0s = ctx.download_changed_global_transforms()
    g0 = ctx.download_global_transforms(want_changed=False)
    assert np.array_equal(rows0, np.arange(n, dtype=np.uint32)) and g0s.tobytes() == g0.tobytes()   # first frame: all
    for frame in range(3):
        dirty = np.nonzero(rng.random(n) < 0.003)[0].astype(np.uint32)
        rng.shuffle(dirty)
        t[dirty] += F(0.25)
        q2 = q[dirty] + rng.normal(scale=0.05, size=(len(dirty), 4)).astype(F)
        q[dirty] = (q2 / np.linalg.norm(q2, axis=1, keepdims=True)).astype(F)
        ctx.upload_transforms_indexed(dirty, t[dirty].reshape(-1), q[dirty].reshape(-1), s[dirty].reshape(-1))
        ctx.propagate(flags)
        changed = np.zeros(n, np.uint8); changed[dirty] = 1
        tree_changed = O.mark_dirty_trees(parent, changed)
        rc, g1, chg = O.propagate_transforms(parent, t.reshape(-1), q.reshape(-1), s.reshape(-1), global_in=g0,
                                             static_opt=static_opt, tree_changed=tree_changed, transform_changed=changed)
        assert rc == 0
        rows, gs = ctx.download_changed_global_transforms()
        exp_rows = np.nonzero(chg)[0].astype(np.uint32)
        assert np.array_equal(rows, exp_rows), (frame, len(rows), len(exp_rows))
        assert gs.tobytes() == g1.reshape(n, 12)[exp_rows].tobytes()
        assert ctx.download_global_transforms(want_changed=False).tobytes() == g1.tobytes()
        g0 = g1
    # nothing dirty: nothing changes, nothing comes back
    ctx.upload_transforms_indexed(np.zeros(0, np.uint32), np.zeros(0, F), np.zeros(0, F), np.zeros(0, F))
    ctx.propagate(flags)
    rows, gs = ctx.download_changed_global_transforms()
    none = np.zeros(n, np.uint8)
    rc, g2, chg = O.propagate_transforms(parent, t.reshape(-1), q.reshape(-1), s.reshape(-1), global_in=g0, static_opt=static_opt,
                                         tree_changed=O.mark_dirty_trees(parent, none), transform_changed=none)
    # without the static-scene optimisation every root with children is re-assigned each frame (systems.rs:522-530)
    assert np.array_equal(rows, np.nonzero(chg)[0].astype(np.uint32)) and (static_opt and len(rows) == 0 or not static_opt)


def test_columns_grow_and_shrink_keep_rows(ctx_factory):
    """mi_columns_resize: growing reallocates the columns (contents of existing rows kept), shrinking keeps them."""
    n0, n1 = 5_000, 70_000
    sc = W.many_cubes(n1, ragged_flags=True)
    frusta = frusta_for([W.many_cubes_camera(1)])
    ctx = ctx_factory()
    ctx.resize(n0)
    ctx.upload_transforms(sc["translation"][:3 * n0], sc["rotation"][:4 * n0], sc["scale"][:3 * n0])
    ctx.upload_bounds(sc["aabb_center"][:3 * n0], sc["aabb_half"][:3 * n0], sc["flags"][:n0], sc["layers"][:n0])
    ctx.resize(n1)   # grows: rows [0, n0) must survive the reallocation
    ctx.upload_transforms(sc["translation"][3 * n0:], sc["rotation"][4 * n0:], sc["scale"][3 * n0:], first_row=n0)
    ctx.upload_bounds(sc["aabb_center"][3 * n0:], sc["aabb_half"][3 * n0:], sc["flags"][n0:], sc["layers"][n0:], first_row=n0)
    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    g_exp, vv_exp, vis_exp, _ = oracle_frame(sc, np.zeros(n1, np.uint8), frusta, None, None)
    assert ctx.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes()
    assert_bits(ctx.download_visibility(0), vis_exp[0], "after growth")
    ctx.resize(n0)   # shrink: the first rows are still there
    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    assert ctx.download_global_transforms(want_changed=False).tobytes() == g_exp[:12 * n0].tobytes()
    assert_bits(ctx.download_visibility(0), vis_exp[0][:n0], "after shrink")
    ctx.resize(0)
    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    assert len(ctx.download_visibility(0)) == 0


def test_error_codes(ctx_factory):
    ctx = ctx_factory()
    ctx.resize(100)
    with pytest.raises(api.MiError) as e:
        ctx.upload_transforms(np.zeros(3 * 200, F), np.zeros(4 * 200, F), np.zeros(3 * 200, F))
    assert e.value.code == api.MI_ERR_INVALID_ARG
    with pytest.raises(api.MiError) as e:
        ctx.upload_transforms_indexed(np.array([5, 100], np.uint32), np.zeros(6, F), np.zeros(8, F), np.zeros(6, F))
    assert e.value.code == api.MI_ERR_INVALID_ARG
    with pytest.raises(api.MiError) as e:
        ctx.download_visibility(0)                      # nothing culled yet
    assert e.value.code == api.MI_ERR_NOT_READY
    with pytest.raises(api.MiError) as e:               # a child whose parent is not in the previous level
        ctx.upload_hierarchy(np.array([B.NO_PARENT] + [0] * 98 + [50], np.uint32), np.array([0, 1, 100], np.uint32))
    assert e.value.code == api.MI_ERR_MALFORMED_HIERARCHY
    with pytest.raises(api.MiError) as e:
        api.hierarchy_sort(np.array([1, 0], np.uint32))  # a 2-cycle
    assert e.value.code == api.MI_ERR_MALFORMED_HIERARCHY
    frusta = frusta_for([W.many_cubes_camera(0)])
    ctx.upload_hierarchy(np.array([B.NO_PARENT] + [0] * 99, np.uint32), np.array([0, 1, 100], np.uint32))
    ctx.propagate_and_cull(frusta)                       # with a hierarchy the frame call is mi_propagate + mi_cull (tests/test_gpu_round3.py)
    with pytest.raises(api.MiError) as e:
        ctx.propagate_and_cull(frusta, flags=B.CULL_WITH_CLUSTERS)   # nothing bound, no cluster view
    assert e.value.code == api.MI_ERR_NOT_READY
    with pytest.raises(api.MiError) as e:
        api.Context(device=4096)
    assert e.value.code == api.MI_ERR_INVALID_ARG


def test_changed_mesh_inputs_wire_format(ctx_factory):
    """MeshInputUniform::world_from_local (transposed affine) + MeshCullingData for exactly the rows whose
    GlobalTransform changed (flat scene: the dirty rows), bit for bit against the restated to_transpose / new()."""
    n = 20_000
    sc = W.many_cubes(n, ragged_flags=True)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    rng = np.random.default_rng(8)
    dirty = np.sort(rng.choice(n, 300, replace=False)).astype(np.uint32)
    t = sc["translation"].reshape(n, 3).copy()
    t[dirty] += F(2.0)
    ctx.upload_transforms_indexed(dirty, t[dirty].reshape(-1), sc["rotation"].reshape(n, 4)[dirty].reshape(-1),
                                  sc["scale"].reshape(n, 3)[dirty].reshape(-1))
    ctx.propagate(0)
    rows, wfl, cull = ctx.download_changed_mesh_inputs()
    assert np.array_equal(rows, dirty)
    g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
    ewfl, ecull = O.mesh_inputs(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], dirty)
    assert wfl.tobytes() == ewfl.tobytes() and cull.tobytes() == ecull.tobytes()
    assert np.isinf(cull.reshape(-1, 8)[:, 4]).any() and not np.isinf(cull.reshape(-1, 8)[:, 4]).all()


def test_non_finite_and_denormal_inputs_match_oracle(ctx_factory):
    """NaN / Inf / zero / denormal components in Transform and Aabb: the comparisons of the reference are written so
    that NaN never culls (`<= 0.0` is false) and GlobalTransform's PartialEq treats NaN as changed; denormals must not
    be flushed.  Whatever the CPU arithmetic does, the device must do bit for bit."""
    n = 4_096
    sc = W.many_cubes(n, radius=30.0, ragged_flags=True)
    rng = np.random.default_rng(12)
    t = sc["translation"].reshape(n, 3); s = sc["scale"].reshape(n, 3); h = sc["aabb_half"].reshape(n, 3)
    q = sc["rotation"].reshape(n, 4); c = sc["aabb_center"].reshape(n, 3)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -3e-45, 3.4e38, 1.17549435e-38], F)
    for arr in (t, s, h, q, c):
        rows = rng.choice(n, 200, replace=False)
        arr[rows, rng.integers(0, arr.shape[1], 200)] = special[rng.integers(0, len(special), 200)]
    frusta = frusta_for([W.many_cubes_camera(0), W.many_cubes_camera(4, yaw=2.0)])
    vv0 = (W.splitmix64(4, n) % np.uint64(4)).astype(np.uint8)
    ctx = ctx_factory()
    upload_scene(ctx, sc, vv0)
    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    g_exp, vv_exp, vis_exp, chg_exp = oracle_frame(sc, vv0, frusta, None, None)
    g = ctx.download_global_transforms(want_changed=False)
    diff = (g.view(np.uint32) != g_exp.view(np.uint32)) & ~(np.isnan(g) & np.isnan(g_exp))   # NaN payload / sign is not specified
    bad = np.nonzero(diff.reshape(-1, 12).any(axis=1))[0]
    detail = [(int(r), g.reshape(-1, 12)[r].view(np.uint32).tolist(), g_exp.reshape(-1, 12)[r].view(np.uint32).tolist()) for r in bad[:3]]
    assert bad.size == 0, f"{bad.size} GlobalTransforms differ, e.g. {detail}"
    for v in range(2):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"view {v}")
    vv, chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, "vv")
    assert_bits(chg, chg_exp, "vv changed")
    # the same rows as a (flat-level + one level) hierarchy: set_if_neq with NaN is always "changed"
    parent = np.full(n, B.NO_PARENT, np.uint32)
    parent[n // 2:] = np.arange(n // 2, dtype=np.uint32)
    new_to_old, pidx, offs = api.hierarchy_sort(parent)
    ctx2 = ctx_factory()
    ctx2.resize(n)
    ctx2.upload_transforms(t[new_to_old].reshape(-1), q[new_to_old].reshape(-1), s[new_to_old].reshape(-1))
    ctx2.upload_hierarchy(pidx, offs)
    for frame in range(2):
        ctx2.propagate(B.PROPAGATE_ALL_DIRTY)
    g2, chg2 = ctx2.download_global_transforms()
    rc, e1, _ = O.propagate_transforms(pidx, t[new_to_old].reshape(-1), q[new_to_old].reshape(-1), s[new_to_old].reshape(-1))
    rc, e2, echg = O.propagate_transforms(pidx, t[new_to_old].reshape(-1), q[new_to_old].reshape(-1), s[new_to_old].reshape(-1), global_in=e1)
    diff = (g2.view(np.uint32) != e2.view(np.uint32)) & ~(np.isnan(g2) & np.isnan(e2))
    bad = np.nonzero(diff.reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"tree: {bad.size} rows differ, first {bad[:5].tolist()}"
    assert_bits(chg2, echg, "tree change ticks with NaN")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5])
def test_hierarchy_shards_reproduce_the_whole_tree(ctx_factory, world):
    """SURVEY 8e, hierarchy: shard by root subtree (sharding.shard_hierarchy), one context per shard as the ranks of an
    N-GPU run would hold them; the owned rows of all shards together are the unsharded GlobalTransforms, bit for bit."""
    from bevy_amd import sharding
    tr = W.gen_tree(9, 4, 60_000)
    n = tr["n"]
    full = ctx_factory()
    full.resize(n)
    full.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
    full.upload_hierarchy(tr["parent"], tr["level_offsets"])
    full.propagate(B.PROPAGATE_ALL_DIRTY)
    g_full = full.download_global_transforms(want_changed=False).reshape(n, 12)
    _, g_orc, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert g_full.tobytes() == g_orc.tobytes()
    got = np.zeros_like(g_full)
    seen = np.zeros(n, np.int64)
    for sh in sharding.shard_hierarchy(tr["parent"], tr["level_offsets"], world):
        rows = sh["rows"].astype(np.int64)
        c = ctx_factory()
        c.resize(len(rows))
        c.upload_transforms(*(tr[k].reshape(n, -1)[rows].reshape(-1) for k in ("translation", "rotation", "scale")))
        c.upload_hierarchy(sh["parent"], sh["level_offsets"])
        c.propagate(B.PROPAGATE_ALL_DIRTY)
        g = c.download_global_transforms(want_changed=False).reshape(-1, 12)
        got[rows[sh["owned"]]] = g[sh["owned"]]
        seen[rows[sh["owned"]]] += 1
    assert np.all(seen == 1)
    assert got.tobytes() == g_full.tobytes()


@pytest.mark.gpu
def test_ten_million_entities_four_views_properties(ctx_factory):
    """configs[3] at full size on one GPU (the 8-GPU run holds an eighth of it per rank): 10 M entities, 4 cameras at
    yaw 0/90/180/270.  The oracle needs minutes for the whole scene, so the full-size run is checked through
    size-independent properties -- rows are independent, so any window of rows culled on its own (fresh context) must
    give the same bits; three windows are compared with the oracle bit for bit (GlobalTransform included); per view the
    sorted list is exactly the set bits of the mask; ViewVisibility is the OR over the views."""
    n = 10_000_000
    radius = 500.0 * 10.0 ** (1.0 / 3.0)
    sc = W.many_cubes(n, radius=radius)
    frusta = frusta_for([W.many_cubes_camera(3, yaw=k * math.pi / 2) for k in range(4)])
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
    vis = [ctx.download_visibility(v) for v in range(4)]
    vv, _ = ctx.download_view_visibility()
    any_vis = np.zeros(n, bool)
    total = 0
    for v in range(4):
        rows = ctx.download_visible_entities(v, 0)[1]
        assert np.array_equal(rows, np.nonzero(vis[v])[0].astype(np.uint32)), f"view {v}: list != set bits of the mask"
        any_vis |= vis[v].astype(bool)
        total += len(rows)
    assert 0.02 * n < total < 0.4 * n
    assert np.array_equal((vv & 1).astype(bool), any_vis), "ViewVisibility = OR over the views"
    assert np.all((vv & 2) == 0), "after MarkNewlyHidden only bit 0 survives"
    g = ctx.download_global_transforms(want_changed=False).reshape(n, 12)
    for lo in (0, 4_999_936, n - 65_536):  # windows on 256-row boundaries, 64 k rows each
        hi = lo + 65_536
        sub = {k: (sc[k].reshape(n, -1)[lo:hi].reshape(-1) if k != "n" else hi - lo) for k in sc}
        g_exp, vv_exp, vis_exp, _ = oracle_frame(sub, np.zeros(hi - lo, np.uint8), frusta, None, None)
        assert g[lo:hi].tobytes() == g_exp.tobytes(), f"GlobalTransform window at {lo}"
        for v in range(4):
            assert_bits(vis[v][lo:hi], vis_exp[v], f"window at {lo}, view {v}")
        assert_bits(vv[lo:hi], vv_exp, f"ViewVisibility window at {lo}")
        c2 = ctx_factory()
        upload_scene(c2, sub)
        c2.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
        for v in range(4):
            assert_bits(c2.download_visibility(v), vis[v][lo:hi], f"window at {lo} culled on its own, view {v}")
        c2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tree", [False, True])
def test_static_frames_change_nothing_and_report_nothing(ctx_factory, tree):
    """0 % dirty frames (configs[1], second run): after the change column was consumed and nothing was marked again,
    mi_propagate returns without launching; GlobalTransforms stay bit-identical and the changed masks read empty --
    what the reference's Changed<Transform> filter + set_if_neq give (systems.rs:45-50, :719).  A later dirty row is
    picked up as usual."""
    if tree:
        tr = W.gen_tree(7, 4)
        n = tr["n"]
        t, r, s = tr["translation"], tr["rotation"], tr["scale"]
    else:
        sc = W.many_cubes(20_000, radius=30.0)
        n = sc["n"]
        t, r, s = sc["translation"], sc["rotation"], sc["scale"]
    ctx = ctx_factory()
    ctx.resize(n)
    ctx.upload_transforms(t, r, s)
    if tree:
        ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
    flags = B.PROPAGATE_STATIC_OPT if tree else 0  # without it the reference re-assigns (and ticks) every root each frame
    ctx.upload_changed(np.ones(n, np.uint8))
    ctx.propagate(flags)
    g0, chg0 = ctx.download_global_transforms()
    assert chg0.all()
    for _ in range(3):
        ctx.propagate(flags)
        g1, chg1 = ctx.download_global_transforms()
        assert g1.tobytes() == g0.tobytes() and not chg1.any()
    # one row moves
    row = n // 2
    t2 = t.reshape(n, 3)[row].copy() + np.float32(1.5)
    ctx.upload_transforms_indexed(np.array([row], np.uint32), t2, r.reshape(n, 4)[row], s.reshape(n, 3)[row])
    ctx.propagate(flags)
    g2, chg2 = ctx.download_global_transforms()
    tt = t.copy().reshape(n, 3)
    tt[row] = t2
    if tree:
        _, g_exp, _ = O.propagate_transforms(tr["parent"], tt.reshape(-1), r, s)
    else:
        g_exp = O.full_frame(tt.reshape(-1), r, s, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], np.zeros(n, np.uint8),
                             frusta_for([W.many_cubes_camera(0)]), None, None)[0]
    assert g2.tobytes() == g_exp.tobytes()
    assert chg2[row] and 1 <= chg2.sum() < n
    ctx.propagate(flags)
    assert not ctx.download_global_transforms()[1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("with_classes", [False, True])
def test_deferred_compaction_matches_inline(ctx_factory, with_classes):
    """MI_CULL_MORE_FRAMES: frame f's lists are built by the tail workgroups of frame f+1's kernel, or by a launch of
    their own at the first entry point that exposes them.  Whatever is read -- after every frame, after a burst of
    frames with nothing read in between, through the batching build, through mi_cull after mi_propagate, after a
    resize -- must be what the inline compaction gives."""
    n = 300_000 + 77
    sc = W.many_cubes(n, radius=300.0, ragged_flags=True)
    cams = [frusta_for([W.many_cubes_camera(f, yaw=0.0), W.many_cubes_camera(f, yaw=2.0)]) for f in range(18)]
    cm = (1 + (np.arange(n) % 3 == 0) * 2 + (np.arange(n) % 5 == 0) * 4).astype(np.uint32)
    bs = W.batching_scene(n, n_sets=5, seed=9)
    classes = (0, 1, 2) if with_classes else (0,)

    def lists(ctx):
        return [ctx.download_visible_entities(v, c)[1].copy() for v in range(2) for c in classes]

    def run(more):
        ctx = ctx_factory()
        upload_scene(ctx, sc)
        if with_classes:
            ctx.upload_visibility_classes(cm)
        ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
        ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
        outs = []
        for f in range(3):  # read after every frame: the join launches the deferred compaction on its own
            ctx.propagate_and_cull(cams[f], flags=B.CULL_END_FRAME | more)
            outs.append(lists(ctx) + [ctx.download_visibility(v).copy() for v in range(2)] + [ctx.download_view_visibility()[0].copy()])
        for f in range(3, 14):  # burst: every compaction rides in the next frame's launch
            ctx.propagate_and_cull(cams[f], flags=B.CULL_END_FRAME | more)
        outs.append(lists(ctx))
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)  # unfused calls, a different number of views in between
        ctx.cull(cams[14][:24], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
        ctx.cull(cams[15], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | more)
        ctx.batch_build(1, 0)                  # enqueues the pending compaction, then reads the list on the same stream
        got = ctx.batch_download()
        outs.append([got["work_items"][0], got["work_items"][1], got["records"]] + lists(ctx))
        ctx.propagate_and_cull(cams[16], flags=B.CULL_END_FRAME | more)
        ctx.resize(n - 1000)                   # joins first (buffers may move), then fewer rows
        ctx.propagate_and_cull(cams[17], flags=B.CULL_END_FRAME | more)
        ctx.synchronize()
        outs.append(lists(ctx))
        return outs

    a, b = run(B.CULL_MORE_FRAMES), run(0)
    assert len(a) == len(b)
    for k, (fa, fb) in enumerate(zip(a, b)):
        assert len(fa) == len(fb)
        for x, y in zip(fa, fb):
            assert x.shape == y.shape and np.array_equal(x, y), f"step {k}"
    assert a[0][0].size > 0 and a[3][0].size > 0
