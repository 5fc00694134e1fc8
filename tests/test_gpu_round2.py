"""GPU regression tests for the round-1 advisor findings: every scenario is driven through the C ABI and
checked against the oracle (or against the inline path of the library where the finding is about a fast path)."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O
from test_gpu_parity import ctx_factory, frusta_for, upload_scene, upload_tree, assert_bits  # noqa: F401

pytestmark = pytest.mark.gpu
F = np.float32


def flat_rows_plus_deep_tree(n_flat=45_000, chain=11, fan_levels=6, seed=3):
    """Level 0 = n_flat flat rows + ONE root; below the root a chain of `chain` single nodes, then a 4-ary fan.  The
    tile planner cuts this into three launches' worth of passes -- pass 0 (> 64 root tiles) on its own, pass 1 (the
    chain) on its own, pass 2 (the fan: chain tiles) riding in pass 1's launch -- so the chain tiles' ancestor chain
    runs up through rows owned by TWO earlier launches."""
    rng = np.random.default_rng(seed)
    parent = [B.NO_PARENT] * (n_flat + 1)
    level_offsets = [0, n_flat + 1]
    prev = [n_flat]  # the root is the last row of level 0
    for _ in range(chain):
        row = len(parent)
        parent.append(prev[0])
        prev = [row]
        level_offsets.append(len(parent))
    for _ in range(fan_levels):
        nxt = []
        for p in prev:
            for _k in range(4):
                nxt.append(len(parent))
                parent.append(p)
        prev = nxt
        level_offsets.append(len(parent))
    n = len(parent)
    q = rng.normal(size=(n, 4)).astype(F)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(F)
    return dict(n=n, parent=np.array(parent, np.uint32), level_offsets=np.array(level_offsets, np.uint32),
                translation=rng.uniform(-3, 3, size=3 * n).astype(F), rotation=q.reshape(-1).astype(F),
                scale=rng.uniform(0.9, 1.1, size=3 * n).astype(F), root=n_flat, n_flat=n_flat)


@pytest.mark.parametrize("static_opt", [True, False])
def test_chain_tiles_below_two_owner_launches(ctx_factory, static_opt):
    """every launch mirrors the rows it owns into the pre-frame snapshot the chain tiles read.  The frames
    below move the root away and back while a local in the chain changes in between -- with a stale snapshot the chain
    tiles would compare against (and, under the static-scene rule, fall back to) values from the first frame."""
    tr = flat_rows_plus_deep_tree()
    n, parent, root = tr["n"], tr["parent"], tr["root"]
    t = tr["translation"].reshape(n, 3).copy()
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    upload_tree(ctx, tr)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    rc, g0, _ = O.propagate_transforms(parent, tr["translation"], tr["rotation"], tr["scale"], static_opt=static_opt)
    assert rc == 0 and ctx.download_global_transforms(want_changed=False).tobytes() == g0.tobytes()
    chain3, chain8 = root + 3, root + 8  # rows of the chain: level 3 is owned by pass 0, level 8 by pass 1
    fan = n - 5
    t_root0 = t[root].copy()
    frames = [
        {root: t_root0 + F(2.0)},                     # root moves away
        {chain3: t[chain3] + F(0.25)},                # an intermediate local changes ...
        {root: t_root0},                              # ... and the root moves back
        {},                                           # nothing changed
        {chain8: t[chain8] - F(0.5), fan: t[fan] + F(1.0)},
        {root: t_root0 + F(2.0), 5: t[5] + F(1.0)},   # root again + a flat row
        {root: t_root0},
    ]
    for f, upd in enumerate(frames):
        rows = np.array(sorted(upd), np.uint32)
        for r_, v in upd.items():
            t[r_] = v
        if rows.size:
            ctx.upload_transforms_indexed(rows, t[rows].reshape(-1), tr["rotation"].reshape(n, 4)[rows].reshape(-1),
                                          tr["scale"].reshape(n, 3)[rows].reshape(-1))
        ctx.propagate(flags)
        changed = np.zeros(n, np.uint8)
        changed[rows] = 1
        rc, g1, chg = O.propagate_transforms(parent, t.reshape(-1), tr["rotation"], tr["scale"], global_in=g0, static_opt=static_opt,
                                             tree_changed=O.mark_dirty_trees(parent, changed), transform_changed=changed)
        assert rc == 0
        g, got_chg = ctx.download_global_transforms()
        bad = np.nonzero((g.view(np.uint32) != g1.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
        assert bad.size == 0, f"frame {f}: {bad.size} rows differ, first {bad[:5].tolist()}"
        assert_bits(got_chg, chg, f"frame {f} change ticks")
        g0 = g1


def test_deferred_compaction_with_a_caller_bound_mask_buffer(ctx_factory):
    """mi_bind_visibility_output + MI_CULL_MORE_FRAMES without the exchange and without class masks: the one
    bound buffer is overwritten by the next frame, so the library must not defer the compaction into that frame."""
    import torch
    n = 200_000 + 13
    sc = W.many_cubes(n, radius=260.0)
    cams = [frusta_for([W.many_cubes_camera(f, yaw=0.7 * f)]) for f in range(6)]
    words = (n + 255) // 256 * 4

    def run(more, bind):
        ctx = ctx_factory()
        upload_scene(ctx, sc)
        buf = torch.zeros(words, dtype=torch.int64, device="cuda")
        if bind:
            ctx.bind_visibility_output(buf.data_ptr(), words, 0)
        outs = []
        for f in range(5):
            ctx.propagate_and_cull(cams[f], flags=B.CULL_END_FRAME | more)
        ctx.propagate_and_cull(cams[5], flags=B.CULL_END_FRAME | more)
        outs.append(ctx.download_visible_entities(0, 0)[1].copy())
        # the frame before the last one: re-run 4 then 5, read the lists of 4 through a deferred join
        ctx.propagate_and_cull(cams[4], flags=B.CULL_END_FRAME | more)
        outs.append(ctx.download_visible_entities(0, 0)[1].copy())
        ctx.synchronize()
        del buf
        return outs

    ref = run(0, False)
    for more, bind in ((B.CULL_MORE_FRAMES, True), (0, True), (B.CULL_MORE_FRAMES, False)):
        got = run(more, bind)
        for a, b in zip(ref, got):
            assert a.size > 0 and np.array_equal(a, b), f"more={more} bind={bind}"


def test_failed_frame_keeps_the_previous_frames_lists(ctx_factory):
    """a cull call that fails validation must not swallow the deferred compaction of the frame before it."""
    n = 120_000
    sc = W.many_cubes(n, radius=240.0)
    fr = frusta_for([W.many_cubes_camera(0)])
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
    want = ctx.download_visible_entities(0, 0)[1].copy()
    vis = ctx.download_visibility(0).copy()
    ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES)
    with pytest.raises(api.MiError) as e:
        ctx.cull(np.zeros(0, F), flags=B.CULL_MORE_FRAMES)  # no views
    assert e.value.code == api.MI_ERR_INVALID_ARG
    got = ctx.download_visible_entities(0, 0)[1]
    assert want.size > 0 and np.array_equal(got, want)
    assert_bits(ctx.download_visibility(0), vis, "masks after the failed call")
    ctx.synchronize()


def test_rows_regrown_inside_the_capacity_are_fresh(ctx_factory):
    """shrink, then regrow within the allocation: the rows that come back are new entities -- default flags /
    layers / ViewVisibility, no batch set, and Added<GlobalTransform> (computed by the next mi_propagate(0))."""
    n = 40_000
    half = n // 2
    sc = W.many_cubes(n, ragged_flags=True)
    fr = frusta_for([W.many_cubes_camera(0)])
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.upload_changed(np.zeros(n, np.uint8))  # the change column is in use
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    ctx.cull(fr, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
    ctx.resize(half)
    ctx.resize(n)  # no reallocation
    sc2 = W.many_cubes(n, radius=333.0)
    ctx.upload_transforms(sc2["translation"][3 * half:], sc2["rotation"][4 * half:], sc2["scale"][3 * half:], first_row=half)
    # bounds / flags deliberately NOT uploaded for the regrown rows: they must read as the defaults of a fresh row
    ctx.propagate(0)
    g, chg = ctx.download_global_transforms()
    g_new, _, _, _ = O.full_frame(sc2["translation"], sc2["rotation"], sc2["scale"], sc2["aabb_center"], sc2["aabb_half"],
                                  sc2["flags"], sc2["layers"], np.zeros(n, np.uint8), fr)
    g_old, _, _, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"],
                                  sc["flags"], sc["layers"], np.zeros(n, np.uint8), fr)
    assert g[:12 * half].tobytes() == g_old[:12 * half].tobytes()
    assert g[12 * half:].tobytes() == g_new[12 * half:].tobytes(), "regrown rows were not recomputed as Added<GlobalTransform>"
    assert not chg[:half].any() and chg[half:].all()
    vv, _ = ctx.download_view_visibility()
    assert not vv[half:].any(), "regrown rows kept a previous occupant's ViewVisibility"
    # default flags = InheritedVisibility only (no Aabb): such a row is visible in every view that shares layer 0
    ctx.cull(fr, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
    assert ctx.download_visibility(0)[half:].all()


def test_first_indexed_upload_keeps_rows_that_were_never_propagated(ctx_factory):
    """ADVICE 5 (second half): the first mi_upload_transforms_indexed starts the change column; rows no propagate has
    consumed yet are still Added<GlobalTransform> and must be computed by the following mi_propagate(0)."""
    n = 10_000
    sc = W.many_cubes(n)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    rows = np.array([7, 4000], np.uint32)
    t = sc["translation"].reshape(n, 3).copy()
    t[rows] += F(1.0)
    ctx.upload_transforms_indexed(rows, t[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1),
                                  sc["scale"].reshape(n, 3)[rows].reshape(-1))
    ctx.propagate(0)
    g, chg = ctx.download_global_transforms()
    g_exp, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
    assert g.tobytes() == g_exp.tobytes() and chg.all()


def test_tile_plans_are_one_launch_for_deep_and_wide_hierarchies(ctx_factory):
    """The planner cuts levels into bands bottom-up; roots and chain bands share ONE launch whatever the depth.  Parity of
    every shape is covered by the tree tests; this pins the launch count the performance rests on."""
    ctx = ctx_factory()
    ctx.debug_set_tile_mode(0)  # (the suite also runs under MI_TEST_TILE_MODE=1, which sweeps level by level: this test is about the tiles)
    for tr, want_launches in ((W.gen_tree(12, 4, 1_000_000), 1), (W.gen_tree(20, 2, 300_000), 1), (W.gen_tree(4, 40), 1),
                              (flat_rows_plus_deep_tree(), 1), (W.gen_tree(3, 1000), None)):
        upload_tree(ctx, tr)
        plan = ctx.debug_tile_plan()
        assert plan["tiles"] > 0 and plan["bands"] >= 1
        if want_launches is not None:
            assert plan["launches"] == want_launches, plan
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        rc, g_exp, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
        assert rc == 0 and ctx.download_global_transforms(want_changed=False).tobytes() == g_exp.tobytes(), plan


@pytest.mark.parametrize("static_opt", [False, True])
def test_very_wide_deepest_level_is_streamed(ctx_factory, static_opt):
    """A very wide deepest level is not tiled: k_propagate_level sweeps it in a launch of its own behind the level above (kernels.h
    STREAM_LEVEL_MIN_ROWS_LAST: 2^23 rows, 2^20 with the test thresholds of mi_debug_set_tile_mode(3)).  Same per-node rule, so: all dirty, a moved root, sparse dirty rows under the
    static-scene rule and a static frame must all be the oracle's bits -- and visibility_propagate walks the same plan."""
    tr = W.gen_tree(3, 1100)  # 1 + 1100 + 1 210 000 rows
    n, parent = tr["n"], tr["parent"]
    assert tr["level_offsets"][-1] - tr["level_offsets"][-2] >= (1 << 20)
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    ctx.debug_set_tile_mode(3)
    upload_tree(ctx, tr)
    plan = ctx.debug_tile_plan()
    assert plan["launches"] == 1 and plan["tiles"] < 100, plan   # the 1.21 M leaves are not in tiles
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    rc, g0, chg0 = O.propagate_transforms(parent, tr["translation"], tr["rotation"], tr["scale"], static_opt=static_opt)
    g, chg = ctx.download_global_transforms()
    assert rc == 0 and g.tobytes() == g0.tobytes()
    assert_bits(chg, chg0, "first frame")
    t = tr["translation"].reshape(n, 3).copy()
    for f, rows in enumerate([np.array([0], np.uint32), np.array([5, 1101 + 7, n - 1, 600_000], np.uint32), np.zeros(0, np.uint32),
                              np.array([1100, 1101, 1_000_000], np.uint32)]):
        t[rows] += F(0.75)
        if rows.size:
            ctx.upload_transforms_indexed(rows, t[rows].reshape(-1), tr["rotation"].reshape(n, 4)[rows].reshape(-1),
                                          tr["scale"].reshape(n, 3)[rows].reshape(-1))
        ctx.propagate(flags)
        changed = np.zeros(n, np.uint8)
        changed[rows] = 1
        rc, g1, chg1 = O.propagate_transforms(parent, t.reshape(-1), tr["rotation"], tr["scale"], global_in=g0, static_opt=static_opt,
                                              tree_changed=O.mark_dirty_trees(parent, changed), transform_changed=changed)
        g, chg = ctx.download_global_transforms()
        bad = np.nonzero((g.view(np.uint32) != g1.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
        assert bad.size == 0, f"frame {f}: {bad.size} rows differ, first {bad[:5].tolist()}"
        assert_bits(chg, chg1, f"frame {f} change ticks")
        g0 = g1
    if not static_opt:
        rng = np.random.default_rng(9)
        vis = rng.choice(np.array([0, 0, 1, 2, 0x80], np.uint8), size=n).astype(np.uint8)
        ctx.upload_visibility(vis)
        ctx.visibility_propagate()
        inh, _ = ctx.download_inherited_visibility()
        rc, inh_exp, _ = O.visibility_propagate(parent, vis, np.ones(n, np.uint8))
        assert rc == 0 and np.array_equal(inh, inh_exp)


def test_frame_results_in_one_call_equal_the_separate_downloads(ctx_factory):
    """mi_download_frame_results = mi_download_changed_global_transforms + mi_download_visible_entities + mi_cluster_download,
    over several frames with different dirty sets, plus the capacity error and the parts switched off."""
    sc, first_light, pr = W.frame_scene(60_000, 6_000, 600, light_range=2.0)
    n, n_l = sc["n"], len(pr) // 4
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ctx = ctx_factory()
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.cluster_upload_objects(pr)
    ctx.cluster_bind_objects_to_rows(first_light, n_l)
    ctx.upload_changed(np.zeros(n, np.uint8))
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    t3 = sc["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(4)
    n_clusters = 16 * 9 * 24
    bufs = api.FrameResultBuffers(n, n, n_clusters, 4 * n_l * 8)
    for f, k in enumerate((1, 0, 700, n // 10)):
        cam = W.many_cubes_camera(3 * f)
        fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
        view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
        rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
        if k:
            t3[rows] += F(0.25)
            ctx.upload_transforms_indexed(rows, t3[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1),
                                          sc["scale"].reshape(n, 3)[rows].reshape(-1))
        ctx.propagate(0)
        ctx.cull(frusta_for([cam]), flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
        ctx.cluster_upload_view(view)
        ctx.cluster_assign_resident()
        got = ctx.download_frame_results(bufs)
        ch_rows, ch_g = ctx.download_changed_global_transforms()
        _, vis = ctx.download_visible_entities(0, 0)
        off, idx, counts, far, total = ctx.cluster_download(n_clusters)
        assert np.array_equal(got["changed_rows"], ch_rows) and np.array_equal(ch_rows, rows), f"frame {f}"
        assert got["changed_global"].tobytes() == ch_g.tobytes()
        assert np.array_equal(got["visible_rows"], vis) and vis.size > 0
        assert got["cluster_total"] == total and np.array_equal(got["cluster_offsets"], off) and np.array_equal(got["cluster_indices"], idx)
        assert np.array_equal(got["cluster_counts"], counts) and got["farthest_z"] == far
    # parts switched off, and a list that does not fit
    only_vis = api.FrameResultBuffers(0, n, 0, 0)
    got = ctx.download_frame_results(only_vis)
    assert np.array_equal(got["visible_rows"], vis) and got["changed_rows"].size == 0
    small = api.FrameResultBuffers(n, 3, 0, 0)
    with pytest.raises(api.MiError) as e:
        ctx.download_frame_results(small)
    assert e.value.code == api.MI_ERR_CAPACITY and small.list_count(0) == vis.size
    # rows without GlobalTransforms and the other way round (sections of the packed window come and go)
    rows_only = api.FrameResultBuffers(n, n, 0, 0, want_globals=False)
    got = ctx.download_frame_results(rows_only)
    assert np.array_equal(got["changed_rows"], ch_rows) and np.array_equal(got["visible_rows"], vis) and got["changed_global"].size == 0
    g_only = api.FrameResultBuffers(n, 0, n_clusters, 4 * n_l * 8, want_rows=False)
    got = ctx.download_frame_results(g_only)
    assert got["changed_global"].tobytes() == ch_g.tobytes() and np.array_equal(got["cluster_indices"], idx)
    assert np.array_equal(got["cluster_offsets"], off) and got["farthest_z"] == far
    # a changed list that does not fit its capacity: the error, the count, and the other lists still delivered
    tight = api.FrameResultBuffers(max(ch_rows.size - 1, 1), n, n_clusters, 4 * n_l * 8)
    with pytest.raises(api.MiError) as e:
        ctx.download_frame_results(tight)
    assert e.value.code == api.MI_ERR_CAPACITY and tight.raw.changed_count == ch_rows.size
    assert np.array_equal(tight.visible_rows[:tight.list_count(0)], vis)
    assert np.array_equal(tight.cluster_indices[:tight.raw.cluster_total], idx)
    # the same frame delivered IN PLACE: pointers into the library's pinned window, nothing copied out
    inplace = api.FrameResultBuffers(n, n, n_clusters, 4 * n_l * 8, in_place=True)
    got = ctx.download_frame_results(inplace)
    assert np.array_equal(got["changed_rows"], ch_rows) and got["changed_global"].tobytes() == ch_g.tobytes()
    assert np.array_equal(got["visible_rows"], vis) and np.array_equal(got["cluster_offsets"], off) and np.array_equal(got["cluster_indices"], idx)
    assert np.array_equal(got["cluster_counts"], counts) and got["farthest_z"] == far


@pytest.mark.gpu
def test_frame_results_beyond_the_packed_window_take_the_copy_path(ctx_factory):
    """More than 8 MB of results (kernels.h PACK_WINDOW_BYTES): the packed launch reports 'does not fit' and the call falls back
    to the counts-then-lists copies; same results as the separate downloads either way, on the frames before and after."""
    n = 200_000
    sc = W.many_cubes(n)
    ctx = ctx_factory()
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.upload_changed(np.zeros(n, np.uint8))
    bufs = api.FrameResultBuffers(n, n, 0, 0)
    cam = W.many_cubes_camera(0)
    t3 = sc["translation"].reshape(n, 3).copy()
    for f, k in enumerate((n, 500, n, 0)):   # 10.4 MB, 26 KB, 10.4 MB, nothing
        rows = np.arange(n, dtype=np.uint32) if k == n else np.arange(0, 4 * k, 4, dtype=np.uint32)
        if k:
            t3[rows] += F(0.125)
            ctx.upload_transforms_indexed(rows, t3[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1),
                                          sc["scale"].reshape(n, 3)[rows].reshape(-1))
        ctx.propagate(0)
        ctx.cull(frusta_for([cam]), flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
        got = ctx.download_frame_results(bufs)
        ch_rows, ch_g = ctx.download_changed_global_transforms()
        _, vis = ctx.download_visible_entities(0, 0)
        assert np.array_equal(got["changed_rows"], rows) and np.array_equal(ch_rows, rows), f"frame {f}"
        assert got["changed_global"].tobytes() == ch_g.tobytes(), f"frame {f}"
        assert np.array_equal(got["visible_rows"], vis) and vis.size > 0, f"frame {f}"
        # in place the window is as big as the results: one launch, one wait at any size
        got = ctx.download_frame_results(api.FrameResultBuffers(n, n, 0, 0, in_place=True))
        assert np.array_equal(got["changed_rows"], rows) and got["changed_global"].tobytes() == ch_g.tobytes(), f"frame {f} in place"
        assert np.array_equal(got["visible_rows"], vis), f"frame {f} in place"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 10_007, 300_000])
def test_fused_frame_over_the_changed_rows_equals_propagate_then_cull(ctx_factory, n):
    """MI_CULL_CHANGED_ROWS: mi_propagate_and_cull propagates only the rows marked changed (sync_simple_transforms' filter,
    systems.rs:45-50).  Frame by frame against mi_propagate(0) + mi_cull on a twin context and against the oracle: GlobalTransforms,
    their change ticks, ViewVisibility with its ticks, the masks and the VisibleEntities list."""
    sc = W.many_cubes(n, ragged_flags=True)
    a, b = ctx_factory(), ctx_factory()
    for c in (a, b):
        upload_scene(c, sc)
    t3 = sc["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(n)
    g_prev = None
    # frame 0: no change column yet -> every row is Added<GlobalTransform>; then sparse, empty, dense and whole-wave dirty sets
    picks = [None, max(1, n // 100), 0, max(1, n // 3), "waves", n]
    for f, k in enumerate(picks):
        cam = frusta_for([W.many_cubes_camera(2 * f)])
        if k is None:
            rows = np.arange(n, dtype=np.uint32)
        else:
            if k == "waves":   # whole waves dirty next to clean ones
                rows = np.nonzero((np.arange(n) // 64) % 3 == 0)[0].astype(np.uint32)
            else:
                rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
            if rows.size:
                t3[rows] += F(0.5)
                for c in (a, b):
                    c.upload_transforms_indexed(rows, t3[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1),
                                                sc["scale"].reshape(n, 3)[rows].reshape(-1))
            elif f == 2:
                for c in (a, b):
                    c.upload_changed(np.zeros(n, np.uint8))
        a.propagate_and_cull(cam, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
        b.propagate(0)
        b.cull(cam, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
        ga, ca = a.download_global_transforms()
        gb, cb = b.download_global_transforms()
        assert ga.tobytes() == gb.tobytes(), f"frame {f}"
        assert_bits(ca, cb, f"frame {f} GlobalTransform ticks")
        changed = np.zeros(n, np.uint8)
        changed[rows] = 1
        exp, exp_chg = O.sync_simple_transforms(t3.reshape(-1), sc["rotation"], sc["scale"], changed,
                                                g_prev if g_prev is not None else np.zeros(12 * n, F))
        assert ga.tobytes() == exp.tobytes(), f"frame {f} vs oracle"
        assert_bits(ca, exp_chg, f"frame {f} ticks vs oracle")
        g_prev = exp
        va, vca = a.download_view_visibility()
        vb, vcb = b.download_view_visibility()
        assert np.array_equal(va, vb) and np.array_equal(vca, vcb), f"frame {f} ViewVisibility"
        assert np.array_equal(a.download_visibility(0), b.download_visibility(0)), f"frame {f} mask"
        assert np.array_equal(a.download_visible_entities(0, 0)[1], b.download_visible_entities(0, 0)[1]), f"frame {f} list"
    with pytest.raises(api.MiError) as e:
        b.cull(cam, flags=B.CULL_BEGIN_FRAME | B.CULL_CHANGED_ROWS)
    assert e.value.code == api.MI_ERR_INVALID_ARG


def test_riding_cluster_walk_takes_the_global_transform_the_frame_leaves(ctx_factory):
    """The light-cluster walk that rides in the frame kernel re-derives each light's ViewVisibility and position itself; it must use
    the GlobalTransform this frame's propagate LEAVES: From(Transform) for the rows it writes, the resident one for the others.
    Transforms changed behind change detection's back (Mut::bypass_change_detection: new values, no mark) make the two differ:
    the reference keeps the stale GlobalTransform, so the changed-rows frame and the cull-only frame must cluster the lights
    where they WERE.  Twin context: mi_propagate(0) + mi_cull + mi_cluster_assign_resident (reads the columns)."""
    sc, first_light, pr = W.frame_scene(30_000, 4_000, 400, light_range=3.0)
    n, n_l = sc["n"], len(pr) // 4
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(0)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    n_clusters = 16 * 9 * 24
    a, b = ctx_factory(), ctx_factory()
    for c in (a, b):
        c.resize(n)
        c.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        c.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        c.cluster_upload_objects(pr)
        c.cluster_bind_objects_to_rows(first_light, n_l)
        c.upload_changed(np.zeros(n, np.uint8))
        c.propagate(B.PROPAGATE_ALL_DIRTY)
        c.cluster_upload_view(view)
    t3 = sc["translation"].reshape(n, 3).copy()
    lights = np.arange(first_light, first_light + n_l)
    t3[lights[::2]] *= F(0.5)                      # every other light moves towards the camera ...
    marked = np.zeros(n, np.uint8)
    marked[lights[::4]] = 1                        # ... but only every fourth is marked changed
    marked[:100] = 1
    for mode in ("changed_rows", "cull_only"):
        for c in (a, b):
            c.upload_transforms(t3.reshape(-1), sc["rotation"], sc["scale"])
            c.upload_changed(marked if mode == "changed_rows" else np.zeros(n, np.uint8))
        if mode == "changed_rows":
            a.propagate_and_cull(frusta_for([cam]), flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS | B.CULL_CHANGED_ROWS)
            b.propagate(0)
        else:
            a.cull(frusta_for([cam]), flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS)
        b.cull(frusta_for([cam]), flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
        b.cluster_assign_resident()
        ra, rb = a.cluster_download(n_clusters), b.cluster_download(n_clusters)
        assert rb[4] > 0 and ra[4] == rb[4], mode
        for x, y in zip(ra[:3], rb[:3]):
            assert np.array_equal(x, y), mode
        assert ra[3] == rb[3], mode
        assert a.download_global_transforms(want_changed=False).tobytes() == b.download_global_transforms(want_changed=False).tobytes()
        t3[lights[1::2]] *= F(0.75)                # second round: other lights move, nothing is marked
