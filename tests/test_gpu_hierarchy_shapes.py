"""GPU parity on the reference's own hierarchy stress shapes (examples/stress_tests/transform_hierarchy.rs:29-160: large_tree,
wide_tree, deep_tree, chain, update_leaves, update_shallow, humanoids_active / _inactive / _mixed) and on SURVEY 8(d) config 5's other
data points (the full 11-level 4-ary tree, a true depth-12 one): every frame kind the example produces -- all dirty, the `update`
system's movers under StaticTransformOptimizations and without, a frame in which nothing moved -- against O.propagate_transforms,
GlobalTransform bits and change ticks.  The planner's worst cases live here: 2 500 dependent levels (chain), a 250 000-row level under
500 parents (wide_tree), 4 000 small trees of which half never move (humanoids_mixed)."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32

SHAPES = [n for n in W.HIERARCHY_SHAPES if n != "tree_4ary_depth12"]


def assert_rows(g, g_exp, what):
    bad = np.nonzero((g.view(np.uint32) != g_exp.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"{what}: {bad.size} rows differ, first {bad[:6].tolist()}"


def assert_bits(a, b, what):
    bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first rows {bad[:8].tolist()}"


def run_shape(sh, static_opt, tile_mode=None, pretest=None):
    n = sh["n"]
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    with api.Context(0) as ctx:
        if tile_mode is not None:
            ctx.debug_set_tile_mode(tile_mode)
        if pretest is not None:
            ctx.debug_set_tile_pretest(pretest)
        ctx.resize(n)
        ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
        ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
        # frame 0: every Transform counts as changed
        ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
        g, chg = ctx.download_global_transforms()
        rc, g_exp, chg_exp = O.propagate_transforms(sh["parent"], sh["translation"], sh["rotation"], sh["scale"], static_opt=static_opt)
        assert rc == 0
        assert_rows(g, g_exp, f"{sh['name']} all dirty")
        assert_bits(chg, chg_exp, f"{sh['name']} all dirty: change ticks")
        # frames 1, 2: the `update` system moved its nodes (transform_hierarchy.rs:250-262)
        t = sh["translation"].copy().reshape(n, 3)
        rot = np.ascontiguousarray(sh["rotation"].reshape(n, 4)[sh["movers"]]).reshape(-1)
        scl = np.ascontiguousarray(sh["scale"].reshape(n, 3)[sh["movers"]]).reshape(-1)
        changed = np.zeros(n, np.uint8)
        changed[sh["movers"]] = 1
        tree_changed = O.mark_dirty_trees(sh["parent"], changed)
        for frame in (1, 2):
            mt = sh["mover_translation"](frame)
            t[sh["movers"]] = mt.reshape(-1, 3)
            if len(sh["movers"]):
                ctx.upload_transforms_indexed(sh["movers"], mt, rot, scl)
            else:
                ctx.upload_changed(np.zeros(n, np.uint8))
            ctx.propagate(flags)
            g, chg = ctx.download_global_transforms()
            rc, g_exp, chg_exp = O.propagate_transforms(sh["parent"], t.reshape(-1), sh["rotation"], sh["scale"], global_in=g_exp, static_opt=static_opt,
                                                        tree_changed=tree_changed, transform_changed=changed)
            assert rc == 0
            assert_rows(g, g_exp, f"{sh['name']} movers frame {frame}")
            assert_bits(chg, chg_exp, f"{sh['name']} movers frame {frame}: change ticks")
        # frame 3: nothing moved
        ctx.upload_changed(np.zeros(n, np.uint8))
        ctx.propagate(flags)
        g, chg = ctx.download_global_transforms()
        none = np.zeros(n, np.uint8)
        rc, g_exp, chg_exp = O.propagate_transforms(sh["parent"], t.reshape(-1), sh["rotation"], sh["scale"], global_in=g_exp, static_opt=static_opt,
                                                    tree_changed=none, transform_changed=none)
        assert rc == 0
        assert_rows(g, g_exp, f"{sh['name']} quiet frame")
        assert_bits(chg, chg_exp, f"{sh['name']} quiet frame: change ticks")
        return ctx.debug_tile_plan()


@pytest.mark.parametrize("static_opt", [True, False])
@pytest.mark.parametrize("name", SHAPES)
def test_reference_hierarchy_shape(name, static_opt):
    sh = W.hierarchy_shape(name)
    plan = run_shape(sh, static_opt)
    print(name, sh["n"], "nodes", sh["n_levels"], "levels", len(sh["movers"]), "movers; plan", plan)


@pytest.mark.parametrize("static_opt", [True, False])
@pytest.mark.parametrize("name", list(W.NARROW_SHAPES))
def test_narrow_hierarchy(name, static_opt):
    """Every level at most a wave wide and more levels than a tile spans: the one-wave walk (k_propagate_narrow) instead of tiles --
    several nodes to a level, parents anywhere in the level above, chunks of whole levels."""
    sh = W.hierarchy_shape(name)
    assert int(np.diff(sh["level_offsets"].astype(np.int64)).max()) <= 64 and sh["n_levels"] > 16
    run_shape(sh, static_opt)
    run_shape(W.hierarchy_shape(name, seed=7, plain_transforms=True), static_opt)


@pytest.mark.parametrize("seed", range(12))
def test_random_narrow_forests(seed):
    """Random forests whose every level fits a wave (the one-wave kernel's domain): several roots, roots without children (their rule is
    sync_simple_transforms': written iff their own Transform changed), level widths from 1 to 64 or held under 16 (the quad form),
    17 to ~400 levels (several LDS chunks), parents anywhere in the level above; frames with random movers, with and without
    StaticTransformOptimizations.  Bits and change ticks against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n_levels = int(rng.integers(17, 400 if seed % 3 else 60))
    cap = 16 if seed % 2 else 64
    widths = rng.integers(1, cap + 1, n_levels)
    starts = np.concatenate([[0], np.cumsum(widths)])
    n = int(starts[-1])
    parent = np.full(n, W.NO_PARENT, np.int64)
    for l in range(1, n_levels):  # (level order by construction: parents sorted, so siblings stay together)
        lo_p = starts[l - 1] + (rng.integers(0, widths[l - 1]) if seed % 4 == 0 else 0)  # some roots stay childless
        parent[starts[l]:starts[l + 1]] = np.sort(rng.integers(lo_p, starts[l], widths[l]))
    new_to_old, p_new, offs = W.level_order(parent)
    assert np.array_equal(new_to_old, np.arange(n)) and int(np.diff(offs.astype(np.int64)).max()) <= 64
    t = (rng.random((n, 3)) * 4 - 2).astype(F)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    s3 = (0.9 + 0.2 * rng.random((n, 3))).astype(F)
    for static_opt in (True, False):
        flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
        with api.Context(0) as ctx:
            ctx.resize(n)
            ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s3.reshape(-1))
            ctx.upload_hierarchy(p_new, offs)
            ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
            g, chg = ctx.download_global_transforms()
            rc, g_exp, chg_exp = O.propagate_transforms(p_new, t.reshape(-1), q.reshape(-1), s3.reshape(-1), static_opt=static_opt)
            assert rc == 0
            assert_rows(g, g_exp, f"seed {seed} all dirty")
            assert_bits(chg, chg_exp, f"seed {seed} all dirty: change ticks")
            tt = t.copy()
            for frame, frac in enumerate((0.3, 0.02, 0.0)):
                moved = np.nonzero(rng.random(n) < frac)[0].astype(np.uint32)
                changed = np.zeros(n, np.uint8)
                changed[moved] = 1
                if len(moved):
                    tt[moved] += F(0.25)
                    ctx.upload_transforms_indexed(moved, np.ascontiguousarray(tt[moved]).reshape(-1), np.ascontiguousarray(q[moved]).reshape(-1),
                                                  np.ascontiguousarray(s3[moved]).reshape(-1))
                else:
                    ctx.upload_changed(changed)
                ctx.propagate(flags)
                g, chg = ctx.download_global_transforms()
                rc, g_exp, chg_exp = O.propagate_transforms(p_new, tt.reshape(-1), q.reshape(-1), s3.reshape(-1), global_in=g_exp, static_opt=static_opt,
                                                            tree_changed=O.mark_dirty_trees(p_new, changed), transform_changed=changed)
                assert rc == 0
                assert_rows(g, g_exp, f"seed {seed} static_opt {static_opt} frame {frame}")
                assert_bits(chg, chg_exp, f"seed {seed} static_opt {static_opt} frame {frame}: change ticks")


def test_reference_hierarchy_shape_with_plain_transforms():
    """Identity rotations and unit scales exactly as spawn_tree leaves them (transform_hierarchy.rs:409-418)."""
    for name in ("humanoids_mixed", "chain"):
        run_shape(W.hierarchy_shape(name, plain_transforms=True), True)


def test_true_depth_12_tree_movers_under_the_static_scene_rule():
    """The 5.6 M-node tree is past TREE_NT_MIN_ROWS (kernels_tree.hip): the tile kernel's streamed level goes past the caches.  The
    change-driven instantiation on it: half the nodes move, StaticTransformOptimizations on."""
    sh = W.hierarchy_shape("tree_4ary_depth12")
    n = sh["n"]
    flags = B.PROPAGATE_STATIC_OPT
    with api.Context(0) as ctx:
        ctx.resize(n)
        ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
        ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
        ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
        rc, g_exp, _ = O.propagate_transforms(sh["parent"], sh["translation"], sh["rotation"], sh["scale"], static_opt=True)
        assert rc == 0
        t = sh["translation"].copy().reshape(n, 3)
        mt = sh["mover_translation"](1)
        t[sh["movers"]] = mt.reshape(-1, 3)
        ctx.upload_transforms_indexed(sh["movers"], mt, np.ascontiguousarray(sh["rotation"].reshape(n, 4)[sh["movers"]]).reshape(-1),
                                      np.ascontiguousarray(sh["scale"].reshape(n, 3)[sh["movers"]]).reshape(-1))
        ctx.propagate(flags)
        g, chg = ctx.download_global_transforms()
    changed = np.zeros(n, np.uint8)
    changed[sh["movers"]] = 1
    rc, g_exp, chg_exp = O.propagate_transforms(sh["parent"], t.reshape(-1), sh["rotation"], sh["scale"], global_in=g_exp, static_opt=True,
                                                tree_changed=O.mark_dirty_trees(sh["parent"], changed), transform_changed=changed)
    assert rc == 0
    assert_rows(g, g_exp, "depth 12, movers")
    assert_bits(chg, chg_exp, "depth 12, movers: change ticks")


def test_true_depth_12_tree_all_dirty():
    """gen_tree(12, 4) in full: 5 592 405 nodes, all dirty."""
    sh = W.hierarchy_shape("tree_4ary_depth12")
    assert sh["n"] == 5_592_405 and sh["n_levels"] == 12
    with api.Context(0) as ctx:
        ctx.resize(sh["n"])
        ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
        ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        g, chg = ctx.download_global_transforms()
    rc, g_exp, chg_exp = O.propagate_transforms(sh["parent"], sh["translation"], sh["rotation"], sh["scale"])
    assert rc == 0
    assert_rows(g, g_exp, "depth 12")
    assert_bits(chg, chg_exp, "depth 12: change ticks")


def _small_forest(rng, n_trees, max_depth, max_children, max_level_width):
    """A forest of small random trees in level order: tree sizes and shapes vary, some roots are childless."""
    parent = []
    for _ in range(n_trees):
        base = len(parent)
        parent.append(W.NO_PARENT)
        level = [base]
        depth = int(rng.integers(0, max_depth + 1))
        for _d in range(depth):
            nxt = []
            for p in level:
                for _c in range(int(rng.integers(0, max_children + 1))):
                    if len(nxt) < max_level_width:
                        nxt.append(len(parent))
                        parent.append(p)
            if not nxt:
                break
            level = nxt
    return np.array(parent, np.int64)


def _run_forest(parent, rng, tile_mode, frames=((0.3, True), (0.02, True), (0.0, True), (0.2, False)), pretest=None):
    new_to_old, p_new, offs = W.level_order(parent)
    n = len(parent)
    t = (rng.random((n, 3)) * 4 - 2).astype(F)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    s3 = (0.9 + 0.2 * rng.random((n, 3))).astype(F)
    with api.Context(0) as ctx:
        ctx.debug_set_tile_mode(tile_mode)
        if pretest is not None:
            ctx.debug_set_tile_pretest(pretest)
        ctx.resize(n)
        ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s3.reshape(-1))
        ctx.upload_hierarchy(p_new, offs)
        plan = ctx.debug_tile_plan()
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        g, chg = ctx.download_global_transforms()
        rc, g_exp, chg_exp = O.propagate_transforms(p_new, t.reshape(-1), q.reshape(-1), s3.reshape(-1))
        assert rc == 0
        assert_rows(g, g_exp, "forest all dirty")
        assert_bits(chg, chg_exp, "forest all dirty: change ticks")
        tt = t.copy()
        for frame, (frac, static_opt) in enumerate(frames):
            moved = np.nonzero(rng.random(n) < frac)[0].astype(np.uint32)
            changed = np.zeros(n, np.uint8)
            changed[moved] = 1
            if len(moved):
                tt[moved] += F(0.25)
                ctx.upload_transforms_indexed(moved, np.ascontiguousarray(tt[moved]).reshape(-1), np.ascontiguousarray(q[moved]).reshape(-1),
                                              np.ascontiguousarray(s3[moved]).reshape(-1))
            else:
                ctx.upload_changed(changed)
            ctx.propagate(B.PROPAGATE_STATIC_OPT if static_opt else 0)
            g, chg = ctx.download_global_transforms()
            rc, g_exp, chg_exp = O.propagate_transforms(p_new, tt.reshape(-1), q.reshape(-1), s3.reshape(-1), global_in=g_exp, static_opt=static_opt,
                                                        tree_changed=O.mark_dirty_trees(p_new, changed), transform_changed=changed)
            assert rc == 0
            assert_rows(g, g_exp, f"forest frame {frame} static_opt {static_opt}")
            assert_bits(chg, chg_exp, f"forest frame {frame}: change ticks")
    return plan, len(offs) - 1


@pytest.mark.parametrize("seed", range(10))
def test_forest_of_small_trees_takes_a_wave_per_tile(seed):
    """Forests every tree of which fits a wave tile (k_propagate_wave_tiles, round 6): rigs of random shape -- narrow enough for the
    quad form (<= 16 rows to a level of a tile) or not, several small trees packed into one tile, childless roots -- with movers under
    StaticTransformOptimizations and without, a quiet frame; and the same forests through the workgroup tiles (tile mode 4)."""
    rng = np.random.default_rng(7000 + seed)
    narrow = seed % 2 == 0
    parent = _small_forest(rng, int(rng.integers(3, 900)), int(rng.integers(5, 13)), 2 if narrow else 4, 6 if narrow else 40)
    state = rng.bit_generator.state
    plan, n_levels = _run_forest(parent, rng, 0)
    rng.bit_generator.state = state
    plan4, _ = _run_forest(parent, rng, 4)
    n_roots = int((parent == W.NO_PARENT).sum())
    counts = np.bincount(np.cumsum(parent == W.NO_PARENT) - 1)  # (spawn order: a tree's nodes follow its root)
    fits = n_levels >= 5 and counts.max() <= 80
    print(f"seed {seed}: {len(parent)} nodes, {n_roots} trees (largest {counts.max()}), {n_levels} levels; plan {plan}, without wave tiles {plan4}")
    if fits and not narrow:
        pass  # (a level of one tree may exceed 64 rows only with > 80 rows in all: every such forest takes the wave tiles)
    if fits:
        assert plan["launches"] == 1 and plan["tiles"] <= n_roots and plan4["tiles"] != plan["tiles"] or plan4["launches"] >= 1, plan  # ONE launch, at most a tile per tree (small trees share tiles)


def test_a_forest_with_one_big_tree_keeps_the_workgroup_tiles():
    """One tree too big for a wave tile: the whole hierarchy takes the workgroup tiles (the planner's rule is all-or-nothing)."""
    rng = np.random.default_rng(99)
    small = _small_forest(rng, 200, 8, 2, 6)
    big = W._parent_map_tree(7, 3)  # 1 093 nodes
    parent = np.concatenate([small, [W.NO_PARENT], big + len(small)])
    plan, _ = _run_forest(parent, rng, 0, frames=((0.3, True), (0.0, True)))
    print("plan", plan)
