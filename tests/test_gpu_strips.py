"""GPU parity of the strips (kernels_tree.hip: k_propagate_strips; planner in ctx_hierarchy.cpp): a deep or lopsided tree in ONE launch of
independent waves, every strip re-evaluating the cone of its rows' ancestors.  The reference's lopsided stress shapes
(transform_hierarchy.rs: large_tree, deep_tree, update_leaves, update_shallow) as planned by default, every other shape with strips
forced (tile mode 5), and random forests at strip widths from 3 rows (every node with four children is cut and handed down, cones of
several rows per level, dozens of bands) to 128 (two rounds per level): all dirty, movers under StaticTransformOptimizations and
without, quiet frames, the flags-first early exit forced on and off -- GlobalTransform bits and change ticks against the oracle."""
import os

import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O
from test_gpu_hierarchy_shapes import run_shape, _run_forest, _small_forest

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture
def strip_width():
    old = os.environ.get("MI_STRIP_W")

    def set_w(w):
        if w is None:
            os.environ.pop("MI_STRIP_W", None)
        else:
            os.environ["MI_STRIP_W"] = str(w)
    yield set_w
    if old is None:
        os.environ.pop("MI_STRIP_W", None)
    else:
        os.environ["MI_STRIP_W"] = old


@pytest.mark.parametrize("name", ["large_tree", "deep_tree", "update_leaves", "update_shallow"])
def test_lopsided_reference_shapes_take_one_launch(name):
    sh = W.hierarchy_shape(name)
    plan = run_shape(sh, True, tile_mode=0)  # (the default plan, whatever MI_TEST_TILE_MODE the suite runs under)
    print(name, sh["n"], "nodes", sh["n_levels"], "levels; plan", plan)
    assert plan["launches"] == 1, plan


@pytest.mark.parametrize("pretest", [1, 2])
@pytest.mark.parametrize("static_opt", [True, False])
@pytest.mark.parametrize("name", ["large_tree", "deep_tree", "wide_tree", "humanoids_mixed", "bundle", "ropes", "update_shallow"])
def test_reference_shapes_through_strips(name, static_opt, pretest):
    sh = W.hierarchy_shape(name)
    plan = run_shape(sh, static_opt, tile_mode=5, pretest=pretest)
    print(name, sh["n"], "nodes", sh["n_levels"], "levels; plan", plan)
    if sh["n_levels"] <= 200:  # (a strip's rounds -- a level each at least -- live in LDS, 248 of them: deeper hierarchies keep the tiles / the one-wave walk)
        assert plan["launches"] == 1, plan


def _lopsided_forest(rng, n_trees, depth, max_children, p_leaf, fan_node_every=0, fan=0):
    """Random trees whose nodes are childless with probability p_leaf; now and then a node with `fan` children (wider than a strip)."""
    parent = []
    for _ in range(n_trees):
        base = len(parent)
        parent.append(W.NO_PARENT)
        level = [base]
        for _d in range(depth):
            nxt = []
            for p in level:
                if rng.random() < p_leaf and len(level) > 1:
                    continue
                k = int(rng.integers(1, max_children + 1))
                if fan_node_every and rng.integers(0, fan_node_every) == 0:
                    k = fan
                for _c in range(k):
                    nxt.append(len(parent))
                    parent.append(p)
            if not nxt or len(parent) > 60000:
                break
            level = nxt
    return np.array(parent, np.int64)


@pytest.mark.parametrize("width", [3, 8, 64, 128, None])  # (None: the planner's own choice)
@pytest.mark.parametrize("seed", range(6))
def test_random_forests_through_strips(seed, width, strip_width):
    rng = np.random.default_rng(9100 + seed)
    strip_width(width)
    kind = seed % 3
    if kind == 0:    # deep and thin: long cones
        parent = _lopsided_forest(rng, int(rng.integers(1, 4)), int(rng.integers(12, 45)), 2, 0.3)
    elif kind == 1:  # bushy with nodes whose fan exceeds any strip
        parent = _lopsided_forest(rng, int(rng.integers(1, 30)), int(rng.integers(4, 9)), 4, 0.4, fan_node_every=40, fan=int(rng.integers(100, 400)))
    else:            # a forest of small trees next to a few big ones
        big = W._parent_map_tree(6, 3)
        small = _small_forest(rng, 300, 6, 3, 30)
        parent = np.concatenate([small, [W.NO_PARENT], big + len(small)])
    state = rng.bit_generator.state
    plan, n_levels = _run_forest(parent, rng, 5, pretest=2 if seed % 2 else None)
    print(f"seed {seed} width {width}: {len(parent)} nodes, {n_levels} levels; plan {plan}")
    assert plan["launches"] == 1, plan
    rng.bit_generator.state = state
    plan4, _ = _run_forest(parent, rng, 4)  # the same frames through the workgroup tiles


def test_strips_after_an_external_global_transform_upload(strip_width):
    """mi_upload_global_transforms between frames: the cones' snapshot is re-taken (the static-scene rule compares against it)."""
    rng = np.random.default_rng(5)
    strip_width(8)
    parent = _lopsided_forest(rng, 2, 14, 3, 0.3)
    new_to_old, p_new, offs = W.level_order(parent)
    n = len(parent)
    t = (rng.random((n, 3)) * 4 - 2).astype(F)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    s3 = np.ones((n, 3), F)
    with api.Context(0) as ctx:
        ctx.debug_set_tile_mode(5)
        ctx.resize(n)
        ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s3.reshape(-1))
        ctx.upload_hierarchy(p_new, offs)
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        g, _ = ctx.download_global_transforms()
        # somebody writes GlobalTransforms behind the system's back, then a few Transforms move
        g2 = g.copy().reshape(n, 12)
        g2[::7, 9:] += F(1.0)
        ctx.upload_global_transforms(g2.reshape(-1))
        moved = np.nonzero(rng.random(n) < 0.05)[0].astype(np.uint32)
        changed = np.zeros(n, np.uint8)
        changed[moved] = 1
        t[moved] += F(0.5)
        ctx.upload_transforms_indexed(moved, np.ascontiguousarray(t[moved]).reshape(-1), np.ascontiguousarray(q[moved]).reshape(-1), np.ascontiguousarray(s3[moved]).reshape(-1))
        ctx.propagate(B.PROPAGATE_STATIC_OPT)
        got, chg = ctx.download_global_transforms()
        rc, g_exp, chg_exp = O.propagate_transforms(p_new, t.reshape(-1), q.reshape(-1), s3.reshape(-1), global_in=g2.reshape(-1), static_opt=True,
                                                    tree_changed=O.mark_dirty_trees(p_new, changed), transform_changed=changed)
        assert rc == 0
        assert (got.view(np.uint32) == g_exp.view(np.uint32)).all()
        assert (np.asarray(chg) == np.asarray(chg_exp)).all()
