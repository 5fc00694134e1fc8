"""GPU parity of the two hierarchy paths (subtree tiles, oversized subtrees cut over several launches / the level-by-level sweep) on
shapes that exercise each planner and kernel branch: the path is forced with the mi_debug_set_tile_mode hook (1 = by levels, 2 =
tiles where they fit), results are compared with the oracle
(propagate_parent_transforms + mark_dirty_trees, crates/bevy_transform/src/systems.rs:111-306,506-748) bit for bit, change
ticks included, over an all-dirty frame and several partially dirty ones, with and without the static-scene rule."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O
from test_gpu_parity import ctx_factory, assert_bits  # noqa: F401

pytestmark = pytest.mark.gpu
F = np.float32


def _levels(spec, seed):
    """spec = children per node for each level below the roots: int (every node) or callable(node_index_in_level) -> int."""
    n_roots, fans = spec
    parent = [B.NO_PARENT] * n_roots
    offs = [0, n_roots]
    prev = list(range(n_roots))
    for fan in fans:
        nxt = []
        for i, p in enumerate(prev):
            for _ in range(fan(i) if callable(fan) else fan):
                nxt.append(len(parent))
                parent.append(p)
        if not nxt:
            break
        prev = nxt
        offs.append(len(parent))
    n = len(parent)
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    return dict(n=n, parent=np.array(parent, np.uint32), level_offsets=np.array(offs, np.uint32),
                translation=rng.uniform(-2, 2, size=3 * n).astype(F), rotation=q.reshape(-1),
                scale=rng.uniform(0.9, 1.1, size=3 * n).astype(F))


SHAPES = {
    "fan_4ary_7_levels": (1, [4] * 6),                                  # the bench's shape, small: chain tiles below a top tile
    "one_node_700_leaves": (1, [700]),                                   # a last level of several batches in one tile
    "wide_second_level": (1, [300, 4]),
    "skewed": (1, [200, lambda i: 300 if i == 0 else 0, 2]),             # one subtree overflows a tile: cut, the rest handed to a second launch
    "skewed_twice": (3, [4, lambda i: 150 if i == 5 else 2, lambda i: 130 if i % 7 == 0 else 1, lambda i: 3 if i % 2 else 0, 2]),  # ... cut again below
    "one_node_9000_children_with_children": (1, [40, lambda i: 9000 if i == 7 else 1, lambda i: 2 if i % 3 == 0 else 0, 1]),  # a wide level handed down whole
    "wide_last_level_under_a_small_top": (1, [400, lambda i: 12 if i == 0 else 1, lambda i: 700 if i < 12 else 1]),  # upper rows fit, 8 400 leaves do not
    "heavy_nodes_deep": (1, [2] * 8 + [lambda i: 200 if i % 97 == 0 else 2, 3, lambda i: 140 if i % 1001 == 0 else 1, 2, 2]),  # cuts inside lower bands
    "forest_3000_roots": (3000, [3, 2]),                                 # many roots per tile, level 0 holds most rows
    "flat_rows_and_trees": (5000, [lambda i: 5 if i % 50 == 0 else 0, 6, 3]),  # flat rows share level 0 with the roots
    "ragged": (7, [lambda i: i % 5, lambda i: (i * 7) % 4, lambda i: 300 if i == 3 else i % 3, 2]),
    "chain_of_30_then_fan": (1, [1] * 30 + [4, 4, 4, 4, 4]),            # deeper than TILE_MAX_CHAIN: dependent launches
    "binary_13_levels": (1, [2] * 12),
}


@pytest.mark.parametrize("static_opt", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_tile_kernels_match_oracle(ctx_factory, shape, mode, static_opt):
    tr = _levels(SHAPES[shape], seed=len(shape))
    n, parent = tr["n"], tr["parent"]
    flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
    ctx = ctx_factory()
    ctx.debug_set_tile_mode(mode)
    ctx.resize(n)
    ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
    ctx.upload_hierarchy(parent, tr["level_offsets"])
    ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
    rc, g0, chg0 = O.propagate_transforms(parent, tr["translation"], tr["rotation"], tr["scale"], static_opt=static_opt)
    assert rc == 0
    g, chg = ctx.download_global_transforms()
    bad = np.nonzero((g.view(np.uint32) != g0.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
    assert bad.size == 0, f"all-dirty frame: {bad.size} rows differ, first {bad[:5].tolist()}"
    assert_bits(chg, chg0, "change ticks of the all-dirty frame")
    t = tr["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(17)
    lo = tr["level_offsets"]
    picks = [np.array([0]),                                              # a root
             np.array([lo[1] % n, lo[len(lo) // 2] % n, n - 1]),         # first row of level 1, a middle level, the last leaf
             np.zeros(0, np.int64),                                      # nothing
             rng.integers(0, n, max(1, n // 100)),                       # 1 % of the rows
             np.array([0, n - 1])]
    for frame, dirty in enumerate(picks):
        dirty = np.unique(dirty).astype(np.uint32)
        t[dirty] += F(0.375)
        if dirty.size:
            ctx.upload_transforms_indexed(dirty, t[dirty].reshape(-1), tr["rotation"].reshape(n, 4)[dirty].reshape(-1),
                                          tr["scale"].reshape(n, 3)[dirty].reshape(-1))
        ctx.propagate(flags)
        changed = np.zeros(n, np.uint8)
        changed[dirty] = 1
        rc, g1, chg1 = O.propagate_transforms(parent, t.reshape(-1), tr["rotation"], tr["scale"], global_in=g0, static_opt=static_opt,
                                              tree_changed=O.mark_dirty_trees(parent, changed), transform_changed=changed)
        assert rc == 0
        g, chg = ctx.download_global_transforms()
        bad = np.nonzero((g.view(np.uint32) != g1.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
        assert bad.size == 0, f"frame {frame}: {bad.size} rows differ, first {bad[:5].tolist()}"
        assert_bits(chg, chg1, f"frame {frame} change ticks")
        g0 = g1


def test_tile_plans_cut_oversized_subtrees_and_the_level_sweep_can_be_forced(ctx_factory):
    def plan(shape, mode):
        tr = _levels(SHAPES[shape], seed=1)
        ctx = ctx_factory()
        ctx.debug_set_tile_mode(mode)
        ctx.resize(tr["n"])
        ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
        ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
        return ctx.debug_tile_plan(), len(tr["level_offsets"]) - 1
    p, levels = plan("fan_4ary_7_levels", 0)
    assert p["launches"] == 1 and levels == 7                       # roots and chain tiles share one launch
    assert plan("fan_4ary_7_levels", 1)[0]["launches"] == 7         # forced: one launch per level
    p, levels = plan("skewed", 4)
    assert p["launches"] == 2 and levels == 4                       # a node with 300 children that have children: its tile is cut, the rest follows
    assert plan("skewed", 1)[0]["launches"] == 4
    assert plan("skewed_twice", 4)[0]["launches"] >= 2
    assert plan("skewed", 0)[0]["launches"] == 1 and plan("skewed_twice", 0)[0]["launches"] == 1  # by default such a hierarchy takes strips: one launch (round 6)
    # by default (mode 0): tiles wherever they fit
    big = W.gen_tree(12, 4, 1_000_000)
    ctx = ctx_factory()
    ctx.debug_set_tile_mode(0)  # (the suite also runs under MI_TEST_TILE_MODE=1)
    ctx.resize(big["n"])
    ctx.upload_transforms(big["translation"], big["rotation"], big["scale"])
    ctx.upload_hierarchy(big["parent"], big["level_offsets"])
    auto = ctx.debug_tile_plan()
    ctx.debug_set_tile_mode(1)
    ctx.upload_hierarchy(big["parent"], big["level_offsets"])
    assert auto["launches"] == 1 and ctx.debug_tile_plan()["launches"] == len(big["level_offsets"]) - 1


def _random_forest(rng, n):
    """Random hierarchy of n nodes: a random share of parentless rows (roots and flat rows), every other node hangs below a
    node drawn from a sliding window before it -- narrow windows give deep chains, wide ones bushy trees."""
    parent_old = np.full(n, B.NO_PARENT, np.uint32)
    root_p = rng.choice([0.001, 0.02, 0.3])
    window = int(rng.choice([1, 3, 40, 2000]))
    for k in range(1, n):
        if rng.random() < root_p:
            continue
        parent_old[k] = rng.integers(max(0, k - window), k)
    return parent_old


@pytest.mark.parametrize("seed", range(12))
def test_random_hierarchies_match_oracle_under_both_kernels(ctx_factory, seed):
    """Differential test over random forests (deep chains, bushy trees, many roots; 1 to ~60 000 nodes): both paths, the
    static-scene rule on and off, an all-dirty frame and three random partially dirty ones, bit for bit with change ticks."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 65, 900, 7000, 60_000]))
    parent_old = _random_forest(rng, n)
    new_to_old, parent, offs = api.hierarchy_sort(parent_old)
    t = rng.normal(size=(n, 3)).astype(F)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F).reshape(-1)
    s = rng.uniform(0.8, 1.25, size=(n, 3)).astype(F).reshape(-1)
    for mode in (1, 2):
        for static_opt in (False, True):
            flags = B.PROPAGATE_STATIC_OPT if static_opt else 0
            ctx = ctx_factory()
            ctx.debug_set_tile_mode(mode)
            ctx.resize(n)
            tt = t.copy()
            ctx.upload_transforms(tt.reshape(-1), q, s)
            ctx.upload_hierarchy(parent, offs)
            ctx.propagate(B.PROPAGATE_ALL_DIRTY | flags)
            rc, g0, chg0 = O.propagate_transforms(parent, tt.reshape(-1), q, s, static_opt=static_opt)
            assert rc == 0
            g, chg = ctx.download_global_transforms()
            assert g.tobytes() == g0.tobytes(), f"seed {seed} mode {mode} static {static_opt}: all-dirty frame"
            assert_bits(chg, chg0, "change ticks of the all-dirty frame")
            ctx.upload_changed(np.zeros(n, np.uint8))  # from here on the change column says what changed (until one is uploaded,
            frng = np.random.default_rng(seed)         # every Transform counts as changed)
            for frame in range(3):
                k = int(frng.choice([0, 1, max(1, n // 50)]))
                dirty = np.unique(frng.integers(0, n, k)).astype(np.uint32) if k else np.zeros(0, np.uint32)
                tt[dirty] += F(0.25)
                if dirty.size:
                    ctx.upload_transforms_indexed(dirty, tt[dirty].reshape(-1), q.reshape(n, 4)[dirty].reshape(-1), s.reshape(n, 3)[dirty].reshape(-1))
                ctx.propagate(flags)
                changed = np.zeros(n, np.uint8)
                changed[dirty] = 1
                rc, g1, chg1 = O.propagate_transforms(parent, tt.reshape(-1), q, s, global_in=g0, static_opt=static_opt,
                                                      tree_changed=O.mark_dirty_trees(parent, changed), transform_changed=changed)
                assert rc == 0
                g, chg = ctx.download_global_transforms()
                bad = np.nonzero((g.view(np.uint32) != g1.view(np.uint32)).reshape(-1, 12).any(axis=1))[0]
                assert bad.size == 0, f"seed {seed} mode {mode} static {static_opt} frame {frame}: {bad.size} rows differ, first {bad[:5].tolist()}"
                assert_bits(chg, chg1, f"seed {seed} mode {mode} frame {frame} change ticks")
                g0 = g1
