"""The strips planner on the CPU (bevy_amd/csrc/strip_plan.h through mi_debug_plan_strips: host code, no device).  A plan is walked the
way k_propagate_strips walks it -- a strip's table entry by entry, a level's results kept for the level below, a row's parent found at
slot parent - pstart of the level above -- with the oracle's own GlobalTransforms standing in for the arithmetic: what is checked is that
every row is owned exactly once, that every parent a strip needs is in the range it evaluated one level up (cone included), and the
table's flags, batches and limits.  Reference shapes (transform_hierarchy.rs) and random forests, at widths 3 .. 128."""
import numpy as np
import pytest

from bevy_amd import api, workloads as W

PARITY, OWNED, ROOT, ABOVE_TOP, BATCH_SHIFT = 1 << 16, 1 << 17, 1 << 18, 1 << 20, 22
TAB_CAP = 256


def walk_plan(parent, level_offsets, plan, width):
    n = len(parent)
    offs = np.asarray(level_offsets, np.int64)
    level_of = np.searchsorted(offs, np.arange(n), side="right") - 1
    owned = np.zeros(n, np.int32)
    in_cone = np.zeros(n, bool)
    rounds = plan["rounds"]
    for first, word, top in plan["strips"].tolist():
        n_entries, n_batches, snap_owner = word & 0xFFFF, (word >> 16) & 0xFF, word >> 31
        assert n_entries + 8 <= TAB_CAP and n_batches % 2 == 0 and n_batches >= 2
        entries = rounds[first:first + n_entries]
        prev = None          # (level, first row, rows) of the level above, as the strip holds it in LDS
        cur = None
        batches, j = 0, 0
        while j < n_entries:
            cnt = max(1, (int(entries[j][2]) >> BATCH_SHIFT) & 7)
            assert j + cnt <= n_entries
            kinds = set()
            for k in range(cnt):
                row0, pstart, info, level = (int(x) for x in entries[j + k])
                rows, slot0 = info & 0x7F, (info >> 8) & 0xFF
                if rows == 0:  # the padding batch
                    assert cnt == 1 and j + 1 == n_entries
                    continue
                if k:
                    assert (int(entries[j + k][2]) >> BATCH_SHIFT) & 7 == 0  # only a batch's first entry carries the count
                if cnt > 1:
                    assert rows <= 16 and (slot0 == 0 or k == 0)  # narrow levels share a batch; only its first may be the tail of a wide level
                kinds.add(bool(info & OWNED))
                assert level == level_of[row0] == level_of[row0 + rows - 1] and rows <= 64 and slot0 % 64 == 0 and slot0 + rows <= width + 63
                assert bool(info & ROOT) == (level == 0) and bool(info & PARITY) == bool(level & 1)
                assert bool(info & OWNED) == (level >= top)
                assert bool(info & ABOVE_TOP) == (level == top - 1)
                if cur is None or cur[0] != level:   # a new level of the strip
                    assert slot0 == 0 and (cur is None or level == cur[0] + 1)
                    prev, cur = cur, [level, row0, 0]
                assert row0 == cur[1] + cur[2] and slot0 == cur[2]  # a level's rounds are consecutive: row i of the level sits in slot i
                cur[2] += rows
                assert cur[2] <= width
                r = np.arange(row0, row0 + rows)
                if level:
                    assert prev is not None and prev[0] == level - 1 and pstart == prev[1]
                    p = parent[r].astype(np.int64)
                    assert (p >= prev[1]).all() and (p < prev[1] + prev[2]).all(), "a parent outside the range the strip evaluated one level up"
                if info & OWNED:
                    owned[r] += 1
                else:
                    in_cone[r] = True
            assert len(kinds) <= 1  # a batch is cone or own, never both
            batches += 1
            j += cnt
        assert batches == n_batches
    assert (owned == 1).all(), f"{(owned != 1).sum()} rows not owned exactly once"
    assert not in_cone[plan["snap_rows"]:].any()  # the snapshot prefix holds every cone row
    # which strips mirror rows into the snapshot: the owners of the cones' rows
    for first, word, top in plan["strips"].tolist():
        entries = rounds[first:first + (word & 0xFFFF)]
        owns = any(in_cone[int(e[0]):int(e[0]) + (int(e[2]) & 0x7F)].any() for e in entries if int(e[2]) & OWNED)
        assert owns == bool(word >> 31)
    return owned, in_cone


@pytest.mark.parametrize("width", [64, 128])
@pytest.mark.parametrize("name", ["large_tree", "deep_tree", "update_leaves", "wide_tree", "humanoids_mixed", "tree_4ary_depth11"])
def test_reference_shapes_plan(name, width):
    sh = W.hierarchy_shape(name)
    plan = api.debug_plan_strips(sh["parent"], sh["level_offsets"], width)
    assert plan is not None
    owned, in_cone = walk_plan(sh["parent"], sh["level_offsets"], plan, width)
    print(name, width, len(plan["strips"]), "strips,", len(plan["rounds"]), "table entries,", int(in_cone.sum()), "rows in some cone")
    if name in ("large_tree", "deep_tree"):
        assert in_cone.sum() < sh["n"] // 4 and plan["snap_rows"] > sh["n"] // 2  # (the snapshot mirrors the cones' rows, not the prefix above the deepest strip)


def test_hierarchies_deeper_than_a_strips_table_are_not_planned():
    for name in ("chain", "ropes"):  # 2 500 and 300 levels; the 70 levels of `bundle` fit
        sh = W.hierarchy_shape(name)
        assert api.debug_plan_strips(sh["parent"], sh["level_offsets"], 64) is None
    sh = W.hierarchy_shape("bundle")
    walk_plan(sh["parent"], sh["level_offsets"], api.debug_plan_strips(sh["parent"], sh["level_offsets"], 64), 64)


def _random_forest(rng, n_trees, depth, max_children, p_leaf, fan_every=0, fan=0):
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
                if fan_every and rng.integers(0, fan_every) == 0:
                    k = fan
                for _c in range(k):
                    nxt.append(len(parent))
                    parent.append(p)
            if not nxt or len(parent) > 40000:
                break
            level = nxt
    return np.array(parent, np.int64)


@pytest.mark.parametrize("width", [3, 8, 17, 64, 100, 128])
@pytest.mark.parametrize("seed", range(8))
def test_random_forests_plan(seed, width):
    rng = np.random.default_rng(4200 + seed)
    kind = seed % 4
    if kind == 0:
        parent = _random_forest(rng, int(rng.integers(1, 4)), int(rng.integers(8, 22)), 2, 0.3)
    elif kind == 1:
        parent = _random_forest(rng, int(rng.integers(1, 40)), int(rng.integers(3, 9)), 4, 0.4, fan_every=40, fan=int(rng.integers(100, 400)))
    elif kind == 2:
        parent = _random_forest(rng, int(rng.integers(200, 900)), int(rng.integers(2, 7)), 3, 0.5)
    else:
        parent = _random_forest(rng, 1, 12, 3, 0.2, fan_every=15, fan=70)
    _, p_new, offs = W.level_order(parent)
    plan = api.debug_plan_strips(p_new, offs, width)
    if plan is None:  # (only a hierarchy too deep for a strip's table may be refused)
        assert len(offs) - 1 + 10 > TAB_CAP - 8 or width < 8
        return
    walk_plan(np.asarray(p_new), offs, plan, width)


def test_malformed_input_is_refused():
    sh = W.hierarchy_shape("wide_tree")
    bad = np.array(sh["parent"], np.uint32).copy()
    bad[-1] = 0  # the last leaf's parent is not in the level above
    with pytest.raises(api.MiError):
        api.debug_plan_strips(bad, sh["level_offsets"], 64)
    with pytest.raises(api.MiError):
        api.debug_plan_strips(sh["parent"], np.array(sh["level_offsets"])[::-1].copy(), 64)  # level ends must not decrease
    assert api.debug_plan_strips(sh["parent"], sh["level_offsets"], 0) is None and api.debug_plan_strips(sh["parent"], sh["level_offsets"], 129) is None
