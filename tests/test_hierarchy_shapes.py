"""The reference's hierarchy stress shapes (examples/stress_tests/transform_hierarchy.rs:29-160) as bevy_amd.workloads generates them:
structure against the reference's own definitions, level order against mi_hierarchy_sort, and the oracle on every shape (CPU only)."""
import numpy as np
import pytest

from bevy_amd import api, workloads as W
import oracle_lib as O

NO_PARENT = W.NO_PARENT


def test_humanoid_rig_structure():
    """HUMANOID_RIG (transform_hierarchy.rs:493-561): 67 entries + root; hips under the root, four spine joints, head with three
    children, two arms hanging from `spine 2` with a hand of five four-joint fingers each, two five-joint legs from the hips."""
    rig = W.HUMANOID_RIG
    assert len(rig) == 67 and all(p <= i for i, p in enumerate(rig))        # spawn order: a parent precedes its children (:351-354)
    kids = {}
    for i, p in enumerate(rig):
        kids.setdefault(p, []).append(i + 1)
    assert kids[0] == [1] and kids[1] == [2, 58, 63] and kids[4] == [5, 10, 34] and kids[6] == [7, 8, 9]
    assert kids[13] == [14, 18, 22, 26, 30] and kids[37] == [38, 42, 46, 50, 54]
    depth = [0] * 68
    for i, p in enumerate(rig):
        depth[i + 1] = depth[p] + 1
    assert max(depth) == 12 and depth[17] == 12 and depth[62] == 6 and depth[67] == 6


def reference_non_uniform(max_depth, max_branch):
    """add_children_non_uniform (transform_hierarchy.rs:455-490), recursively as the reference writes it (small inputs only)."""
    tree = []

    def add(parent, depth):
        for _ in range(max_branch):
            tree.append(parent)
            depth -= 1
            if depth == 0:
                return
            add(len(tree), depth)
    add(0, max_depth)
    return tree


@pytest.mark.parametrize("depth,branch", [(1, 3), (2, 2), (5, 2), (6, 3), (9, 8), (12, 2)])
def test_non_uniform_tree_is_the_references_recursion(depth, branch):
    assert W._parent_map_non_uniform(depth, branch).tolist() == reference_non_uniform(depth, branch)


def test_shape_sizes():
    sizes = {name: W.hierarchy_shape(name) for name in W.HIERARCHY_SHAPES if name != "tree_4ary_depth12"}
    assert sizes["wide_tree"]["n"] == 1 + 500 + 250_000 and sizes["wide_tree"]["n_levels"] == 3
    assert sizes["chain"]["n"] == 2500 and sizes["chain"]["n_levels"] == 2500
    assert sizes["update_leaves"]["n"] == 2 ** 18 - 1 and sizes["update_shallow"]["n"] == 2 ** 18 - 1
    assert sizes["tree_4ary_depth11"]["n"] == 1_398_101                   # SURVEY 8(d) config 5's second data point
    assert sizes["deep_tree"]["n_levels"] == 26 and sizes["large_tree"]["n_levels"] == 19
    for name in ("humanoids_active", "humanoids_inactive", "humanoids_mixed"):
        assert sizes[name]["n"] == 4000 * 68 and sizes[name]["n_levels"] == 13
    # the update filters (probability per node, depth window, inactive rigs never): transform_hierarchy.rs:398-405, 307-321
    lv = sizes["update_leaves"]
    depth = np.repeat(np.arange(lv["n_levels"]), np.diff(lv["level_offsets"].astype(np.int64)))
    assert depth[lv["movers"]].min() == 17 and 0.45 < len(lv["movers"]) / 2 ** 17 < 0.55
    sh = sizes["update_shallow"]
    assert depth[sh["movers"]].max() <= 8 and depth[sh["movers"]].min() >= 1
    assert len(sizes["humanoids_active"]["movers"]) == 4000 * 67            # probability 1.0: every node but the rigs' roots
    assert len(sizes["humanoids_inactive"]["movers"]) == 10 * 67 and len(sizes["humanoids_mixed"]["movers"]) == 2000 * 67


@pytest.mark.parametrize("name", ["large_tree", "deep_tree", "chain", "humanoids_mixed", "wide_tree"])
def test_level_order_is_what_the_library_computes(name):
    (kind, a, b), _ = W.HIERARCHY_SHAPES[name]
    if kind == "humanoids":
        rig = np.array(W.HUMANOID_RIG, np.int64)
        parent = np.full((50, 68), NO_PARENT, np.int64)
        parent[:, 1:] = np.arange(50)[:, None] * 68 + rig[None, :]
        parent = parent.reshape(-1)
    else:
        pm = W._parent_map_tree(a, b) if kind == "tree" else W._parent_map_non_uniform(a, b)
        parent = np.concatenate([[NO_PARENT], pm])
    n2o, p_new, offs = W.level_order(parent)
    n2o_lib, p_lib, offs_lib = api.hierarchy_sort(parent.astype(np.uint32))
    assert np.array_equal(n2o, n2o_lib) and np.array_equal(p_new, p_lib) and np.array_equal(offs, offs_lib)


@pytest.mark.parametrize("name", ["chain", "deep_tree", "humanoids_inactive"])
def test_oracle_propagates_the_shapes(name):
    """A node's GlobalTransform from the full propagate agrees with the chain product TransformHelper::compute_global_transform forms
    (helper.rs:38-72: parent * (.. * child), the other association -- approximately, as the reference's own test compares them,
    helper.rs:97-146) at the deepest rows of each shape; a movers frame under the static-scene rule changes exactly the movers' subtrees."""
    sh = W.hierarchy_shape(name)
    n = sh["n"]
    rc, g, chg = O.propagate_transforms(sh["parent"], sh["translation"], sh["rotation"], sh["scale"])
    assert rc == 0 and chg.all()
    for row in (n - 1, n // 2, int(sh["level_offsets"][-2])):
        one = O.compute_global_transform(sh["parent"], sh["translation"], sh["rotation"], sh["scale"], row)
        ref = g[12 * row:12 * row + 12]
        assert np.allclose(one, ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max())), (row, one, ref)
    t2 = sh["translation"].copy().reshape(n, 3)
    t2[sh["movers"]] = sh["mover_translation"](1).reshape(-1, 3)
    changed = np.zeros(n, np.uint8)
    changed[sh["movers"]] = 1
    tree_changed = O.mark_dirty_trees(sh["parent"], changed)
    rc, g2, chg2 = O.propagate_transforms(sh["parent"], t2.reshape(-1), sh["rotation"], sh["scale"], global_in=g, static_opt=True,
                                          tree_changed=tree_changed, transform_changed=changed)
    assert rc == 0
    differs = (g2.view(np.uint32) != g.view(np.uint32)).reshape(n, 12).any(axis=1)
    below = changed.astype(bool)
    for lo, hi in zip(sh["level_offsets"][1:-1], sh["level_offsets"][2:]):   # closure of the movers under "child of"
        rows = np.arange(int(lo), int(hi))
        below[rows] |= below[sh["parent"][rows]]
    assert not (differs & ~below).any() and differs.sum() > 0.5 * below.sum()


@pytest.mark.parametrize("name", list(W.NARROW_SHAPES))
def test_narrow_shapes_are_what_the_one_wave_kernel_takes(name):
    """bevy_amd.workloads.NARROW_SHAPES (not the reference's): every level at most a wave wide, more levels than a tile spans; level
    order as the library computes it; the oracle propagates them like any hierarchy."""
    sh = W.hierarchy_shape(name)
    widths = np.diff(sh["level_offsets"].astype(np.int64))
    assert widths.max() <= 64 and sh["n_levels"] > 16 and widths.min() >= 1
    n2o_lib, p_lib, offs_lib = api.hierarchy_sort(sh["parent"].astype(np.uint32))
    assert np.array_equal(n2o_lib, np.arange(sh["n"])) and np.array_equal(p_lib, sh["parent"]) and np.array_equal(offs_lib, sh["level_offsets"])
    rc, g, chg = O.propagate_transforms(sh["parent"], sh["translation"], sh["rotation"], sh["scale"])
    assert rc == 0 and chg.all()
    if name.startswith("bundle"):  # chains side by side: every parent sits where its child does
        lo = sh["level_offsets"].astype(np.int64)
        for l in range(2, sh["n_levels"]):
            assert np.array_equal(sh["parent"][lo[l]:lo[l + 1]].astype(np.int64) - lo[l - 1], np.arange(lo[l + 1] - lo[l]))
