"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the
propagate -> cull -> cluster path.  CPU only.

Reference tests restated here (paths relative to /root/reference):
  crates/bevy_camera/src/primitives.rs:462-857        frustum / sphere / OBB / half-space vectors
  benches/benches/bevy_camera/primitives.rs:41-52     intersects_obb sanity assert
  crates/bevy_transform/src/systems.rs:827-1221       propagate semantics (exact assert_eq!)
  crates/bevy_transform/src/helper.rs:97-146          TRS chain, helper == systems
  crates/bevy_camera/src/visibility/mod.rs:1313-1448  ViewVisibility 2-bit lifecycle
  crates/bevy_light/src/cluster/test.rs               cluster tiling
"""
import math

import numpy as np
import pytest

import oracle_lib as O

F = np.float32
IDENT = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], F)
PI = math.pi


def quat_axis(axis, angle):
    """Quat::from_rotation_{x,y,z}: (sin(a/2) on the axis, cos(a/2)), computed in f32."""
    s, c = F(math.sin(F(angle) * F(0.5))), F(math.cos(F(angle) * F(0.5)))
    q = np.zeros(4, F)
    q["xyz".index(axis)] = s
    q[3] = c
    return q


def affine_rt(q, t):
    return O.transform_to_affine(np.array(t, F), q, np.ones(3, F))


def affine_t(t):
    return affine_rt(np.array([0, 0, 0, 1], F), t)


# ---------------------------------------------------------------- primitives.rs:462-611

import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "primitives_known_answers.json")) as _f:
    GOLDEN = json.load(_f)  # literals of the reference's tests, see the file's _provenance


def _golden_frustum(name):
    return O.frustum_from_planes([tuple(p) for p in GOLDEN["frusta"][name]["planes"]])


def big_frustum():
    return _golden_frustum("big_frustum")


def frustum():
    return _golden_frustum("frustum")


def long_frustum():
    return _golden_frustum("long_frustum")


SPHERE_CASES = [(n, globals()[f], tuple(c), r, e) for n, f, c, r, e in GOLDEN["intersects_sphere"]["cases"]]


@pytest.mark.parametrize("name,fr,center,radius,expect", SPHERE_CASES, ids=[c[0] for c in SPHERE_CASES])
def test_intersects_sphere_known_answers(name, fr, center, radius, expect):
    assert O.intersects_sphere(fr(), center, radius, True) is expect


# ---------------------------------------------------------------- primitives.rs:613-685

def test_sphere_intersects_obb_vectors():
    assert O.sphere_intersects_obb((0, 0, 0), 1.0, (0, 0, 0), (0.5, 0.5, 0.5), IDENT)
    assert O.sphere_intersects_obb((1, 0, 0), 0.0, (0, 0, 0), (1, 0, 0), IDENT)
    assert O.sphere_intersects_obb((0, 0, 0), 10.0, (1, 1, 1), (0, 0, 0), IDENT)
    tr = affine_rt(quat_axis("y", PI), (5.0, 0.0, 0.0))
    assert O.sphere_intersects_obb((5, 0, 0), 1.0, (0, 0, 0), (0, 0, 0), tr)


# ---------------------------------------------------------------- primitives.rs:711-799

def contains_aabb_test_frustum():
    return O.compute_frustum_perspective(F(math.radians(90.0)), 1.0, 1.0, 100.0, affine_t((2.0, 2.0, 0.0)))


def contains_aabb_test_frustum_with_rotation():
    half_extent_world = F(math.sqrt(F((49.5 * 49.5) * 0.5))) + F(math.sqrt(F(0.5)))
    near = F(50.5) - half_extent_world
    far = near + F(2.0) * half_extent_world
    fov = F(2.0) * F(math.atan(half_extent_world / near))
    return O.compute_frustum_perspective(fov, 1.0, near, far, IDENT)


def test_contains_aabb_vectors():
    fr = contains_aabb_test_frustum()
    assert O.contains_aabb(fr, (0, 0, 0), (0.99, 0.99, 49.49), affine_t((2.0, 2.0, -50.5)))
    assert not O.contains_aabb(fr, (0, 0, 0), (0.99, 0.99, 49.6), affine_t((2.0, 2.0, -50.5)))
    assert not O.contains_aabb(fr, (0, 0, 0), (0.99, 0.99, 0.99), affine_t((0.0, 0.0, 49.6)))
    fr = contains_aabb_test_frustum_with_rotation()
    model = affine_rt(quat_axis("x", PI / 4.0), (0.0, 0.0, -50.5))
    assert O.contains_aabb(fr, (0, 0, 0), (0.99, 0.99, 49.49), model)
    assert not O.contains_aabb(fr, (0, 0, 0), (0.99, 0.99, 49.6), model)


# ---------------------------------------------------------------- primitives.rs:801-857

def test_identity_optimized_equivalence():
    n = np.array([1.0, 1.0, 1.0, -1.0], F)
    n = n / F(np.sqrt(np.sum(n * n, dtype=F)))  # Vec4::normalize
    cases = [((0, 0, 0), (1, 1, 1), O.half_space_new((1.0, 0.0, 0.0, -0.5))),
             ((2.0, -1.0, 0.5), (1.0, 2.0, 0.5), O.half_space_new(n)),
             ((1, 1, 1), (0, 0, 0), O.half_space_new((0.0, 0.0, 1.0, -2.0)))]
    for c, h, hs in cases:
        assert O.is_in_half_space(c, h, hs, IDENT) == O.is_in_half_space_identity(c, h, hs)


def test_intersects_obb_identity_matches_standard():
    aabbs = [((0, 0, 0), (0.5, 0.5, 0.5)), ((1.0, 0.0, 0.5), (0.9, 0.9, 0.9)), ((100, 100, 100), (1, 1, 1))]
    for fr in (frustum(), long_frustum(), big_frustum()):
        for c, h in aabbs:
            assert O.intersects_obb(fr, c, h, IDENT, True, True) == O.intersects_obb_identity(fr, c, h)


def test_bench_primitives_assert():
    # benches/benches/bevy_camera/primitives.rs:41-52: default perspective frustum at identity,
    # unit-ish aabb rotated 45deg about Y and pushed down -Z must intersect (near+far on).
    fr = O.compute_frustum_perspective(F(PI / 4.0), 1.0, 0.1, 1000.0, IDENT)
    model = affine_rt(quat_axis("y", PI / 4.0), (0.0, 0.0, -5.0))
    assert O.intersects_obb(fr, (0, 0, 0), (1, 1, 1), model, True, True)


def test_half_space_new_normalises():
    hs = O.half_space_new((0.0, 0.0, 2.0, 4.0))
    assert hs.tolist() == [0.0, 0.0, 1.0, 2.0]


# ---------------------------------------------------------------- systems.rs:888-925 did_propagate

def T(x, y, z):
    return (np.array([x, y, z], F), np.array([0, 0, 0, 1], F), np.ones(3, F))


def cols(*trs):
    t = np.concatenate([x[0] for x in trs]).astype(F)
    r = np.concatenate([x[1] for x in trs]).astype(F)
    s = np.concatenate([x[2] for x in trs]).astype(F)
    return t, r, s


def gt_from_xyz(x, y, z):
    return O.transform_to_affine(*T(x, y, z))


def test_did_propagate():
    # flat root (row 0), parent (row 1) with two children (rows 2,3)
    t, r, s = cols(T(1, 0, 0), T(1, 0, 0), T(0, 2, 0), T(0, 0, 3))
    parent = np.array([O.NO_PARENT, O.NO_PARENT, 1, 1], np.uint32)
    rc, g, changed = O.propagate_transforms(parent, t, r, s, static_opt=True)
    assert rc == 0
    g = g.reshape(4, 12)
    exp0 = O.affine_mul(gt_from_xyz(1, 0, 0), O.transform_to_affine(*T(0, 2, 0)))
    exp1 = O.affine_mul(gt_from_xyz(1, 0, 0), O.transform_to_affine(*T(0, 0, 3)))
    assert np.array_equal(g[2], exp0) and np.array_equal(g[3], exp1)
    assert g[2][9:].tolist() == [1.0, 2.0, 0.0] and g[3][9:].tolist() == [1.0, 0.0, 3.0]
    assert changed.tolist() == [1, 1, 1, 1]


def test_correct_parent_removed():
    # systems.rs:827-886: root(3.3) <- parent(4.4) <- child(5.5); then orphan parent, then child
    def off(o):
        return T(o, o, o)
    t, r, s = cols(off(3.3), off(4.4), off(5.5))
    rc, g, _ = O.propagate_transforms(np.array([O.NO_PARENT, 0, 1], np.uint32), t, r, s)
    assert np.array_equal(g.reshape(3, 12)[1], gt_from_xyz(F(4.4) + F(3.3), F(4.4) + F(3.3), F(4.4) + F(3.3)))
    rc, g, _ = O.propagate_transforms(np.array([O.NO_PARENT, O.NO_PARENT, 1], np.uint32), t, r, s, global_in=g)
    assert np.array_equal(g.reshape(3, 12)[1], gt_from_xyz(4.4, 4.4, 4.4))
    rc, g, _ = O.propagate_transforms(np.array([O.NO_PARENT] * 3, np.uint32), t, r, s, global_in=g)
    assert np.array_equal(g.reshape(3, 12)[2], gt_from_xyz(5.5, 5.5, 5.5))


def test_correct_transforms_when_no_children():
    # systems.rs:1048-1097: parent(1,0,0) <- identity <- identity: all three equal from_translation
    t, r, s = cols(T(1, 0, 0), T(0, 0, 0), T(0, 0, 0))
    rc, g, _ = O.propagate_transforms(np.array([O.NO_PARENT, 0, 1], np.uint32), t, r, s)
    for row in g.reshape(3, 12):
        assert np.array_equal(row, gt_from_xyz(1, 0, 0))


def test_panic_when_hierarchy_cycle():
    # systems.rs:1099-1164: the reference panics; the oracle reports -1
    t, r, s = cols(T(0, 0, 0), T(0, 0, 0), T(0, 0, 0))
    rc, _, _ = O.propagate_transforms(np.array([O.NO_PARENT, 2, 1], np.uint32), t, r, s)
    assert rc == -1


def test_global_transform_not_overwritten_after_reparenting():
    # systems.rs:1167-1221
    t, r, s = cols(T(1, 1, 1), T(1, 1, 1))
    parent = np.array([O.NO_PARENT, 0], np.uint32)
    rc, g1, _ = O.propagate_transforms(parent, t, r, s, static_opt=True)
    assert np.allclose(g1.reshape(2, 12)[0][9:], 1.0, atol=0.1) and np.allclose(g1.reshape(2, 12)[1][9:], 2.0, atol=0.1)
    rc, g2, changed = O.propagate_transforms(parent, t, r, s, global_in=g1, static_opt=True)
    assert np.array_equal(g1, g2)
    assert changed.tolist() == [1, 0]  # root is re-assigned, child's set_if_neq sees no change


def test_helper_matches_systems():
    # helper.rs:97-146: TRS chain; helper (chain product) vs systems within approx default epsilon
    TAU = F(2 * PI)

    def trs(tv, axis, angle, sc):
        return (np.array(tv, F), quat_axis(axis, angle), np.full(3, sc, F))
    chain = [trs((1, 0, 0), "y", TAU / F(4), 2.0), trs((0, 1, 0), "z", TAU / F(3), 1.5), trs((0, 0, 1), "x", TAU / F(2), 0.3)]
    t, r, s = cols(*chain)
    parent = np.array([O.NO_PARENT, 0, 1], np.uint32)
    rc, g, _ = O.propagate_transforms(parent, t, r, s)
    helper = O.compute_global_transform(parent, t, r, s, 2)
    assert np.allclose(g.reshape(3, 12)[2], helper, atol=np.finfo(F).eps * 8, rtol=0)
    # single transform case is exact
    rc, g1, _ = O.propagate_transforms(parent[:1], t[:3], r[:4], s[:3])
    assert np.array_equal(g1, O.compute_global_transform(parent[:1], t[:3], r[:4], s[:3], 0))


def test_static_optimization_skips_clean_subtrees():
    # systems.rs:708-714: with static optimisation a clean subtree under an unchanged parent G is not touched
    t, r, s = cols(T(1, 0, 0), T(0, 1, 0), T(0, 0, 1), T(5, 5, 5))
    parent = np.array([O.NO_PARENT, 0, 1, 1], np.uint32)
    rc, g, _ = O.propagate_transforms(parent, t, r, s)
    stale = g.copy()
    stale[36:48] = 777.0  # poison row 3's G: a skipped row must keep it
    tree_changed = O.mark_dirty_trees(parent, np.array([0, 0, 1, 0], np.uint8))
    assert tree_changed.tolist() == [1, 1, 1, 0]
    rc, g2, changed = O.propagate_transforms(parent, t, r, s, global_in=stale, static_opt=True, tree_changed=tree_changed)
    assert np.all(g2[36:48] == 777.0)
    assert changed.tolist() == [1, 0, 0, 0]
    # without the optimisation everything is recomputed
    rc, g3, changed3 = O.propagate_transforms(parent, t, r, s, global_in=stale, static_opt=False)
    assert np.array_equal(g3, g) and changed3.tolist() == [1, 0, 0, 1]


# ---------------------------------------------------------------- visibility/mod.rs:1313-1448

def test_view_visibility_lifecycle():
    flags = np.array([O.FLAG_INHERITED_VISIBLE], np.uint8)
    vv = np.array([0], np.uint8)

    def frame(vv, mark):
        vv = O.reset_view_visibility(flags, vv)
        changed = 0
        if mark:  # SetViewVisibility::set_visible through check_visibility with culling disabled
            fl = np.array([O.FLAG_INHERITED_VISIBLE | O.FLAG_NO_FRUSTUM_CULLING], np.uint8)
            vv, _, chg = O.check_visibility(np.tile(IDENT, 1), np.zeros(3, F), np.zeros(3, F), fl,
                                            np.ones(1, np.uint32), vv, frustum())
            changed |= int(chg[0])
        vv, chg = O.mark_newly_hidden(flags, vv)
        return vv, bool(changed | int(chg[0]))

    vv, ch = frame(vv, False); assert not (vv[0] & 1) and not ch   # frame 1
    vv, ch = frame(vv, True);  assert (vv[0] & 1) and ch           # frame 2: hidden -> visible
    vv, ch = frame(vv, True);  assert (vv[0] & 1) and not ch       # frame 3: still visible
    vv, ch = frame(vv, False); assert not (vv[0] & 1) and ch       # frame 4: becomes hidden
    vv, ch = frame(vv, False); assert not (vv[0] & 1) and not ch   # frame 5
    assert vv[0] == 0


def test_no_cpu_culling_rows():
    flags = np.array([O.FLAG_NO_CPU_CULLING | O.FLAG_INHERITED_VISIBLE, O.FLAG_NO_CPU_CULLING], np.uint8)
    vv, chg = O.check_visibility_gpu_culling(flags, np.array([0, 3], np.uint8))
    assert vv.tolist() == [3, 0] and chg.tolist() == [1, 1]
    vv2, chg2 = O.check_visibility_gpu_culling(flags, vv)
    assert vv2.tolist() == [3, 0] and chg2.tolist() == [0, 0]
    assert O.reset_view_visibility(flags, vv).tolist() == [3, 0]  # Without<NoCpuCulling>


def test_visible_entities_sorted_by_entity_bits():
    vis = np.array([1, 0, 1, 1, 1], np.uint8)
    cls = np.array([1, 1, 2, 3, 1], np.uint32)
    keys = np.array([50, 40, 30, 20, 10], np.uint64)
    k, rows = O.visible_entities_sorted(vis, cls, 0, keys)
    assert k.tolist() == [10, 20, 50] and rows.tolist() == [4, 3, 0]
    k, rows = O.visible_entities_sorted(vis, cls, 1, keys)
    assert k.tolist() == [20, 30]


# ---------------------------------------------------------------- cluster/test.rs

def _check_tiling(w, h):
    dims = O.cluster_dimensions_fixed_z(4096, 24, w, h)
    tile, d = O.clusters_update(w, h, dims)
    assert tile[0] * d[0] >= w and tile[1] * d[1] >= h
    assert tile[0] * (d[0] - 1) < w and tile[1] * (d[1] - 1) < h
    assert (tile[0] - 1) * d[0] < w and (tile[1] - 1) * d[1] < h
    assert d[0] <= w and d[1] <= h
    assert d[0] * d[1] * d[2] <= 4096


def test_default_cluster_setup_small_screensizes():
    for x in range(1, 100):
        for y in range(1, 100):
            _check_tiling(x, y)


def test_default_cluster_setup_small_x():
    for x in range(1, 10):
        for y in range(1, 5000, 7):  # strided: the reference sweeps every y; same assertions
            _check_tiling(x, y)
            _check_tiling(y, x)


def test_default_cluster_dims_1080p():
    assert O.cluster_dimensions_fixed_z(4096, 24, 1920, 1080) == (17, 9, 24)
    tile, dims = O.clusters_update(1920, 1080, (16, 9, 24))
    assert tile == (120, 120) and dims == (16, 9, 24)


# ---------------------------------------------------------------- visibility/mod.rs:950-1282
# visibility_propagate_system scenarios (Visibility: 0 Inherited, 1 Hidden, 2 Visible; 0x80 no components)
INH, HID, VIS, NONE = 0, 1, 2, 0x80
NP = O.NO_PARENT


def _propagate(parent, vis, inherited=None):
    inherited = np.zeros(len(vis), np.uint8) if inherited is None else inherited  # InheritedVisibility::default() = HIDDEN
    rc, inh, chg = O.visibility_propagate(np.array(parent, np.uint32), np.array(vis, np.uint8), inherited)
    assert rc == 0
    return inh, chg


def test_visibility_propagation_reference_tree():
    # mod.rs:950-1037: root1 Hidden {child1 {gc}, child2 Hidden {gc}}, root2 Inherited {child1 {gc}, child2 Hidden {gc}}
    parent = [NP, 0, 0, 1, 2, NP, 5, 5, 6, 7]
    vis = [HID, INH, HID, INH, INH, INH, INH, HID, INH, INH]
    inh, _ = _propagate(parent, vis)
    assert inh.tolist() == [0, 0, 0, 0, 0, 1, 1, 0, 1, 0]


def test_visibility_propagation_parent_change_and_removal():
    # mod.rs:1040-1092: child2 re-parented from a Hidden to a Visible parent
    vis = [HID, VIS, INH, INH]
    inh, _ = _propagate([NP, NP, 0, 0], vis)
    assert inh.tolist() == [0, 1, 0, 0]
    inh, _ = _propagate([NP, NP, 0, 1], vis, inh)
    assert inh.tolist() == [0, 1, 0, 1]
    # mod.rs:1094-1125: ChildOf removed -> an inheriting entity without a parent is visible
    inh, _ = _propagate([NP, 0], [HID, INH])
    assert inh.tolist() == [0, 0]
    inh, _ = _propagate([NP, NP], [HID, INH], inh)
    assert inh.tolist() == [0, 1]


def test_visibility_propagation_unconditional_visible():
    # mod.rs:1127-1187
    parent = [NP, 0, 0, 1, 2, NP, NP]
    vis = [VIS, INH, HID, VIS, VIS, INH, HID]
    inh, _ = _propagate(parent, vis)
    assert inh.tolist() == [1, 1, 0, 1, 1, 1, 0]


def test_visibility_propagation_change_detection():
    # mod.rs:1189-1262: chain id1 -> id2 -> id3(Hidden) -> id4
    parent = [NP, 0, 1, 2]
    vis = [INH, INH, HID, INH]
    inh, _ = _propagate(parent, vis)
    assert inh.tolist() == [1, 1, 0, 0]
    vis[0] = HID
    inh, chg = _propagate(parent, vis, inh)
    assert chg.tolist() == [1, 1, 0, 0]
    inh, chg = _propagate(parent, vis, inh)
    assert chg.tolist() == [0, 0, 0, 0]
    vis[2] = INH
    inh, chg = _propagate(parent, vis, inh)
    assert chg.tolist() == [0, 0, 0, 0]
    vis[1] = VIS
    inh, chg = _propagate(parent, vis, inh)
    assert chg.tolist() == [0, 1, 1, 1] and inh.tolist() == [0, 1, 1, 1]
    inh, chg = _propagate(parent, vis, inh)
    assert chg.tolist() == [0, 0, 0, 0]


def test_visibility_propagation_with_invalid_parent():
    # mod.rs:1264-1279: the parent has no visibility components
    inh, _ = _propagate([NP, 0], [NONE, INH])
    assert inh[1] == 1


# ---------------------------------------------------------------- visibility/range.rs:159-161,225-284
def test_visibility_range_is_visible_at_all():
    g = np.tile(IDENT, 4)
    g[9::12] = [0.0, 20.0, 24.999, 25.0]          # translations along x
    c = np.zeros(12, F)
    c[0::3] = 100.0                                # aabb centers far away (only used with use_aabb)
    flags = np.array([O.FLAG_HAS_VISIBILITY_RANGE] * 4, np.uint8)
    rng = np.tile(np.array([20.0, 25.0], F), 4)    # start_margin.start = 20, end_margin.end = 25
    r = O.check_visibility_ranges(g, c, flags, rng, np.zeros(3, F))
    assert r[0].tolist() == [0, 1, 1, 0]           # distance >= start && distance < end
    flags2 = flags | np.uint8(O.FLAG_RANGE_USE_AABB | O.FLAG_HAS_AABB)
    rng2 = np.tile(np.array([100.0, 126.0], F), 4)
    r = O.check_visibility_ranges(g, c, flags2, rng2, np.zeros(3, F))
    assert r[0].tolist() == [1, 1, 1, 1]           # model position = affine * aabb.center = 100 + tx
    no_range = np.array([0, 0, 0, 0], np.uint8)
    assert not O.check_visibility_ranges(g, c, no_range, rng, np.zeros(3, F)).any()
