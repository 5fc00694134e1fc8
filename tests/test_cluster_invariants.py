"""CPU tests that anchor assign_objects_to_clusters without leaning on its own restatement:

* the invariants of tests/cluster_invariants.py, computed in float64 from the view's definition, over the oracle's output for
  perspective / orthographic / rotated cameras, mixed object types, layers, degenerate grids -- and negative controls showing
  that the checker really rejects a dropped entry, a reordered list, a wrong count and an over-wide assignment;
* the default ClusterConfig path of the reference (FixedZ 4096 x 24, MaxClusterableObjectRange, dynamic_resizing;
  crates/bevy_light/src/cluster/assign.rs:324-404, cluster/mod.rs:288-382): the product's host helper
  mi_cluster_config_resolve against the oracle's twin, literal expectations derived by hand from the reference's formulas,
  and the two-frame feedback loop SURVEY.md 8d config 3 asks for;
* the UBO sort + truncate of the gather (assign.rs:297-321).
The same invariants run over the HIP output in tests/test_gpu_cluster.py.
"""
import math

import numpy as np
import pytest

import oracle_lib as O
import cluster_invariants as CI
from bevy_amd import api, workloads as W
from test_abi_and_host import ortho_clip_from_view

F = np.float32
PERSP = O.perspective_infinite_reverse(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)


def rand_lights(n, box, rmax, seed, rmin=0.05):
    r = np.random.default_rng(seed)
    p = r.uniform(-box, box, size=(n, 3))
    rr = r.uniform(rmin, rmax, size=(n, 1))
    return np.concatenate([p, rr], 1).astype(F).reshape(-1)


def oracle_view(cam, dims=(16, 9, 24), far=1000.0, fsd=5.0, ortho=False, screen=(1920, 1080), view_mask=1):
    cfv = ortho_clip_from_view(-60.0, 60.0, -33.75, 33.75, 0.1, 1000.0) if ortho else PERSP
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    return O.cluster_view_setup(cam, cfv, fr, screen[0], screen[1], dims, fsd, far, view_mask)


def run_oracle(view, lights, types=None, layers=None, sd=None, sc=None):
    return O.assign_objects_to_clusters(view, lights, types, layers, sd, sc)


CASES = [
    ("many_lights 20k r0.3", dict(lights=W.many_lights(20_000, 50.0, 0.3))),
    ("many_lights 3k r10", dict(lights=W.many_lights(3_000, 50.0, 10.0))),
    ("random box40", dict(lights=rand_lights(3000, 40, 8, 2))),
    ("random box200 r60", dict(lights=rand_lights(2000, 200, 60, 3))),
    ("near the eye", dict(lights=rand_lights(2000, 10, 3, 4))),
    ("rotated + translated camera", dict(lights=rand_lights(3000, 40, 8, 5), cam=W.many_cubes_camera(300, yaw=1.0, position=(3.0, -2.0, 5.0)))),
    ("32x16x8", dict(lights=rand_lights(2000, 40, 8, 6), dims=(32, 16, 8))),
    ("single cluster", dict(lights=rand_lights(500, 40, 8, 7), dims=(1, 1, 1))),
    ("7x5x3 far 100", dict(lights=rand_lights(2000, 40, 8, 8), dims=(7, 5, 3), far=100.0)),
    ("17x9 on 1920x1080 (tiles do not divide the screen)", dict(lights=rand_lights(2000, 40, 8, 9), dims=(17, 9, 24))),
    ("orthographic", dict(lights=rand_lights(3000, 50, 8, 10), ortho=True)),
    ("orthographic rotated", dict(lights=rand_lights(3000, 50, 8, 11), ortho=True, cam=W.many_cubes_camera(100, yaw=0.6))),
]


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_invariants_hold_for_the_oracle(name, case):
    case = dict(case)
    lights = case.pop("lights")
    view = oracle_view(case.pop("cam", W.many_cubes_camera(0)), **case)
    off, idx, counts, far, total = run_oracle(view, lights)
    rep = CI.check_all(view, lights, None, None, off, idx, counts, far, total)
    if name != "single cluster":
        assert rep["pairs"] > rep["listed_objects"] > 0 and rep["centre_checked"] > 0


def test_invariants_with_mixed_types_layers_and_spot_cones():
    rng = np.random.default_rng(5)
    n = 3000
    pos = rng.uniform(-60, 60, size=(n, 3)).astype(F)
    rr = np.where(rng.random(n) < 0.1, rng.uniform(20, 200, n), rng.uniform(0.5, 12, n)).astype(F)
    pr = np.concatenate([pos, rr[:, None]], axis=1).astype(F).reshape(-1)
    types = np.sort(rng.integers(0, 6, n)).astype(np.uint8)
    layers = np.where(rng.random(n) < 0.1, 2, 1).astype(np.uint32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F).reshape(-1)
    ang = rng.uniform(0.1, 1.4, n).astype(F)
    sc = np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(F).reshape(-1)
    for cam in (W.many_cubes_camera(0), W.many_cubes_camera(12, yaw=2.2, position=(5.0, 1.0, -3.0))):
        for ortho in (False, True):
            view = oracle_view(cam, ortho=ortho)
            off, idx, counts, far, total = run_oracle(view, pr, types, layers, d, sc)
            rep = CI.check_all(view, pr, types, layers, off, idx, counts, far, total)
            assert rep["superset_hits"] > 1000
            # a spot light is listed in a subset of what the same sphere gets as a point light
            as_points = types.copy()
            as_points[types == 1] = 0
            # (types must stay grouped for the structure check: compare pair sets instead of re-checking structure)
            off2, idx2, _, _, _ = run_oracle(view, pr, as_points, layers, d, sc)
            o1, c1 = CI.expand_lists(off, idx)
            o2, c2 = CI.expand_lists(off2, idx2)
            spot = types[o1] == 1
            assert set(zip(o1[spot].tolist(), c1[spot].tolist())) <= set(zip(o2.tolist(), c2.tolist()))
            assert set(zip(o1[~spot].tolist(), c1[~spot].tolist())) == set(z for z in zip(o2.tolist(), c2.tolist()) if types[z[0]] != 1)


def test_the_checker_rejects_wrong_assignments():
    """Negative controls: each corruption of a correct assignment trips the invariant that is meant to see it."""
    lights = rand_lights(1500, 40, 8, 21)
    view = oracle_view(W.many_cubes_camera(0))
    off, idx, counts, far, total = run_oracle(view, lights)
    CI.check_all(view, lights, None, None, off, idx, counts, far, total)
    g = CI.view_geometry(view)
    obj, cl = CI.expand_lists(off, idx)

    def rebuild(keep):
        o, c = obj[keep], cl[keep]
        n_cl = len(off) - 1
        cnt = np.bincount(c, minlength=n_cl)
        off2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        counts2 = np.zeros((n_cl, 6), np.uint32)
        counts2[:, 0] = cnt
        return off2, o.astype(np.uint32), counts2, int(cnt.sum())

    # 1. drop every pair of the cluster that holds some light's centre -> `centre` (and `superset`) must fire
    lo, hi, cv = CI.object_boxes(g, lights)
    victim = int(obj[len(obj) // 2])
    keep = ~((obj == victim))
    off2, idx2, counts2, tot2 = rebuild(keep)
    with pytest.raises(AssertionError):
        CI.check_all(view, lights, None, None, off2, idx2, counts2, far, tot2)
    # 2. drop ONE non-centre pair of a big light -> only the sample-point superset test can see it
    big = int(np.argmax(np.bincount(obj)))
    pairs_of_big = np.nonzero(obj == big)[0]
    surely_in, _ = CI.passes_early_outs(g, lights, None, 1e-4)
    hits = 0
    for k in pairs_of_big[:: max(1, len(pairs_of_big) // 12)]:
        keep = np.ones(len(obj), bool)
        keep[k] = False
        off2, idx2, counts2, tot2 = rebuild(keep)
        try:
            CI.check_superset(g, lights, None, off2, idx2, surely_in)
        except AssertionError:
            hits += 1
    assert hits >= 6, f"the superset check noticed only {hits} of ~12 dropped interior pairs"
    # 3. a swapped pair inside one cluster's list -> `structure`
    c_big = int(np.argmax(np.diff(off.astype(np.int64))))
    bad = idx.copy()
    a = int(off[c_big])
    bad[a], bad[a + 1] = bad[a + 1], bad[a]
    with pytest.raises(AssertionError, match="ascending"):
        CI.check_structure(len(off) - 1, off, bad, counts, total, None)
    # 4. a count that does not match
    bad_counts = counts.copy()
    bad_counts[c_big, 0] += 1
    with pytest.raises(AssertionError):
        CI.check_structure(len(off) - 1, off, idx, bad_counts, total, None)
    # 5. an extra pair far outside the object's box -> `box`; in a z slice the sphere does not reach -> `z slab`
    small = int(np.argmin(np.where(np.bincount(obj, minlength=len(lights) // 4) > 0, np.bincount(obj, minlength=len(lights) // 4), 1 << 30)))
    far_cluster = (0 * 16 + 0) * 24 + 23 if cl[obj == small][0] % 24 < 12 else 0
    o3 = np.concatenate([obj, [small]])
    c3 = np.concatenate([cl, [far_cluster]])
    order = np.lexsort((o3, c3))
    o3, c3 = o3[order], c3[order]
    cnt = np.bincount(c3, minlength=len(off) - 1)
    off3 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    with pytest.raises(AssertionError, match="box|z slice"):
        CI.check_exclusion_and_box(g, lights, None, off3, o3.astype(np.uint32))
        CI.check_z_slab(g, lights, off3, o3.astype(np.uint32))
    # 6. a culled light that is listed anyway -> `exclusion`
    _, surely_out = CI.passes_early_outs(g, lights, None, 1e-4)
    ghost = int(np.nonzero(surely_out)[0][0])
    o4 = np.concatenate([obj, [ghost]])
    c4 = np.concatenate([cl, [100]])
    order = np.lexsort((o4, c4))
    cnt = np.bincount(c4, minlength=len(off) - 1)
    off4 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    with pytest.raises(AssertionError, match="outside the frustum"):
        CI.check_exclusion_and_box(g, lights, None, off4, o4[order].astype(np.uint32))


# ---- ClusterConfig -> per-frame view parameters (assign.rs:324-404) ---------------------------------------------

def mi_config(**kw):
    cfg = api.cluster_config_default()
    for k, v in kw.items():
        if k == "dimensions":
            cfg.dimensions[:] = v
        else:
            setattr(cfg, k, v)
    return cfg


def orc_config(mi):
    cfg = O.ClusterConfig()
    for name, _ in O.ClusterConfig._fields_:
        if name == "dimensions":
            cfg.dimensions[:] = list(mi.dimensions)
        else:
            setattr(cfg, name, getattr(mi, name))
    return cfg


def history(farthest=None, total=None):
    h = api.ClusterHistory()
    if farthest is not None:
        h.has_farthest_z, h.farthest_z = 1, farthest
    if total is not None:
        h.has_total_cluster_index_count, h.total_cluster_index_count = 1, total
    return h


def test_default_config_is_the_references_default():
    """ClusterConfig::default(), crates/bevy_light/src/cluster/mod.rs:288-308."""
    for cfg in (api.cluster_config_default(), O.cluster_config_default()):
        assert (cfg.kind, cfg.total, cfg.z_slices, cfg.far_z_mode, cfg.dynamic_resizing) == (3, 4096, 24, 0, 1)
        assert cfg.first_slice_depth == 5.0


def test_config_resolve_literal_expectations():
    """Hand-derived from the reference's formulas (values are exact in f32)."""
    cfg = api.cluster_config_default()
    # first frame: no statistics.  1920x1080: per_layer = 4096/24 = 170.67; y = sqrt(170.67 / 1.7778) = 9.798 -> 9;
    # x = (9.798 * 1.7778) as u32 = 17 (mod.rs:323-333); far_z = DEFAULT_FAR_DEPTH (assign.rs:37,351-353)
    r = api.cluster_config_resolve(cfg, None, 1920, 1080)
    assert (r.active, tuple(r.requested_dims), r.first_slice_depth, r.far_z) == (1, (17, 9, 24), 5.0, 1000.0)
    # second frame: last_frame_farthest_z feeds far_z; 10 000 indices <= MAX_INDICES: no resizing
    r = api.cluster_config_resolve(cfg, history(50.25, 10_000), 1920, 1080)
    assert (tuple(r.requested_dims), r.far_z) == ((17, 9, 24), 50.25)
    # 65 536 indices: index_ratio = 16384/65536 = 0.25, xy_ratio = 0.5 -> (8, 4) (assign.rs:396-403)
    r = api.cluster_config_resolve(cfg, history(50.25, 65_536), 1920, 1080)
    assert tuple(r.requested_dims) == (8, 4, 24)
    # a huge count cannot shrink below one cluster per axis (.max(1))
    r = api.cluster_config_resolve(cfg, history(50.25, 1 << 40), 1920, 1080)
    assert tuple(r.requested_dims) == (1, 1, 24)
    # exactly MAX_INDICES is not "greater than" (:387-388)
    r = api.cluster_config_resolve(cfg, history(50.25, 16_384), 1920, 1080)
    assert tuple(r.requested_dims) == (17, 9, 24)
    # Constant far, no dynamic resizing: statistics are ignored
    xyz = mi_config(kind=api.CLUSTER_CONFIG_XYZ, dimensions=[16, 9, 24], far_z_mode=api.CLUSTER_FAR_Z_CONSTANT, far_z_constant=1000.0,
                    dynamic_resizing=0)
    r = api.cluster_config_resolve(xyz, history(3.0, 10 ** 9), 1920, 1080)
    assert (tuple(r.requested_dims), r.far_z, r.first_slice_depth) == ((16, 9, 24), 1000.0, 5.0)
    # Single: one cluster, first_slice_depth 0, MaxClusterableObjectRange, never resized (mod.rs:349-382)
    single = mi_config(kind=api.CLUSTER_CONFIG_SINGLE)
    r = api.cluster_config_resolve(single, history(77.0, 10 ** 9), 1920, 1080)
    assert (tuple(r.requested_dims), r.first_slice_depth, r.far_z) == ((1, 1, 1), 0.0, 77.0)
    # None and an empty viewport clear the clusters (assign.rs:328-339)
    assert api.cluster_config_resolve(mi_config(kind=api.CLUSTER_CONFIG_NONE), None, 1920, 1080).active == 0
    assert api.cluster_config_resolve(cfg, None, 0, 1080).active == 0
    # portrait and extreme aspect ratios (the `check extremes` branch, mod.rs:335-343)
    # portrait: aspect 0.5625, y = sqrt(170.67 / 0.5625) = 17.42 -> 17, x = (17.42 * 0.5625) as u32 = 9
    assert tuple(api.cluster_config_resolve(cfg, None, 1080, 1920).requested_dims) == (9, 17, 24)
    # aspect 1000: y = sqrt(0.17067) = 0.413 -> 0 => x = per_layer as u32 = 170, y = 1
    assert tuple(api.cluster_config_resolve(cfg, None, 100_000, 100).requested_dims) == (170, 1, 24)
    # aspect 0.001: x = (413.1 * 0.001) as u32 = 0 => x = 1, y = per_layer as u32 = 170
    assert tuple(api.cluster_config_resolve(cfg, None, 100, 100_000).requested_dims) == (1, 170, 24)


def test_config_resolve_matches_the_oracle_twin():
    rng = np.random.default_rng(3)
    n_checked = 0
    for trial in range(400):
        kind = int(rng.integers(0, 4))
        cfg = mi_config(kind=kind, dimensions=[int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 30))],
                        total=int(rng.integers(1, 5000)), z_slices=int(rng.integers(1, 40)),
                        first_slice_depth=float(F(rng.uniform(0.1, 20))), far_z_mode=int(rng.integers(0, 2)),
                        far_z_constant=float(F(rng.uniform(10, 2000))), dynamic_resizing=int(rng.integers(0, 2)))
        w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 3000))
        if trial % 50 == 0:
            w = 0
        farthest = None if rng.random() < 0.3 else float(F(rng.uniform(1, 500)))
        total = None if rng.random() < 0.3 else int(rng.integers(0, 400_000))
        max_idx = int(rng.choice([16_384, 1 << 20]))
        got = api.cluster_config_resolve(cfg, history(farthest, total), w, h, max_idx)
        want = O.cluster_config_resolve(orc_config(cfg), farthest, total, w, h, max_idx)
        if want is None:
            assert got.active == 0
            continue
        n_checked += 1
        assert got.active == 1 and tuple(got.requested_dims) == want[0], (trial, tuple(got.requested_dims), want)
        assert F(got.first_slice_depth).tobytes() == F(want[1]).tobytes() and F(got.far_z).tobytes() == F(want[2]).tobytes()
    assert n_checked > 200


def two_frames(lights, cam, screen=(1920, 1080)):
    """The system's feedback loop on the oracle: frame k's statistics configure frame k+1 (assign.rs:350-355,384-404,810-811)."""
    cfg = O.cluster_config_default()
    fr = api.compute_frustum(PERSP, cam, W.CAMERA_FAR)
    farthest, total, frames = None, None, []
    for _ in range(3):
        req, fsd, far = O.cluster_config_resolve(cfg, farthest, total, screen[0], screen[1])
        view = O.cluster_view_setup(cam, PERSP, fr, screen[0], screen[1], req, fsd, far)
        off, idx, counts, farthest, total = O.assign_objects_to_clusters(view, lights)
        frames.append((req, far, view, off, idx, counts, farthest, total))
    return frames


def test_default_config_two_frame_feedback_on_the_oracle():
    """SURVEY.md 8d config 3, secondary check: many_lights under the DEFAULT ClusterConfig for consecutive frames."""
    cam = W.many_cubes_camera(0)
    lights = W.many_lights(100_000, 50.0, 0.3)
    f0, f1, f2 = two_frames(lights, cam)
    assert f0[0] == (17, 9, 24) and f0[1] == 1000.0
    # every light sits on the R = 50 shell: the farthest reach is 50 + 0.3 (give or take the f32 rounding of positions)
    assert abs(f0[6] - 50.3) < 1e-3 and f1[1] == pytest.approx(f0[6]) and f1[0] == (17, 9, 24)
    assert f0[7] < 16_384  # range 0.3: few indices, no resizing
    for fr_ in (f0, f1):
        CI.check_all(fr_[2], lights, None, None, fr_[3], fr_[4], fr_[5], fr_[6], fr_[7], superset=False)
    # with far_z pulled in from 1000 to 50.3 the shell of lights moves from the middle z slices to the last ones
    z0 = CI.cluster_coords(CI.view_geometry(f0[2]), CI.expand_lists(f0[3], f0[4])[1])[2]
    z1 = CI.cluster_coords(CI.view_geometry(f1[2]), CI.expand_lists(f1[3], f1[4])[1])[2]
    assert f1[7] != f0[7] and z0.max() < 23 and z1.max() == 23 and z1.min() > z0.max()
    assert f2[0] == f1[0] and f2[1] == f1[1] and f2[7] == f1[7]  # steady state
    # big lights: frame 0 overflows MAX_INDICES, frame 1 runs on a coarser grid
    big = W.many_lights(20_000, 50.0, 6.0)
    g0, g1, g2 = two_frames(big, cam)
    assert g0[7] > 16_384
    ratio = F(math.sqrt(F(16_384) / F(g0[7])))
    assert g1[0] == (max(int(math.floor(F(17) * ratio)), 1), max(int(math.floor(F(9) * ratio)), 1), 24)
    assert g1[0][0] < 17 and g1[7] < g0[7]
    CI.check_all(g1[2], big, None, None, g1[3], g1[4], g1[5], g1[6], g1[7], superset=False)


def test_sort_truncate_matches_the_oracle_and_the_rule():
    """assign.rs:297-321 with ClusterableObjectType::ordering() :108-128."""
    rng = np.random.default_rng(11)
    n = 700
    types = rng.integers(0, 6, n).astype(np.uint8)
    shadow = rng.integers(0, 2, n).astype(np.uint8)
    vol = rng.integers(0, 2, n).astype(np.uint8)
    ent = rng.permutation(np.arange(1000, 1000 + n)).astype(np.uint64) | (rng.integers(0, 3, n).astype(np.uint64) << np.uint64(32))
    got = api.cluster_sort_truncate(types, shadow, vol, ent, 204, False)
    want = O.cluster_sort_truncate(types, shadow, vol, ent, 204, False)
    assert np.array_equal(got, want) and len(got) == 204
    light = types < 2
    keys = sorted(range(n), key=lambda i: (int(types[i]), bool(light[i] and not shadow[i]), bool(light[i] and not vol[i]), int(ent[i])))
    assert got.tolist() == keys[:204]
    # nothing happens with storage buffers, or when the objects fit
    assert np.array_equal(api.cluster_sort_truncate(types, shadow, vol, ent, 204, True), np.arange(n))
    assert np.array_equal(api.cluster_sort_truncate(types[:100], shadow[:100], vol[:100], ent[:100], 204, False), np.arange(100))
    assert np.array_equal(O.cluster_sort_truncate(types, shadow, vol, ent, 204, True), np.arange(n))


def test_layers_above_31_select_the_same_objects_as_a_filtered_list():
    """orc_assign_objects_to_clusters_layers64: RenderLayers::intersects over the first u64 word (render_layers.rs:121-135).  The lists
    over objects with (lo, hi) layer words against a view's two words equal the lists over just the objects that intersect it."""
    rng = np.random.default_rng(9)
    n = 1500
    pos = rng.uniform(-50, 50, size=(n, 3)).astype(F)
    pr = np.concatenate([pos, rng.uniform(0.5, 15, n).astype(F)[:, None]], axis=1).astype(F)
    lo = np.where(rng.random(n) < 0.5, 1, 0).astype(np.uint32) | np.where(rng.random(n) < 0.1, 1 << 7, 0).astype(np.uint32)
    hi = np.where(rng.random(n) < 0.4, 1 << 8, 0).astype(np.uint32) | np.where(rng.random(n) < 0.1, 1 << 31, 0).astype(np.uint32)
    view = oracle_view(W.many_cubes_camera(0))
    for vlo, vhi in ((1, 0), (0, 1 << 8), (1 << 7, 1 << 31), (0, 0)):
        view.view_layer_mask, view.view_layer_mask_hi = vlo, vhi
        off, idx, counts, far, total = O.assign_objects_to_clusters(view, pr.reshape(-1), None, lo, layer_mask_hi=hi)
        sel = np.nonzero((lo & np.uint32(vlo)) | (hi & np.uint32(vhi)))[0]
        view.view_layer_mask, view.view_layer_mask_hi = 1, 0
        off2, idx2, counts2, far2, total2 = O.assign_objects_to_clusters(view, pr[sel].reshape(-1).copy())
        assert total == total2 and np.array_equal(off, off2) and np.array_equal(idx, sel[idx2].astype(np.uint32)), (vlo, vhi)
        if vlo == 0 and vhi == 0:
            assert total == 0
