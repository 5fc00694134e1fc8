"""Invariants of assign_objects_to_clusters that do not depend on any restatement of its refinement loops.

The reference has no test that exercises the assignment: these checks are the independent anchor.  They
are computed in float64 numpy straight from the view's definition -- camera transform, projection, grid dimensions and
the z-slice formula of the reference (crates/bevy_light/src/cluster/assign.rs:903-920, 1046-1062) -- and from the light
spheres, and they are applied to BOTH the oracle's output (CPU tests) and the HIP library's output (GPU tests):

  structure   offsets start at 0, are monotone and end at the total; every cluster's list is strictly ascending in object
              index (= the reference's push order); ClusterableObjectCounts match the listed objects' types; the total
              is the sum.
  exclusion   an object on a layer the view does not render, or whose sphere lies outside the frustum by a margin,
              or (row-bound lights) that is not visible, appears in no cluster.            [assign.rs:489,496,194]
  box         every cluster an object is in lies inside the object's min/max cluster box -- the view-space AABB of its
              sphere pushed through the projection (:501-526) -- recomputed here in float64 with one cluster of slack
              in x / y / z for rounding at cell boundaries.
  centre      an object whose centre lies inside the frustum (by a margin) is listed in the cluster that contains its
              centre.
  superset    for sample points strictly inside every cluster cell: a light whose centre is in front of the eye plane
              and whose sphere contains the point (by a margin) must be listed in that cluster -- the assignment may
              be conservative, never lossy.  (Centres BEHIND the eye are excluded: the reference derives their centre
              cluster from an NDC position divided by a negative w and its walk is not conservative there; that quirk
              is reproduced bit for bit by the parity tests, it is not an invariant.)
  z slab      every listed (object, cluster) pair: the sphere's depth interval overlaps the cluster's z slice (the z step
              of the refinement is exact).  The distance from the sphere to the cluster's view-space AABB
              (compute_aabb_for_cluster, :834-900) is reported as a diagnostic only: the reference's x / y steps are
              conservative by design (a one-column box is emitted without any plane test), so "AABB touches sphere"
              is not an invariant of the reference.  (Spot lights are additionally cone-culled per cluster, so they
              are subject to `structure`, `exclusion`, `box` and `z slab` only, not to `superset` / `centre`.)
  farthest_z  = max(0, max over the objects that pass the two early-outs of -view_z + range * scale_z)  (:558-561).
"""
import numpy as np

D = np.float64


def _mat4(a16):
    return np.asarray(a16, D).reshape(4, 4).T  # column-major storage -> math matrix


def view_geometry(view):
    """Everything the checks need from a (mi_ / orc_) cluster view struct, as float64."""
    g = dict(dims=tuple(int(d) for d in view.dims), ortho=bool(view.is_orthographic), near=D(view.near_), far=D(view.far_),
             vfw=_mat4(view.view_from_world), cfv=_mat4(view.clip_from_view), vfc=_mat4(view.view_from_clip),
             scale=np.array(list(view.view_from_world_scale), D), scale_max=D(view.view_from_world_scale_max),
             frustum=np.array(list(view.frustum), D).reshape(6, 4), layer_mask=int(view.view_layer_mask),
             screen=tuple(int(s) for s in view.screen_size), tile=tuple(int(t) for t in view.tile_size))
    return g


def slice_depths(g):
    """View-space depth (positive) of the z-slice boundaries 0..Z (z_slice_to_view_z, assign.rs:903-920)."""
    Z = g["dims"][2]
    k = np.arange(Z + 1, dtype=D)
    if g["ortho"]:
        return g["near"] + (g["far"] - g["near"]) * k / Z
    d = np.zeros(Z + 1, D)
    if Z > 1:
        d[1:] = g["near"] * (g["far"] / g["near"]) ** ((k[1:] - 1) / (Z - 1))
    else:
        d[1] = g["far"]
    return d


def depth_to_slice(g, depth):
    """view_z_to_z_slice (assign.rs:1046-1062) in float64, unclamped float result."""
    Z = g["dims"][2]
    depth = np.asarray(depth, D)
    if g["ortho"]:
        return np.floor((depth - g["near"]) * Z / (g["far"] - g["near"]))
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.log(np.maximum(depth, 1e-300) / g["near"]) * (Z - 1) / np.log(g["far"] / g["near"]) + 1.0 if Z > 1 else np.zeros_like(depth)
    return np.where(depth <= g["near"], 0.0, np.floor(s))


def ndc_of(g, p_view):
    """(N,3) view-space points -> (N,3) NDC."""
    p4 = np.concatenate([p_view, np.ones((len(p_view), 1), D)], axis=1)
    c = p4 @ g["cfv"].T
    return c[:, :3] / c[:, 3:4]


def view_of_ndc(g, ndc_xy, depth):
    """View-space point at NDC (x, y) and view depth `depth` (positive, along -z)."""
    ndc_xy = np.asarray(ndc_xy, D)
    depth = np.asarray(depth, D)
    P = g["cfv"]
    if g["ortho"]:
        x = (ndc_xy[..., 0] - P[0, 3]) / P[0, 0]
        y = (ndc_xy[..., 1] - P[1, 3]) / P[1, 1]
    else:
        x = ndc_xy[..., 0] * depth / P[0, 0]
        y = ndc_xy[..., 1] * depth / P[1, 1]
    return np.stack([x, y, -depth * np.ones_like(x)], axis=-1)


def cluster_coords(g, index):
    dx, dy, dz = g["dims"]
    index = np.asarray(index, np.int64)
    z = index % dz
    xy = index // dz
    return xy % dx, xy // dx, z


def expand_lists(offsets, indices):
    """CSR -> (object, cluster) pair arrays."""
    counts = np.diff(offsets.astype(np.int64))
    cl = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    return indices.astype(np.int64), cl


def check_structure(n_clusters, offsets, indices, counts, total, obj_type):
    offsets = np.asarray(offsets, np.int64)
    assert len(offsets) == n_clusters + 1 and offsets[0] == 0, "offsets must start at 0"
    assert (np.diff(offsets) >= 0).all(), "offsets not monotone"
    assert offsets[-1] == total == len(indices), f"offsets end {offsets[-1]}, total {total}, {len(indices)} indices"
    obj, cl = expand_lists(offsets, indices)
    if len(obj) > 1:
        same = cl[1:] == cl[:-1]
        assert (obj[1:][same] > obj[:-1][same]).all(), "a cluster's list is not strictly ascending in object index"
    counts = np.asarray(counts, np.int64).reshape(n_clusters, 6)
    assert (counts.sum(axis=1) == np.diff(offsets)).all(), "ClusterableObjectCounts do not add up to the list lengths"
    ty = np.zeros(len(obj), np.int64) if obj_type is None else np.asarray(obj_type, np.int64)[obj]
    want = np.zeros((n_clusters, 6), np.int64)
    np.add.at(want, (cl, ty), 1)
    assert (want == counts).all(), "per-type counts do not match the listed objects"
    if len(obj) > 1:  # gather order groups the types: within a cluster the types are non-decreasing
        assert (ty[1:][same] >= ty[:-1][same]).all() or obj_type is None


def passes_early_outs(g, pos_range, layer_mask, margin):
    """-> (surely_in, surely_out) per object for the layer + frustum-vs-sphere tests (assign.rs:489,496)."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    layers = np.ones(len(pr), np.int64) if layer_mask is None else np.asarray(layer_mask, np.int64)
    on_layer = (layers & g["layer_mask"]) != 0
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    d = c4 @ g["frustum"].T + pr[:, 3:4]          # n.c + d + r per plane; inside iff > 0 for all six
    tol = margin * (1.0 + np.abs(pr[:, :3]).max(axis=1, keepdims=True) + pr[:, 3:4])
    surely_in = on_layer & (d > tol).all(axis=1)
    surely_out = ~on_layer | (d < -tol).any(axis=1)
    return surely_in, surely_out


def object_boxes(g, pos_range):
    """float64 restatement of the min/max cluster box (assign.rs:501-526, 948-1036) as float cluster coordinates."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    cv = (c4 @ g["vfw"].T)[:, :3]
    he = np.abs(g["scale"])[None, :] * pr[:, 3:4]
    vmin, vmax = cv - he, cv + he
    tiny = -np.finfo(np.float32).tiny
    vmin[:, 2] = np.minimum(vmin[:, 2], tiny)
    vmax[:, 2] = np.minimum(vmax[:, 2], tiny)
    corners = [vmin, np.stack([vmin[:, 0], vmin[:, 1], vmax[:, 2]], 1), np.stack([vmax[:, 0], vmax[:, 1], vmin[:, 2]], 1), vmax]
    ndcs = np.stack([ndc_of(g, c) for c in corners])
    nmin, nmax = np.clip(ndcs.min(axis=0)[:, :2], -1, 1), np.clip(ndcs.max(axis=0)[:, :2], -1, 1)
    dx, dy, dz = g["dims"]

    def to_cluster(nxy, vz):
        fx = np.clip(nxy[:, 0] * 0.5 + 0.5, 0, 1) * dx
        fy = np.clip(nxy[:, 1] * -0.5 + 0.5, 0, 1) * dy
        zs = np.clip(depth_to_slice(g, -vz), 0, dz - 1)
        return np.stack([np.minimum(np.floor(fx), dx - 1), np.minimum(np.floor(fy), dy - 1), zs], axis=1), np.stack([fx, fy], 1)
    a, fa = to_cluster(nmin, vmin[:, 2])
    b, fb = to_cluster(nmax, vmax[:, 2])
    return np.minimum(a, b), np.maximum(a, b), cv


def check_exclusion_and_box(g, pos_range, layer_mask, offsets, indices, visible=None, margin=1e-4, slack=1):
    obj, cl = expand_lists(np.asarray(offsets), np.asarray(indices))
    surely_in, surely_out = passes_early_outs(g, pos_range, layer_mask, margin)
    listed = np.zeros(len(surely_in), bool)
    listed[obj] = True
    bad = np.nonzero(listed & surely_out)[0]
    assert bad.size == 0, f"{bad.size} objects outside the frustum / off-layer are listed, first {bad[:5].tolist()}"
    if visible is not None:
        bad = np.nonzero(listed & ~np.asarray(visible, bool))[0]
        assert bad.size == 0, f"{bad.size} invisible objects are listed, first {bad[:5].tolist()}"
    lo, hi, _ = object_boxes(g, pos_range)
    cx, cy, cz = cluster_coords(g, cl)
    for k, cc in enumerate((cx, cy, cz)):
        out = (cc < lo[obj, k] - slack) | (cc > hi[obj, k] + slack)
        assert not out.any(), (f"axis {k}: {int(out.sum())} listed clusters lie outside the object's min/max box, first object "
                               f"{int(obj[out][0])} cluster {int(cl[out][0])}")
    return surely_in, listed


def check_centre(g, pos_range, obj_type, offsets, indices, surely_in, visible=None, margin=1e-3):
    """An object whose centre is well inside the frustum is in the cluster that holds its centre."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    dx, dy, dz = g["dims"]
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    cv = (c4 @ g["vfw"].T)[:, :3]
    depth = -cv[:, 2]
    ndc = ndc_of(g, cv)
    fx, fy = (ndc[:, 0] * 0.5 + 0.5) * dx, (ndc[:, 1] * -0.5 + 0.5) * dy
    bounds = slice_depths(g)
    with np.errstate(invalid="ignore"):
        zs = np.clip(depth_to_slice(g, depth), 0, dz - 1).astype(np.int64)
    # stay away from every cell boundary so that float32 rounding cannot move the centre to a neighbour
    frac_ok = (np.abs(fx - np.round(fx)) > margin) & (np.abs(fy - np.round(fy)) > margin)
    zb = np.minimum(np.abs(depth - bounds[zs]), np.abs(depth - bounds[np.minimum(zs + 1, dz)]))
    inside = (fx > 0) & (fx < dx) & (fy > 0) & (fy < dy) & (depth > bounds[0]) & (depth < bounds[-1]) & frac_ok & (zb > margin * (1 + depth))
    sel = surely_in & inside
    if obj_type is not None:
        sel &= np.asarray(obj_type) != 1  # spot lights are additionally cone-culled per cluster
    if visible is not None:
        sel &= np.asarray(visible, bool)
    want_cluster = (np.floor(fy).astype(np.int64) * dx + np.floor(fx).astype(np.int64)) * dz + zs
    obj, cl = expand_lists(np.asarray(offsets), np.asarray(indices))
    have = set(zip(obj.tolist(), cl.tolist()))
    missing = [int(i) for i in np.nonzero(sel)[0] if (int(i), int(want_cluster[i])) not in have]
    assert not missing, f"{len(missing)} objects are not listed in the cluster that contains their centre, first {missing[:5]}"
    return int(sel.sum())


def cell_sample_points(g, per_axis=2, inset=0.2):
    """(C, K, 3) view-space points strictly inside each cluster cell (inset from the cell's faces)."""
    dx, dy, dz = g["dims"]
    fr = np.linspace(inset, 1.0 - inset, per_axis)
    bounds = slice_depths(g)
    pts = np.zeros((dy, dx, dz, per_axis ** 3, 3), D)
    k = 0
    for a in fr:
        for b in fr:
            for c in fr:
                ndc_x = (np.arange(dx) + a) / dx * 2.0 - 1.0
                ndc_y = (1.0 - (np.arange(dy) + b) / dy) * 2.0 - 1.0
                depth = bounds[:-1] + c * (bounds[1:] - bounds[:-1])
                if not g["ortho"]:
                    depth = np.maximum(depth, 1e-6)
                NX, NY, DZ = np.meshgrid(ndc_x, ndc_y, depth, indexing="xy")  # (dy, dx, dz)
                pts[:, :, :, k, :] = view_of_ndc(g, np.stack([NX, NY], -1), DZ)
                k += 1
    return pts.reshape(dy * dx * dz, per_axis ** 3, 3)


def check_superset(g, pos_range, obj_type, offsets, indices, surely_in, visible=None, margin=1e-3, per_axis=2, chunk=256):
    """Brute force: a (non-spot) light whose view-space sphere contains a point strictly inside a cell is listed there."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    cv = (c4 @ g["vfw"].T)[:, :3]
    radius = pr[:, 3] * g["scale_max"]
    sel = surely_in.copy()
    if obj_type is not None:
        sel &= np.asarray(obj_type) != 1
    if visible is not None:
        sel &= np.asarray(visible, bool)
    # Only lights whose CENTRE is in front of the eye plane.  For a centre behind it the reference derives the centre
    # cluster from an NDC position divided by a negative w (assign.rs:583-605) -- mirrored in x / y -- and its row /
    # column walk is then not conservative (whole rows can be skipped).  That is reference behaviour, reproduced bit for
    # bit by the parity tests; it is not an invariant, so those lights are left to `structure`, `box` and `z slab`.
    sel &= -cv[:, 2] > margin * (1.0 + radius)
    cand = np.nonzero(sel)[0]
    pts = cell_sample_points(g, per_axis)             # (C, K, 3)
    C_ = pts.shape[0]
    obj, cl = expand_lists(np.asarray(offsets), np.asarray(indices))
    listed = np.zeros((len(pr), C_), bool) if len(pr) * C_ <= 64_000_000 else None
    if listed is not None:
        listed[obj, cl] = True
    else:
        have = set(zip(obj.tolist(), cl.tolist()))
    n_hits, missing = 0, []
    for s in range(0, len(cand), chunk):
        ids = cand[s:s + chunk]
        d2 = ((pts[None, :, :, :] - cv[ids][:, None, None, :]) ** 2).sum(axis=-1)     # (chunk, C, K)
        r_in = radius[ids] * (1.0 - margin)
        hit = (d2 < (r_in * r_in)[:, None, None]).any(axis=-1)                       # (chunk, C)
        n_hits += int(hit.sum())
        ii, cc = np.nonzero(hit)
        if listed is not None:
            ok = listed[ids[ii], cc]
            for i_, c_ in zip(ids[ii][~ok].tolist(), cc[~ok].tolist()):
                missing.append((i_, c_))
        else:
            for i_, c_ in zip(ids[ii].tolist(), cc.tolist()):
                if (i_, c_) not in have:
                    missing.append((i_, c_))
    assert not missing, f"{len(missing)} (object, cluster) pairs are missing although the sphere reaches into the cell, first {missing[:5]}"
    return n_hits


def cluster_aabbs(g):
    """compute_aabb_for_cluster (assign.rs:834-900) in float64 -> (C,3) min, (C,3) max in view space."""
    dx, dy, dz = g["dims"]
    tsx, tsy = g["tile"]
    sw, sh = g["screen"]
    lo = np.zeros((dy, dx, dz, 3), D)
    hi = np.zeros((dy, dx, dz, 3), D)
    X, Y, Zi = np.meshgrid(np.arange(dx, dtype=D), np.arange(dy, dtype=D), np.arange(dz, dtype=D), indexing="xy")

    def screen_to_ndc(px, py):
        return np.stack([px / sw * 2.0 - 1.0, (1.0 - py / sh) * 2.0 - 1.0], -1)
    pmin, pmax = screen_to_ndc(X * tsx, Y * tsy), screen_to_ndc((X + 1) * tsx, (Y + 1) * tsy)
    if g["ortho"]:
        a = view_of_ndc(g, pmin, np.zeros_like(X))
        b = view_of_ndc(g, pmax, np.zeros_like(X))
        a[..., 2] = -g["near"] + (g["near"] - g["far"]) * Zi / dz
        b[..., 2] = -g["near"] + (g["near"] - g["far"]) * (Zi + 1) / dz
        lo, hi = np.minimum(a, b), np.maximum(a, b)
    else:
        ratio = g["far"] / g["near"]
        with np.errstate(invalid="ignore", divide="ignore"):
            near_d = np.where(Zi == 0, 0.0, g["near"] * ratio ** ((Zi - 1) / max(dz - 1, 1)))
            far_d = g["far"] * np.ones_like(Zi) if dz == 1 else g["near"] * ratio ** (Zi / (dz - 1))
        corners = [view_of_ndc(g, p, d) for p in (pmin, pmax) for d in (near_d, far_d)]
        lo, hi = np.minimum.reduce(corners), np.maximum.reduce(corners)
    return lo.reshape(-1, 3), hi.reshape(-1, 3)


def check_z_slab(g, pos_range, offsets, indices, margin=1e-4):
    """Every listed pair: the sphere's depth interval overlaps the slab of the cluster's z slice.  This part of the
    refinement is exact in the reference -- a slice other than the centre's is kept only if the sphere reaches the
    nearer boundary plane (project_to_plane_z, assign.rs:606-624,1094-1113) -- so it is a necessary condition with no
    slack beyond rounding.  (x / y are conservative by construction: a one-column box is emitted without a plane test.)"""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    depth = -(c4 @ g["vfw"].T)[:, 2]
    radius = pr[:, 3] * g["scale_max"]
    bounds = slice_depths(g)
    obj, cl = expand_lists(np.asarray(offsets), np.asarray(indices))
    _, _, cz = cluster_coords(g, cl)
    lo_s, hi_s = bounds[cz], bounds[cz + 1]
    if not g["ortho"]:
        hi_s = np.where(cz == g["dims"][2] - 1, np.inf, hi_s)  # the last slice is open-ended (view_z_to_z_slice clamps)
    tol = margin * (1.0 + np.abs(depth[obj]) + radius[obj])
    bad = (depth[obj] + radius[obj] < lo_s - tol) | (depth[obj] - radius[obj] > hi_s + tol)
    assert not bad.any(), (f"{int(bad.sum())} listed pairs lie in a z slice the sphere does not reach, first object "
                           f"{int(obj[bad][0])} cluster {int(cl[bad][0])}")


def aabb_looseness(g, pos_range, offsets, indices):
    """Diagnostic, not an invariant: max over the listed pairs of distance(sphere centre, cluster AABB of
    compute_aabb_for_cluster) / radius.  <= 1 would mean every listed cluster's AABB touches the sphere; the reference's
    refinement is looser than that by design (e.g. a one-column box skips the x test), so this is only reported."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    cv = (c4 @ g["vfw"].T)[:, :3]
    radius = pr[:, 3] * g["scale_max"]
    lo, hi = cluster_aabbs(g)
    obj, cl = expand_lists(np.asarray(offsets), np.asarray(indices))
    if not len(obj):
        return 0.0
    p = cv[obj]
    dist = np.sqrt(((p - np.clip(p, lo[cl], hi[cl])) ** 2).sum(axis=1))
    return float((dist / np.maximum(radius[obj], 1e-30)).max())


def expected_farthest_z(g, pos_range, layer_mask, margin=1e-4, visible=None):
    """-> (lower, upper) bounds of last_frame_farthest_z: objects surely past / possibly past the two early-outs."""
    pr = np.asarray(pos_range, D).reshape(-1, 4)
    surely_in, surely_out = passes_early_outs(g, pos_range, layer_mask, margin)
    if visible is not None:
        vis = np.asarray(visible, bool)
        surely_in, surely_out = surely_in & vis, surely_out | ~vis
    c4 = np.concatenate([pr[:, :3], np.ones((len(pr), 1), D)], axis=1)
    fz = -(c4 @ g["vfw"].T)[:, 2] + pr[:, 3] * g["scale"][2]
    lo = max(0.0, float(fz[surely_in].max())) if surely_in.any() else 0.0
    maybe = ~surely_out
    hi = max(0.0, float(fz[maybe].max())) if maybe.any() else 0.0
    return lo, hi


def check_all(view, pos_range, obj_type, layer_mask, offsets, indices, counts, farthest_z, total, visible=None,
              superset=True, per_axis=2):
    """Runs every invariant; returns a small report (how much each check actually exercised)."""
    g = view_geometry(view)
    n_clusters = g["dims"][0] * g["dims"][1] * g["dims"][2]
    check_structure(n_clusters, offsets, indices, counts, total, obj_type)
    surely_in, listed = check_exclusion_and_box(g, pos_range, layer_mask, offsets, indices, visible)
    n_centre = check_centre(g, pos_range, obj_type, offsets, indices, surely_in, visible)
    n_hits = check_superset(g, pos_range, obj_type, offsets, indices, surely_in, visible, per_axis=per_axis) if superset else 0
    check_z_slab(g, pos_range, offsets, indices)
    worst = aabb_looseness(g, pos_range, offsets, indices)
    lo, hi = expected_farthest_z(g, pos_range, layer_mask, visible=visible)
    tol = 1e-5 * (1.0 + abs(hi))
    assert lo - tol <= farthest_z <= hi + tol, f"farthest_z {farthest_z} outside [{lo}, {hi}]"
    return dict(listed_objects=int(listed.sum()), pairs=int(total), centre_checked=n_centre, superset_hits=n_hits,
                aabb_distance_over_radius=worst)
