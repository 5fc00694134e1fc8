"""Synthetic inputs shaped like the reference's stress tests.  Pure numpy, seeded,
no reference files are read.  These only GENERATE columns; all computation on them happens in the
HIP library (or, in tests, in the oracle).

  many_cubes    examples/stress_tests/many_cubes.rs:192-212,574-587   Fibonacci sphere of unit cubes
  many_lights   examples/stress_tests/many_lights.rs:48-51,71-126     100k point lights on a shell
  gen_tree      examples/stress_tests/transform_hierarchy.rs:440-453  uniform b-ary tree, BFS parents
"""
import math

import numpy as np

F = np.float32
NO_PARENT = 0xFFFFFFFF
EPSILON = 0.36


def splitmix64(seed, n, start=0):
    """pseudo-random uint64 number start..start+n of the splitmix64(seed) sequence (vectorised, so any
    slice of a big scene can be generated on its own)."""
    idx = (np.arange(start + 1, start + n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed))
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, n, start=0):
    return (splitmix64(seed, n, start) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def random_unit_quats(seed, n, start=0):
    """Uniform on S^3 (Marsaglia / Shoemake), normalised in f64 then cast -> from_quat's precondition holds."""
    u1, u2, u3 = uniform01(seed, n, start), uniform01(seed + 1, n, start), uniform01(seed + 2, n, start)
    q = np.stack([np.sqrt(1 - u1) * np.sin(2 * np.pi * u2), np.sqrt(1 - u1) * np.cos(2 * np.pi * u2),
                  np.sqrt(u1) * np.sin(2 * np.pi * u3), np.sqrt(u1) * np.cos(2 * np.pi * u3)], axis=1)
    return q.astype(F)


def fibonacci_sphere(n, radius, f32_acos=False, start=0, count=None):
    """many_cubes.rs:574-587 (f64) / many_lights.rs:120-126 (acos evaluated in f32).  Points start..start+count
    of the n-point spiral."""
    count = n - start if count is None else count
    i = np.arange(start, start + count, dtype=np.float64)
    golden = 0.5 * (1.0 + math.sqrt(5.0))
    theta = 2.0 * np.pi * (i / golden)
    arg = 1.0 - 2.0 * (i + EPSILON) / (n - 1.0 + 2.0 * EPSILON)
    phi = np.arccos(arg.astype(F)).astype(np.float64) if f32_acos else np.arccos(arg)
    p = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    return (radius * p).astype(F)


def many_cubes(n, radius=500.0, seed=42, ragged_flags=False, start=0, count=None):
    """Flat scene: n unit cubes on a sphere (rows start..start+count of it).  Returns a dict of contiguous columns."""
    total = n
    count = total - start if count is None else count
    t = fibonacci_sphere(total, radius, start=start, count=count)
    r = random_unit_quats(seed, count, start)
    n = count
    s = np.ones((n, 3), F)
    c = np.zeros((n, 3), F)
    h = np.full((n, 3), 0.5, F)
    flags = np.full(n, 0x01 | 0x04, np.uint8)  # InheritedVisibility | HAS_AABB
    layers = np.ones(n, np.uint32)
    if ragged_flags:  # exercise every branch of the visibility closure
        rnd = splitmix64(seed + 7, n)
        flags[(rnd % np.uint64(17)) == 0] &= ~np.uint8(0x01)              # hidden by inheritance
        flags[(rnd % np.uint64(19)) == 1] |= np.uint8(0x02)               # NoFrustumCulling
        sph = (rnd % np.uint64(23)) == 2
        flags[sph] = (flags[sph] & ~np.uint8(0x04)) | np.uint8(0x08)      # Sphere instead of Aabb
        c[sph] = t[sph]
        h[sph, 0] = 0.75
        flags[(rnd % np.uint64(29)) == 3] &= ~np.uint8(0x04)              # neither Aabb nor Sphere
        flags[(rnd % np.uint64(31)) == 4] |= np.uint8(0x10)               # NoCpuCulling
        layers[(rnd % np.uint64(13)) == 5] = 2                            # on render layer 1 only
        layers[(rnd % np.uint64(37)) == 6] = 3
        s[:, :] = (0.5 + uniform01(seed + 9, 3 * n).reshape(n, 3)).astype(F)
        c[~sph] = (uniform01(seed + 11, 3 * n).reshape(n, 3)[~sph] - 0.5).astype(F)
    return dict(n=n, translation=np.ascontiguousarray(t).reshape(-1), rotation=np.ascontiguousarray(r).reshape(-1),
                scale=np.ascontiguousarray(s).reshape(-1), aabb_center=np.ascontiguousarray(c).reshape(-1),
                aabb_half=np.ascontiguousarray(h).reshape(-1), flags=flags, layers=layers)


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], np.float64)


def quat_axis(axis, angle):
    q = np.zeros(4, np.float64)
    q["xyz".index(axis)] = math.sin(angle * 0.5)
    q[3] = math.cos(angle * 0.5)
    return q


def affine_from_quat_translation(q, t):
    """col-major 3x4 (x_axis,y_axis,z_axis,translation) of a rotation+translation, f32."""
    x, y, z, w = [float(v) for v in q]
    m = np.array([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y),
                  2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x),
                  2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y),
                  t[0], t[1], t[2]], np.float64)
    return m.astype(F)


def many_cubes_camera(frame=0, yaw=0.0, position=(0.0, 0.0, 0.0)):
    """Camera of many_cubes.rs:590-603: at the origin, rotate_z(d) then rotate_x(d) every frame (d = 0.15/60),
    plus an optional yaw (config 4: four cameras at 0/90/180/270 deg)."""
    d = 0.15 / 60.0
    q = quat_axis("y", yaw)
    for _ in range(frame):
        q = quat_mul(quat_axis("z", d), q)
        q = quat_mul(quat_axis("x", d), q)
    q = q / np.linalg.norm(q)
    return affine_from_quat_translation(q, position)


CAMERA_FOV = math.pi / 4.0
CAMERA_ASPECT = 16.0 / 9.0
CAMERA_NEAR = 0.1
CAMERA_FAR = 1000.0


def many_lights(n=100_000, radius=50.0, light_range=0.3):
    p = fibonacci_sphere(n, radius, f32_acos=True)
    pos_range = np.concatenate([p, np.full((n, 1), light_range, F)], axis=1)
    return np.ascontiguousarray(pos_range).reshape(-1)


def concat_scenes(*scenes):
    """Row-wise concatenation of scene dicts (same columns as many_cubes)."""
    out = dict(n=sum(sc["n"] for sc in scenes))
    for k in ("translation", "rotation", "scale", "aabb_center", "aabb_half", "flags", "layers"):
        out[k] = np.ascontiguousarray(np.concatenate([sc[k] for sc in scenes]))
    return out


def light_rows(pos_range, seed=77):
    """Point lights as ROWS of a scene: every light is an entity with a Transform, a world-space bounding Sphere
    (update_point_light_bounding_spheres, crates/bevy_light/src/point_light.rs:195-208: centre = translation, radius =
    range) and a ViewVisibility like any other entity; check_visibility culls it through the Sphere branch
    (visibility/mod.rs:838-843) and assign_objects_to_clusters gathers the visible ones (assign.rs:190-215)."""
    pr = np.asarray(pos_range, F).reshape(-1, 4)
    n = len(pr)
    h = np.zeros((n, 3), F)
    h[:, 0] = pr[:, 3]
    return dict(n=n, translation=np.ascontiguousarray(pr[:, :3]).reshape(-1), rotation=np.ascontiguousarray(random_unit_quats(seed, n)).reshape(-1),
                scale=np.ones(3 * n, F), aabb_center=np.ascontiguousarray(pr[:, :3]).reshape(-1), aabb_half=h.reshape(-1),
                flags=np.full(n, 0x01 | 0x08, np.uint8), layers=np.ones(n, np.uint32))


def frame_scene(n_entities=1_000_000, n_lights=100_000, n_meshes=10_000, light_range=0.3, ragged_flags=False):
    """The scene of BASELINE.json's metric -- "propagate + cull + cluster at 1M entities" -- in ONE context: configs[1]'s
    many_cubes sphere (n_entities unit cubes, R = 500) followed by configs[2]'s many_lights set (n_meshes cubes on
    R = 40, n_lights point lights on R = 50, range 0.3).  All three share the camera at the origin.  Returns
    (scene, first_light_row, pos_range of the lights)."""
    cubes = many_cubes(n_entities, ragged_flags=ragged_flags)
    meshes = many_cubes(max(n_meshes, 1), radius=40.0, seed=43)
    if n_meshes == 0:
        meshes = {k: (v[:0] if k != "n" else 0) for k, v in meshes.items()}
    pr = many_lights(n_lights, 50.0, light_range)
    sc = concat_scenes(cubes, meshes, light_rows(pr))
    return sc, n_entities + n_meshes, pr


def gen_tree(depth, branch, max_nodes=None, seed=42):
    """Uniform tree (transform_hierarchy.rs:440-453): node i>0 has parent (i-1)//branch; rows are already in
    level (BFS) order.  Local transforms: translation on a radius-32 circle (:266-269,416-422) plus a seeded
    small rotation and a uniform scale in [0.9,1.1] so the chain product is non-trivial."""
    total = sum(branch ** i for i in range(depth))
    n = total if max_nodes is None else min(total, max_nodes)
    idx = np.arange(n, dtype=np.int64)
    parent = np.where(idx == 0, NO_PARENT, (idx - 1) // branch).astype(np.uint32)
    level_offsets = [0]
    acc, k = 0, 0
    while acc < n:
        acc = min(n, acc + branch ** k)
        level_offsets.append(acc)
        k += 1
    slot = np.where(idx == 0, 0, (idx - 1) % branch).astype(np.float64)
    a = slot / branch
    t = np.stack([32.0 * np.cos(a), 32.0 * np.sin(a), np.zeros(n)], axis=1).astype(F)
    small = (uniform01(seed, 3 * n).reshape(n, 3) - 0.5) * 0.4
    q = np.concatenate([small, np.ones((n, 1))], axis=1)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    sc = (0.9 + 0.2 * uniform01(seed + 5, n)).astype(F)
    s = np.repeat(sc[:, None], 3, axis=1)
    return dict(n=n, parent=parent, level_offsets=np.array(level_offsets, np.uint32),
                translation=np.ascontiguousarray(t).reshape(-1), rotation=np.ascontiguousarray(q).reshape(-1),
                scale=np.ascontiguousarray(s).reshape(-1))


# ---- the reference's own hierarchy stress shapes (examples/stress_tests/transform_hierarchy.rs:29-160) -------------------------------

# HUMANOID_RIG (transform_hierarchy.rs:493-561): parent map of a mixamo-like rig, root excluded -- 68 nodes per rig.  Restated as
# structure: spine + head, two arms of (shoulder, arm, forearm, hand, 5 fingers x 4 joints), two legs of 5 joints.
def _humanoid_rig():
    rig = [0, 1, 2, 3, 4, 5, 6, 6, 6]          # hips, spine x3, neck, head, head top + two eyes (nodes 1..9)
    for _side in range(2):                       # shoulder (child of spine 2 = node 4), arm, forearm, hand, then five fingers
        shoulder = len(rig) + 1
        rig += [4, shoulder, shoulder + 1, shoulder + 2]
        hand = shoulder + 3
        for _finger in range(5):
            first = len(rig) + 1
            rig += [hand, first, first + 1, first + 2]
    for _side in range(2):                       # upper leg (child of hips = node 1), leg, foot, toe base, toe end
        upper = len(rig) + 1
        rig += [1, upper, upper + 1, upper + 2, upper + 3]
    return rig


HUMANOID_RIG = _humanoid_rig()
assert len(HUMANOID_RIG) == 67

# name -> (test case, update filter (probability, min_depth, max_depth)), transform_hierarchy.rs:29-160
HIERARCHY_SHAPES = {
    "large_tree": (("non_uniform", 18, 8), (0.5, 0, None)),
    "wide_tree": (("tree", 3, 500), (0.5, 0, None)),
    "deep_tree": (("non_uniform", 25, 2), (0.5, 0, None)),
    "chain": (("tree", 2500, 1), (0.5, 0, None)),
    "update_leaves": (("tree", 18, 2), (0.5, 17, None)),
    "update_shallow": (("tree", 18, 2), (0.5, 0, 8)),
    "humanoids_active": (("humanoids", 4000, 0), (1.0, 0, None)),
    "humanoids_inactive": (("humanoids", 10, 3990), (1.0, 0, None)),
    "humanoids_mixed": (("humanoids", 2000, 2000), (1.0, 0, None)),
    # SURVEY 8(d) config 5's other data points: the full 11-level 4-ary tree and a true depth-12 one
    "tree_4ary_depth11": (("tree", 11, 4), (0.5, 0, None)),
    "tree_4ary_depth12": (("tree", 12, 4), (0.5, 0, None)),
}


# not the reference's: hierarchies as narrow as `chain` but with several nodes to a level (ropes, a handful of rigs) -- every level at
# most a wave wide, which is what the one-wave kernel takes (kernels_tree.hip, k_propagate_narrow)
NARROW_SHAPES = {
    "ropes": (("narrow", 300, 64), (0.3, 0, None)),
    "ropes_few_movers": (("narrow", 90, 17), (0.02, 0, None)),
    "strands": (("narrow", 120, 9), (0.4, 0, None)),      # <= 16 to a level: a node per quad of lanes
    "bundle": (("bundle", 70, 12), (0.1, 0, None)),       # 12 chains side by side under one root: every parent sits where its child does
    "bundle_wide": (("bundle", 40, 50), (0.1, 0, None)),
}


def _parent_map_narrow(n_levels, max_width, rng):
    """Level l holds 1..max_width nodes (level 0: the single root), each under a seeded parent of level l - 1."""
    widths = np.concatenate([[1], rng.integers(1, max_width + 1, n_levels - 1)])
    starts = np.concatenate([[0], np.cumsum(widths)])
    pm = [starts[l - 1] + np.sort(rng.integers(0, widths[l - 1], widths[l])) for l in range(1, n_levels)]
    return np.concatenate(pm).astype(np.int64)


def _parent_map_bundle(n_levels, n_ropes):
    """A root, n_ropes children, and under each a chain down to level n_levels - 1."""
    first = np.zeros(n_ropes, np.int64)
    rest = (1 + np.arange((n_levels - 2) * n_ropes, dtype=np.int64))
    return np.concatenate([first, rest])


def _parent_map_tree(depth, branch):
    """gen_tree (transform_hierarchy.rs:440-453): 0,0,..,1,1,.. -- every one of the first sum(branch^i, i < depth-1) nodes has `branch` children."""
    inner = sum(branch ** i for i in range(depth - 1))
    return np.repeat(np.arange(inner, dtype=np.int64), branch)


def _parent_map_non_uniform(max_depth, max_branch):
    """gen_non_uniform_tree (transform_hierarchy.rs:455-490): child k of a node gets a subtree one level shallower than child k - 1;
    the recursion returns as soon as the remaining depth reaches zero.  Iterative restatement (an explicit stack of loop states)."""
    tree = []
    stack = [[0, max_depth, 0]]  # parent, curr_depth (decremented per sibling), siblings pushed so far
    while stack:
        frame = stack[-1]
        if frame[2] == max_branch:
            stack.pop()
            continue
        tree.append(frame[0])
        frame[2] += 1
        frame[1] -= 1
        if frame[1] == 0:
            stack.pop()       # `return`: the remaining siblings of this node are never pushed
            continue
        stack.append([len(tree), frame[1], 0])
    return np.array(tree, np.int64)


def level_order(parent):
    """Rows of an arbitrary forest in level (BFS) order, the order mi_upload_hierarchy asks for: level l = the children of level l - 1's
    rows, parent by parent, siblings in the caller's order (what mi_hierarchy_sort computes; restated here in numpy so that the harness
    does not need the library to build a scene).  Returns (new_to_old, parent in new numbering, level_offsets)."""
    parent = np.asarray(parent, np.int64)
    n = len(parent)
    has = parent != NO_PARENT
    kids_all = np.nonzero(has)[0]
    order = np.argsort(parent[kids_all], kind="stable")
    kids = kids_all[order]
    counts = np.bincount(parent[kids_all], minlength=n).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)])
    level = np.nonzero(~has)[0]
    out, offs = [level], [0, len(level)]
    while True:
        cnt = counts[level]
        tot = int(cnt.sum())
        if tot == 0:
            break
        first = np.repeat(starts[level], cnt)
        within = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        level = kids[first + within]
        out.append(level)
        offs.append(offs[-1] + tot)
    new_to_old = np.concatenate(out)
    assert len(new_to_old) == n, "cycle or dangling parent"
    old_to_new = np.empty(n, np.int64)
    old_to_new[new_to_old] = np.arange(n)
    p_new = np.where(parent[new_to_old] == NO_PARENT, NO_PARENT, old_to_new[np.where(has, parent, 0)][new_to_old])
    return new_to_old.astype(np.uint32), p_new.astype(np.uint32), np.array(offs, np.uint32)


def hierarchy_shape(name, seed=42, plain_transforms=False, scale_rigs=1.0):
    """One of the reference's hierarchy stress configurations (transform_hierarchy.rs:29-160) as level-ordered rows.
      structure   gen_tree / gen_non_uniform_tree / HUMANOID_RIG x (active + inactive), spawn_tree's placement: a child's translation is
                  (32 cos a, 32 sin a, 0), a = its slot / its parent's child count (:395-422); a rig's root at a seeded (x, y) in
                  [-250, 250)^2 (:291-325; the reference draws from rand::rng(), unseeded)
      movers      the nodes that carry UpdateValue: every non-root node whose depth passes the filter, with the filter's probability
                  (drawn per node at insertion, :398-405; seeded here), never a rig of the inactive group
      mover_translation(frame)   what the `update` system leaves in the movers' Transforms in frame k (:250-262: a += dt * 0.1, dt = 1/60)
    plain_transforms = identity rotation and unit scale like the reference; otherwise the seeded small rotation and uniform scale of
    gen_tree() above so that the products along a chain are not trivially exact."""
    (kind, a, b), (prob, min_depth, max_depth) = HIERARCHY_SHAPES[name] if name in HIERARCHY_SHAPES else NARROW_SHAPES[name]
    rng = np.random.default_rng(seed)
    if kind == "humanoids":
        n_rigs = int(round((a + b) * scale_rigs))
        n_active = int(round(a * scale_rigs)) if b else n_rigs
        rig = np.array(HUMANOID_RIG, np.int64)
        per = len(rig) + 1
        base = np.arange(n_rigs, dtype=np.int64)[:, None] * per
        parent = np.full((n_rigs, per), NO_PARENT, np.int64)
        parent[:, 1:] = base + rig[None, :]
        parent = parent.reshape(-1)
        rig_active = np.repeat(np.arange(n_rigs) < n_active, per)
        root_xy = rng.random((n_rigs, 2)) * 500.0 - 250.0
    else:
        pm = _parent_map_tree(a, b) if kind == "tree" else _parent_map_narrow(a, b, rng) if kind == "narrow" else _parent_map_bundle(a, b) if kind == "bundle" else _parent_map_non_uniform(a, b)
        parent = np.concatenate([[NO_PARENT], pm]).astype(np.int64)
        rig_active = np.ones(len(parent), bool)
        root_xy = None
    n = len(parent)
    # spawn_tree: slot / child_count per node (in spawn order), depth
    has = parent != NO_PARENT
    child_count = np.bincount(parent[has], minlength=n)
    order = np.argsort(parent[has], kind="stable")
    kids = np.nonzero(has)[0][order]
    starts = np.concatenate([[0], np.cumsum(child_count)])
    slot = np.zeros(n, np.int64)
    slot[kids] = np.arange(len(kids)) - np.repeat(starts[:-1], child_count)
    sep = np.where(has, slot / np.maximum(child_count[np.where(has, parent, 0)], 1), 0.0)
    new_to_old, p_new, offs = level_order(parent)
    depth_new = np.repeat(np.arange(len(offs) - 1), np.diff(offs.astype(np.int64)))
    sep_new = sep[new_to_old]
    t = np.stack([32.0 * np.cos(sep_new), 32.0 * np.sin(sep_new), np.zeros(n)], axis=1)
    is_root = p_new == NO_PARENT
    t[is_root] = 0.0
    if root_xy is not None:
        t[is_root, :2] = root_xy[(new_to_old[is_root] // (len(HUMANOID_RIG) + 1))]
    draw = rng.random(n)
    mover = (~is_root) & rig_active[new_to_old] & (draw <= prob) & (depth_new >= min_depth) & (depth_new <= (max_depth if max_depth is not None else 1 << 30))
    movers = np.nonzero(mover)[0].astype(np.uint32)
    if plain_transforms:
        q = np.tile(np.array([0, 0, 0, 1], F), (n, 1))
        s3 = np.ones((n, 3), F)
    else:
        small = (uniform01(seed, 3 * n).reshape(n, 3) - 0.5) * 0.4
        q = np.concatenate([small, np.ones((n, 1))], axis=1)
        q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
        s3 = np.repeat((0.95 + 0.1 * uniform01(seed + 5, n)).astype(F)[:, None], 3, axis=1)
    sep_movers = sep_new[movers]

    def mover_translation(frame):
        ang = (sep_movers + np.float64(frame) * (0.1 / 60.0)).astype(F)  # UpdateValue is an f32 the system adds to each frame
        return np.ascontiguousarray(np.stack([np.cos(ang) * F(32.0), np.sin(ang) * F(32.0), np.zeros(len(ang), F)], axis=1).astype(F)).reshape(-1)

    return dict(name=name, n=n, parent=p_new, level_offsets=offs, translation=np.ascontiguousarray(t.astype(F)).reshape(-1),
                rotation=np.ascontiguousarray(q).reshape(-1), scale=np.ascontiguousarray(s3).reshape(-1), movers=movers,
                mover_translation=mover_translation, depth=int(len(offs) - 2), n_levels=int(len(offs) - 1))


def batching_scene(n_rows, n_sets=7, max_bins=40, seed=42, unbatched_fraction=0.05):
    """Synthetic render-phase binning for `n_rows` mesh rows: every row names a batch set (or
    NO_BATCH_SET = not multidrawable), a RenderBinIndex inside it and an InputUniformIndex; every set has a
    RenderBinIndex -> metadata-index table with holes and a GpuBinMetadata array whose indirect_parameters_offset is a
    permutation (bins are created and destroyed over time in the reference, render_phase/mod.rs:268-400)."""
    r = splitmix64(seed, 3 * n_rows + 4 * n_sets + 16)
    rng = np.random.default_rng(int(r[0] & 0xFFFFFFFF))
    bins_per_set = 1 + (r[1:1 + n_sets] % np.uint64(max_bins)).astype(np.int64)
    if n_sets > 2:
        bins_per_set[1] = max(1, max_bins * 8)  # one set with more than 256 bins: the two-level scan
    set_indexed = ((r[1 + n_sets:1 + 2 * n_sets] >> np.uint64(7)) & np.uint64(1)).astype(np.uint8)
    bin_table_offset = np.zeros(n_sets + 1, np.uint32)
    meta_offset = np.zeros(n_sets + 1, np.uint32)
    tables, metas = [], []
    for s in range(n_sets):
        b = int(bins_per_set[s])
        slots = b + int(rng.integers(0, 4))  # holes in the RenderBinIndex space
        live = np.sort(rng.choice(slots, b, replace=False)).astype(np.uint32)
        table = np.full(slots, 0xFFFFFFFF, np.uint32)
        meta_of_bin = rng.permutation(b).astype(np.uint32)
        table[live] = meta_of_bin
        meta = np.zeros((b, 3), np.uint32)
        meta[meta_of_bin, 1] = live                      # bin_index: the reverse map
        meta[:, 0] = rng.permutation(b).astype(np.uint32)  # indirect_parameters_offset, relative to the set
        meta[:, 2] = 0xABCD                               # stale instance_count: must be overwritten
        tables.append(table)
        metas.append(meta)
        bin_table_offset[s + 1] = bin_table_offset[s] + slots
        meta_offset[s + 1] = meta_offset[s] + b
    row_set = (r[16 + 2 * n_sets:16 + 2 * n_sets + n_rows] % np.uint64(max(n_sets, 1))).astype(np.uint32)
    pick = r[16 + 2 * n_sets + n_rows:16 + 2 * n_sets + 2 * n_rows]
    row_bin = np.zeros(n_rows, np.uint32)
    for s in range(n_sets):
        live = np.nonzero(tables[s] != 0xFFFFFFFF)[0].astype(np.uint32)
        m = row_set == s
        # skewed: most rows of a set share a few bins (instances of the same mesh + material)
        k = (pick[m] % np.uint64(len(live))).astype(np.int64)
        k = np.minimum(k, (pick[m] >> np.uint64(32)) % np.uint64(len(live))).astype(np.int64)
        row_bin[m] = live[k]
    unb = uniform01(seed + 1, n_rows) < unbatched_fraction
    row_set[unb] = 0xFFFFFFFF
    row_input = rng.permutation(n_rows).astype(np.uint32)
    return dict(row_set=row_set, row_bin=row_bin, row_input=row_input, set_indexed=set_indexed,
                bin_table_offset=bin_table_offset, bin_table=np.concatenate(tables) if tables else np.zeros(0, np.uint32),
                meta_offset=meta_offset, bin_metadata=np.concatenate(metas) if metas else np.zeros((0, 3), np.uint32))


def phase_scene(n_rows, n_sets=7, max_bins=40, n_unbatchable_bins=5, n_batchable_bins=6, seed=42, cpu_fraction=0.15, no_input_fraction=0.1):
    """batching_scene plus the CPU-built part of the same binned phase (gpu_preprocessing.rs:2135-2357): a share of the rows
    sits in unbatchable bins (some of them without an input uniform index: get_binned_index() == None) or in
    batchable-but-not-multidrawable bins; bins are numbered in the phase's sorted key order and carry their mesh class."""
    sc = batching_scene(n_rows, n_sets=n_sets, max_bins=max_bins, seed=seed, unbatched_fraction=0.03)
    r = splitmix64(seed + 101, 3 * n_rows + n_unbatchable_bins + n_batchable_bins + 8)
    u = uniform01(seed + 102, n_rows)
    kind = np.zeros(n_rows, np.uint8)
    cpu_bin = np.zeros(n_rows, np.uint32)
    if n_unbatchable_bins:
        m = u < cpu_fraction * 0.5
        kind[m] = 2
        cpu_bin[m] = (r[:n_rows][m] % np.uint64(n_unbatchable_bins)).astype(np.uint32)
    if n_batchable_bins:
        m = (u >= cpu_fraction * 0.5) & (u < cpu_fraction)
        kind[m] = 1
        cpu_bin[m] = (r[n_rows:2 * n_rows][m] % np.uint64(n_batchable_bins)).astype(np.uint32)
    row_input = sc["row_input"].copy()
    no_input = (kind == 2) & (uniform01(seed + 103, n_rows) < no_input_fraction)
    row_input[no_input] = 0xFFFFFFFF
    sc.update(row_kind=kind, row_cpu_bin=cpu_bin, row_input=row_input,
              unbatchable_indexed=((r[3 * n_rows:3 * n_rows + n_unbatchable_bins] >> np.uint64(9)) & np.uint64(1)).astype(np.uint8),
              batchable_indexed=((r[3 * n_rows + n_unbatchable_bins:3 * n_rows + n_unbatchable_bins + n_batchable_bins] >> np.uint64(9))
                                 & np.uint64(1)).astype(np.uint8))
    return sc


def sorted_items(n_items, seed=42, run=6, n_set_keys=5, n_bin_keys=4, no_input_fraction=0.04, no_meta_fraction=0.06):
    """A sorted phase's items u32[n,4] = (input_index, batch_set_key, bin_key, flags): runs of equal batch-set keys with shorter
    runs of equal bin keys inside (what a depth-sorted transparent phase with instanced meshes looks like), a few items that
    are not part of the pipeline (no input index) and a few without compare data."""
    r = splitmix64(seed + 7, 4 * n_items + 8)
    items = np.zeros((n_items, 4), np.uint32)
    items[:, 0] = (r[:n_items] & np.uint64(0xFFFFF)).astype(np.uint32)
    block = np.arange(n_items) // max(run, 1)
    sub = np.arange(n_items) // max(run // 3, 1)
    sk = splitmix64(seed + 8, int(block.max()) + 2 if n_items else 1)
    bk = splitmix64(seed + 9, int(sub.max()) + 2 if n_items else 1)
    if n_items:
        items[:, 1] = (sk[block] % np.uint64(n_set_keys)).astype(np.uint32)
        items[:, 2] = (bk[sub] % np.uint64(n_bin_keys)).astype(np.uint32)
        indexed = ((sk[block] >> np.uint64(11)) & np.uint64(1)).astype(np.uint32)  # the mesh class follows the batch set
        u = uniform01(seed + 10, n_items)
        has_meta = (uniform01(seed + 11, n_items) >= no_meta_fraction).astype(np.uint32)
        items[:, 3] = indexed | (has_meta << 1)
        items[u < no_input_fraction, 0] = 0xFFFFFFFF
    return items
