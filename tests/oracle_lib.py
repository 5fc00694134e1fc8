"""ctypes binding of the CPU oracle (oracle/libbevy_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (bevy_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libbevy_oracle.so")

FLAG_INHERITED_VISIBLE = 0x01
FLAG_NO_FRUSTUM_CULLING = 0x02
FLAG_HAS_AABB = 0x04
FLAG_HAS_SPHERE = 0x08
FLAG_NO_CPU_CULLING = 0x10
FLAG_HAS_VISIBILITY_RANGE = 0x20
FLAG_RANGE_USE_AABB = 0x40
FLAG_SHADOW_CASTER = 0x80
VIEW_FLAG_NO_CPU_CULLING = 0x01
VIEW_FLAG_SHADOW = 0x02
VIEW_FLAG_SKIP_NEAR = 0x04
VIEW_FLAG_TEST_FAR = 0x08
VIEW_FLAG_LIGHT_SPHERE = 0x10
VIEW_FLAG_RANGES = 0x20
VIEW_FLAG_RANGES_NO_ORIGIN = 0x40
NO_PARENT = 0xFFFFFFFF
MAX_CLUSTER_DIM = 4096


def build():
    """(Re)build the oracle .so if it is missing or older than its sources."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("bevy_oracle.c", "bevy_oracle.h", "Makefile")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return _SO


# ---- parity-margin variants (tests/test_parity_margin.py): the same oracle with ONE of glam's lane orders swapped for the order a
# wrong memory of glam 0.33.2 would give (the ORC_* switches at the top of oracle/bevy_oracle.c), or with every a*b+c fused.
PARITY_VARIANTS = {
    "dot4_left_to_right": ["-DORC_DOT4_LEFT_TO_RIGHT"],
    "dot3_pairwise": ["-DORC_DOT3_PAIRWISE"],
    "dot3_x_yz": ["-DORC_DOT3_X_YZ"],
    "length_rsqrt": ["-DORC_LENGTH_RSQRT"],
    "mat3_columns_zyx": ["-DORC_MAT3_COLUMNS_ZYX"],
    "fma": ["-ffp-contract=fast", "-mfma"],
}
_BASE_CFLAGS = ["-O2", "-std=c99", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-msse2", "-mfpmath=sse", "-Wno-unused-function"]


def build_variant(name):
    """oracle/libbevy_oracle_<name>.so: bevy_oracle.c with PARITY_VARIANTS[name] appended to the oracle's own flags."""
    so = os.path.join(ORACLE_DIR, f"libbevy_oracle_{name}.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("bevy_oracle.c", "batching_oracle.c", "bevy_oracle.h")] + [os.path.abspath(__file__)]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.run(["gcc"] + _BASE_CFLAGS + PARITY_VARIANTS[name] + [os.path.join(ORACLE_DIR, "bevy_oracle.c"), os.path.join(ORACLE_DIR, "batching_oracle.c"),
                                                                      "-o", so, "-shared", "-lm", "-lpthread"], check=True, capture_output=True)
    return so


class variant:
    """with oracle_lib.variant("dot4_left_to_right"): every oracle_lib function runs the variant build; the oracle proper comes
    back on exit.  (Every binding goes through lib(), so swapping the handle is enough.)"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _lib
        self.saved = _lib
        _lib = None if self.name is None else _prepare(C.CDLL(build_variant(self.name)))
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


class ClusterView(C.Structure):
    _fields_ = [
        ("dims", C.c_uint32 * 3),
        ("tile_size", C.c_uint32 * 2),
        ("screen_size", C.c_uint32 * 2),
        ("is_orthographic", C.c_uint32),
        ("view_layer_mask", C.c_uint32),
        ("near_", C.c_float),
        ("far_", C.c_float),
        ("cluster_factors", C.c_float * 2),
        ("view_from_world", C.c_float * 16),
        ("clip_from_view", C.c_float * 16),
        ("view_from_clip", C.c_float * 16),
        ("view_from_world_scale", C.c_float * 3),
        ("view_from_world_scale_max", C.c_float),
        ("frustum", C.c_float * 24),
        ("n_x_planes", C.c_uint32),
        ("n_y_planes", C.c_uint32),
        ("n_z_planes", C.c_uint32),
        ("x_planes", C.c_float * ((MAX_CLUSTER_DIM + 1) * 4)),
        ("y_planes", C.c_float * ((MAX_CLUSTER_DIM + 1) * 4)),
        ("z_planes", C.c_float * ((MAX_CLUSTER_DIM + 1) * 4)),
        ("view_layer_mask_hi", C.c_uint32),
    ]


class ClusterConfig(C.Structure):
    """orc_cluster_config"""
    _fields_ = [("kind", C.c_uint32), ("dimensions", C.c_uint32 * 3), ("total", C.c_uint32), ("z_slices", C.c_uint32),
                ("first_slice_depth", C.c_float), ("far_z_mode", C.c_uint32), ("far_z_constant", C.c_float),
                ("dynamic_resizing", C.c_uint32)]


class View(C.Structure):
    """orc_view"""
    _fields_ = [("frustum", C.c_float * 24), ("layer_mask", C.c_uint32), ("flags", C.c_uint32),
                ("position", C.c_float * 3), ("light_sphere", C.c_float * 4)]


def make_views(frusta, layer_masks=None, flags=None, positions=None, light_spheres=None):
    """-> ctypes array of orc_view from per-view numpy columns."""
    fr = np.ascontiguousarray(frusta, np.float32).reshape(-1, 24)
    nv = len(fr)
    arr = (View * nv)()
    for v in range(nv):
        arr[v].frustum[:] = fr[v].tolist()
        arr[v].layer_mask = int(layer_masks[v]) if layer_masks is not None else 1
        arr[v].flags = int(flags[v]) if flags is not None else 0
        if positions is not None:
            arr[v].position[:] = np.asarray(positions, np.float32).reshape(-1, 3)[v].tolist()
        if light_spheres is not None:
            arr[v].light_sphere[:] = np.asarray(light_spheres, np.float32).reshape(-1, 4)[v].tolist()
    return arr


_lib = None


def _prepare(l):
    l.orc_radius_vec3a.restype = C.c_float
    l.orc_bench_flat_frame.restype = C.c_double
    l.orc_bench_flat_frame2.restype = C.c_double
    l.orc_assign_objects_to_clusters.restype = C.c_uint64
    l.orc_assign_objects_to_clusters_layers64.restype = C.c_uint64
    l.orc_visible_entities_sorted.restype = C.c_uint32
    return l


def lib():
    global _lib
    if _lib is None:
        _lib = _prepare(C.CDLL(build()))
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, ty):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ty))


def fp(a):
    return _p(a, C.c_float)


def u8p(a):
    return _p(a, C.c_uint8)


def u32p(a):
    return _p(a, C.c_uint32)


def u64p(a):
    return _p(a, C.c_uint64)


# ---- primitive wrappers ------------------------------------------------------------------

def transform_to_affine(t, r, s):
    t, tp = _f(t); r, rp = _f(r); s, sp = _f(s)
    out = np.zeros(12, np.float32)
    lib().orc_transform_to_affine(tp, rp, sp, fp(out))
    return out


def affine_mul(a, b):
    a, ap = _f(a); b, bp = _f(b)
    out = np.zeros(12, np.float32)
    lib().orc_affine_mul(ap, bp, fp(out))
    return out


def half_space_new(nd):
    nd, p = _f(nd)
    out = np.zeros(4, np.float32)
    lib().orc_half_space_new(p, fp(out))
    return out


def frustum_from_planes(planes):
    """HalfSpace::new applied to six raw Vec4s (the form the reference tests use)."""
    return np.concatenate([half_space_new(p) for p in planes]).astype(np.float32)


def compute_frustum_perspective(fov, aspect, near, far, camera_affine):
    cam, cp = _f(camera_affine)
    out = np.zeros(24, np.float32)
    lib().orc_compute_frustum_perspective(C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far), cp, fp(out))
    return out


def perspective_infinite_reverse(fov, aspect, near):
    out = np.zeros(16, np.float32)
    lib().orc_perspective_infinite_reverse(C.c_float(fov), C.c_float(aspect), C.c_float(near), fp(out))
    return out


def intersects_sphere(frustum, center, radius, intersect_far):
    fr, frp = _f(frustum); c, cp = _f(center)
    return bool(lib().orc_frustum_intersects_sphere(frp, cp, C.c_float(radius), int(intersect_far)))


def intersects_obb(frustum, center, half, wfl, near, far):
    fr, frp = _f(frustum); c, cp = _f(center); h, hp = _f(half); w, wp = _f(wfl)
    return bool(lib().orc_frustum_intersects_obb(frp, cp, hp, wp, int(near), int(far)))


def intersects_obb_identity(frustum, center, half):
    fr, frp = _f(frustum); c, cp = _f(center); h, hp = _f(half)
    return bool(lib().orc_frustum_intersects_obb_identity(frp, cp, hp))


def contains_aabb(frustum, center, half, wfl):
    fr, frp = _f(frustum); c, cp = _f(center); h, hp = _f(half); w, wp = _f(wfl)
    return bool(lib().orc_frustum_contains_aabb(frp, cp, hp, wp))


def is_in_half_space(center, half, hs, wfl):
    c, cp = _f(center); h, hp = _f(half); s, sp = _f(hs); w, wp = _f(wfl)
    return bool(lib().orc_aabb_is_in_half_space(cp, hp, sp, wp))


def is_in_half_space_identity(center, half, hs):
    c, cp = _f(center); h, hp = _f(half); s, sp = _f(hs)
    return bool(lib().orc_aabb_is_in_half_space_identity(cp, hp, sp))


def sphere_intersects_obb(sc, sr, center, half, wfl):
    s, sp = _f(sc); c, cp = _f(center); h, hp = _f(half); w, wp = _f(wfl)
    return bool(lib().orc_sphere_intersects_obb(sp, C.c_float(sr), cp, hp, wp))


# ---- systems -------------------------------------------------------------------------------

def sync_simple_transforms(t, r, s, dirty=None, global_in=None):
    n = len(t) // 3
    g = np.zeros(12 * n, np.float32) if global_in is None else global_in.copy()
    changed = np.zeros(n, np.uint8)
    lib().orc_sync_simple_transforms(n, fp(t), fp(r), fp(s), u8p(dirty), fp(g), u8p(changed))
    return g, changed


def mark_dirty_trees(parent, changed):
    n = len(parent)
    tc = np.zeros(n, np.uint8)
    lib().orc_mark_dirty_trees(n, u32p(parent), u8p(changed), u8p(tc))
    return tc


def propagate_transforms(parent, t, r, s, global_in=None, static_opt=False, tree_changed=None,
                         transform_changed=None):
    n = len(parent)
    g = np.zeros(12 * n, np.float32) if global_in is None else global_in.copy()
    changed = np.zeros(n, np.uint8)
    rc = lib().orc_propagate_transforms(n, u32p(parent), fp(t), fp(r), fp(s), int(static_opt),
                                        u8p(tree_changed), u8p(transform_changed), fp(g), u8p(changed))
    return rc, g, changed


def compute_global_transform(parent, t, r, s, row):
    out = np.zeros(12, np.float32)
    rc = lib().orc_compute_global_transform(len(parent), u32p(parent), fp(t), fp(r), fp(s), int(row), fp(out))
    assert rc == 0
    return out


def reset_view_visibility(flags, vv):
    vv = vv.copy()
    lib().orc_reset_view_visibility(len(vv), u8p(flags), u8p(vv))
    return vv


def check_visibility(g, c, h, flags, layers, vv, frusta, view_masks=None, view_flags=None, in_range=None):
    n = len(flags)
    nv = len(frusta) // 24
    vv = vv.copy()
    vis = np.zeros(nv * n, np.uint8)
    chg = np.zeros(n, np.uint8)
    lib().orc_check_visibility(n, fp(g), fp(c), fp(h), u8p(flags), u32p(layers), u8p(in_range), u8p(vv),
                               fp(frusta), u32p(view_masks), u8p(view_flags), nv, u8p(vis), u8p(chg))
    return vv, vis.reshape(nv, n), chg


def check_visibility_layers64(g, c, h, flags, layers, layers_hi, vv, frusta, view_masks, view_masks_hi):
    n = len(flags)
    nv = len(frusta) // 24
    vv = vv.copy()
    vis = np.zeros(nv * n, np.uint8)
    chg = np.zeros(n, np.uint8)
    lib().orc_check_visibility_layers64(n, fp(g), fp(c), fp(h), u8p(flags), u32p(layers), u32p(layers_hi), None, u8p(vv), fp(frusta),
                                        u32p(view_masks), u32p(view_masks_hi), None, nv, u8p(vis), u8p(chg))
    return vv, vis.reshape(nv, n), chg


def check_visibility_ranges(g, c, flags, range_start_end, view_positions):
    n = len(flags)
    vp = np.ascontiguousarray(view_positions, np.float32).reshape(-1)
    nv = len(vp) // 3
    out = np.zeros(nv * n, np.uint8)
    lib().orc_check_visibility_ranges(n, fp(g), fp(c), u8p(flags), fp(range_start_end), fp(vp), nv, u8p(out))
    return out.reshape(nv, n)


def check_visibility_views(g, c, h, flags, layers, range_start_end, vv, views):
    """views: ctypes array from make_views()."""
    n = len(flags)
    nv = len(views)
    vv = vv.copy()
    vis = np.zeros(nv * n, np.uint8)
    chg = np.zeros(n, np.uint8)
    lib().orc_check_visibility_views(n, fp(g), fp(c), fp(h), u8p(flags), u32p(layers), fp(range_start_end), u8p(vv),
                                     views, nv, u8p(vis), u8p(chg))
    return vv, vis.reshape(nv, n), chg


def visibility_margin_census(g, c, h, flags, layers, frusta):
    """-> uint64[6]: deciding plane-test values within 1 / 4 / 16 / 64 / 1024 ulps of their `<= 0.0` (cumulative), and all of them."""
    n = len(flags)
    fr = np.ascontiguousarray(frusta, np.float32).reshape(-1, 24)
    hist = np.zeros(6, np.uint64)
    lib().orc_visibility_margin_census(n, fp(g), fp(c), fp(h), u8p(flags), u32p(layers), fp(fr), None, len(fr), u64p(hist))
    return hist


def visibility_propagate(parent, visibility, inherited):
    """-> (rc, inherited_after, changed)"""
    n = len(visibility)
    inh = np.ascontiguousarray(inherited, np.uint8).copy()
    chg = np.zeros(n, np.uint8)
    rc = lib().orc_visibility_propagate(n, u32p(np.ascontiguousarray(parent, np.uint32)) if parent is not None else None,
                                        u8p(np.ascontiguousarray(visibility, np.uint8)), u8p(inh), u8p(chg))
    return rc, inh, chg


def check_visibility_gpu_culling(flags, vv):
    vv = vv.copy()
    chg = np.zeros(len(vv), np.uint8)
    lib().orc_check_visibility_gpu_culling(len(vv), u8p(flags), u8p(vv), u8p(chg))
    return vv, chg


def mark_newly_hidden(flags, vv):
    vv = vv.copy()
    chg = np.zeros(len(vv), np.uint8)
    lib().orc_mark_newly_hidden(len(vv), u8p(flags), u8p(vv), u8p(chg))
    return vv, chg


def visible_entities_sorted(visible, class_mask, class_bit, entity_keys):
    n = len(visible)
    keys = np.zeros(n, np.uint64)
    rows = np.zeros(n, np.uint32)
    m = lib().orc_visible_entities_sorted(n, u8p(visible), u32p(class_mask), int(class_bit), u64p(entity_keys),
                                          u64p(keys), u32p(rows))
    return keys[:m].copy(), rows[:m].copy()


def full_frame(t, r, s, c, h, flags, layers, vv, frusta, view_masks=None, view_flags=None):
    """reset -> sync_simple (all dirty) -> check_visibility -> gpu_culling rows -> mark_newly_hidden."""
    g, _ = sync_simple_transforms(t, r, s)
    vv1 = reset_view_visibility(flags, vv)
    vv2, vis, chg = check_visibility(g, c, h, flags, layers, vv1, frusta, view_masks, view_flags)
    vv3, chg2 = check_visibility_gpu_culling(flags, vv2)
    vv4, chg3 = mark_newly_hidden(flags, vv3)
    return g, vv4, vis, (chg | chg2 | chg3)


# ---- clustering ------------------------------------------------------------------------------

def cluster_dimensions_fixed_z(total, z_slices, w, h):
    out = (C.c_uint32 * 3)()
    lib().orc_cluster_dimensions_fixed_z(total, z_slices, w, h, out)
    return tuple(out)


def clusters_update(w, h, req):
    tile = (C.c_uint32 * 2)()
    dims = (C.c_uint32 * 3)()
    lib().orc_clusters_update(w, h, (C.c_uint32 * 3)(*req), tile, dims)
    return tuple(tile), tuple(dims)


def cluster_config_default():
    cfg = ClusterConfig()
    lib().orc_cluster_config_default(C.byref(cfg))
    return cfg


def cluster_config_resolve(config, last_farthest_z, last_total, w, h, max_indices=16384):
    """-> None (view cleared) or (requested_dims, first_slice_depth, far_z); last_* = None for Option::None."""
    req = (C.c_uint32 * 3)()
    fsd, far = C.c_float(0), C.c_float(0)
    lf = C.byref(C.c_float(last_farthest_z)) if last_farthest_z is not None else None
    lt = C.byref(C.c_uint64(last_total)) if last_total is not None else None
    active = lib().orc_cluster_config_resolve(C.byref(config), lf, lt, w, h, C.c_uint64(max_indices), req, C.byref(fsd), C.byref(far))
    return (tuple(req), float(fsd.value), float(far.value)) if active else None


def cluster_sort_truncate(obj_type, shadow_maps_enabled, volumetric, entity_bits, max_objects, supports_storage_buffers):
    n = len(entity_bits)
    order = np.zeros(max(n, 1), np.uint32)
    lib().orc_cluster_sort_truncate.restype = C.c_uint32
    m = lib().orc_cluster_sort_truncate(n, u8p(obj_type), u8p(shadow_maps_enabled), u8p(volumetric), u64p(entity_bits), max_objects,
                                        int(bool(supports_storage_buffers)), u32p(order))
    return order[:m]


def cluster_view_setup(camera_affine, clip_from_view, frustum, w, h, requested_dims, first_slice_depth, far_z,
                       view_layer_mask=1):
    view = ClusterView()
    cam, camp = _f(camera_affine); cfv, cfvp = _f(clip_from_view); fr, frp = _f(frustum)
    lib().orc_cluster_view_setup(camp, cfvp, frp, w, h, (C.c_uint32 * 3)(*requested_dims),
                                 C.c_float(first_slice_depth), C.c_float(far_z), view_layer_mask, C.byref(view))
    return view


def cluster_aabb_sphere(view, x, y, z):
    out = np.zeros(4, np.float32)
    lib().orc_cluster_aabb_sphere(C.byref(view), x, y, z, fp(out))
    return out


def assign_objects_to_clusters(view, pos_range, obj_type=None, layer_mask=None, spot_dir=None, spot_sin_cos=None,
                               capacity=None, layer_mask_hi=None):
    """layer_mask_hi: the objects' RenderLayers 32..63 (matched against view.view_layer_mask_hi)."""
    n = len(pos_range) // 4
    ncl = view.dims[0] * view.dims[1] * view.dims[2]
    offsets = np.zeros(ncl + 1, np.uint32)
    counts = np.zeros(6 * ncl, np.uint32)
    far = C.c_float(0)
    if capacity is None:
        total = lib().orc_assign_objects_to_clusters_layers64(C.byref(view), n, fp(pos_range), u8p(obj_type), u32p(layer_mask), u32p(layer_mask_hi),
                                                              fp(spot_dir), fp(spot_sin_cos), u32p(offsets), None,
                                                              C.c_uint64(0), u32p(counts), C.byref(far))
        capacity = int(total)
    indices = np.zeros(max(capacity, 1), np.uint32)
    total = lib().orc_assign_objects_to_clusters_layers64(C.byref(view), n, fp(pos_range), u8p(obj_type), u32p(layer_mask), u32p(layer_mask_hi),
                                                          fp(spot_dir), fp(spot_sin_cos), u32p(offsets), u32p(indices),
                                                          C.c_uint64(capacity), u32p(counts), C.byref(far))
    return offsets, indices[:min(int(total), capacity)], counts.reshape(ncl, 6), float(far.value), int(total)


def bench_assign_objects_to_clusters(view, pos_range, iters, obj_type=None, spot_dir=None, spot_sin_cos=None):
    """The reference's one-walk, push-per-cluster form of the assignment, `iters` frames; -> (seconds, total entries, farthest_z)."""
    n = len(pos_range) // 4
    total, far = C.c_uint64(0), C.c_float(0)
    lib().orc_bench_assign_objects_to_clusters.restype = C.c_double
    secs = lib().orc_bench_assign_objects_to_clusters(C.byref(view), n, fp(pos_range), u8p(obj_type), fp(spot_dir), fp(spot_sin_cos), int(iters),
                                                      C.byref(total), C.byref(far))
    return float(secs), int(total.value), float(far.value)


def mesh_inputs(g, c, h, flags, rows):
    """-> (world_from_local f32[len(rows)*12], culling f32[len(rows)*8]) for the listed rows."""
    m = len(rows)
    wfl = np.zeros(12 * max(m, 1), np.float32)
    cull = np.zeros(8 * max(m, 1), np.float32)
    g = np.ascontiguousarray(g, np.float32); c = np.ascontiguousarray(c, np.float32); h = np.ascontiguousarray(h, np.float32)
    for k, r in enumerate(rows):
        r = int(r)
        lib().orc_mesh_inputs(fp(g[12 * r:12 * r + 12].copy()), fp(c[3 * r:3 * r + 3].copy()), fp(h[3 * r:3 * r + 3].copy()),
                              int(flags[r] & FLAG_HAS_AABB != 0), fp(wfl[12 * k:12 * k + 12]), fp(cull[8 * k:8 * k + 8]))
    return wfl[:12 * m], cull[:8 * m]


def cluster_bindings_storage(offsets, counts, indices, remap=None):
    ncl = len(offsets) - 1
    oc = np.zeros(8 * ncl, np.uint32)
    idx = np.zeros(max(len(indices), 1), np.uint32)
    lib().orc_cluster_bindings_storage(ncl, u32p(np.ascontiguousarray(offsets, np.uint32)),
                                       u32p(np.ascontiguousarray(counts, np.uint32).reshape(-1)),
                                       u32p(np.ascontiguousarray(indices, np.uint32)),
                                       u32p(np.ascontiguousarray(remap, np.uint32)) if remap is not None else None,
                                       u32p(oc), u32p(idx))
    return oc, idx[:len(indices)]


def bench_flat_frame(t, r, s, c, h, flags, layers, frusta, threads, iters, fused_visibility=False):
    n = len(flags)
    nv = len(frusta) // 24
    g = np.zeros(12 * n, np.float32)
    vv = np.zeros(n, np.uint8)
    vis = np.zeros(nv * n, np.uint8)
    secs = lib().orc_bench_flat_frame2(n, fp(t), fp(r), fp(s), fp(c), fp(h), u8p(flags), u32p(layers), fp(g), u8p(vv),
                                       u8p(vis), fp(frusta), None, None, nv, int(threads), int(iters), 1 if fused_visibility else 0)
    return float(secs), g, vv, vis.reshape(nv, n)


# ---- batching work-item build (oracle/batching_oracle.c) -----------------------------------------------------------
NO_BATCH_SET = 0xFFFFFFFF


class BatchInitial(C.Structure):
    _fields_ = [("work_item_index", C.c_uint32 * 2), ("indirect_parameters_index", C.c_uint32 * 2),
                ("batch_set_index", C.c_uint32 * 2), ("output_mesh_uniform_index", C.c_uint32)]


class BatchTotals(C.Structure):
    _fields_ = [("work_item_len", C.c_uint32 * 2), ("indirect_parameters_len", C.c_uint32 * 2),
                ("batch_set_len", C.c_uint32 * 2), ("data_buffer_len", C.c_uint32), ("n_records", C.c_uint32)]


def unpack_bins(base_work_item, base_indirect, instances, bin_metadata, bin_table, n_out):
    """instances u32[k,2] (input_uniform_index, bin_index); bin_metadata u32[m,3]; -> work items u32[n_out,2]"""
    inst = np.ascontiguousarray(instances, np.uint32).reshape(-1, 2)
    out = np.zeros((n_out, 2), np.uint32)
    lib().orc_unpack_bins(C.c_uint32(base_work_item), C.c_uint32(base_indirect), C.c_uint32(len(inst)),
                          inst.ctypes.data_as(C.c_void_p), np.ascontiguousarray(bin_metadata, np.uint32).ctypes.data_as(C.c_void_p),
                          u32p(np.ascontiguousarray(bin_table, np.uint32)), out.ctypes.data_as(C.c_void_p))
    return out


def allocate_uniforms(batch_set_index, first_indirect, first_output, bin_metadata, n_out):
    meta = np.ascontiguousarray(bin_metadata, np.uint32).reshape(-1, 3)
    out = np.full((n_out, 5), 0xDEADBEEF, np.uint32)
    fan = np.zeros(len(meta) // 256 + 2, np.uint32)
    lib().orc_allocate_uniforms(C.c_uint32(batch_set_index), C.c_uint32(len(meta)), C.c_uint32(first_indirect), C.c_uint32(first_output),
                                meta.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), u32p(fan))
    return out


def batch_build(rows, row_set, row_bin, row_input, set_indexed, bin_table_offset, bin_table, meta_offset, bin_metadata,
                initial=None):
    """-> dict(work_items=[ni, ix], metadata=[ni, ix], batch_sets=[ni, ix], records u32[k,8], totals, bin_metadata u32[m,3])"""
    rows = np.ascontiguousarray(rows, np.uint32)
    n_sets = len(set_indexed)
    meta = np.ascontiguousarray(bin_metadata, np.uint32).reshape(-1, 3).copy()
    ini = initial if initial is not None else BatchInitial()
    cap_items = [int(ini.work_item_index[c]) + len(rows) + 1 for c in range(2)]
    cap_meta = [int(ini.indirect_parameters_index[c]) + len(meta) + 1 for c in range(2)]
    cap_sets = [int(ini.batch_set_index[c]) + n_sets + 1 for c in range(2)]
    wi = [np.zeros((cap_items[c], 2), np.uint32) for c in range(2)]
    md = [np.zeros((cap_meta[c], 5), np.uint32) for c in range(2)]
    bs = [np.zeros((cap_sets[c], 2), np.uint32) for c in range(2)]
    rec = np.zeros((n_sets + 1, 8), np.uint32)
    tot = BatchTotals()
    P2 = C.c_void_p * 2
    lib().orc_batch_build.restype = C.c_uint32
    lib().orc_batch_build(C.c_uint32(len(rows)), u32p(rows), u32p(np.ascontiguousarray(row_set, np.uint32)),
                          u32p(np.ascontiguousarray(row_bin, np.uint32)), u32p(np.ascontiguousarray(row_input, np.uint32)),
                          C.c_uint32(n_sets), u8p(np.ascontiguousarray(set_indexed, np.uint8)),
                          u32p(np.ascontiguousarray(bin_table_offset, np.uint32)), u32p(np.ascontiguousarray(bin_table, np.uint32)),
                          u32p(np.ascontiguousarray(meta_offset, np.uint32)), meta.ctypes.data_as(C.c_void_p), C.byref(ini),
                          P2(wi[0].ctypes.data, wi[1].ctypes.data), P2(md[0].ctypes.data, md[1].ctypes.data),
                          P2(bs[0].ctypes.data, bs[1].ctypes.data), rec.ctypes.data_as(C.c_void_p), C.byref(tot))
    return dict(work_items=[wi[c][:tot.work_item_len[c]] for c in range(2)],
                metadata=[md[c][:tot.indirect_parameters_len[c]] for c in range(2)],
                batch_sets=[bs[c][:tot.batch_set_len[c]] for c in range(2)], records=rec[:tot.n_records],
                totals=dict(work_item_len=list(tot.work_item_len), indirect_parameters_len=list(tot.indirect_parameters_len),
                            batch_set_len=list(tot.batch_set_len), data_buffer_len=int(tot.data_buffer_len)),
                bin_metadata=meta)


ROW_MULTIDRAWABLE, ROW_BATCHABLE, ROW_UNBATCHABLE = 0, 1, 2
NO_INPUT_INDEX = 0xFFFFFFFF
NO_INDEX = 0xFFFFFFFF
RECORD_BATCHABLE_BIN = 0x80000000
ITEM_INDEXED, ITEM_HAS_COMPARE_DATA = 1, 2


def _totals_dict(tot):
    return dict(work_item_len=list(tot.work_item_len), indirect_parameters_len=list(tot.indirect_parameters_len),
                batch_set_len=list(tot.batch_set_len), data_buffer_len=int(tot.data_buffer_len))


def initial_from_totals(t):
    ini = BatchInitial()
    for c in range(2):
        ini.work_item_index[c] = t["work_item_len"][c]
        ini.indirect_parameters_index[c] = t["indirect_parameters_len"][c]
        ini.batch_set_index[c] = t["batch_set_len"][c]
    ini.output_mesh_uniform_index = t["data_buffer_len"]
    return ini


def batch_cpu_bins(rows, row_kind, row_bin, row_input, unbatchable_indexed, batchable_indexed, no_indirect_drawing=False, initial=None):
    """The unbatchable + batchable loops (gpu_preprocessing.rs:2135-2357) for one view's list.
    -> dict(work_items, metadata, batch_sets (per class), unbatchable u32[k,2] (bin, instance_index), records u32[k,8], totals)"""
    rows = np.ascontiguousarray(rows, np.uint32)
    ini = initial if initial is not None else BatchInitial()
    n = len(rows)
    wi = [np.zeros((int(ini.work_item_index[c]) + n + 1, 2), np.uint32) for c in range(2)]
    md = [np.zeros((int(ini.indirect_parameters_index[c]) + n + 1, 5), np.uint32) for c in range(2)]
    bs = [np.zeros((int(ini.batch_set_index[c]) + n + 1, 2), np.uint32) for c in range(2)]
    unb = np.zeros((n + 1, 2), np.uint32)
    rec = np.zeros((len(batchable_indexed) + 1, 8), np.uint32)
    n_unb = C.c_uint32(0)
    tot = BatchTotals()
    P2 = C.c_void_p * 2
    lib().orc_batch_cpu_bins.restype = C.c_uint32
    lib().orc_batch_cpu_bins(C.c_uint32(n), u32p(rows), u8p(np.ascontiguousarray(row_kind, np.uint8)),
                             u32p(np.ascontiguousarray(row_bin, np.uint32)), u32p(np.ascontiguousarray(row_input, np.uint32)),
                             C.c_uint32(len(unbatchable_indexed)), u8p(np.ascontiguousarray(unbatchable_indexed, np.uint8)),
                             C.c_uint32(len(batchable_indexed)), u8p(np.ascontiguousarray(batchable_indexed, np.uint8)),
                             C.c_int(int(bool(no_indirect_drawing))), C.byref(ini), P2(wi[0].ctypes.data, wi[1].ctypes.data),
                             P2(md[0].ctypes.data, md[1].ctypes.data), P2(bs[0].ctypes.data, bs[1].ctypes.data),
                             unb.ctypes.data_as(C.c_void_p), C.byref(n_unb), rec.ctypes.data_as(C.c_void_p), C.byref(tot))
    return dict(work_items=[wi[c][:tot.work_item_len[c]] for c in range(2)],
                metadata=[md[c][:tot.indirect_parameters_len[c]] for c in range(2)],
                batch_sets=[bs[c][:tot.batch_set_len[c]] for c in range(2)], unbatchable=unb[:n_unb.value],
                records=rec[:tot.n_records], totals=_totals_dict(tot))


def batch_phase(rows, ph, no_indirect_drawing=False, initial=None):
    """One view's whole binned phase as the reference builds it: unbatchables, batchables (CPU loops), then -- with indirect
    drawing -- the multidrawable batch sets starting from the lengths the CPU loops left (gpu_preprocessing.rs:2431-2447).
    `ph` is a workloads.phase_scene dict.  Arrays are absolute (index 0 = start of the buffer; below `initial` zeros)."""
    cpu = batch_cpu_bins(rows, ph["row_kind"], ph["row_cpu_bin"], ph["row_input"], ph["unbatchable_indexed"], ph["batchable_indexed"],
                         no_indirect_drawing, initial)
    if no_indirect_drawing:
        cpu["bin_metadata"] = np.ascontiguousarray(ph["bin_metadata"], np.uint32).reshape(-1, 3).copy()
        cpu["bin_metadata"][:, 2] = 0
        return cpu
    multi_rows = np.ascontiguousarray(rows, np.uint32)
    row_set = np.where(np.asarray(ph["row_kind"]) == ROW_MULTIDRAWABLE, ph["row_set"], NO_BATCH_SET).astype(np.uint32)
    md = batch_build(multi_rows, row_set, ph["row_bin"], ph["row_input"], ph["set_indexed"], ph["bin_table_offset"], ph["bin_table"],
                     ph["meta_offset"], ph["bin_metadata"], initial_from_totals(cpu["totals"]))
    out = dict(unbatchable=cpu["unbatchable"], records=np.concatenate([cpu["records"], md["records"]]), totals=md["totals"],
               bin_metadata=md["bin_metadata"])
    for k in ("work_items", "metadata", "batch_sets"):
        out[k] = []
        for c in range(2):
            a = md[k][c].copy()
            a[:len(cpu[k][c])] = cpu[k][c]  # the multidrawable pass leaves the region below its `initial` zero
            out[k].append(a)
    return out


def _sorted_call(fn, items, automatic_batching, *rest):
    it = np.ascontiguousarray(items, np.uint32).reshape(-1, 4)
    return it, [C.c_uint32(len(it)), it.ctypes.data_as(C.c_void_p), C.c_int(int(bool(automatic_batching)))] + list(rest)


def batch_sorted(items, automatic_batching=True, no_indirect_drawing=False, initial=None):
    """gpu_preprocessing::batch_and_prepare_sorted_render_phase over items u32[n,4] (input_index, batch_set_key, bin_key, flags)."""
    ini = initial if initial is not None else BatchInitial()
    n = len(np.asarray(items).reshape(-1, 4))
    wi = [np.zeros((int(ini.work_item_index[c]) + n + 1, 2), np.uint32) for c in range(2)]
    md = [np.zeros((int(ini.indirect_parameters_index[c]) + n + 1, 5), np.uint32) for c in range(2)]
    bs = [np.zeros((int(ini.batch_set_index[c]) + n + 1, 2), np.uint32) for c in range(2)]
    bat = np.zeros((n + 1, 6), np.uint32)
    tot = BatchTotals()
    P2 = C.c_void_p * 2
    it, args = _sorted_call(None, items, automatic_batching, C.c_int(int(bool(no_indirect_drawing))), C.byref(ini),
                            P2(wi[0].ctypes.data, wi[1].ctypes.data), P2(md[0].ctypes.data, md[1].ctypes.data),
                            P2(bs[0].ctypes.data, bs[1].ctypes.data), bat.ctypes.data_as(C.c_void_p), C.byref(tot))
    lib().orc_batch_sorted.restype = C.c_uint32
    k = lib().orc_batch_sorted(*args)
    return dict(work_items=[wi[c][:tot.work_item_len[c]] for c in range(2)],
                metadata=[md[c][:tot.indirect_parameters_len[c]] for c in range(2)],
                batch_sets=[bs[c][:tot.batch_set_len[c]] for c in range(2)], batches=bat[:k], totals=_totals_dict(tot))


def batch_sorted_merge(items, automatic_batching=True, first_index=0):
    """batching::batch_and_prepare_sorted_render_phase (mod.rs:219-244) under no_gpu_preprocessing.rs:76-103.
    -> (batches u32[k,6], buffer_len)"""
    n = len(np.asarray(items).reshape(-1, 4))
    bat = np.zeros((n + 1, 6), np.uint32)
    blen = C.c_uint32(0)
    it, args = _sorted_call(None, items, automatic_batching, C.c_uint32(first_index), bat.ctypes.data_as(C.c_void_p), C.byref(blen))
    lib().orc_batch_sorted_merge.restype = C.c_uint32
    k = lib().orc_batch_sorted_merge(*args)
    return bat[:k], int(blen.value)


def bench_tree_frame(parent, level_offsets, t, r, s, threads, iters):
    """Level-parallel propagate (all Transforms changed) on a persistent pool; -> (seconds, G)."""
    n = len(parent)
    g = np.zeros(12 * n, np.float32)
    lo = np.ascontiguousarray(level_offsets, np.uint32)
    lib().orc_bench_tree_frame.restype = C.c_double
    secs = lib().orc_bench_tree_frame(C.c_uint32(n), u32p(np.ascontiguousarray(parent, np.uint32)), u32p(lo), C.c_uint32(len(lo) - 1),
                                      fp(t), fp(r), fp(s), fp(g), int(threads), int(iters))
    return float(secs), g
