"""ctypes binding of include/bevy_mi355x.h.  Thin by design: one Python method per C entry point,
numpy arrays in / out, status codes turned into MiError.  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_LIB_VARIANT=<name>: an experiment build (bevy_amd.build.build(variant=name)) for A/B measurements on the GPU box
_LIB_PATH = os.path.join(_HERE, "libbevy_mi355x_%s.so" % os.environ["MI_LIB_VARIANT"] if os.environ.get("MI_LIB_VARIANT") else "libbevy_mi355x.so")

MI_OK = 0
MI_ERR_INVALID_ARG = -1
MI_ERR_DEVICE = -2
MI_ERR_OUT_OF_MEMORY = -3
MI_ERR_MALFORMED_HIERARCHY = -4
MI_ERR_NOT_READY = -5
MI_ERR_CAPACITY = -6


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bevy_mi355x error {code}: {msg}")
        self.code = code


class ClusterView(C.Structure):
    """mi_cluster_view"""
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
        ("x_planes", C.POINTER(C.c_float)),
        ("y_planes", C.POINTER(C.c_float)),
        ("z_planes", C.POINTER(C.c_float)),
        ("cluster_spheres", C.POINTER(C.c_float)),
        ("view_layer_mask_hi", C.c_uint32),
    ]

    @property
    def n_clusters(self):
        return self.dims[0] * self.dims[1] * self.dims[2]


class ClusterConfig(C.Structure):
    """mi_cluster_config"""
    _fields_ = [("kind", C.c_uint32), ("dimensions", C.c_uint32 * 3), ("total", C.c_uint32), ("z_slices", C.c_uint32),
                ("first_slice_depth", C.c_float), ("far_z_mode", C.c_uint32), ("far_z_constant", C.c_float),
                ("dynamic_resizing", C.c_uint32)]


class ClusterHistory(C.Structure):
    """mi_cluster_history"""
    _fields_ = [("has_farthest_z", C.c_uint32), ("farthest_z", C.c_float), ("has_total_cluster_index_count", C.c_uint32),
                ("reserved", C.c_uint32), ("total_cluster_index_count", C.c_uint64)]


class ClusterResolved(C.Structure):
    """mi_cluster_resolved"""
    _fields_ = [("active", C.c_uint32), ("requested_dims", C.c_uint32 * 3), ("first_slice_depth", C.c_float), ("far_z", C.c_float)]


CLUSTER_CONFIG_NONE, CLUSTER_CONFIG_SINGLE, CLUSTER_CONFIG_XYZ, CLUSTER_CONFIG_FIXED_Z = 0, 1, 2, 3
CLUSTER_FAR_Z_MAX_CLUSTERABLE_OBJECT_RANGE, CLUSTER_FAR_Z_CONSTANT = 0, 1
VIEW_CLUSTER_BINDINGS_MAX_INDICES = 16384
MAX_UNIFORM_BUFFER_CLUSTERABLE_OBJECTS = 204


RESULTS_CHANGED_ROWS, RESULTS_CHANGED_GLOBALS, RESULTS_CLUSTERS, RESULTS_CLUSTER_INDICES, RESULTS_IN_PLACE = 0x1, 0x2, 0x4, 0x8, 0x10
RESULTS_MAX_LISTS = 16


class VisibleList(C.Structure):
    """mi_visible_list"""
    _fields_ = [("view", C.c_uint32), ("class_bit", C.c_uint32), ("capacity", C.c_uint32), ("count", C.c_uint32), ("rows", C.POINTER(C.c_uint32))]


class FrameResults(C.Structure):
    """mi_frame_results"""
    _fields_ = [("flags", C.c_uint32), ("n_lists", C.c_uint32), ("lists", C.POINTER(VisibleList)), ("changed_capacity", C.c_uint32),
                ("reserved", C.c_uint32), ("cluster_capacity", C.c_uint64),
                ("changed_rows", C.POINTER(C.c_uint32)), ("changed_global12", C.POINTER(C.c_float)),
                ("cluster_offsets", C.POINTER(C.c_uint32)), ("cluster_counts", C.POINTER(C.c_uint32)), ("cluster_indices", C.POINTER(C.c_uint32)),
                ("changed_count", C.c_uint32), ("farthest_z", C.c_float), ("cluster_total", C.c_uint64)]


class FrameResultBuffers:
    """The caller's side of Context.download_frame_results, allocated once (an ECS system would hand in its own slices).
    changed_capacity / n_clusters = 0 switch those parts off; lists = [(view, class_bit, capacity), ...] (default: the camera's
    list of class 0 when visible_capacity > 0).  in_place: no buffers -- the library returns pointers into its pinned window."""

    def __init__(self, changed_capacity, visible_capacity, n_clusters, cluster_capacity, view=0, class_bit=0, lists=None, in_place=False,
                 want_rows=True, want_globals=True):
        if lists is None:
            lists = [(view, class_bit, visible_capacity)] if visible_capacity else []
        self.in_place = in_place
        self.n_clusters = n_clusters
        self.list_spec = list(lists)
        r = self.raw = FrameResults()
        r.flags = ((RESULTS_CHANGED_ROWS if want_rows else 0) | (RESULTS_CHANGED_GLOBALS if want_globals else 0) if changed_capacity else 0) | \
                  ((RESULTS_CLUSTERS | RESULTS_CLUSTER_INDICES) if n_clusters else 0) | (RESULTS_IN_PLACE if in_place else 0)
        r.changed_capacity, r.cluster_capacity = changed_capacity, cluster_capacity if n_clusters else 0
        self._lists = (VisibleList * max(len(lists), 1))()
        self.list_rows = []
        for k, (v, cb, cap) in enumerate(lists):
            self._lists[k].view, self._lists[k].class_bit, self._lists[k].capacity = v, cb, cap
            if not in_place:
                a = np.zeros(max(cap, 1), np.uint32)
                self.list_rows.append(a)
                self._lists[k].rows = _ptr(a, C.c_uint32)
        r.n_lists = len(lists)
        r.lists = C.cast(self._lists, C.POINTER(VisibleList)) if lists else None
        if not in_place:
            self.changed_rows = np.zeros(max(changed_capacity, 1), np.uint32)
            self.changed_global = np.zeros(12 * max(changed_capacity, 1), np.float32)
            self.cluster_offsets = np.zeros(n_clusters + 1, np.uint32) if n_clusters else None
            self.cluster_counts = np.zeros(6 * n_clusters, np.uint32) if n_clusters else None
            self.cluster_indices = np.zeros(max(cluster_capacity, 1), np.uint32) if n_clusters else None
            r.changed_rows = _ptr(self.changed_rows, C.c_uint32)
            r.changed_global12 = _ptr(self.changed_global, C.c_float)
            if n_clusters:
                r.cluster_offsets = _ptr(self.cluster_offsets, C.c_uint32)
                r.cluster_counts = _ptr(self.cluster_counts, C.c_uint32)
                r.cluster_indices = _ptr(self.cluster_indices, C.c_uint32)

    @property
    def visible_rows(self):
        return self.list_rows[0]

    def list_count(self, k=0):
        return int(self._lists[k].count)


class UploadWindow(C.Structure):
    """mi_upload_window"""
    _fields_ = [("rows", C.POINTER(C.c_uint32)), ("translation", C.POINTER(C.c_float)), ("rotation", C.POINTER(C.c_float)),
                ("scale", C.POINTER(C.c_float)), ("capacity", C.c_uint32), ("flags", C.c_uint32), ("token", C.c_uint64)]


UPLOAD_DENSE = 0x1
UPLOAD_TRANSLATION, UPLOAD_ROTATION, UPLOAD_SCALE = 0x2, 0x4, 0x8


class View(C.Structure):
    """mi_view"""
    _fields_ = [("frustum", C.c_float * 24), ("layer_mask", C.c_uint32), ("flags", C.c_uint32),
                ("position", C.c_float * 3), ("light_sphere", C.c_float * 4), ("layer_mask_hi", C.c_uint32), ("reserved", C.c_uint32 * 2)]


def make_views(frusta, layer_masks=None, flags=None, positions=None, light_spheres=None, layer_masks_hi=None):
    """Per-view numpy columns -> ctypes array of mi_view."""
    fr = np.ascontiguousarray(frusta, np.float32).reshape(-1, 24)
    nv = len(fr)
    arr = (View * nv)()
    pos = None if positions is None else np.asarray(positions, np.float32).reshape(-1, 3)
    sph = None if light_spheres is None else np.asarray(light_spheres, np.float32).reshape(-1, 4)
    for v in range(nv):
        arr[v].frustum[:] = fr[v].tolist()
        arr[v].layer_mask = int(layer_masks[v]) if layer_masks is not None else 1
        arr[v].flags = int(flags[v]) if flags is not None else 0
        arr[v].layer_mask_hi = int(layer_masks_hi[v]) if layer_masks_hi is not None else 0
        if pos is not None:
            arr[v].position[:] = pos[v].tolist()
        if sph is not None:
            arr[v].light_sphere[:] = sph[v].tolist()
    return arr


_lib = None

# every symbol include/bevy_mi355x.h declares (tests/test_abi.py checks header <-> library <-> this list)
BATCH_ROW_MULTIDRAWABLE, BATCH_ROW_BATCHABLE, BATCH_ROW_UNBATCHABLE, BATCH_ROW_NONE = 0, 1, 2, 3
BATCH_NO_INDIRECT_DRAWING = 0x1
SORTED_AUTOMATIC_BATCHING, SORTED_NO_INDIRECT_DRAWING, SORTED_NO_GPU_PREPROCESSING = 0x1, 0x2, 0x4
NO_INPUT_INDEX = 0xFFFFFFFF

ABI_SYMBOLS = [
    "mi_abi_version", "mi_ctx_create", "mi_ctx_destroy", "mi_last_error_string", "mi_synchronize",
    "mi_columns_resize", "mi_upload_transforms", "mi_upload_transforms_indexed", "mi_map_upload_window", "mi_commit_upload_window", "mi_upload_global_transforms", "mi_upload_bounds", "mi_upload_render_layers_hi",
    "mi_upload_view_visibility", "mi_upload_visibility_classes", "mi_upload_entity_keys", "mi_upload_changed",
    "mi_upload_visibility_ranges", "mi_upload_visibility", "mi_upload_hierarchy", "mi_hierarchy_sort", "mi_hierarchy_advice_for", "mi_propagate",
    "mi_visibility_propagate", "mi_download_inherited_visibility",
    "mi_visibility_begin_frame", "mi_cull", "mi_cull_views", "mi_propagate_and_cull", "mi_propagate_and_cull_views",
    "mi_visibility_end_frame", "mi_check_light_mesh_visibility",
    "mi_download_global_transforms", "mi_download_changed_global_transforms", "mi_download_frame_results", "mi_download_changed_mesh_inputs", "mi_download_visibility", "mi_download_view_visibility",
    "mi_download_visible_entities", "mi_cluster_view_dims", "mi_cluster_view_build",
    "mi_cluster_dimensions_fixed_z", "mi_cluster_assign", "mi_cluster_upload_objects", "mi_cluster_upload_object_layers_hi", "mi_cluster_upload_view",
    "mi_cluster_assign_resident", "mi_cluster_select_view", "mi_cluster_download", "mi_cluster_download_bindings",
    "mi_cluster_config_default", "mi_cluster_config_resolve", "mi_cluster_sort_truncate", "mi_cluster_bind_objects_to_rows", "mi_cluster_bind_objects_to_row_list",
    "mi_cluster_assign_frame",
    "mi_batch_upload_rows", "mi_batch_upload_sets", "mi_batch_upload_row_bins", "mi_batch_upload_bins", "mi_batch_build", "mi_batch_build_phase",
    "mi_batch_sorted_build", "mi_batch_download_totals", "mi_batch_download", "mi_perspective_clip_from_view", "mi_compute_frustum",
    "mi_bind_visibility_output", "mi_exchange_set_mode", "mi_exchange_configure", "mi_exchange_configure_multi", "mi_exchange_configure_owned", "mi_exchange_group_flush", "mi_exchange_last", "mi_exchange_download", "mi_device_buffer",
]
# include/bevy_mi355x_debug.h: instrumentation and test hooks, exported by the same library, not part of the boundary
DEBUG_SYMBOLS = [
    "mi_timer_begin", "mi_timer_end", "mi_profile_enable", "mi_profile_filter", "mi_profile_sample", "mi_profile_burst", "mi_profile_read",
    "mi_profile_kernel_name", "mi_debug_set_tile_mode", "mi_debug_tree_trace", "mi_debug_tile_plan", "mi_debug_tile_groups", "mi_debug_strip_plan", "mi_debug_plan_strips", "mi_debug_exchange_times", "mi_debug_logf", "mi_debug_set_sphere_path", "mi_debug_set_row_summary", "mi_debug_set_tree_cull", "mi_debug_set_walk_inrow", "mi_debug_set_chunked_frames", "mi_debug_chunked_counts", "mi_debug_set_tile_pretest", "mi_debug_set_sorted_one_wg_limit", "mi_debug_set_static_cull_order", "mi_debug_static_cull_counts", "mi_debug_cluster_download_unjoined",
]


def lib_path():
    return _LIB_PATH


def load_library():
    """Loads libbevy_mi355x.so.  Fails loudly if it has not been built (python -m bevy_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MiError(MI_ERR_DEVICE, f"{_LIB_PATH} is missing: build it with `python -m bevy_amd.build` "
                                     "(there is no CPU fallback)")
    lib = C.CDLL(_LIB_PATH)
    lib.mi_last_error_string.restype = C.c_char_p
    lib.mi_last_error_string.argtypes = [C.c_void_p]
    lib.mi_profile_kernel_name.restype = C.c_char_p
    for name in ABI_SYMBOLS + DEBUG_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("mi_last_error_string", "mi_profile_kernel_name"):
            fn.restype = C.c_int32
    _lib = lib
    return lib


class PreparedFrusta:
    """A frusta array converted once (contiguous f32 + its ctypes pointer) for per-frame calls."""

    def __init__(self, frusta):
        self.array = np.ascontiguousarray(frusta, dtype=np.float32).reshape(-1)
        self.n_views = len(self.array) // 24
        self.pointer = self.array.ctypes.data_as(C.POINTER(C.c_float))


def _ptr(a, ty):
    if a is None:
        return None
    if isinstance(a, PreparedFrusta):
        return a.pointer
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need a C-contiguous numpy array"
    return a.ctypes.data_as(C.POINTER(ty))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def _u32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint64)


def _host_check(rc, what):
    if rc != MI_OK:
        raise MiError(rc, f"{what} failed")


# ---- pure host helpers -------------------------------------------------------------------------

def perspective_clip_from_view(fov, aspect, near):
    out = np.zeros(16, np.float32)
    _host_check(load_library().mi_perspective_clip_from_view(C.c_float(fov), C.c_float(aspect), C.c_float(near),
                                                             _ptr(out, C.c_float)), "mi_perspective_clip_from_view")
    return out


def compute_frustum(clip_from_view, camera_affine, far):
    cfv, cam = _f32(clip_from_view), _f32(camera_affine)
    out = np.zeros(24, np.float32)
    _host_check(load_library().mi_compute_frustum(_ptr(cfv, C.c_float), _ptr(cam, C.c_float), C.c_float(far),
                                                  _ptr(out, C.c_float)), "mi_compute_frustum")
    return out


def cluster_dimensions_fixed_z(total, z_slices, w, h):
    out = (C.c_uint32 * 3)()
    _host_check(load_library().mi_cluster_dimensions_fixed_z(total, z_slices, w, h, out), "mi_cluster_dimensions_fixed_z")
    return tuple(out)


def cluster_config_default():
    cfg = ClusterConfig()
    _host_check(load_library().mi_cluster_config_default(C.byref(cfg)), "mi_cluster_config_default")
    return cfg


def cluster_config_resolve(config, history, w, h, max_indices=VIEW_CLUSTER_BINDINGS_MAX_INDICES):
    out = ClusterResolved()
    _host_check(load_library().mi_cluster_config_resolve(C.byref(config), C.byref(history) if history is not None else None, w, h,
                                                         C.c_uint64(max_indices), C.byref(out)), "mi_cluster_config_resolve")
    return out


def cluster_sort_truncate(obj_type, shadow_maps_enabled, volumetric, entity_bits, max_objects, supports_storage_buffers):
    ty, sh, vo, en = _u8(obj_type), _u8(shadow_maps_enabled), _u8(volumetric), _u64(entity_bits)
    n = len(en)
    order = np.zeros(max(n, 1), np.uint32)
    out_n = C.c_uint32(0)
    _host_check(load_library().mi_cluster_sort_truncate(n, _ptr(ty, C.c_uint8), _ptr(sh, C.c_uint8), _ptr(vo, C.c_uint8),
                                                        _ptr(en, C.c_uint64), max_objects, int(bool(supports_storage_buffers)),
                                                        _ptr(order, C.c_uint32), C.byref(out_n)), "mi_cluster_sort_truncate")
    return order[:out_n.value]


def cluster_view_build(camera_affine, clip_from_view, frustum, w, h, requested_dims, first_slice_depth, far_z,
                       view_layer_mask=1, with_spheres=True):
    """Returns (ClusterView, keepalive) -- keepalive owns the plane / sphere storage the view points into."""
    lib = load_library()
    req = (C.c_uint32 * 3)(*requested_dims)
    tile = (C.c_uint32 * 2)()
    dims = (C.c_uint32 * 3)()
    _host_check(lib.mi_cluster_view_dims(w, h, req, tile, dims), "mi_cluster_view_dims")
    planes = np.zeros((dims[0] + dims[1] + dims[2] + 3) * 4, np.float32)
    spheres = np.zeros(dims[0] * dims[1] * dims[2] * 4, np.float32) if with_spheres else None
    cam, cfv, fr = _f32(camera_affine), _f32(clip_from_view), _f32(frustum)
    view = ClusterView()
    _host_check(lib.mi_cluster_view_build(_ptr(cam, C.c_float), _ptr(cfv, C.c_float), _ptr(fr, C.c_float), w, h, req,
                                          C.c_float(first_slice_depth), C.c_float(far_z), view_layer_mask,
                                          _ptr(planes, C.c_float), _ptr(spheres, C.c_float), C.byref(view)),
                "mi_cluster_view_build")
    return view, (planes, spheres)


def hierarchy_sort(parent):
    """-> (new_to_old, parent_idx_new, level_offsets)"""
    parent = _u32(parent)
    n = len(parent)
    new_to_old = np.zeros(max(n, 1), np.uint32)
    pidx = np.zeros(max(n, 1), np.uint32)
    cap = n + 2
    offs = np.zeros(cap, np.uint32)
    nl = C.c_uint32(0)
    rc = load_library().mi_hierarchy_sort(n, _ptr(parent, C.c_uint32), _ptr(new_to_old, C.c_uint32),
                                          _ptr(pidx, C.c_uint32), _ptr(offs, C.c_uint32), cap, C.byref(nl))
    if rc != MI_OK:
        raise MiError(rc, "mi_hierarchy_sort: malformed hierarchy" if rc == MI_ERR_MALFORMED_HIERARCHY else "mi_hierarchy_sort")
    return new_to_old[:n], pidx[:n], offs[:nl.value + 1].copy()


class HierarchyAdvice(C.Structure):
    """mi_hierarchy_advice"""
    _fields_ = [("plan", C.c_uint32), ("keep_on_host", C.c_uint32), ("n_levels", C.c_uint32), ("widest_level", C.c_uint32),
                ("est_device_us", C.c_float), ("est_host_us", C.c_float)]


def hierarchy_advice(level_offsets):
    """How mi_propagate would walk a hierarchy of these level sizes and whether the stock CPU systems should keep it (a hierarchy no
    wider than a wave: a chain, a rope) -> dict(plan, keep_on_host, n_levels, widest_level, est_device_us, est_host_us).  No context."""
    offs = _u32(level_offsets)
    a = HierarchyAdvice()
    rc = load_library().mi_hierarchy_advice_for(max(len(offs) - 1, 0), _ptr(offs, C.c_uint32), C.byref(a))
    if rc != MI_OK:
        raise MiError(rc, "mi_hierarchy_advice_for")
    return {k: getattr(a, k) for k, _ in HierarchyAdvice._fields_}


def debug_plan_strips(parent, level_offsets, width=64):
    """The strips plan of a hierarchy (mi_debug_plan_strips: host code, no context) -> None when it cannot be planned, else
    dict(strips: (n, 3) uint32 [first entry, entries | batches << 16 | bit 31, first own level], rounds: (m, 4) uint32
    [row0, pstart, info, level], bands, snap_rows)."""
    par, offs = _u32(parent), _u32(level_offsets)
    counts = np.zeros(4, np.uint32)
    lib = load_library()
    rc = lib.mi_debug_plan_strips(len(offs) - 1, _ptr(offs, C.c_uint32), _ptr(par, C.c_uint32), int(width), None, 0, None, 0, _ptr(counts, C.c_uint32))
    if rc != MI_OK:
        raise MiError(rc, "mi_debug_plan_strips")
    if not counts[0]:
        return None
    strips, rounds = np.zeros((int(counts[0]), 3), np.uint32), np.zeros((int(counts[1]), 4), np.uint32)
    rc = lib.mi_debug_plan_strips(len(offs) - 1, _ptr(offs, C.c_uint32), _ptr(par, C.c_uint32), int(width), _ptr(strips, C.c_uint32), len(strips),
                                  _ptr(rounds, C.c_uint32), len(rounds), _ptr(counts, C.c_uint32))
    if rc != MI_OK:
        raise MiError(rc, "mi_debug_plan_strips")
    return dict(strips=strips, rounds=rounds, bands=int(counts[2]), snap_rows=int(counts[3]))


# ---- device context ------------------------------------------------------------------------------

class Context:
    """One mi_ctx: device-resident component columns + the systems of the render-prep path."""

    def __init__(self, device=0, stream=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.mi_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != MI_OK:
            msg = self._lib.mi_last_error_string(None).decode()
            self._h = None
            raise MiError(rc, msg)
        self.n = 0
        self.n_views = 0
        # test harness only: run a whole test session under one tile kernel (the library itself reads no environment)
        if os.environ.get("MI_TEST_TILE_MODE"):
            self.debug_set_tile_mode(int(os.environ["MI_TEST_TILE_MODE"]))
        if os.environ.get("MI_TEST_SORTED_TILED"):
            self.debug_set_sorted_one_wg_limit(0)
        if os.environ.get("MI_TEST_TILE_PRETEST"):
            self.debug_set_tile_pretest(int(os.environ["MI_TEST_TILE_PRETEST"]))
        if os.environ.get("MI_TEST_CHUNKED_FRAMES"):  # ... or with upload windows never (1) / always (2) in pieces and GlobalTransforms sent back ahead of the frame
            self.debug_set_chunked_frames(int(os.environ["MI_TEST_CHUNKED_FRAMES"]))
        if os.environ.get("MI_TEST_WALK_INROW"):  # ... or with the riding cluster walk in workgroups of its own (1)
            self.debug_set_walk_inrow(int(os.environ["MI_TEST_WALK_INROW"]))
        if os.environ.get("MI_TEST_TREE_CULL"):  # ... or with the all-dirty hierarchy frame fused into the tile launches (2)
            self.debug_set_tree_cull(int(os.environ["MI_TEST_TREE_CULL"]))
        if os.environ.get("MI_TEST_ROW_SUMMARY"):  # ... or without the per-wave Aabb / flags / layers summary (1)
            self.debug_set_row_summary(int(os.environ["MI_TEST_ROW_SUMMARY"]))
        if os.environ.get("MI_TEST_SPHERE_PATH"):  # ... or with the world-sphere cull path forced on (2) / off (1)
            self.debug_set_sphere_path(int(os.environ["MI_TEST_SPHERE_PATH"]))
        if os.environ.get("MI_TEST_STATIC_CULL_ORDER"):
            self.debug_set_static_cull_order(int(os.environ["MI_TEST_STATIC_CULL_ORDER"]))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != MI_OK:
            raise MiError(rc, self._lib.mi_last_error_string(self._h).decode())

    # columns
    def resize(self, n):
        self._ck(self._lib.mi_columns_resize(self._h, int(n)))
        self.n = int(n)

    def upload_transforms(self, translation, rotation, scale, first_row=0):
        t, r, s = _f32(translation), _f32(rotation), _f32(scale)
        n = len(t) // 3
        self._ck(self._lib.mi_upload_transforms(self._h, first_row, n, _ptr(t, C.c_float), _ptr(r, C.c_float),
                                                _ptr(s, C.c_float)))

    def upload_transforms_indexed(self, rows, translation, rotation, scale):
        rw, t, r, s = _u32(rows), _f32(translation), _f32(rotation), _f32(scale)
        self._ck(self._lib.mi_upload_transforms_indexed(self._h, len(rw), _ptr(rw, C.c_uint32), _ptr(t, C.c_float),
                                                        _ptr(r, C.c_float), _ptr(s, C.c_float)))

    def map_upload_window(self, capacity, dense=False, components="trs"):
        """-> (window, rows u32[capacity] or None, translation f32[3 capacity], rotation f32[4 capacity], scale f32[3 capacity]): numpy views
        on the library's pinned memory, to be filled in place and handed over with commit_upload_window.  components: which of
        translation / rotation / scale the window carries ("trs" = all; "r" = rotations only, ...): the others come back as None."""
        w = UploadWindow()
        comp = 0 if set(components) == set("trs") else sum({"t": UPLOAD_TRANSLATION, "r": UPLOAD_ROTATION, "s": UPLOAD_SCALE}[c] for c in set(components))
        self._ck(self._lib.mi_map_upload_window(self._h, int(capacity), (UPLOAD_DENSE if dense else 0) | comp, C.byref(w)))
        arr = lambda p, k, dt: (np.ctypeslib.as_array(p, shape=(k * capacity,)).view(dt) if capacity else np.zeros(0, dt)) if p else None
        return w, (None if dense else arr(w.rows, 1, np.uint32)), arr(w.translation, 3, np.float32), arr(w.rotation, 4, np.float32), arr(w.scale, 3, np.float32)

    def commit_upload_window(self, window, n, first_row=0):
        self._ck(self._lib.mi_commit_upload_window(self._h, C.byref(window), int(n), int(first_row)))

    def upload_global_transforms(self, g, first_row=0):
        g = _f32(g)
        self._ck(self._lib.mi_upload_global_transforms(self._h, first_row, len(g) // 12, _ptr(g, C.c_float)))

    def upload_bounds(self, center, half, flags=None, layer_mask=None, first_row=0):
        c, h, f, l = _f32(center), _f32(half), _u8(flags), _u32(layer_mask)
        self._ck(self._lib.mi_upload_bounds(self._h, first_row, len(c) // 3, _ptr(c, C.c_float), _ptr(h, C.c_float),
                                            _ptr(f, C.c_uint8), _ptr(l, C.c_uint32)))

    def upload_render_layers_hi(self, layer_mask_hi, first_row=0):
        """RenderLayers 32..63 of the rows (the low word goes with upload_bounds)."""
        m = _u32(layer_mask_hi)
        self._ck(self._lib.mi_upload_render_layers_hi(self._h, first_row, len(m), _ptr(m, C.c_uint32)))

    def upload_view_visibility(self, vv, first_row=0):
        vv = _u8(vv)
        self._ck(self._lib.mi_upload_view_visibility(self._h, first_row, len(vv), _ptr(vv, C.c_uint8)))

    def upload_visibility_classes(self, class_mask, first_row=0):
        cm = _u32(class_mask)
        self._ck(self._lib.mi_upload_visibility_classes(self._h, first_row, len(cm), _ptr(cm, C.c_uint32)))

    def upload_entity_keys(self, keys, first_row=0):
        k = _u64(keys)
        self._ck(self._lib.mi_upload_entity_keys(self._h, first_row, len(k), _ptr(k, C.c_uint64)))

    def upload_changed(self, changed, first_row=0):
        ch = _u8(changed)
        self._ck(self._lib.mi_upload_changed(self._h, first_row, len(ch), _ptr(ch, C.c_uint8)))

    def upload_changed_all(self):
        """Every row counts as changed (the 100 %-dirty frame)."""
        if getattr(self, "_ones", None) is None or len(self._ones) != self.n:
            self._ones = np.ones(self.n, np.uint8)
        self.upload_changed(self._ones)

    def upload_visibility_ranges(self, start_end, first_row=0):
        """start_end: f32[2n] (start_margin.start, end_margin.end); None = no VisibleEntityRanges resource."""
        if start_end is None:
            self._ck(self._lib.mi_upload_visibility_ranges(self._h, 0, 0, None))
            return
        a = _f32(start_end)
        self._ck(self._lib.mi_upload_visibility_ranges(self._h, first_row, len(a) // 2, _ptr(a, C.c_float)))

    def upload_visibility(self, visibility, first_row=0):
        v = _u8(visibility)
        self._ck(self._lib.mi_upload_visibility(self._h, first_row, len(v), _ptr(v, C.c_uint8)))

    def visibility_propagate(self):
        self._ck(self._lib.mi_visibility_propagate(self._h))

    def download_inherited_visibility(self, first_row=0, n=None):
        n = self.n - first_row if n is None else n
        out = np.zeros(n, np.uint8)
        chg = np.zeros((n + 31) // 32, np.uint32)
        self._ck(self._lib.mi_download_inherited_visibility(self._h, first_row, n, _ptr(out, C.c_uint8), _ptr(chg, C.c_uint32)))
        return out, unpack_bits(chg, n)

    def upload_hierarchy(self, parent_idx, level_offsets):
        if parent_idx is None:
            self._ck(self._lib.mi_upload_hierarchy(self._h, self.n, None, None, 1))
            return
        p, lo = _u32(parent_idx), _u32(level_offsets)
        self._ck(self._lib.mi_upload_hierarchy(self._h, len(p), _ptr(p, C.c_uint32), _ptr(lo, C.c_uint32), len(lo) - 1))

    # systems
    def propagate(self, flags=0x1):
        self._ck(self._lib.mi_propagate(self._h, int(flags)))

    def visibility_begin_frame(self):
        self._ck(self._lib.mi_visibility_begin_frame(self._h))

    def visibility_end_frame(self):
        self._ck(self._lib.mi_visibility_end_frame(self._h))

    def _views(self, frusta, view_masks, view_flags):
        if isinstance(frusta, PreparedFrusta):  # per-frame hot path: no numpy work
            self.n_views = frusta.n_views
            return frusta, _u32(view_masks), _u8(view_flags), frusta.n_views
        fr = _f32(frusta).reshape(-1)
        nv = len(fr) // 24
        self.n_views = nv
        return fr, _u32(view_masks), _u8(view_flags), nv

    def cull(self, frusta, view_masks=None, view_flags=None, flags=0):
        fr, vm, vf, nv = self._views(frusta, view_masks, view_flags)
        self._ck(self._lib.mi_cull(self._h, _ptr(fr, C.c_float), _ptr(vm, C.c_uint32), _ptr(vf, C.c_uint8), nv,
                                   int(flags)))

    def cull_views(self, views, flags=0):
        """views: ctypes array from make_views()."""
        self.n_views = len(views)
        self._ck(self._lib.mi_cull_views(self._h, views, len(views), int(flags)))

    def check_light_mesh_visibility(self, shadow_views, flags=0, want_any=True):
        """mi_check_light_mesh_visibility: the frame's shadow views behind the camera pass.  Returns (per-view 0/1 arrays [n_views, n],
        the rows set_visible() was called on) -- both unpacked from the bitmasks the call delivers in one device wait."""
        nv = 0 if shadow_views is None else len(shadow_views)
        w32 = (self.n + 31) // 32
        masks = np.zeros(max(nv * w32, 1), np.uint32)
        any_ = np.zeros(max(w32, 1), np.uint32) if want_any else None
        self._ck(self._lib.mi_check_light_mesh_visibility(self._h, shadow_views if nv else None, nv, int(flags), _ptr(masks, C.c_uint32),
                                                          _ptr(any_, C.c_uint32)))
        if nv:
            self.n_views = nv
        per_view = np.stack([unpack_bits(masks[v * w32:(v + 1) * w32], self.n) for v in range(nv)]) if nv else np.zeros((0, self.n), np.uint8)
        return per_view, (unpack_bits(any_[:w32], self.n) if want_any else None)

    def propagate_and_cull_views(self, views, flags=0):
        self.n_views = len(views)
        self._ck(self._lib.mi_propagate_and_cull_views(self._h, views, len(views), int(flags)))

    def propagate_and_cull(self, frusta, view_masks=None, view_flags=None, flags=0):
        fr, vm, vf, nv = self._views(frusta, view_masks, view_flags)
        self._ck(self._lib.mi_propagate_and_cull(self._h, _ptr(fr, C.c_float), _ptr(vm, C.c_uint32),
                                                 _ptr(vf, C.c_uint8), nv, int(flags)))

    # results
    def download_global_transforms(self, first_row=0, n=None, want_changed=True):
        n = self.n - first_row if n is None else n
        g = np.zeros(12 * n, np.float32)
        chg = np.zeros((n + 31) // 32, np.uint32) if want_changed else None
        self._ck(self._lib.mi_download_global_transforms(self._h, first_row, n, _ptr(g, C.c_float), _ptr(chg, C.c_uint32)))
        return (g, unpack_bits(chg, n)) if want_changed else g

    def download_changed_global_transforms(self):
        """-> (rows ascending, G[12 * len(rows)]) of the rows whose GlobalTransform changed in the last propagate."""
        cnt = C.c_uint32(0)
        rc = self._lib.mi_download_changed_global_transforms(self._h, None, None, 0, C.byref(cnt))
        if rc not in (MI_OK, MI_ERR_CAPACITY):
            self._ck(rc)
        m = cnt.value
        rows = np.zeros(max(m, 1), np.uint32)
        g = np.zeros(12 * max(m, 1), np.float32)
        self._ck(self._lib.mi_download_changed_global_transforms(self._h, _ptr(rows, C.c_uint32), _ptr(g, C.c_float), m, C.byref(cnt)))
        return rows[:m], g[:12 * m]

    def download_frame_results(self, bufs):
        """mi_download_frame_results into `bufs` (FrameResultBuffers); -> dict of views on its arrays (in place: on the library's
        pinned window -- valid until the next call on this context), cut to the counts."""
        rc = self._lib.mi_download_frame_results(self._h, C.byref(bufs.raw))
        self._ck(rc)
        r = bufs.raw
        want_rows, want_g = r.flags & RESULTS_CHANGED_ROWS, r.flags & RESULTS_CHANGED_GLOBALS

        def arr(ptr, count, dtype):
            if not ptr or not count:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).view(dtype)
        if bufs.in_place:
            out = {"changed_rows": arr(r.changed_rows, r.changed_count if want_rows else 0, np.uint32),
                   "changed_global": arr(r.changed_global12, 12 * r.changed_count if want_g else 0, np.float32),
                   "lists": [arr(bufs._lists[k].rows, bufs._lists[k].count, np.uint32) for k in range(r.n_lists)]}
        else:
            out = {"changed_rows": bufs.changed_rows[:r.changed_count if want_rows else 0],
                   "changed_global": bufs.changed_global[:12 * r.changed_count if want_g else 0],
                   "lists": [bufs.list_rows[k][:bufs._lists[k].count] for k in range(r.n_lists)]}
        out["visible_rows"] = out["lists"][0] if out["lists"] else np.zeros(0, np.uint32)
        if bufs.n_clusters:
            C_ = bufs.n_clusters
            if bufs.in_place:
                out.update(cluster_offsets=arr(r.cluster_offsets, C_ + 1, np.uint32), cluster_counts=arr(r.cluster_counts, 6 * C_, np.uint32).reshape(C_, 6),
                           cluster_indices=arr(r.cluster_indices, int(r.cluster_total), np.uint32))
            else:
                out.update(cluster_offsets=bufs.cluster_offsets, cluster_counts=bufs.cluster_counts.reshape(C_, 6),
                           cluster_indices=bufs.cluster_indices[:r.cluster_total])
            out.update(cluster_total=int(r.cluster_total), farthest_z=float(r.farthest_z))
        return out

    def download_changed_mesh_inputs(self):
        """-> (rows, world_from_local f32[12m] transposed 3x4, culling f32[8m]) for rows whose GlobalTransform changed."""
        cnt = C.c_uint32(0)
        rc = self._lib.mi_download_changed_mesh_inputs(self._h, None, None, None, 0, C.byref(cnt))
        if rc not in (MI_OK, MI_ERR_CAPACITY):
            self._ck(rc)
        m = cnt.value
        rows = np.zeros(max(m, 1), np.uint32)
        wfl = np.zeros(12 * max(m, 1), np.float32)
        cull = np.zeros(8 * max(m, 1), np.float32)
        self._ck(self._lib.mi_download_changed_mesh_inputs(self._h, _ptr(rows, C.c_uint32), _ptr(wfl, C.c_float),
                                                           _ptr(cull, C.c_float), m, C.byref(cnt)))
        return rows[:m], wfl[:12 * m], cull[:8 * m]

    def download_visibility(self, view=0):
        bm = np.zeros((self.n + 31) // 32, np.uint32)
        self._ck(self._lib.mi_download_visibility(self._h, view, _ptr(bm, C.c_uint32)))
        return unpack_bits(bm, self.n)

    def download_view_visibility(self, first_row=0, n=None):
        n = self.n - first_row if n is None else n
        vv = np.zeros(n, np.uint8)
        chg = np.zeros((n + 31) // 32, np.uint32)
        self._ck(self._lib.mi_download_view_visibility(self._h, first_row, n, _ptr(vv, C.c_uint8), _ptr(chg, C.c_uint32)))
        return vv, unpack_bits(chg, n)

    def download_visible_entities(self, view=0, class_bit=0):
        cnt = C.c_uint32(0)
        rc = self._lib.mi_download_visible_entities(self._h, view, class_bit, None, None, 0, C.byref(cnt))
        if rc not in (MI_OK, MI_ERR_CAPACITY):
            self._ck(rc)
        m = cnt.value
        keys = np.zeros(max(m, 1), np.uint64)
        rows = np.zeros(max(m, 1), np.uint32)
        self._ck(self._lib.mi_download_visible_entities(self._h, view, class_bit, _ptr(keys, C.c_uint64),
                                                        _ptr(rows, C.c_uint32), m, C.byref(cnt)))
        return keys[:m], rows[:m]

    # clustering
    def cluster_assign(self, view, pos_range, obj_type=None, layer_mask=None, spot_dir=None, spot_sin_cos=None):
        self.cluster_upload_objects(pos_range, obj_type, layer_mask, spot_dir, spot_sin_cos)
        self.cluster_upload_view(view)
        self.cluster_assign_resident(want_total=True)
        return self.cluster_download(view.n_clusters)

    def cluster_upload_objects(self, pos_range, obj_type=None, layer_mask=None, spot_dir=None, spot_sin_cos=None):
        pr, ty, lm, sd, sc = _f32(pos_range), _u8(obj_type), _u32(layer_mask), _f32(spot_dir), _f32(spot_sin_cos)
        self._ck(self._lib.mi_cluster_upload_objects(self._h, len(pr) // 4, _ptr(pr, C.c_float), _ptr(ty, C.c_uint8),
                                                     _ptr(lm, C.c_uint32), _ptr(sd, C.c_float), _ptr(sc, C.c_float)))
        self._cluster_n = len(pr) // 4

    def cluster_upload_object_layers_hi(self, layer_mask_hi):
        """RenderLayers 32..63 of the uploaded objects (None clears the column); after cluster_upload_objects, which clears it too."""
        hi = _u32(layer_mask_hi)
        n = len(hi) if hi is not None else self._cluster_n
        self._ck(self._lib.mi_cluster_upload_object_layers_hi(self._h, n, _ptr(hi, C.c_uint32)))

    def cluster_select_view(self, slot):
        """Makes view slot `slot` (0 .. 7) the one the cluster calls and MI_CULL_WITH_CLUSTERS refer to (several clustered cameras)."""
        self._ck(self._lib.mi_cluster_select_view(self._h, int(slot)))

    def cluster_upload_view(self, view):
        self._ck(self._lib.mi_cluster_upload_view(self._h, C.byref(view)))

    def cluster_bind_objects_to_rows(self, first_row, n_objects):
        self._ck(self._lib.mi_cluster_bind_objects_to_rows(self._h, first_row, n_objects))

    def cluster_bind_objects_to_row_list(self, rows):
        rows = _u32(rows)
        self._ck(self._lib.mi_cluster_bind_objects_to_row_list(self._h, len(rows), _ptr(rows, C.c_uint32)))

    def cluster_assign_frame(self, config, history, camera_affine, clip_from_view, frustum, w, h, view_layer_mask=1,
                             max_indices=VIEW_CLUSTER_BINDINGS_MAX_INDICES):
        """-> (ClusterView used, active).  `history` (ClusterHistory) is updated in place."""
        cam, cfv, fr = _f32(camera_affine), _f32(clip_from_view), _f32(frustum)
        view = ClusterView()
        active = C.c_uint32(0)
        self._ck(self._lib.mi_cluster_assign_frame(self._h, C.byref(config), C.byref(history), _ptr(cam, C.c_float), _ptr(cfv, C.c_float),
                                                   _ptr(fr, C.c_float), w, h, view_layer_mask, C.c_uint64(max_indices), C.byref(view),
                                                   C.byref(active)))
        return view, bool(active.value)

    def cluster_assign_resident(self, want_total=False):
        tot = C.c_uint64(0)
        self._ck(self._lib.mi_cluster_assign_resident(self._h, C.byref(tot) if want_total else None))
        return tot.value

    def cluster_download(self, n_clusters):
        tot = C.c_uint64(0)
        far = C.c_float(0)
        offsets = np.zeros(n_clusters + 1, np.uint32)
        counts = np.zeros(6 * n_clusters, np.uint32)
        self._ck(self._lib.mi_cluster_download(self._h, _ptr(offsets, C.c_uint32), None, C.c_uint64(0),
                                               _ptr(counts, C.c_uint32), C.byref(tot), C.byref(far)))
        indices = np.zeros(max(tot.value, 1), np.uint32)
        self._ck(self._lib.mi_cluster_download(self._h, None, _ptr(indices, C.c_uint32), C.c_uint64(len(indices)), None,
                                               C.byref(tot), None))
        return offsets, indices[:tot.value], counts.reshape(n_clusters, 6), float(far.value), int(tot.value)

    def cluster_download_bindings(self, n_clusters, remap=None):
        """-> (offsets_and_counts u32[C, 8], index_list u32[total]) in the storage-buffer wire format."""
        rm = _u32(remap)
        tot = C.c_uint64(0)
        oc = np.zeros(8 * n_clusters, np.uint32)
        self._ck(self._lib.mi_cluster_download_bindings(self._h, _ptr(rm, C.c_uint32), 0 if rm is None else len(rm),
                                                        _ptr(oc, C.c_uint32), None, C.c_uint64(0), C.byref(tot)))
        idx = np.zeros(max(tot.value, 1), np.uint32)
        self._ck(self._lib.mi_cluster_download_bindings(self._h, _ptr(rm, C.c_uint32), 0 if rm is None else len(rm), None,
                                                        _ptr(idx, C.c_uint32), C.c_uint64(len(idx)), C.byref(tot)))
        return oc.reshape(n_clusters, 8), idx[:tot.value]

    # batching work-item build
    def batch_upload_rows(self, batch_set, bin_index, input_uniform_index, first_row=0):
        bs, bi, iu = _u32(batch_set), _u32(bin_index), _u32(input_uniform_index)
        self._ck(self._lib.mi_batch_upload_rows(self._h, first_row, len(bs), _ptr(bs, C.c_uint32), _ptr(bi, C.c_uint32),
                                                _ptr(iu, C.c_uint32)))

    def batch_upload_sets(self, set_indexed, bin_table_offset, bin_table, meta_offset, bin_metadata):
        si = np.ascontiguousarray(set_indexed, np.uint8)
        bto, bt, mo = _u32(bin_table_offset), _u32(bin_table), _u32(meta_offset)
        bm = np.ascontiguousarray(bin_metadata, np.uint32).reshape(-1)
        self._ck(self._lib.mi_batch_upload_sets(self._h, len(si), _ptr(si, C.c_uint8), _ptr(bto, C.c_uint32), _ptr(bt, C.c_uint32),
                                                _ptr(mo, C.c_uint32), bm.ctypes.data_as(C.c_void_p) if bm.size else None))

    def batch_upload_row_bins(self, kind, cpu_bin, first_row=0):
        """Per row: MI_BATCH_ROW_* and, for batchable / unbatchable rows, the bin (gpu_preprocessing.rs:2135-2357)."""
        k = np.ascontiguousarray(kind, np.uint8)
        b = _u32(cpu_bin)
        self._ck(self._lib.mi_batch_upload_row_bins(self._h, first_row, len(k), _ptr(k, C.c_uint8), _ptr(b, C.c_uint32)))

    def batch_upload_bins(self, unbatchable_indexed, batchable_indexed):
        u = np.ascontiguousarray(unbatchable_indexed, np.uint8)
        b = np.ascontiguousarray(batchable_indexed, np.uint8)
        self._ck(self._lib.mi_batch_upload_bins(self._h, len(u), _ptr(u, C.c_uint8) if len(u) else None, len(b),
                                                _ptr(b, C.c_uint8) if len(b) else None))

    def batch_build(self, view=0, class_bit=0, initial=None, no_indirect_drawing=False):
        """initial: 7 ints (work_item_index[2], indirect_parameters_index[2], batch_set_index[2], output_mesh_uniform_index)."""
        ini = None if initial is None else (C.c_uint32 * 7)(*[int(x) for x in initial])
        if no_indirect_drawing:
            self._ck(self._lib.mi_batch_build_phase(self._h, view, class_bit, ini, C.c_uint32(BATCH_NO_INDIRECT_DRAWING)))
        else:
            self._ck(self._lib.mi_batch_build(self._h, view, class_bit, ini))

    def batch_sorted_build(self, items, automatic_batching=True, no_indirect_drawing=False, no_gpu_preprocessing=False, initial=None):
        """items u32[n, 4] = (input_index, batch_set_key, bin_key, flags) in the phase's sorted order."""
        it = np.ascontiguousarray(items, np.uint32).reshape(-1, 4)
        ini = None if initial is None else (C.c_uint32 * 7)(*[int(x) for x in initial])
        flags = (SORTED_AUTOMATIC_BATCHING if automatic_batching else 0) | (SORTED_NO_INDIRECT_DRAWING if no_indirect_drawing else 0) \
            | (SORTED_NO_GPU_PREPROCESSING if no_gpu_preprocessing else 0)
        self._ck(self._lib.mi_batch_sorted_build(self._h, len(it), it.ctypes.data_as(C.c_void_p) if len(it) else None, ini, C.c_uint32(flags)))

    def batch_download(self):
        """-> dict: work_items / metadata / batch_sets (lists indexed by mesh class, u32 arrays), records u32[k, 8], unbatchable
        u32[k, 2], batches u32[k, 6] (sorted builds), totals, bin_metadata."""
        tot = (C.c_uint32 * 9)()
        self._ck(self._lib.mi_batch_download_totals(self._h, tot))

        def get(what, cls, words):
            cnt = C.c_uint32(0)
            self._lib.mi_batch_download(self._h, what, cls, None, 0, C.byref(cnt))  # length only (capacity error expected)
            out = np.zeros((max(cnt.value, 1), words), np.uint32)
            self._ck(self._lib.mi_batch_download(self._h, what, cls, out.ctypes.data_as(C.c_void_p), len(out), C.byref(cnt)))
            return out[:cnt.value]
        return dict(work_items=[get(0, c, 2) for c in range(2)], metadata=[get(1, c, 5) for c in range(2)],
                    batch_sets=[get(2, c, 2) for c in range(2)], records=get(3, 0, 8), bin_metadata=get(4, 0, 3),
                    unbatchable=get(5, 0, 2), batches=get(6, 0, 6),
                    totals=dict(work_item_len=[tot[0], tot[1]], indirect_parameters_len=[tot[2], tot[3]],
                                batch_set_len=[tot[4], tot[5]], data_buffer_len=int(tot[6])))

    # interop / timing
    def bind_visibility_output(self, device_ptr, words_per_view, word_offset):
        self._ck(self._lib.mi_bind_visibility_output(self._h, C.c_void_p(device_ptr), C.c_uint64(words_per_view),
                                                     C.c_uint64(word_offset)))

    def exchange_configure(self, comms, fn_all_gather, bufs, words_per_view, word_offset, block_bytes, rank):
        """comms: one ncclComm_t handle or a list of them (used round-robin by frame); None / empty switches the exchange
        off.  bufs: list of device pointers of the gathered buffers."""
        if comms is None:
            comms = []
        elif not isinstance(comms, (list, tuple)):
            comms = [comms]
        comms = [c for c in comms if c]
        bufs = list(bufs or [])
        arr = (C.c_void_p * max(len(bufs), 1))(*bufs)
        carr = (C.c_void_p * max(len(comms), 1))(*comms)
        self._ck(self._lib.mi_exchange_configure_multi(self._h, carr, len(comms), C.c_void_p(fn_all_gather), arr, len(bufs),
                                                       C.c_uint64(words_per_view), C.c_uint64(word_offset),
                                                       C.c_uint64(block_bytes), C.c_uint32(rank)))

    def exchange_set_mode(self, mode):
        """0 = MI_EXCHANGE_SIMPLE (default), 1 = MI_EXCHANGE_PIPELINED; before exchange_configure."""
        self._ck(self._lib.mi_exchange_set_mode(self._h, int(mode)))

    def exchange_download(self, n_bytes):
        """The most recent frame's gathered buffer on the host ([rank][view][word] as uint64), one copy behind its all-gather."""
        out = np.zeros((int(n_bytes) + 7) // 8, np.uint64)
        self._ck(self._lib.mi_exchange_download(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint64(int(n_bytes))))
        return out

    def exchange_last(self, wait=True):
        p = C.c_void_p()
        self._ck(self._lib.mi_exchange_last(self._h, C.byref(p), 1 if wait else 0))
        return p.value

    def device_buffer(self, which):
        p = C.c_void_p()
        nbytes = C.c_uint64(0)
        self._ck(self._lib.mi_device_buffer(self._h, which, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def synchronize(self):
        self._ck(self._lib.mi_synchronize(self._h))

    def timer_begin(self):
        self._ck(self._lib.mi_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float(0)
        self._ck(self._lib.mi_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        self._ck(self._lib.mi_profile_enable(self._h, 1 if on else 0))

    def profile_filter(self, kernel_names=None):
        """Only bracket the named kernels (None = all)."""
        mask = 0
        if kernel_names:
            k = 0
            while True:
                name = self._lib.mi_profile_kernel_name(k)
                if not name:
                    break
                if name.decode() in kernel_names:
                    mask |= 1 << k
                k += 1
        self._ck(self._lib.mi_profile_filter(self._h, C.c_uint64(mask)))

    def profile_sample(self, every_n=1):
        self._ck(self._lib.mi_profile_sample(self._h, int(every_n)))

    def profile_burst(self, first_n=0):
        self._ck(self._lib.mi_profile_burst(self._h, int(first_n)))

    def profile_read(self):
        n = C.c_uint32(64)
        launches = (C.c_uint64 * 64)()
        ms = (C.c_double * 64)()
        self._ck(self._lib.mi_profile_read(self._h, C.byref(n), launches, ms))
        out = {}
        for k in range(n.value):
            name = self._lib.mi_profile_kernel_name(k)
            if name and launches[k]:
                out[name.decode()] = {"launches": int(launches[k]), "total_ms": float(ms[k]),
                                      "avg_us": 1e3 * float(ms[k]) / int(launches[k])}
        return out

    def debug_set_tile_mode(self, mode):
        """What the next upload_hierarchy plans: 0 subtree tiles where they fit, 1 level by level whatever the shape, 2 as 0, 3 as 0 with the
        streamed-level thresholds at their test values (2^20 / 2^21 rows) -- test / bench hook."""
        self._ck(self._lib.mi_debug_set_tile_mode(self._h, int(mode)))

    def debug_tree_trace(self, n_tiles=0):
        """Development hook: n_tiles == 0 enables the per-tile phase timestamps of the light tile kernel; otherwise returns
        them as an (n_tiles, 8) uint64 array of 10 ns ticks; n_tiles < 0 switches the stamps off."""
        if not n_tiles:
            self._ck(self._lib.mi_debug_tree_trace(self._h, 1, None, 0))
            return None
        if n_tiles < 0:  # off again
            self._ck(self._lib.mi_debug_tree_trace(self._h, 0, None, 0))
            return None
        out = np.zeros((n_tiles, 8), dtype=np.uint64)
        self._ck(self._lib.mi_debug_tree_trace(self._h, 0, out.ctypes.data_as(C.c_void_p), int(n_tiles)))
        return out

    def debug_tile_plan(self):
        """-> dict(launches, tiles, chain_tiles, bands) of the current hierarchy's tile plan (test hook)."""
        v = [C.c_uint32(0) for _ in range(4)]
        self._ck(self._lib.mi_debug_tile_plan(self._h, *[C.byref(x) for x in v]))
        return dict(launches=v[0].value, tiles=v[1].value, chain_tiles=v[2].value, bands=v[3].value)

    def debug_exchange_times(self, reset=False):
        """-> dict(frames, begin_ns, wait_ns, end_ns, worker_ns): what the exchange cost the calling thread (wait_ns = device back-pressure)."""
        out = (C.c_double * 5)()
        self._ck(self._lib.mi_debug_exchange_times(self._h, out, 1 if reset else 0))
        return dict(frames=out[0], begin_ns=out[1], wait_ns=out[2], end_ns=out[3], worker_ns=out[4])

    def debug_tile_groups(self, cap_tiles=1 << 20):
        """-> (groups [n][4]: first tile, tiles, chain tiles, deep; tiles [n][3]: levels, rows, chain length | 0x100 roots) of the tile plan."""
        g = np.zeros(4 * 64, np.uint32)
        ng = C.c_uint32(0)
        t = np.zeros(3 * cap_tiles, np.uint32)
        self._ck(self._lib.mi_debug_tile_groups(self._h, _ptr(g, C.c_uint32), 64, C.byref(ng), _ptr(t, C.c_uint32), cap_tiles))
        groups = g.reshape(-1, 4)[:ng.value]
        n_tiles = int((groups[:, 0] + groups[:, 1]).max()) if len(groups) else 0
        return groups, t.reshape(-1, 3)[:n_tiles]

    def debug_set_sorted_one_wg_limit(self, items):
        """Sorted phases up to `items` long take the single-workgroup kernel; 0 = always the tiled form (test / bench hook)."""
        self._ck(self._lib.mi_debug_set_sorted_one_wg_limit(self._h, C.c_uint32(int(items) & 0xFFFFFFFF)))

    def debug_strip_plan(self):
        """(rounds per strip, cone rounds per strip, rounds of the whole table) of the current plan's strips; empty arrays without strips."""
        n, tot = C.c_uint32(0), C.c_uint32(0)
        self._ck(self._lib.mi_debug_strip_plan(self._h, None, None, 0, C.byref(n), C.byref(tot)))
        r, cr = np.zeros(max(n.value, 1), np.uint32), np.zeros(max(n.value, 1), np.uint32)
        self._ck(self._lib.mi_debug_strip_plan(self._h, _ptr(r, C.c_uint32), _ptr(cr, C.c_uint32), n.value, C.byref(n), C.byref(tot)))
        return r[:n.value], cr[:n.value], tot.value

    def debug_set_tile_pretest(self, mode):
        """0 = the light tiles test their flags first when few rows changed (default), 1 = never, 2 = always (test / bench hook)."""
        self._ck(self._lib.mi_debug_set_tile_pretest(self._h, int(mode)))

    def debug_set_chunked_frames(self, mode):
        """Dense windows that carry the whole table go out in pieces and their GlobalTransforms are fetched ahead of the frame: 0 = tables of 262144 rows and more (default), 1 = never, 2 = any row count, fetching ahead at once."""
        self._ck(self._lib.mi_debug_set_chunked_frames(self._h, int(mode)))

    def debug_chunked_counts(self):
        """(dense windows that went out as pieces of a sequence, result downloads that handed out GlobalTransforms fetched ahead of an
        all-rows frame, ... written ahead by the indexed window of a changed-rows frame)."""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._ck(self._lib.mi_debug_chunked_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def debug_set_walk_inrow(self, mode):
        """0 = objects bound to a row range are walked by the frame kernel's own row workgroups (default), 1 = by extra workgroups."""
        self._ck(self._lib.mi_debug_set_walk_inrow(self._h, int(mode)))

    def debug_set_tree_cull(self, mode):
        """0 = the all-dirty hierarchy frame culls inside its tile launches when it has one view (default), 1 = never, 2 = whenever that applies (test / bench hook)."""
        self._ck(self._lib.mi_debug_set_tree_cull(self._h, int(mode)))

    def debug_set_row_summary(self, mode):
        """0 = 64 aligned rows that agree in Aabb / flags / RenderLayers read a 32-byte summary instead of their columns (default),
        1 = off (test / bench hook; results are identical)."""
        self._ck(self._lib.mi_debug_set_row_summary(self._h, int(mode)))

    def debug_set_static_cull_order(self, mode):
        """0 = cull-only frames of a static scene run over the cell order from the second eligible frame on (default), 1 = never, 2 = at once, any row count, 3 = as 2 with the list kernels' runs capped at three (long runs, as beyond 16.7 M rows)."""
        self._ck(self._lib.mi_debug_set_static_cull_order(self._h, int(mode)))

    def debug_cluster_download_unjoined(self, n_clusters, capacity):
        """(offsets, indices) as the last fill that RAN left them; a pending fill is not launched (test hook)."""
        offsets = np.zeros(n_clusters + 1, np.uint32)
        indices = np.zeros(max(capacity, 1), np.uint32)
        tot = C.c_uint64(0)
        self._ck(self._lib.mi_debug_cluster_download_unjoined(self._h, _ptr(offsets, C.c_uint32), _ptr(indices, C.c_uint32), C.c_uint64(len(indices)),
                                                              C.byref(tot)))
        return offsets, indices[:tot.value]

    def debug_static_cull_counts(self):
        """(cell orders built, frames that ran over one)."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self._ck(self._lib.mi_debug_static_cull_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def debug_set_sphere_path(self, mode):
        """0 = the world-sphere cull path from the second quiet frame on (default), 1 = never, 2 = at once (test / bench hook)."""
        self._ck(self._lib.mi_debug_set_sphere_path(self._h, int(mode)))

    def debug_logf(self, x):
        x = _f32(x)
        out = np.zeros_like(x)
        self._ck(self._lib.mi_debug_logf(self._h, _ptr(x, C.c_float), _ptr(out, C.c_float), len(x)))
        return out


def unpack_bits(words, n):
    """u32 little-endian bit words -> uint8[n] of 0/1"""
    if words is None:
        return None
    b = np.unpackbits(words.view(np.uint8), bitorder="little")
    return b[:n].copy()
