"""bevy_amd -- MI355X-native render-prep path for Bevy (transform propagate -> frustum cull ->
light-cluster assign) as hand-written HIP kernels behind a C ABI (include/bevy_mi355x.h).

The product is `libbevy_mi355x.so` (csrc/); this package is only the plumbing that tests and
bench.py use to reach the C ABI from Python (ctypes).  There is NO CPU fallback: every compute entry
point needs a gfx950 device and raises `MiError` otherwise.
"""
from .api import (  # noqa: F401
    ClusterView,
    Context,
    MiError,
    cluster_dimensions_fixed_z,
    cluster_view_build,
    compute_frustum,
    hierarchy_sort,
    lib_path,
    load_library,
    perspective_clip_from_view,
)
from . import workloads  # noqa: F401

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
VIEW_KIND_CASCADE = 0x02 | 0x04 | 0x08
VIEW_KIND_CUBE_FACE_OR_SPOT = 0x02 | 0x08 | 0x10
VISIBILITY_INHERITED, VISIBILITY_HIDDEN, VISIBILITY_VISIBLE, VISIBILITY_NONE = 0, 1, 2, 0x80
CULL_BEGIN_FRAME = 0x1
CULL_END_FRAME = 0x2
CULL_MORE_FRAMES = 0x4  # another cull frame follows at once: defer the compaction into its launch (MI_CULL_MORE_FRAMES)
CULL_WITH_CLUSTERS = 0x8  # the frame also assigns the row-bound lights to clusters (MI_CULL_WITH_CLUSTERS)
CULL_CLUSTERS_CONCURRENT = 0x10  # ... on the cluster stream, next to the frame kernel (MI_CULL_CLUSTERS_CONCURRENT)
CULL_CHANGED_ROWS = 0x20  # mi_propagate_and_cull: only rows whose change byte is set are propagated (MI_CULL_CHANGED_ROWS)
CULL_STATIC_OPT = 0x40  # mi_propagate_and_cull with a hierarchy: StaticTransformOptimizations for the propagate part (MI_CULL_STATIC_OPT)
PROPAGATE_ALL_DIRTY = 0x1
PROPAGATE_STATIC_OPT = 0x2
NO_PARENT = 0xFFFFFFFF
