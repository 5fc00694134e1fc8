"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, refuses to run without a gfx950 device (no CPU fallback), and its pure-host helpers
(camera frusta, per-view cluster constants, hierarchy flattening) agree bit-for-bit with the oracle."""
import math
import os
import re

import numpy as np
import pytest

import bevy_amd
from bevy_amd import api, build as mi_build, workloads as W
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def lib():
    mi_build.build()
    return api.load_library()


def header_symbols(name="bevy_mi355x.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bevy_mi355x.h but not exported"
    assert sorted(api.ABI_SYMBOLS) == syms, "bevy_amd.api.ABI_SYMBOLS out of sync with the header"
    assert lib.mi_abi_version() == 3


def test_debug_header_declares_every_hook(lib):
    """Instrumentation and test hooks live in include/bevy_mi355x_debug.h, not in the boundary header -- and nothing the library
    exports under the mi_ prefix is declared in neither."""
    import subprocess
    dbg = header_symbols("bevy_mi355x_debug.h")
    assert sorted(api.DEBUG_SYMBOLS) == dbg
    for s in dbg:
        assert hasattr(lib, s), f"{s} declared in include/bevy_mi355x_debug.h but not exported"
    assert not set(dbg) & set(header_symbols()), "a debug hook is also declared in the boundary header"
    out = subprocess.run(["nm", "-D", "--defined-only", api.lib_path()], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("mi_")}
    assert exported == set(dbg) | set(header_symbols()), exported ^ (set(dbg) | set(header_symbols()))


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "bevy_mi355x.h")).read()
    assert len(re.findall(r"\.rs:\d+", text)) >= 30


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.MiError) as e:
        api.Context(0)
    assert e.value.code in (api.MI_ERR_DEVICE, api.MI_ERR_INVALID_ARG)
    assert "HIP device" in str(e.value) or "gfx950" in str(e.value)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under bevy_amd/ or include/ may reference it."""
    for base in ("bevy_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h")):
                    src = open(os.path.join(dp, f)).read()
                    assert "oracle_lib" not in src and "bevy_oracle" not in src and "libbevy_oracle" not in src, f


# ---- host helpers vs oracle -------------------------------------------------------------------

def cameras():
    yield W.many_cubes_camera(0)
    yield W.many_cubes_camera(17, yaw=1.3, position=(3.0, -2.0, 11.0))
    q = W.quat_axis("y", 0.7)
    cam = W.affine_from_quat_translation(q, (1.0, 2.0, 3.0)).astype(np.float64)
    cam[:9] *= 1.7  # uniformly scaled camera
    yield cam.astype(F)


def test_compute_frustum_matches_oracle(lib):
    for cam in cameras():
        for fov, aspect, near, far in [(W.CAMERA_FOV, W.CAMERA_ASPECT, 0.1, 1000.0), (math.radians(90.0), 1.0, 1.0, 100.0)]:
            cfv = api.perspective_clip_from_view(fov, aspect, near)
            assert np.array_equal(cfv, O.perspective_infinite_reverse(fov, aspect, near))
            got = api.compute_frustum(cfv, cam, far)
            exp = O.compute_frustum_perspective(fov, aspect, near, far, cam)
            assert got.tobytes() == exp.tobytes()


def test_cluster_dims_match_oracle(lib):
    for w, h in [(1920, 1080), (1, 1), (7, 4999), (640, 480), (99, 3)]:
        assert api.cluster_dimensions_fixed_z(4096, 24, w, h) == O.cluster_dimensions_fixed_z(4096, 24, w, h)


def ortho_clip_from_view(left, right, bottom, top, near, far):
    """Mat4::orthographic_rh with near/far swapped for reversed z (OrthographicProjection::get_clip_from_view)."""
    n, f = far, near
    rcp_w, rcp_h, r = 1.0 / (right - left), 1.0 / (top - bottom), 1.0 / (n - f)
    m = np.zeros(16, F)
    m[0] = rcp_w + rcp_w; m[5] = rcp_h + rcp_h; m[10] = r
    m[12] = -(left + right) * rcp_w; m[13] = -(top + bottom) * rcp_h; m[14] = r * n; m[15] = 1.0
    return m


@pytest.mark.parametrize("ortho", [False, True])
def test_cluster_view_build_matches_oracle(lib, ortho):
    for cam in cameras():
        if ortho:
            cfv = ortho_clip_from_view(-10.0, 10.0, -5.6, 5.6, 0.1, 1000.0)
        else:
            cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, 0.1)
        fr = api.compute_frustum(cfv, cam, 1000.0)
        for req, fsd, far_z in [((16, 9, 24), 5.0, 1000.0), ((17, 9, 24), 5.0, 77.5), ((4, 4, 1), 5.0, 50.0)]:
            view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, req, fsd, far_z, 1)
            ov = O.cluster_view_setup(cam, cfv, fr, 1920, 1080, req, fsd, far_z, 1)
            assert tuple(view.dims) == tuple(ov.dims) and tuple(view.tile_size) == tuple(ov.tile_size)
            for name in ("near_", "far_", "view_from_world_scale_max", "is_orthographic"):
                assert getattr(view, name) == getattr(ov, name), name
            for name in ("cluster_factors", "view_from_world", "clip_from_view", "view_from_clip", "view_from_world_scale", "frustum"):
                assert bytes(getattr(view, name)) == bytes(getattr(ov, name)), name
            dx, dy, dz = view.dims
            planes = keep[0]
            assert planes[:4 * (dx + 1)].tobytes() == bytes(ov.x_planes)[:16 * (dx + 1)]
            assert planes[4 * (dx + 1):4 * (dx + dy + 2)].tobytes() == bytes(ov.y_planes)[:16 * (dy + 1)]
            assert planes[4 * (dx + dy + 2):].tobytes() == bytes(ov.z_planes)[:16 * (dz + 1)]
            spheres = keep[1].reshape(-1, 4)
            for (x, y, z) in [(0, 0, 0), (dx - 1, dy - 1, dz - 1), (dx // 2, dy // 3, dz // 2), (1 % dx, 0, dz - 1)]:
                exp = O.cluster_aabb_sphere(ov, x, y, z)
                assert spheres[(y * dx + x) * dz + z].tobytes() == exp.tobytes()


def test_hierarchy_sort_orders_levels(lib):
    rng = np.random.default_rng(42)
    n = 5000
    # random forest in arbitrary row order: parent index is any earlier node in a hidden order
    hidden = rng.permutation(n)
    parent = np.full(n, O.NO_PARENT, np.uint32)
    for k in range(1, n):
        if rng.random() < 0.02:
            continue  # another root / flat entity
        parent[hidden[k]] = hidden[rng.integers(0, k)]
    new_to_old, pidx, offs = api.hierarchy_sort(parent)
    assert sorted(new_to_old.tolist()) == list(range(n))
    assert offs[0] == 0 and offs[-1] == n and np.all(np.diff(offs.astype(np.int64)) > 0)
    old_to_new = np.empty(n, np.int64); old_to_new[new_to_old] = np.arange(n)
    for l in range(len(offs) - 1):
        rows = np.arange(offs[l], offs[l + 1])
        if l == 0:
            assert np.all(pidx[rows] == O.NO_PARENT)
        else:
            p = pidx[rows].astype(np.int64)
            assert np.all((p >= offs[l - 1]) & (p < offs[l])) and np.all(np.diff(p) >= 0)
    # same tree: parent relation preserved
    has = parent != O.NO_PARENT
    assert np.array_equal(pidx[old_to_new[np.nonzero(has)[0]]], old_to_new[parent[has]].astype(np.uint32))
    # stable: siblings keep caller order
    for p_old in np.unique(parent[has])[:50]:
        kids_old = np.nonzero(parent == p_old)[0]
        assert np.all(np.diff(old_to_new[kids_old]) > 0)


def test_hierarchy_sort_rejects_cycles(lib):
    with pytest.raises(api.MiError) as e:
        api.hierarchy_sort(np.array([O.NO_PARENT, 2, 1], np.uint32))
    assert e.value.code == api.MI_ERR_MALFORMED_HIERARCHY
    with pytest.raises(api.MiError):
        api.hierarchy_sort(np.array([O.NO_PARENT, 7], np.uint32))


def test_gen_tree_is_level_ordered():
    tr = W.gen_tree(6, 4)
    assert tr["n"] == sum(4 ** i for i in range(6))
    n2o, pidx, offs = api.hierarchy_sort(tr["parent"])
    assert np.array_equal(n2o, np.arange(tr["n"])) and np.array_equal(pidx, tr["parent"])
    assert np.array_equal(offs, tr["level_offsets"])


def test_header_is_plain_c_and_wire_structs_have_the_reference_sizes(tmp_path):
    """include/bevy_mi355x.h compiles as C99 on its own (no torch / HIP / C++ types at the boundary) and the structs that
    are GPU wire formats in the reference have the sizes of their #[repr(C)] originals: PreprocessWorkItem 8 B,
    IndirectParametersMetadata 20 B, IndirectBatchSet 8 B (gpu_preprocessing.rs:783-965), GpuBinMetadata 12 B
    (mesh_preprocess_types.wesl:130-149); mi_view is the 144-byte block the kernels read."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "bevy_mi355x.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(mi_view), sizeof(mi_preprocess_work_item), sizeof(mi_indirect_parameters_metadata), '
                   'sizeof(mi_indirect_batch_set), sizeof(mi_bin_metadata), sizeof(mi_batch_set_record), sizeof(mi_batch_initial), '
                   'sizeof(mi_batch_totals)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [144, 8, 20, 8, 12, 32, 28, 36]  # mi_batch_totals: 9 words since n_unbatchable was added


def test_python_constants_mirror_the_header_defines():
    """bevy_amd/__init__.py repeats the header's flag values for the harness; a define that moves must not leave them behind."""
    import re
    import bevy_amd as B
    text = open(os.path.join(ROOT, "include", "bevy_mi355x.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+MI_([A-Z0-9_]+)\s+(0x[0-9A-Fa-f]+|\d+)u?\b", text)}
    checked = 0
    for name in dir(B):
        if name.isupper() and name in defines and isinstance(getattr(B, name), int):
            assert getattr(B, name) == defines[name], (name, getattr(B, name), defines[name])
            checked += 1
    assert checked >= 12, checked
    for must in ("CULL_BEGIN_FRAME", "CULL_END_FRAME", "CULL_MORE_FRAMES", "CULL_WITH_CLUSTERS", "CULL_CHANGED_ROWS", "PROPAGATE_ALL_DIRTY",
                 "PROPAGATE_STATIC_OPT"):
        assert must in defines and getattr(B, must) == defines[must], must


def test_oracle_mark_dirty_trees_is_the_ancestor_closure():
    """mark_dirty_trees (systems.rs:111-306) marks exactly the changed rows and all their ancestors -- the definition the device's
    climb (plain byte marks, early stop at a marked node) is held to through the oracle.  Brute force over random forests."""
    import oracle_lib as O
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(1, 400))
        parent = np.full(n, 0xFFFFFFFF, np.uint32)
        for k in range(1, n):
            if rng.random() > 0.1:
                parent[k] = rng.integers(max(0, k - int(rng.choice([1, 4, 60]))), k)
        changed = (rng.random(n) < rng.choice([0.0, 0.02, 0.3])).astype(np.uint8)
        want = np.zeros(n, np.uint8)
        for k in np.nonzero(changed)[0]:
            r = int(k)
            while r != 0xFFFFFFFF and not want[r]:
                want[r] = 1
                r = int(parent[r])
        got = O.mark_dirty_trees(parent, changed)
        assert np.array_equal(got != 0, want != 0), trial


def test_hierarchy_advice_keeps_only_narrow_hierarchies_on_the_host(lib):
    """mi_hierarchy_advice_for (pure host code): the planner's question "is this hierarchy one wave's walk?" answered before anything is
    uploaded.  The reference's `chain` (2 500 levels of one node) and a rope stay with the stock systems -- levels x 0.32 us on the device
    against 20 ns per node on a core --; everything with rows to run side by side is the device's, whatever its depth."""
    from bevy_amd import api, workloads as W
    chain = api.hierarchy_advice(W.hierarchy_shape("chain")["level_offsets"])
    assert chain["plan"] == 2 and chain["keep_on_host"] == 1 and chain["n_levels"] == 2500 and chain["widest_level"] == 1
    assert chain["est_device_us"] > 10 * chain["est_host_us"]
    ropes = api.hierarchy_advice(W.hierarchy_shape("ropes")["level_offsets"])
    assert ropes["plan"] == 2 and ropes["keep_on_host"] == 1 and ropes["widest_level"] <= 64
    for name in ("humanoids_active", "deep_tree", "large_tree", "wide_tree", "update_leaves"):
        a = api.hierarchy_advice(W.hierarchy_shape(name)["level_offsets"])
        assert a["plan"] == 1 and a["keep_on_host"] == 0, (name, a)
    flat = api.hierarchy_advice([0, 1_000_000])
    assert flat["plan"] == 0 and flat["keep_on_host"] == 0
    # 16 levels of one node fit a tile: not the one-wave plan; 17 do not
    assert api.hierarchy_advice(list(range(17)))["plan"] == 1 and api.hierarchy_advice(list(range(18)))["plan"] == 2
    # an empty level is refused by the one-wave plan, a decreasing offset by the function
    assert api.hierarchy_advice([0, 1, 1] + list(range(2, 30)))["plan"] == 1
    with pytest.raises(api.MiError):
        api.hierarchy_advice([0, 5, 3])
