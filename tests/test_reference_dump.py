"""Oracle (CPU) and HIP path (GPU) against tests/golden/reference_dump.migd -- what Bevy's OWN systems produced on the inputs of
the committed fixtures (tools/golden_dump/README.md).

The dump needs a Rust toolchain to produce and the development image has none, so every test here skips while the file is absent.
The round trip of the container and the consistency of the exported inputs with the committed fixtures are tested regardless."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools", "golden_dump"))
import migd  # noqa: E402
import oracle_lib as O  # noqa: E402

DUMP = os.path.join(HERE, "golden", "reference_dump.migd")
INPUTS = os.path.join(ROOT, "tools", "golden_dump", "inputs.migd")
needs_dump = pytest.mark.skipif(not os.path.exists(DUMP), reason="tests/golden/reference_dump.migd absent: no Rust toolchain in "
                                "this image; produce it with tools/golden_dump (README.md there)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_container_round_trip(tmp_path):
    arrays = {"a.u8": np.arange(7, dtype=np.uint8), "b.u32": np.array([1, 2 ** 32 - 1], np.uint32),
              "c.f32": np.array([0.0, -0.0, np.nan, 1.5], np.float32), "d.u64": np.array([2 ** 63 + 5], np.uint64), "e.empty": np.zeros(0, np.float32)}
    p = str(tmp_path / "x.migd")
    migd.write(p, arrays)
    back = migd.read(p)
    assert list(back) == list(arrays)
    for k in arrays:
        assert back[k].dtype == arrays[k].dtype and back[k].tobytes() == arrays[k].tobytes()


def test_exported_inputs_are_the_fixture_inputs():
    inp = migd.read(INPUTS)
    flat, tree, cl = (np.load(os.path.join(HERE, "golden", f)) for f in ("flat_frame_777.npz", "tree_6x3.npz", "cluster_3000.npz"))
    for k in ("translation", "rotation", "scale", "aabb_center", "aabb_half", "flags", "layers"):
        assert inp["flat." + k].tobytes() == flat[k].tobytes(), k
    for k in ("parent", "translation", "rotation", "scale"):
        assert inp["tree." + k].tobytes() == tree[k].tobytes(), k
    assert inp["cluster.lights_pos_range"].tobytes() == cl["lights"].tobytes()
    assert inp["cluster.camera"].tobytes() == cl["camera"].tobytes()
    import export_inputs
    for k, v in export_inputs.spot_scene().items():  # case 4 is seeded in the export script itself
        assert inp[k].dtype == v.dtype and inp[k].tobytes() == v.tobytes(), k
    kind = inp["cluster2.type"]
    assert np.all(np.diff(kind.astype(np.int32)) >= 0) and 0 < int(kind.sum()) < kind.size  # points, then spots: the gather order


def _spot_back(inp):
    """GlobalTransform::back() of every light of case 4, the way the reference forms it: (matrix3 * Vec3::Z).normalize() with glam's
    normalize = v * (1 / length) (global_transform.rs: local_z; oracle/bevy_oracle.c:306-310 restates the same for the camera)."""
    n = inp["cluster2.type"].size
    pr = inp["cluster2.lights_pos_range"].reshape(n, 4)
    rc, g, _ = O.propagate_transforms(np.full(n, 0xFFFFFFFF, np.uint32), np.ascontiguousarray(pr[:, :3]).reshape(-1), inp["cluster2.rotation"],
                                      np.ones(3 * n, np.float32))
    assert rc == 0
    z = g.reshape(n, 12)[:, 6:9].astype(np.float32)
    f = np.float32
    dot = (z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1]).astype(f) + (z[:, 2] * z[:, 2]).astype(f)
    recip = (f(1.0) / np.sqrt(dot.astype(f)).astype(f)).astype(f)
    return (z * recip[:, None]).astype(f)


def _cameras(inp, ref, key):
    """The cameras' GlobalTransforms: the ones the reference's systems saw when the dump carries them (the tool builds a camera's
    Transform with `Transform::from_matrix`, whose decomposition need not give the fixture's affine back bit for bit), else the
    fixture's.  Either way they are the same cameras."""
    fixture = inp[key]
    seen = None if ref is None else ref.get(key.replace("cameras", "camera").replace(".camera", ".camera_global"))
    if seen is None:
        return fixture
    assert seen.shape == fixture.shape and np.allclose(seen, fixture, rtol=1e-5, atol=1e-5), key
    return seen


def _spot_views(inp, ref=None):
    fov, aspect, near, far = inp["camera.fov_aspect_near_far"]
    cfv = O.perspective_infinite_reverse(np.float32(fov), aspect, near)
    w, h, dx, dy, dz = (int(x) for x in inp["cluster.screen_dims_z"])
    first, far_z = inp["cluster.first_slice_depth_far_z"]
    views = []
    for cam in _cameras(inp, ref, "cluster2.cameras").reshape(-1, 12):
        fr = O.compute_frustum_perspective(np.float32(fov), aspect, near, far, cam)
        views.append((cam, cfv, fr, O.cluster_view_setup(cam, cfv, fr, w, h, (dx, dy, dz), float(first), float(far_z))))
    return views, (w, h, (dx, dy, dz), float(first), float(far_z))


def test_case_4_is_a_real_case():
    """Without the dump: the spot / two-camera inputs do exercise the cone test -- the spot lights reach fewer clusters than the same
    lights taken as point lights -- and the two cameras see different lists."""
    inp = migd.read(INPUTS)
    back = _spot_back(inp)
    assert np.allclose(np.linalg.norm(back, axis=1), 1.0, atol=1e-6)
    sin_cos = np.stack([np.sin(inp["cluster2.outer_angle"]), np.cos(inp["cluster2.outer_angle"])], axis=1).astype(np.float32).reshape(-1)
    views, _ = _spot_views(inp)
    totals = []
    for _, _, _, view in views:
        off, idx, _, _, total = O.assign_objects_to_clusters(view, inp["cluster2.lights_pos_range"], obj_type=inp["cluster2.type"],
                                                            spot_dir=back.reshape(-1), spot_sin_cos=sin_cos)
        _, _, _, _, total_points = O.assign_objects_to_clusters(view, inp["cluster2.lights_pos_range"])
        assert 0 < total < total_points
        totals.append((total, idx.tobytes()))
    assert totals[0] != totals[1]


def _flat_oracle(inp, ref=None):
    fov, aspect, near, far = inp["camera.fov_aspect_near_far"]
    cams = _cameras(inp, ref, "flat.cameras").reshape(-1, 12)
    frusta = np.concatenate([O.compute_frustum_perspective(np.float32(fov), aspect, near, far, c) for c in cams])
    n = inp["flat.flags"].size
    g, vv, vis, _ = O.full_frame(inp["flat.translation"], inp["flat.rotation"], inp["flat.scale"], inp["flat.aabb_center"],
                                 inp["flat.aabb_half"], inp["flat.flags"], inp["flat.layers"], np.zeros(n, np.uint8), frusta,
                                 inp["flat.view_masks"], None)
    return frusta, g, vv, vis


@needs_dump
def test_oracle_hierarchy_matches_the_reference():
    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    rc, g, _ = O.propagate_transforms(inp["tree.parent"], inp["tree.translation"], inp["tree.rotation"], inp["tree.scale"])
    assert rc == 0
    assert np.array_equal(bits(g).reshape(-1), bits(ref["tree.global"]))


@needs_dump
def test_oracle_flat_frame_matches_the_reference():
    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    frusta, g, vv, vis = _flat_oracle(inp, ref)
    for v in range(inp["flat.view_masks"].size):
        assert np.array_equal(bits(frusta[24 * v:24 * v + 24]), bits(ref[f"flat.frustum.{v}"])), f"frustum {v}"
    assert np.array_equal(bits(g).reshape(-1), bits(ref["flat.global"]))
    assert np.array_equal((vv & 1).astype(np.uint8), ref["flat.view_visible"])
    # VisibleEntities hold rows that have a visibility class and are not NoCpuCulling (visibility/mod.rs:785, 848-856)
    cpu_culled = (inp["flat.flags"] & 0x10) == 0
    for v in range(inp["flat.view_masks"].size):
        assert np.array_equal(np.nonzero(vis[v].astype(bool) & cpu_culled)[0].astype(np.uint32), ref[f"flat.visible_rows.{v}"]), f"view {v}"


@needs_dump
def test_oracle_clusters_match_the_reference():
    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    fov, aspect, near, far = inp["camera.fov_aspect_near_far"]
    cam = _cameras(inp, ref, "cluster.camera")
    cfv = O.perspective_infinite_reverse(np.float32(fov), aspect, near)
    fr = O.compute_frustum_perspective(np.float32(fov), aspect, near, far, cam)
    w, h, dx, dy, dz = (int(x) for x in inp["cluster.screen_dims_z"])
    first, far_z = inp["cluster.first_slice_depth_far_z"]
    view = O.cluster_view_setup(cam, cfv, fr, w, h, (dx, dy, dz), float(first), float(far_z))
    off, idx, _, farthest, total = O.assign_objects_to_clusters(view, inp["cluster.lights_pos_range"])
    assert tuple(ref["cluster.dims"]) == tuple(view.dims)
    assert np.array_equal(off, ref["cluster.offsets"]) and np.array_equal(idx, ref["cluster.indices"])
    assert int(total) == int(ref["cluster.total"][0])
    assert bits(np.float32(farthest)) == bits(ref["cluster.farthest_z"])[0]


@needs_dump
def test_oracle_spot_clusters_of_two_cameras_match_the_reference():
    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    back = _spot_back(inp)
    spots = inp["cluster2.type"] == 1
    assert np.array_equal(bits(back[spots]), bits(ref["cluster2.spot_back"].reshape(-1, 3)[spots]))  # the direction the cone test reads
    views, _ = _spot_views(inp, ref)
    for v, (_, _, _, view) in enumerate(views):
        off, idx, _, farthest, total = O.assign_objects_to_clusters(view, inp["cluster2.lights_pos_range"], obj_type=inp["cluster2.type"],
                                                                   spot_dir=ref["cluster2.spot_back"], spot_sin_cos=ref["cluster2.sin_cos"])
        assert tuple(ref[f"cluster2.dims.{v}"]) == tuple(view.dims)
        assert np.array_equal(off, ref[f"cluster2.offsets.{v}"]) and np.array_equal(idx, ref[f"cluster2.indices.{v}"]), f"camera {v}"
        assert int(total) == int(ref[f"cluster2.total.{v}"][0])
        assert bits(np.float32(farthest)) == bits(ref[f"cluster2.farthest_z.{v}"])[0]


@needs_dump
@pytest.mark.gpu
def test_hip_spot_clusters_of_two_cameras_match_the_reference():
    from bevy_amd import api

    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    views, (w, h, dims, first, far_z) = _spot_views(inp, ref)
    ctx = api.Context(device=0)
    for v, (cam, cfv, fr, _) in enumerate(views):
        view, _ = api.cluster_view_build(cam, cfv, fr, w, h, dims, first, far_z)
        off, idx, *_ = ctx.cluster_assign(view, inp["cluster2.lights_pos_range"], inp["cluster2.type"], None, ref["cluster2.spot_back"],
                                          ref["cluster2.sin_cos"])
        assert np.array_equal(off, ref[f"cluster2.offsets.{v}"]) and np.array_equal(idx, ref[f"cluster2.indices.{v}"]), f"camera {v}"


@pytest.mark.gpu
def test_hip_matches_the_oracle_on_case_4():
    """The same comparison against the oracle, which needs no dump: spot lights, two cameras, list for list."""
    from bevy_amd import api

    inp = migd.read(INPUTS)
    back = _spot_back(inp).reshape(-1)
    sin_cos = np.stack([np.sin(inp["cluster2.outer_angle"]), np.cos(inp["cluster2.outer_angle"])], axis=1).astype(np.float32).reshape(-1)
    views, (w, h, dims, first, far_z) = _spot_views(inp)
    ctx = api.Context(device=0)
    for v, (cam, cfv, fr, oview) in enumerate(views):
        off0, idx0, *_ = O.assign_objects_to_clusters(oview, inp["cluster2.lights_pos_range"], obj_type=inp["cluster2.type"], spot_dir=back,
                                                      spot_sin_cos=sin_cos)
        view, _ = api.cluster_view_build(cam, cfv, fr, w, h, dims, first, far_z)
        off, idx, *_ = ctx.cluster_assign(view, inp["cluster2.lights_pos_range"], inp["cluster2.type"], None, back, sin_cos)
        assert np.array_equal(off, off0) and np.array_equal(idx, idx0), f"camera {v}"


@needs_dump
@pytest.mark.gpu
def test_hip_path_matches_the_reference():
    import bevy_amd as B
    from bevy_amd import api

    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    frusta, *_ = _flat_oracle(inp, ref)
    n = inp["flat.flags"].size
    ctx = api.Context(device=0)
    ctx.resize(n)
    ctx.upload_transforms(inp["flat.translation"], inp["flat.rotation"], inp["flat.scale"])
    ctx.upload_bounds(inp["flat.aabb_center"], inp["flat.aabb_half"], inp["flat.flags"], inp["flat.layers"])
    ctx.propagate_and_cull(frusta, inp["flat.view_masks"], flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
    g, _ = ctx.download_global_transforms()
    vv, _ = ctx.download_view_visibility()
    assert np.array_equal(bits(g).reshape(-1), bits(ref["flat.global"]))
    assert np.array_equal((vv & 1).astype(np.uint8), ref["flat.view_visible"])


def test_the_dump_tests_pass_on_a_stand_in_dump_made_by_the_oracle(tmp_path, monkeypatch):
    """The tests above have never met a real dump (no Rust here).  This runs them on a stand-in the ORACLE produced, under the
    names tools/golden_dump/src/main.rs writes -- so that a wrong key, shape or dtype in the tests shows today, not on the day
    somebody runs the tool -- with cameras nudged by one ulp, as `Transform::from_matrix` may nudge them: the tests must follow
    the cameras the dump says the systems saw."""
    import re
    import sys as _sys

    inp = migd.read(INPUTS)
    nudge = lambda a: np.nextafter(a.astype(np.float32), np.float32(np.inf)).astype(np.float32)  # noqa: E731
    ref = {"flat.camera_global": nudge(inp["flat.cameras"]), "cluster.camera_global": nudge(inp["cluster.camera"]),
           "cluster2.camera_global": nudge(inp["cluster2.cameras"])}
    rc, g, _ = O.propagate_transforms(inp["tree.parent"], inp["tree.translation"], inp["tree.rotation"], inp["tree.scale"])
    assert rc == 0
    ref["tree.global"] = np.ascontiguousarray(g, np.float32).reshape(-1)
    frusta, g, vv, vis = _flat_oracle(inp, ref)
    ref["flat.global"] = np.ascontiguousarray(g, np.float32).reshape(-1)
    ref["flat.view_visible"] = (vv & 1).astype(np.uint8)
    cpu_culled = (inp["flat.flags"] & 0x10) == 0
    for v in range(inp["flat.view_masks"].size):
        ref[f"flat.frustum.{v}"] = np.ascontiguousarray(frusta[24 * v:24 * v + 24], np.float32)
        ref[f"flat.visible_rows.{v}"] = np.nonzero(vis[v].astype(bool) & cpu_culled)[0].astype(np.uint32)
    fov, aspect, near, far = inp["camera.fov_aspect_near_far"]
    cfv = O.perspective_infinite_reverse(np.float32(fov), aspect, near)
    cam = ref["cluster.camera_global"]
    fr = O.compute_frustum_perspective(np.float32(fov), aspect, near, far, cam)
    w, h, dx, dy, dz = (int(x) for x in inp["cluster.screen_dims_z"])
    first, far_z = inp["cluster.first_slice_depth_far_z"]
    view = O.cluster_view_setup(cam, cfv, fr, w, h, (dx, dy, dz), float(first), float(far_z))
    off, idx, _, farthest, total = O.assign_objects_to_clusters(view, inp["cluster.lights_pos_range"])
    ref.update({"cluster.dims": np.array(view.dims, np.uint32), "cluster.near_far": np.array([first, far_z], np.float32), "cluster.offsets": off,
                "cluster.indices": idx, "cluster.farthest_z": np.array([farthest], np.float32), "cluster.total": np.array([total], np.uint64)})
    back = _spot_back(inp)
    ref["cluster2.spot_back"] = back.reshape(-1)
    ref["cluster2.sin_cos"] = np.stack([np.sin(inp["cluster2.outer_angle"]), np.cos(inp["cluster2.outer_angle"])], axis=1).astype(np.float32).reshape(-1)
    views, _ = _spot_views(inp, ref)
    for v, (_, _, _, view) in enumerate(views):
        off, idx, _, farthest, total = O.assign_objects_to_clusters(view, inp["cluster2.lights_pos_range"], obj_type=inp["cluster2.type"],
                                                                   spot_dir=ref["cluster2.spot_back"], spot_sin_cos=ref["cluster2.sin_cos"])
        ref.update({f"cluster2.dims.{v}": np.array(view.dims, np.uint32), f"cluster2.offsets.{v}": off, f"cluster2.indices.{v}": idx,
                    f"cluster2.farthest_z.{v}": np.array([farthest], np.float32), f"cluster2.total.{v}": np.array([total], np.uint64)})
    # the stand-in carries exactly the arrays the Rust tool writes
    src = open(os.path.join(ROOT, "tools", "golden_dump", "src", "main.rs")).read()
    names = set(re.findall(r'out\.insert\(\s*"([a-z0-9_.]+)"', src))
    for pattern in re.findall(r'out\.insert\(format!\("([a-z0-9_.]+)\.\{v\}"\)', src):
        names |= {f"{pattern}.{v}" for v in range(2)}
    assert names == set(ref), names ^ set(ref)
    path = str(tmp_path / "stand_in.migd")
    migd.write(path, ref)
    me = _sys.modules[__name__]
    monkeypatch.setattr(me, "DUMP", path)
    test_oracle_hierarchy_matches_the_reference()
    test_oracle_flat_frame_matches_the_reference()
    test_oracle_clusters_match_the_reference()
    test_oracle_spot_clusters_of_two_cameras_match_the_reference()
    # ... and they do look at the dump: a flipped bit in it fails them
    broken = dict(ref)
    broken["cluster2.indices.1"] = ref["cluster2.indices.1"].copy()
    broken["cluster2.indices.1"][0] ^= 1
    migd.write(path, broken)
    with pytest.raises(AssertionError):
        test_oracle_spot_clusters_of_two_cameras_match_the_reference()
