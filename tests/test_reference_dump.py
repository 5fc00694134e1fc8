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


def _flat_oracle(inp):
    fov, aspect, near, far = inp["camera.fov_aspect_near_far"]
    cams = inp["flat.cameras"].reshape(-1, 12)
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
    frusta, g, vv, vis = _flat_oracle(inp)
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
    cam = inp["cluster.camera"]
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
@pytest.mark.gpu
def test_hip_path_matches_the_reference():
    import bevy_amd as B
    from bevy_amd import api

    inp, ref = migd.read(INPUTS), migd.read(DUMP)
    frusta, *_ = _flat_oracle(inp)
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
