"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_fixtures.py).
CPU: the oracle still reproduces them bit for bit (guards the checker against compiler / libm drift).
GPU: the HIP path reproduces them through the C ABI without the oracle in the loop."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return dict(np.load(os.path.join(HERE, "golden", name)))


def test_oracle_reproduces_flat_fixture():
    import oracle_lib as O
    fx = load("flat_frame_777.npz")
    g, vv, vis, chg = O.full_frame(fx["translation"], fx["rotation"], fx["scale"], fx["aabb_center"], fx["aabb_half"],
                                   fx["flags"], fx["layers"], fx["vv0"], fx["frusta"], fx["view_masks"], None)
    assert np.array_equal(g.view(np.uint32), fx["global_bits"])
    assert np.array_equal(vv, fx["vv"]) and np.array_equal(vis, fx["visible"]) and np.array_equal(chg, fx["vv_changed"])


def test_oracle_reproduces_tree_fixture():
    import oracle_lib as O
    fx = load("tree_6x3.npz")
    rc, g, chg = O.propagate_transforms(fx["parent"], fx["translation"], fx["rotation"], fx["scale"])
    assert rc == 0 and np.array_equal(g.view(np.uint32), fx["global_bits"]) and np.array_equal(chg, fx["changed"])


def test_oracle_reproduces_cluster_fixture():
    import oracle_lib as O
    fx = load("cluster_3000.npz")
    view = O.cluster_view_setup(fx["camera"], fx["clip_from_view"], fx["frustum"], 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    off, idx, counts, far, total = O.assign_objects_to_clusters(view, fx["lights"])
    assert total == int(fx["total"]) and np.array_equal(off, fx["offsets"]) and np.array_equal(idx, fx["indices"])
    assert np.array_equal(counts, fx["counts"]) and np.float32(far) == fx["farthest_z"]


@pytest.mark.gpu
def test_hip_reproduces_flat_fixture():
    import bevy_amd as B
    from bevy_amd import api
    fx = load("flat_frame_777.npz")
    n = int(fx["n"])
    with api.Context(0) as ctx:
        ctx.resize(n)
        ctx.upload_transforms(fx["translation"], fx["rotation"], fx["scale"])
        ctx.upload_bounds(fx["aabb_center"], fx["aabb_half"], fx["flags"], fx["layers"])
        ctx.upload_view_visibility(fx["vv0"])
        ctx.propagate_and_cull(fx["frusta"], fx["view_masks"], flags=B.CULL_END_FRAME)
        assert np.array_equal(ctx.download_global_transforms(want_changed=False).view(np.uint32), fx["global_bits"])
        for v in range(2):
            assert np.array_equal(ctx.download_visibility(v), fx["visible"][v])
        vv, chg = ctx.download_view_visibility()
        assert np.array_equal(vv, fx["vv"]) and np.array_equal(chg, fx["vv_changed"])


@pytest.mark.gpu
def test_hip_reproduces_tree_fixture():
    import bevy_amd as B
    from bevy_amd import api
    fx = load("tree_6x3.npz")
    with api.Context(0) as ctx:
        ctx.resize(int(fx["n"]))
        ctx.upload_transforms(fx["translation"], fx["rotation"], fx["scale"])
        ctx.upload_hierarchy(fx["parent"], fx["level_offsets"])
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        g, chg = ctx.download_global_transforms()
        assert np.array_equal(g.view(np.uint32), fx["global_bits"]) and np.array_equal(chg, fx["changed"])


@pytest.mark.gpu
def test_hip_reproduces_cluster_fixture():
    from bevy_amd import api
    fx = load("cluster_3000.npz")
    view, keep = api.cluster_view_build(fx["camera"], fx["clip_from_view"], fx["frustum"], 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    with api.Context(0) as ctx:
        off, idx, counts, far, total = ctx.cluster_assign(view, fx["lights"])
    assert total == int(fx["total"]) and np.array_equal(off, fx["offsets"]) and np.array_equal(idx, fx["indices"])
    assert np.array_equal(counts, fx["counts"]) and np.float32(far) == fx["farthest_z"]
