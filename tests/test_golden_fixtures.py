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


def _batch_inputs(fx):
    return (fx["row_set"], fx["row_bin"], fx["row_input"], fx["set_indexed"], fx["bin_table_offset"], fx["bin_table"], fx["meta_offset"],
            fx["bin_metadata"])


def _check_batching(res, fx):
    tot = res["totals"]
    assert list(fx["totals"]) == tot["work_item_len"] + tot["indirect_parameters_len"] + tot["batch_set_len"] + [tot["data_buffer_len"]]
    assert np.array_equal(res["records"], fx["records"]) and np.array_equal(res["bin_metadata"], fx["bin_metadata_out"])
    for c in range(2):
        assert np.array_equal(res["work_items"][c], fx[f"work_items_{c}"])
        assert np.array_equal(res["metadata"][c], fx[f"metadata_{c}"])
        assert np.array_equal(res["batch_sets"][c], fx[f"batch_sets_{c}"])


def test_oracle_reproduces_batching_fixture():
    import oracle_lib as O
    fx = load("batching_777.npz")
    ini = O.BatchInitial()
    i = [int(x) for x in fx["initial"]]
    ini.work_item_index[0], ini.work_item_index[1], ini.indirect_parameters_index[0], ini.indirect_parameters_index[1] = i[:4]
    ini.batch_set_index[0], ini.batch_set_index[1], ini.output_mesh_uniform_index = i[4:]
    _check_batching(O.batch_build(fx["rows"], *_batch_inputs(fx), ini), fx)
    assert len(fx["records"]) >= 2 and len(fx["rows"]) > 20


@pytest.mark.gpu
def test_hip_reproduces_batching_fixture():
    import bevy_amd as B
    from bevy_amd import api
    flat, fx = load("flat_frame_777.npz"), load("batching_777.npz")
    n = int(flat["n"])
    with api.Context(0) as ctx:
        ctx.resize(n)
        ctx.upload_transforms(flat["translation"], flat["rotation"], flat["scale"])
        ctx.upload_bounds(flat["aabb_center"], flat["aabb_half"], flat["flags"], flat["layers"])
        ctx.upload_view_visibility(flat["vv0"])
        ctx.batch_upload_rows(fx["row_set"], fx["row_bin"], fx["row_input"])
        ctx.batch_upload_sets(fx["set_indexed"], fx["bin_table_offset"], fx["bin_table"], fx["meta_offset"], fx["bin_metadata"])
        ctx.propagate_and_cull(flat["frusta"], flat["view_masks"], flags=B.CULL_END_FRAME)
        assert np.array_equal(ctx.download_visible_entities(0, 0)[1], fx["rows"])
        ctx.batch_build(0, 0, [int(x) for x in fx["initial"]])
        _check_batching(ctx.batch_download(), fx)
