"""GPU parity of the batching work-item build (mi_batch_build) against oracle/batching_oracle.c:
every output array bit-exact (u32), from the VisibleEntities list the cull pass left on the device."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx_factory():
    made = []

    def make():
        c = api.Context()
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


def upload_scene(ctx, sc):
    n = len(sc["flags"])
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])


def assert_same(got, exp):
    assert got["totals"] == exp["totals"]
    assert np.array_equal(got["records"], exp["records"])
    assert np.array_equal(got["bin_metadata"], exp["bin_metadata"])
    for c in range(2):
        for key in ("work_items", "metadata", "batch_sets"):
            assert got[key][c].shape == exp[key][c].shape, (key, c)
            assert np.array_equal(got[key][c], exp[key][c]), (key, c)


INITIAL = (3, 10, 2, 7, 1, 4, 13)


def oracle_initial(ini):
    o = O.BatchInitial()
    if ini is not None:
        o.work_item_index[0], o.work_item_index[1] = ini[0], ini[1]
        o.indirect_parameters_index[0], o.indirect_parameters_index[1] = ini[2], ini[3]
        o.batch_set_index[0], o.batch_set_index[1] = ini[4], ini[5]
        o.output_mesh_uniform_index = ini[6]
    return o


def test_batch_build_one_bin(ctx_factory):
    """many_cubes itself: one mesh, one material -> one batch set with one bin holding every visible instance."""
    n = 200_000
    sc = W.many_cubes(n, radius=30.0)
    bs = W.batching_scene(n, n_sets=1, max_bins=1, seed=3, unbatched_fraction=0.0)
    assert len(bs["bin_metadata"]) == 1
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
    ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
    ctx.propagate_and_cull(frusta_for([W.many_cubes_camera(0)]), flags=B.CULL_END_FRAME)
    ctx.batch_build(0, 0)
    got = ctx.batch_download()
    rows = ctx.download_visible_entities(0, 0)[1]
    exp = O.batch_build(rows, bs["row_set"], bs["row_bin"], bs["row_input"], bs["set_indexed"], bs["bin_table_offset"],
                        bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
    assert_same(got, exp)
    assert got["bin_metadata"][0, 2] == len(rows) > 1000


@pytest.mark.parametrize("n,n_sets,radius,initial", [(1, 1, 5.0, None), (5000, 7, 30.0, INITIAL), (70_001, 40, 60.0, None),
                                                      (300_000, 300, 40.0, INITIAL), (300_000, 1, 40.0, None),
                                                      (40_000, 65536, 20.0, None)])
def test_batch_build_matches_oracle(ctx_factory, n, n_sets, radius, initial):
    sc = W.many_cubes(n, radius=radius, ragged_flags=True)
    bs = W.batching_scene(n, n_sets=n_sets, max_bins=40 if n_sets < 1000 else 3, seed=n + n_sets)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
    ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
    views = frusta_for([W.many_cubes_camera(0), W.many_cubes_camera(0, yaw=2.5)])
    for frame in range(2):  # second frame: scratch and counters are reused
        ctx.propagate_and_cull(views, flags=B.CULL_END_FRAME)
        for v in range(2):
            ctx.batch_build(v, 0, initial)
            got = ctx.batch_download()
            rows = ctx.download_visible_entities(v, 0)[1]
            exp = O.batch_build(rows, bs["row_set"], bs["row_bin"], bs["row_input"], bs["set_indexed"], bs["bin_table_offset"],
                                bs["bin_table"], bs["meta_offset"], bs["bin_metadata"], oracle_initial(initial))
            assert_same(got, exp)
            if n >= 5000 and v == 0:
                assert len(rows) > 0 and len(exp["records"]) >= 1
        views = frusta_for([W.many_cubes_camera(7), W.many_cubes_camera(7, yaw=2.5)])


def test_batch_build_classes_and_key_order(ctx_factory):
    """A second visibility class, and Entity keys that are not in row order (general compaction path: the list is in key
    order, the work items follow it)."""
    n = 20_000
    sc = W.many_cubes(n, radius=25.0)
    bs = W.batching_scene(n, n_sets=9, seed=5)
    cm = (1 + (np.arange(n) % 3 == 0) * 4).astype(np.uint32)
    keys = np.random.default_rng(1).permutation(n).astype(np.uint64) + np.uint64(1 << 32)
    for use_keys in (False, True):
        ctx = ctx_factory()
        upload_scene(ctx, sc)
        ctx.upload_visibility_classes(cm)
        if use_keys:
            ctx.upload_entity_keys(keys)
        ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
        ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bs["bin_metadata"])
        ctx.propagate_and_cull(frusta_for([W.many_cubes_camera(0)]), flags=B.CULL_END_FRAME)
        for class_bit in (0, 2, 5):  # 5: no row carries it
            ctx.batch_build(0, class_bit, INITIAL)
            got = ctx.batch_download()
            rows = ctx.download_visible_entities(0, class_bit)[1]
            exp = O.batch_build(rows, bs["row_set"], bs["row_bin"], bs["row_input"], bs["set_indexed"], bs["bin_table_offset"],
                                bs["bin_table"], bs["meta_offset"], bs["bin_metadata"], oracle_initial(INITIAL))
            assert_same(got, exp)
            if class_bit == 2:
                assert 0 < len(rows) < n and np.all(cm[rows] & 4)
        if use_keys:
            assert not np.all(np.diff(rows.astype(np.int64)) > 0) or len(rows) < 2


def test_batch_build_errors(ctx_factory):
    ctx = ctx_factory()
    sc = W.many_cubes(100, radius=5.0)
    upload_scene(ctx, sc)
    with pytest.raises(api.MiError):
        ctx.batch_build(0, 0)  # before a cull
    ctx.propagate_and_cull(frusta_for([W.many_cubes_camera(0)]), flags=B.CULL_END_FRAME)
    with pytest.raises(api.MiError):
        ctx.batch_build(0, 0)  # before the uploads
    bs = W.batching_scene(100, n_sets=2, seed=1)
    bad = bs["bin_metadata"].copy()
    bad[0, 0] = 10_000
    with pytest.raises(api.MiError):
        ctx.batch_upload_sets(bs["set_indexed"], bs["bin_table_offset"], bs["bin_table"], bs["meta_offset"], bad)
    ctx.batch_upload_sets(np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros((0, 3), np.uint32))
    ctx.batch_upload_rows(bs["row_set"], bs["row_bin"], bs["row_input"])
    ctx.batch_build(0, 0)  # no batch sets at all: nothing appended
    got = ctx.batch_download()
    assert got["totals"]["data_buffer_len"] == 0 and len(got["records"]) == 0
    with pytest.raises(api.MiError):
        ctx.batch_build(3, 0)  # no such view


# ---- the whole binned phase (unbatchables, batchables, multidrawables) and the sorted phases ---------------------------------
def assert_same_phase(got, exp):
    assert_same(got, exp)
    assert np.array_equal(got["unbatchable"], exp["unbatchable"])


def upload_phase(ctx, ph):
    ctx.batch_upload_rows(ph["row_set"], ph["row_bin"], ph["row_input"])
    ctx.batch_upload_sets(ph["set_indexed"], ph["bin_table_offset"], ph["bin_table"], ph["meta_offset"], ph["bin_metadata"])
    ctx.batch_upload_row_bins(ph["row_kind"], ph["row_cpu_bin"])
    ctx.batch_upload_bins(ph["unbatchable_indexed"], ph["batchable_indexed"])


@pytest.mark.parametrize("n,n_sets,n_unb,n_bat,radius,initial", [
    (1, 1, 1, 1, 5.0, None), (5000, 7, 5, 6, 30.0, INITIAL), (70_001, 40, 30, 50, 60.0, None), (300_000, 1, 3, 2, 40.0, INITIAL),
    (60_000, 300, 20, 20, 30.0, INITIAL),   # more than 256 buckets: the two-pass partition
    (20_000, 0, 9, 4, 25.0, None),          # a phase without multidrawable sets
    (20_000, 6, 0, 0, 25.0, None)])
def test_phase_build_matches_oracle(ctx_factory, n, n_sets, n_unb, n_bat, radius, initial):
    """One view's whole binned phase in the reference's order -- unbatchable bins, batchable bins, multidrawable batch sets
    (gpu_preprocessing.rs:2135-2455) -- in three launches, every output array bit-exact, with and without indirect drawing."""
    sc = W.many_cubes(n, radius=radius, ragged_flags=True)
    ph = W.phase_scene(n, n_sets=n_sets, n_unbatchable_bins=n_unb, n_batchable_bins=n_bat, seed=n + n_sets, cpu_fraction=0.3)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    upload_phase(ctx, ph)
    views = frusta_for([W.many_cubes_camera(0), W.many_cubes_camera(0, yaw=2.5)])
    for frame in range(2):  # second frame: scratch, the instance counters and the bucket tables are reused
        ctx.propagate_and_cull(views, flags=B.CULL_END_FRAME)
        for v in range(2):
            rows = ctx.download_visible_entities(v, 0)[1]
            for no_indirect in (False, True, False):
                ctx.batch_build(v, 0, initial, no_indirect_drawing=no_indirect)
                assert_same_phase(ctx.batch_download(), O.batch_phase(rows, ph, no_indirect, oracle_initial(initial)))
        views = frusta_for([W.many_cubes_camera(7), W.many_cubes_camera(7, yaw=2.5)])
    if n >= 5000:
        exp = O.batch_phase(rows, ph, False, oracle_initial(initial))
        assert len(rows) > 100 and (n_unb == 0 or len(exp["unbatchable"]) > 0) and (n_bat == 0 or (exp["records"][:, 0] >= 0x80000000).any())


def test_phase_tables_can_change_between_builds(ctx_factory):
    n = 30_000
    sc = W.many_cubes(n, radius=25.0)
    ctx = ctx_factory()
    upload_scene(ctx, sc)
    ctx.propagate_and_cull(frusta_for([W.many_cubes_camera(0)]), flags=B.CULL_END_FRAME)
    rows = ctx.download_visible_entities(0, 0)[1]
    for seed, n_sets, n_unb, n_bat in ((1, 5, 4, 3), (2, 9, 0, 7), (3, 2, 6, 0), (4, 5, 4, 3)):
        ph = W.phase_scene(n, n_sets=n_sets, n_unbatchable_bins=n_unb, n_batchable_bins=n_bat, seed=seed, cpu_fraction=0.4)
        upload_phase(ctx, ph)
        ctx.batch_build(0, 0)
        assert_same_phase(ctx.batch_download(), O.batch_phase(rows, ph))


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 700, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 9000, 12_289, 100_000])
def test_sorted_build_matches_oracle(ctx_factory, n):
    """batch_and_prepare_sorted_render_phase (gpu_preprocessing.rs:1850-2061) and the range merge of batching/mod.rs:219-244."""
    ctx = ctx_factory()
    ctx.resize(1)
    items = W.sorted_items(n, seed=n + 1)
    for automatic, no_indirect, initial in ((True, False, None), (True, False, INITIAL), (True, True, INITIAL), (False, False, None)):
        ctx.batch_sorted_build(items, automatic, no_indirect, False, initial)
        got = ctx.batch_download()
        exp = O.batch_sorted(items, automatic, no_indirect, oracle_initial(initial))
        assert got["totals"] == exp["totals"]
        assert np.array_equal(got["batches"], exp["batches"])
        for c in range(2):
            for key in ("work_items", "metadata", "batch_sets"):
                assert np.array_equal(got[key][c], exp[key][c]), (key, c, automatic, no_indirect)
    for automatic in (True, False):
        ctx.batch_sorted_build(items, automatic, False, True, (0, 0, 0, 0, 0, 0, 41))
        got = ctx.batch_download()
        b, blen = O.batch_sorted_merge(items, automatic, 41)
        assert np.array_equal(got["batches"], b) and got["totals"]["data_buffer_len"] == blen
    if n >= 700:
        assert 0 < len(exp["batches"]) < n


def test_sorted_build_long_runs_cross_chunks(ctx_factory):
    """Batch sets longer than a thread's items, than a tile (4 096 items), and one that spans the whole phase; every item a set."""
    ctx = ctx_factory()
    ctx.resize(1)
    for n, run in ((5000, 3000), (4096, 100_000), (3000, 1), (20_000, 9000), (30_000, 1_000_000), (8192, 17), (8200, 1)):
        items = W.sorted_items(n, seed=3, run=run, no_input_fraction=0.0, no_meta_fraction=0.0)
        ctx.batch_sorted_build(items, True, False, False, INITIAL)
        got = ctx.batch_download()
        exp = O.batch_sorted(items, True, False, oracle_initial(INITIAL))
        assert np.array_equal(got["batches"], exp["batches"]) and got["totals"] == exp["totals"]
        for c in range(2):
            for key in ("work_items", "metadata", "batch_sets"):
                assert np.array_equal(got[key][c], exp[key][c]), (key, c, n, run)
