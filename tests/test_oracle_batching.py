"""CPU checks of oracle/batching_oracle.c (batching work-item build).  The reference's own test of these
structures is a proptest of invariants (render_phase/mod.rs:2356-2700) and the shaders carry no golden values, so the
oracle is checked (a) against an independent numpy statement of what each shader computes and (b) against the same
invariants the proptest asserts: instance_count of a bin == instances in it, work items <-> instances one to one,
MeshUniform ranges of the bins tile the batch set's range."""
import numpy as np
import pytest

from bevy_amd import workloads as W
import oracle_lib as O


@pytest.mark.parametrize("bins", [1, 2, 255, 256, 257, 512, 513, 1000, 65536 + 300])
def test_allocate_uniforms_is_an_exclusive_prefix_in_metadata_order(bins):
    rng = np.random.default_rng(bins)
    meta = np.zeros((bins, 3), np.uint32)
    meta[:, 0] = rng.permutation(bins)
    meta[:, 2] = rng.integers(0, 50, bins)
    first_indirect, first_output, bsi = 17, 1000, 5
    out = O.allocate_uniforms(bsi, first_indirect, first_output, meta, first_indirect + bins + 3)
    excl = first_output + np.concatenate([[0], np.cumsum(meta[:-1, 2], dtype=np.uint64)]).astype(np.uint32)
    got = out[first_indirect + meta[:, 0]]
    assert np.array_equal(got[:, 0], excl)
    assert np.all(got[:, 1] == bsi) and np.all(got[:, 2:] == 0)
    untouched = np.ones(len(out), bool)
    untouched[first_indirect + meta[:, 0]] = False
    assert np.all(out[untouched] == 0xDEADBEEF)


def test_unpack_bins_matches_definition():
    rng = np.random.default_rng(3)
    bins, k = 37, 5000
    table = np.full(bins + 5, 0xFFFFFFFF, np.uint32)
    live = np.sort(rng.choice(bins + 5, bins, replace=False))
    table[live] = rng.permutation(bins)
    meta = np.zeros((bins, 3), np.uint32)
    meta[:, 0] = rng.permutation(bins)
    inst = np.stack([rng.permutation(k).astype(np.uint32), live[rng.integers(0, bins, k)].astype(np.uint32)], 1)
    out = O.unpack_bins(11, 400, inst, meta, table, 11 + k)
    assert np.all(out[:11] == 0)
    assert np.array_equal(out[11:, 0], inst[:, 0])
    assert np.array_equal(out[11:, 1], 400 + meta[table[inst[:, 1]], 0])


def check_build_invariants(sc, rows, res, initial):
    row_set, row_bin, row_input = sc["row_set"], sc["row_bin"], sc["row_input"]
    n_sets = len(sc["set_indexed"])
    meta = res["bin_metadata"]
    listed = np.zeros(len(row_set), bool)
    listed[rows] = True
    # instance_count of every bin == visible rows naming it (the proptest's central invariant)
    for s in range(n_sets):
        t = sc["bin_table"][sc["bin_table_offset"][s]:sc["bin_table_offset"][s + 1]]
        m = meta[sc["meta_offset"][s]:sc["meta_offset"][s + 1]]
        sel = listed & (row_set == s)
        cnt = np.bincount(t[row_bin[sel]], minlength=len(m))
        assert np.array_equal(m[:, 2], cnt)
    recs = res["records"]
    nonempty = [s for s in range(n_sets) if np.any(listed & (row_set == s))]
    assert list(recs[:, 0]) == nonempty
    out_cursor = int(initial.output_mesh_uniform_index)
    cur = {c: [int(initial.work_item_index[c]), int(initial.indirect_parameters_index[c]), int(initial.batch_set_index[c])] for c in (0, 1)}
    for rec in recs:
        s, cls, index, first_wi, count, first_ip, batch_count, first_out = (int(x) for x in rec)
        assert cls == int(sc["set_indexed"][s])
        assert (first_wi, first_ip, index) == tuple(cur[cls]) and first_out == out_cursor
        m = meta[sc["meta_offset"][s]:sc["meta_offset"][s + 1]]
        t = sc["bin_table"][sc["bin_table_offset"][s]:sc["bin_table_offset"][s + 1]]
        assert batch_count == len(m) and count == int(m[:, 2].sum())
        assert tuple(res["batch_sets"][cls][index]) == (0, first_ip)
        # work items <-> the set's visible rows, in list order
        set_rows = rows[row_set[rows] == s]
        wi = res["work_items"][cls][first_wi:first_wi + count]
        assert np.array_equal(wi[:, 0], row_input[set_rows])
        assert np.array_equal(wi[:, 1], first_ip + m[t[row_bin[set_rows]], 0])
        # MeshUniform ranges of the bins tile [first_out, first_out + count) in metadata order
        md = res["metadata"][cls][first_ip + m[:, 0]]
        assert np.array_equal(md[:, 0], first_out + np.concatenate([[0], np.cumsum(m[:-1, 2])]).astype(np.uint32))
        assert np.all(md[:, 1] == index) and np.all(md[:, 2:] == 0)
        cur[cls][0] += count
        cur[cls][1] += batch_count
        cur[cls][2] += 1
        out_cursor += count
    tot = res["totals"]
    assert tot["data_buffer_len"] == out_cursor
    for c in (0, 1):
        assert [tot["work_item_len"][c], tot["indirect_parameters_len"][c], tot["batch_set_len"][c]] == cur[c]


@pytest.mark.parametrize("n_rows,n_sets,frac", [(0, 3, 0.5), (1, 1, 1.0), (5000, 7, 0.3), (20000, 40, 0.05), (3000, 5, 1.0)])
def test_batch_build_invariants(n_rows, n_sets, frac):
    sc = W.batching_scene(max(n_rows, 1), n_sets=n_sets, seed=n_rows + n_sets)
    rng = np.random.default_rng(n_rows)
    rows = np.nonzero(rng.random(max(n_rows, 1)) < frac)[0].astype(np.uint32) if n_rows else np.zeros(0, np.uint32)
    ini = O.BatchInitial()
    ini.work_item_index[0], ini.work_item_index[1] = 3, 10
    ini.indirect_parameters_index[0], ini.indirect_parameters_index[1] = 2, 7
    ini.batch_set_index[0], ini.batch_set_index[1] = 1, 4
    ini.output_mesh_uniform_index = 13
    res = O.batch_build(rows, sc["row_set"], sc["row_bin"], sc["row_input"], sc["set_indexed"], sc["bin_table_offset"],
                        sc["bin_table"], sc["meta_offset"], sc["bin_metadata"], ini)
    check_build_invariants(sc, rows, res, ini)
    if n_rows >= 3000:
        assert len(res["records"]) >= 2


# ---- the CPU-built part of a binned phase and the sorted phases ------------------------------------------------------------------
import batching_model as M  # noqa: E402

INI = ((3, 10), (2, 7), (1, 4), 13)


def _initial(t=INI):
    ini = O.BatchInitial()
    for c in range(2):
        ini.work_item_index[c], ini.indirect_parameters_index[c], ini.batch_set_index[c] = t[0][c], t[1][c], t[2][c]
    ini.output_mesh_uniform_index = t[3]
    return ini


def _same(a, b, keys):
    assert a["totals"] == b["totals"]
    for k in keys:
        if isinstance(a[k], list):
            for c in range(2):
                assert np.array_equal(a[k][c], b[k][c]), (k, c)
        else:
            assert np.array_equal(a[k], b[k]), k


def test_unbatchable_and_batchable_loops_hand_worked():
    """Five listed rows: rows 0, 3 in unbatchable bin 1 (indexed; row 3 has no input index), rows 1, 2, 4 in batchable bin 0
    (non-indexed).  Worked through gpu_preprocessing.rs:2148-2353 by hand, indirect mode, empty buffers."""
    kind = np.array([2, 1, 1, 2, 1], np.uint8)
    rbin = np.array([1, 0, 0, 1, 0], np.uint32)
    rin = np.array([50, 51, 52, 0xFFFFFFFF, 54], np.uint32)
    r = O.batch_cpu_bins(np.arange(5), kind, rbin, rin, [0, 1], [0], False)
    # unbatchable bin 1: allocate(2) -> metadata[indexed] = [m0, zero]; row 0: output 0, indirect 0; row 3 skipped
    assert r["metadata"][1].tolist() == [[0, 0xFFFFFFFF, 0, 0, 0], [0, 0, 0, 0, 0]]
    assert r["work_items"][1].tolist() == [[50, 0]] and r["batch_sets"][1].tolist() == [[0, 0]]
    assert r["unbatchable"].tolist() == [[1, 0]]
    # batchable bin 0: outputs 1, 2, 3; one indirect slot (non-indexed 0), batch set index 0; every work item names slot 0
    assert r["metadata"][0].tolist() == [[1, 0, 0, 0, 0]] and r["batch_sets"][0].tolist() == [[0, 0]]
    assert r["work_items"][0].tolist() == [[51, 0], [52, 0], [54, 0]]
    assert r["records"].tolist() == [[0x80000000, 0, 0, 0, 3, 0, 1, 1]]
    assert r["totals"] == dict(work_item_len=[3, 1], indirect_parameters_len=[1, 2], batch_set_len=[1, 1], data_buffer_len=4)
    # direct mode: no indirect parameters, work items carry the output index
    d = O.batch_cpu_bins(np.arange(5), kind, rbin, rin, [0, 1], [0], True)
    assert d["work_items"][1].tolist() == [[50, 0]] and d["work_items"][0].tolist() == [[51, 1], [52, 2], [54, 3]]
    assert d["unbatchable"].tolist() == [[1, 0]] and d["records"].tolist() == [[0x80000000, 0, 0, 0, 3, 0xFFFFFFFF, 1, 1]]
    assert d["totals"] == dict(work_item_len=[3, 1], indirect_parameters_len=[0, 0], batch_set_len=[0, 0], data_buffer_len=4)


@pytest.mark.parametrize("n_rows,frac,no_indirect", [(0, 0.5, False), (40, 1.0, False), (3000, 0.4, False), (3000, 0.4, True), (20000, 0.1, False)])
def test_cpu_bins_match_the_prefix_sum_model(n_rows, frac, no_indirect):
    ph = W.phase_scene(max(n_rows, 1), seed=n_rows + 3)
    rows = np.nonzero(W.uniform01(n_rows + 1, max(n_rows, 1)) < frac)[0].astype(np.uint32) if n_rows else np.zeros(0, np.uint32)
    got = O.batch_cpu_bins(rows, ph["row_kind"], ph["row_cpu_bin"], ph["row_input"], ph["unbatchable_indexed"], ph["batchable_indexed"],
                           no_indirect, _initial())
    exp = M.cpu_bins(rows, ph["row_kind"], ph["row_cpu_bin"], ph["row_input"], ph["unbatchable_indexed"], ph["batchable_indexed"],
                     no_indirect, INI)
    _same(got, exp, ("work_items", "metadata", "batch_sets", "unbatchable", "records"))
    # invariants in the spirit of the reference's proptest: every listed row with an input index owns exactly one MeshUniform
    # slot, slots are handed out contiguously in bin order, and every work item points at a live indirect slot of its class
    k = sum(len(got["work_items"][c]) - INI[0][c] for c in range(2))
    assert got["totals"]["data_buffer_len"] == INI[3] + k
    if not no_indirect:
        for c in range(2):
            wi = got["work_items"][c][INI[0][c]:]
            assert np.all(wi[:, 1] >= INI[1][c]) and np.all(wi[:, 1] < got["totals"]["indirect_parameters_len"][c])
            base = got["metadata"][c][wi[:, 1], 0]
            assert np.all(base >= INI[3]) and np.all(base < got["totals"]["data_buffer_len"])


def test_whole_phase_is_cpu_bins_then_multidrawables():
    ph = W.phase_scene(6000, seed=17)
    rows = np.nonzero(W.uniform01(9, 6000) < 0.5)[0].astype(np.uint32)
    res = O.batch_phase(rows, ph, initial=_initial())
    cpu = O.batch_cpu_bins(rows, ph["row_kind"], ph["row_cpu_bin"], ph["row_input"], ph["unbatchable_indexed"], ph["batchable_indexed"], False, _initial())
    # the multidrawable pass starts where the CPU loops stopped (gpu_preprocessing.rs:2431-2447) and leaves their entries alone
    for c in range(2):
        assert np.array_equal(res["work_items"][c][:len(cpu["work_items"][c])], cpu["work_items"][c])
        assert len(res["work_items"][c]) >= len(cpu["work_items"][c])
    multi = res["records"][len(cpu["records"]):]
    assert np.all(multi[:, 0] < 0x80000000) and (len(multi) == 0 or multi[0, 7] == cpu["totals"]["data_buffer_len"])
    n_multi = int(((ph["row_kind"][rows] == 0) & (ph["row_set"][rows] != 0xFFFFFFFF)).sum())
    assert res["totals"]["data_buffer_len"] == cpu["totals"]["data_buffer_len"] + n_multi


def test_sorted_phase_hand_worked():
    """Items: A A B | (no input) | A C(no compare data) with set key 7 throughout; A/B/C = bin keys 1/2/3; all non-indexed.
    gpu_preprocessing.rs:1897-2061 by hand, indirect mode."""
    F = O.ITEM_HAS_COMPARE_DATA
    items = np.array([[10, 7, 1, F], [11, 7, 1, F], [12, 7, 2, F], [0xFFFFFFFF, 7, 2, F], [14, 7, 1, F], [15, 7, 3, 0]], np.uint32)
    r = O.batch_sorted(items, True, False)
    # item 0: new set, indirect 0; item 1: BatchOk; item 2: BreakBatch -> indirect 1; item 3 flushes (instances 0..3, indirect 0..2);
    # item 4: new set, indirect 2; item 5 has no compare data -> BreakBatchSet: flush (3..4, 2..3), new set indirect 3; end: flush (4..5, 3..4)
    assert r["work_items"][0].tolist() == [[10, 0], [11, 0], [12, 1], [14, 2], [15, 3]]
    assert r["metadata"][0][:, 0].tolist() == [0, 2, 3, 4] and np.all(r["metadata"][0][:, 1] == 0xFFFFFFFF)
    assert r["batches"].tolist() == [[0, 0, 3, 0, 2, 0], [4, 3, 4, 2, 3, 0], [5, 4, 5, 3, 4, 0]]
    assert r["batch_sets"][0].tolist() == [[0, 0], [0, 2], [0, 3]]
    # without indirect drawing a different mesh is a new batch set, and work items carry the output index
    d = O.batch_sorted(items, True, True)
    assert d["work_items"][0].tolist() == [[10, 0], [11, 1], [12, 2], [14, 3], [15, 4]]
    assert d["batches"][:, :3].tolist() == [[0, 0, 2], [2, 2, 3], [4, 3, 4], [5, 4, 5]] and np.all(d["batches"][:, 3:5] == 0xFFFFFFFF)
    # I::AUTOMATIC_BATCHING == false: every item is its own batch set
    assert len(O.batch_sorted(items, False, False)["batches"]) == 5
    # the range merge of batching/mod.rs:219-244 (no GPU preprocessing): items 0-1 merge, 2 alone, 3 skipped, 4 alone, 5 alone
    b, blen = O.batch_sorted_merge(items, True, 100)
    assert b[:, :3].tolist() == [[0, 100, 102], [2, 102, 103], [4, 103, 104], [5, 104, 105]] and blen == 105


@pytest.mark.parametrize("n,automatic,no_indirect", [(0, True, False), (1, True, False), (700, True, False), (700, True, True),
                                                     (700, False, False), (9000, True, False)])
def test_sorted_phase_matches_the_prefix_sum_model(n, automatic, no_indirect):
    items = W.sorted_items(n, seed=n + 1)
    _same(O.batch_sorted(items, automatic, no_indirect, _initial()), M.sorted_phase(items, automatic, no_indirect, INI),
          ("work_items", "metadata", "batch_sets", "batches"))
    b, blen = O.batch_sorted_merge(items, automatic, 5)
    mb, mlen = M.sorted_merge(items, automatic, 5)
    assert np.array_equal(b, mb) and blen == mlen
    # the ranges tile the instance buffer: every item with data is in exactly one batch
    assert (b[:, 2] - b[:, 1]).sum() == blen - 5 and (len(b) == 0 or (b[0, 1] == 5 and np.all(b[1:, 1] == b[:-1, 2])))
