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
