"""N>1 path on CPU: world_size-2 gloo run of the entity-range shard + in-place all-gather of the packed
ViewVisibility bitmask (bevy_amd/sharding.py).  The kernels need a GPU, so each rank fills its block of the
gathered buffer from the oracle's per-row visibility (the oracle is the checker here, never the product);
what is under test is the row partition, the [gpu][view][words] layout and the collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bevy_amd import api, sharding, workloads as W
import oracle_lib as O

N_ROWS = 10_000 + 37
N_VIEWS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene_and_frusta():
    sc = W.many_cubes(N_ROWS, radius=120.0, ragged_flags=True)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    frusta = np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(0, yaw=v * 1.1), W.CAMERA_FAR) for v in range(N_VIEWS)])
    return sc, frusta


def _visible(sc, frusta, lo, hi):
    s3, s4 = slice(3 * lo, 3 * hi), slice(4 * lo, 4 * hi)
    n = hi - lo
    _, _, vis, _ = O.full_frame(sc["translation"][s3], sc["rotation"][s4], sc["scale"][s3], sc["aabb_center"][s3],
                                sc["aabb_half"][s3], sc["flags"][lo:hi], sc["layers"][lo:hi], np.zeros(n, np.uint8), frusta)
    return vis


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, frusta = _scene_and_frusta()
        lo, hi = sharding.shard_rows(N_ROWS, world, rank)
        w = sharding.words_per_shard(N_ROWS, world)
        full = torch.zeros(sharding.gathered_words(N_ROWS, world, N_VIEWS), dtype=torch.int64)
        wpv, woff = sharding.block_offset_words(N_ROWS, world, N_VIEWS, rank)
        assert wpv == w and woff == rank * N_VIEWS * w
        vis = _visible(sc, frusta, lo, hi)
        mine = np.zeros((N_VIEWS, w), np.uint64)
        for v in range(N_VIEWS):
            bits = np.zeros(w * 64, np.uint8)
            bits[:hi - lo] = vis[v]
            mine[v] = np.packbits(bits, bitorder="little").view(np.uint64)
        full[woff:woff + N_VIEWS * w] = torch.from_numpy(mine.reshape(-1).view(np.int64))
        sharding.all_gather_visibility(full, N_ROWS, world, N_VIEWS, rank)
        ret[rank] = full.numpy().copy()
        # the pipelined gatherer bench.py uses (two alternating buffers); on CPU tensors it goes through gloo
        g = sharding.MaskGatherer(N_ROWS, world, N_VIEWS, rank, device=None)
        assert g.mode == "torch.distributed" and g.bind_args(0)[1:] == (wpv, woff)
        for frame in range(3):
            g.before_kernels(frame)
            buf = g.buffer(frame)
            buf.zero_()
            buf[woff:woff + N_VIEWS * w] = torch.from_numpy((mine.reshape(-1) + np.uint64(frame)).view(np.int64))
            out = g.after_kernels(frame)
            ret[(rank, frame)] = out.numpy().copy()
        g.close()
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    for n in (0, 1, 255, 256, 257, 10_000, 1_000_000, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert all(lo % sharding.ROW_ALIGN == 0 for lo, hi in spans if hi > lo)
            assert all(hi - lo <= sharding.words_per_shard(n, world) * 64 for lo, hi in spans)


def test_all_gather_visibility_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world + 3 * world
    assert np.array_equal(ret[0], ret[1]), "ranks disagree after the all-gather"
    for frame in range(3):
        assert np.array_equal(ret[(0, frame)], ret[(1, frame)]), f"pipelined gatherer: ranks disagree at frame {frame}"
    assert np.array_equal(ret[(0, 0)], ret[0]), "pipelined gatherer differs from the plain all-gather"
    assert not np.array_equal(ret[(0, 1)], ret[(0, 0)])
    sc, frusta = _scene_and_frusta()
    expect = _visible(sc, frusta, 0, N_ROWS)
    for v in range(N_VIEWS):
        got = sharding.unpack_view(ret[0], N_ROWS, world, N_VIEWS, v)
        assert np.array_equal(got, expect[v]), f"view {v}"
    assert expect.sum() > 0


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("shape", ["one_tree", "forest", "chain"])
def test_shard_hierarchy_partition(world, shape):
    """Hierarchy sharding by root subtree: every row owned exactly once, every held row's ancestors held,
    level order kept, loads balanced; and propagating each shard on its own reproduces the unsharded result bit for
    bit (checked with the oracle on CPU -- no collective is involved)."""
    if shape == "one_tree":
        tr = W.gen_tree(7, 4)                      # 5461 nodes, one root: must be opened up
    elif shape == "forest":
        tr = W.gen_tree(6, 3)
        # 9 copies of a small tree -> a forest (roots first, then level by level)
        parts, n1 = [], tr["n"]
        lv1 = tr["level_offsets"].astype(np.int64)
        k = 9
        parent = np.zeros(k * n1, np.uint32)
        t, r, s = (np.zeros((k * n1, d), np.float32) for d in (3, 4, 3))
        lv = [0]
        pos = 0
        new_index = np.zeros((k, n1), np.int64)
        for l in range(len(lv1) - 1):
            for c in range(k):
                cnt = lv1[l + 1] - lv1[l]
                new_index[c, lv1[l]:lv1[l + 1]] = np.arange(pos, pos + cnt)
                pos += cnt
            lv.append(pos)
        for c in range(k):
            p1 = tr["parent"]
            parent[new_index[c]] = np.where(p1 == W.NO_PARENT, W.NO_PARENT, new_index[c][np.where(p1 == W.NO_PARENT, 0, p1)])
            t[new_index[c]] = tr["translation"].reshape(-1, 3) * (1 + 0.1 * c)
            r[new_index[c]] = tr["rotation"].reshape(-1, 4)
            s[new_index[c]] = tr["scale"].reshape(-1, 3)
        tr = dict(n=k * n1, parent=parent, level_offsets=np.array(lv, np.uint32), translation=t.reshape(-1), rotation=r.reshape(-1),
                  scale=s.reshape(-1))
    else:
        n = 40
        tr = W.gen_tree(n, 1)
    n = tr["n"]
    shards = sharding.shard_hierarchy(tr["parent"], tr["level_offsets"], world)
    owners = np.zeros(n, np.int64)
    _, g_full, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    g_full = g_full.reshape(n, 12)
    got = np.zeros_like(g_full)
    for r, sh in enumerate(shards):
        rows = sh["rows"].astype(np.int64)
        owners[rows[sh["owned"]]] += 1
        assert np.all(np.diff(rows) > 0)
        lp = sh["parent"]
        assert np.all((lp == W.NO_PARENT) | (lp < np.arange(len(rows))))          # parents precede children
        assert np.array_equal(tr["parent"][rows] == W.NO_PARENT, lp == W.NO_PARENT)
        ok = lp != W.NO_PARENT
        assert np.array_equal(rows[lp[ok]], tr["parent"][rows][ok])             # and are the same nodes
        lv = sh["level_offsets"].astype(np.int64)
        assert lv[0] == 0 and lv[-1] == len(rows) and np.all(np.diff(lv) > 0) or len(rows) == 0
        if len(rows):
            t3, r4, s3 = (tr[k].reshape(n, -1)[rows].reshape(-1) for k in ("translation", "rotation", "scale"))
            _, g, _ = O.propagate_transforms(lp, t3, r4, s3)
            got[rows[sh["owned"]]] = g.reshape(-1, 12)[sh["owned"]]
    assert np.all(owners == 1)
    assert got.tobytes() == g_full.tobytes()
    held = [len(sh["rows"]) for sh in shards]
    if shape != "chain" and world > 1:
        assert max(held) <= 1.35 * n / world + 64, held


def test_shard_hierarchy_degenerate_inputs():
    """Nothing to shard, one node, fewer trees than ranks: every row still owned exactly once, empty ranks are empty."""
    cases = [(np.zeros(0, np.uint32), np.array([0], np.uint32)),
             (np.array([W.NO_PARENT], np.uint32), np.array([0, 1], np.uint32)),
             (np.full(5, W.NO_PARENT, np.uint32), np.array([0, 5], np.uint32))]
    for parent, lv in cases:
        for world in (1, 2, 8):
            shards = sharding.shard_hierarchy(parent, lv, world)
            assert len(shards) == world
            owned = np.zeros(len(parent), np.int64)
            for sh in shards:
                owned[sh["rows"][sh["owned"]].astype(np.int64)] += 1
                assert sh["level_offsets"][0] == 0 and sh["level_offsets"][-1] == len(sh["rows"])
            assert np.all(owned == 1)
