"""SURVEY.md section 8e, row 3 on CPU: assign_objects_to_clusters with the gathered object list range-sharded over the ranks
(bevy_amd/sharding.py: shard_objects, merge_cluster_assignments), world sizes 2 and 3 over gloo.  The kernels need a GPU, so each rank
assigns ITS objects with the oracle (the checker here, never the product); what is under test is the partition, the three collectives
and the merge: offsets, the six per-type counts, farthest_z and -- the point -- the INDEX ORDER inside every cluster must equal the
unsharded push order (crates/bevy_light/src/cluster/assign.rs:740-800)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from bevy_amd import api, sharding, workloads as W
import oracle_lib as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def scene(n_point, n_spot, seed=9):
    """The gathered list as the reference builds it: point lights first, then spot lights (assign.rs:190-296)."""
    rng = np.random.default_rng(seed)
    n = n_point + n_spot
    pos = rng.normal(0.0, 14.0, (n, 3)).astype(np.float32)
    pos[:, 2] -= 25.0
    rng_col = (0.5 + 6.0 * rng.random(n)).astype(np.float32)
    pr = np.concatenate([pos, rng_col[:, None]], axis=1).astype(np.float32).reshape(-1)
    ty = np.concatenate([np.zeros(n_point, np.uint8), np.ones(n_spot, np.uint8)])
    layers = rng.choice(np.array([1, 1, 1, 3, 2], np.uint32), n)
    d = rng.normal(0.0, 1.0, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ang = (0.2 + 0.9 * rng.random(n)).astype(np.float32)
    sc = np.stack([np.sin(ang), np.cos(ang)], axis=1).astype(np.float32)
    return pr, ty, layers, d.reshape(-1).astype(np.float32), sc.reshape(-1)


def view_for():
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(7)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    return O.cluster_view_setup(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)


def _worker(rank, world, port, n_point, n_spot, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pr, ty, layers, sd, sc = scene(n_point, n_spot)
        n = n_point + n_spot
        view = view_for()
        lo, hi = sharding.shard_objects(n, world, rank)
        local = O.assign_objects_to_clusters(view, pr[4 * lo:4 * hi], ty[lo:hi], layers[lo:hi], sd[3 * lo:3 * hi], sc[2 * lo:2 * hi])
        off, idx, counts, far, total = sharding.merge_cluster_assignments(local, lo, world, rank)
        e_off, e_idx, e_counts, e_far, e_total = O.assign_objects_to_clusters(view, pr, ty, layers, sd, sc)
        ok = (np.array_equal(off, e_off) and np.array_equal(idx, e_idx) and np.array_equal(counts, e_counts) and total == e_total
              and np.float32(far) == np.float32(e_far))
        ret[rank] = (bool(ok), int(total), int(hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_point,n_spot", [(2, 3000, 500), (3, 1000, 333), (2, 1, 0), (3, 2, 0), (2, 0, 0)])
def test_sharded_assignment_equals_the_unsharded_push_order(world, n_point, n_spot):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_point, n_spot, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        ok, total, mine = ret[rank]
        assert ok, f"rank {rank} of {world}: merged assignment differs from the unsharded one"
    assert sum(ret[r][2] for r in range(world)) == n_point + n_spot
    if n_point >= 1000:
        assert ret[0][1] > 1000  # (the scene does put lights into clusters)


def test_object_ranges_are_contiguous_and_cover_the_list():
    for n in (0, 1, 5, 100_000, 100_001):
        for world in (1, 2, 3, 8):
            ranges = [sharding.shard_objects(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
