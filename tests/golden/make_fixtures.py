"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for small cases of every stage.

    python tests/golden/make_fixtures.py

The reference itself (Rust) cannot run in this image, so these fixtures are ORACLE outputs, not reference
outputs: they pin the oracle against drift (compiler, libm) on CPU (`-m "not gpu"`) and give the `-m gpu` tests a
checker-independent target.  The reference's own literal known-answer vectors live in
primitives_known_answers.json (provenance inside)."""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from bevy_amd import workloads as W  # noqa: E402


def cameras():
    return [W.many_cubes_camera(0), W.many_cubes_camera(5, yaw=math.pi / 2)]


def frusta_of(cams):
    return np.concatenate([O.compute_frustum_perspective(np.float32(W.CAMERA_FOV), W.CAMERA_ASPECT, W.CAMERA_NEAR,
                                                         W.CAMERA_FAR, cam) for cam in cams])


def flat_case(n=777):
    sc = W.many_cubes(n, radius=40.0, ragged_flags=True)
    fr = frusta_of(cameras())
    vv0 = (W.splitmix64(3, n) % np.uint64(4)).astype(np.uint8)
    g, vv, vis, chg = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"],
                                   sc["flags"], sc["layers"], vv0, fr, np.array([1, 3], np.uint32), None)
    return dict(n=n, frusta=fr, view_masks=np.array([1, 3], np.uint32), vv0=vv0, global_bits=g.view(np.uint32), vv=vv,
                visible=vis, vv_changed=chg, **{k: v for k, v in sc.items() if k != "n"})


def tree_case():
    tr = W.gen_tree(6, 3)
    rc, g, chg = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert rc == 0
    return dict(global_bits=g.view(np.uint32), changed=chg, **tr)


def cluster_case(n=3000):
    cam = W.many_cubes_camera(0)
    cfv = O.perspective_infinite_reverse(np.float32(W.CAMERA_FOV), W.CAMERA_ASPECT, W.CAMERA_NEAR)
    fr = frusta_of([cam])
    view = O.cluster_view_setup(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    lights = W.many_lights(n, 50.0, 2.5)
    off, idx, counts, far, total = O.assign_objects_to_clusters(view, lights)
    return dict(camera=cam, clip_from_view=cfv, frustum=fr, lights=lights, offsets=off, indices=idx, counts=counts,
                farthest_z=np.float32(far), total=np.uint64(total))


def main():
    np.savez_compressed(os.path.join(HERE, "flat_frame_777.npz"), **flat_case())
    np.savez_compressed(os.path.join(HERE, "tree_6x3.npz"), **tree_case())
    np.savez_compressed(os.path.join(HERE, "cluster_3000.npz"), **cluster_case())
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
