#!/usr/bin/env python
"""Writes tools/golden_dump/inputs.migd: the INPUT halves of the committed fixtures (tests/golden/*.npz) plus the frame
parameters, for the Rust tool to feed to the real Bevy systems.

    python tools/golden_dump/export_inputs.py

Nothing here touches the oracle: the inputs are seeded generator outputs that the fixtures already hold."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import migd  # noqa: E402
from bevy_amd import workloads as W  # noqa: E402


def spot_scene(seed=4242, n=1500):
    """Case 4 of the dump: point and spot lights (points first: the gather order), unit quaternions, outer angles, two cameras.
    Seeded here (no fixture holds it); tests/test_reference_dump.py regenerates it and compares."""
    rng = np.random.default_rng(seed)
    kind = np.sort(rng.integers(0, 2, n)).astype(np.uint8)
    pos = rng.uniform(-40, 40, size=(n, 3)).astype(np.float32)
    rng_ = rng.uniform(0.5, 12.0, n).astype(np.float32)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    q[kind == 0] = np.array([0, 0, 0, 1], np.float32)  # point lights: identity rotation
    outer = rng.uniform(0.1, 1.3, n).astype(np.float32)
    cams = np.concatenate([W.many_cubes_camera(0), W.many_cubes_camera(7, yaw=1.1)]).astype(np.float32)
    return {"cluster2.lights_pos_range": np.concatenate([pos, rng_[:, None]], axis=1).reshape(-1), "cluster2.type": kind,
            "cluster2.rotation": q.reshape(-1), "cluster2.outer_angle": outer, "cluster2.cameras": cams}


def main():
    g = os.path.join(ROOT, "tests", "golden")
    flat, tree, cl = (np.load(os.path.join(g, f)) for f in ("flat_frame_777.npz", "tree_6x3.npz", "cluster_3000.npz"))
    out = {}
    for k in ("translation", "rotation", "scale", "aabb_center", "aabb_half", "flags", "layers", "vv0", "view_masks"):
        out["flat." + k] = flat[k]
    out["flat.cameras"] = np.concatenate([W.many_cubes_camera(0), W.many_cubes_camera(5, yaw=np.pi / 2)]).astype(np.float32)
    out["camera.fov_aspect_near_far"] = np.array([W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR, W.CAMERA_FAR], np.float32)
    for k in ("parent", "translation", "rotation", "scale"):
        out["tree." + k] = tree[k]
    out["cluster.camera"] = cl["camera"]
    out["cluster.lights_pos_range"] = cl["lights"]
    out["cluster.screen_dims_z"] = np.array([1920, 1080, 16, 9, 24], np.uint32)
    out["cluster.first_slice_depth_far_z"] = np.array([5.0, 1000.0], np.float32)
    out.update(spot_scene())
    migd.write(os.path.join(HERE, "inputs.migd"), out)
    print("wrote", os.path.join(HERE, "inputs.migd"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
