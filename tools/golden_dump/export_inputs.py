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
    migd.write(os.path.join(HERE, "inputs.migd"), out)
    print("wrote", os.path.join(HERE, "inputs.migd"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
