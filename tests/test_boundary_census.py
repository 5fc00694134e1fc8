"""The boundary census (tests/boundary_census.py): how many ViewVisibility flags of the BASELINE configs would flip if glam's
Vec4::dot / Vec3A::dot / Mat3A * Vec3A were evaluated in an order other than the one the oracle and the kernels restate from
memory.  CPU only.  The numpy emulation is first checked against the oracle (same flags with the `used` orders), then the census
is recomputed and compared with the committed tests/golden/boundary_census.json (MI_UPDATE_GOLDEN=1 rewrites it), which DESIGN.md
section 3 and bench.py (`parity_census`) quote."""
import json
import os

import numpy as np

from bevy_amd import api, workloads as W
import boundary_census as BC
import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "boundary_census.json")
F = np.float32


def frusta_for(cams, far=W.CAMERA_FAR):
    cfv = O.perspective_infinite_reverse(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([O.compute_frustum_perspective(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR, far, cam) for cam in cams])


def oracle_flags(sc, frusta):
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    vv = O.reset_view_visibility(sc["flags"], np.zeros(sc["n"], np.uint8))
    _, vis, _ = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], vv, frusta)
    return g, vis


def scenes():
    """(name, scene, frusta, view kinds): configs[1], configs[2]'s rows of the metric frame, three windows of configs[3], and a
    ragged scene under directional-light cascades (OBB only, far plane tested)."""
    for frame in (0, 37, 150):
        yield f"configs[1] many_cubes 1M x 1 view, camera frame {frame}", W.many_cubes(1_000_000), frusta_for([W.many_cubes_camera(frame)]), None
    sc, first_light, pr = W.frame_scene(0, 100_000, 10_000)
    yield "configs[2] 10k meshes + 100k light spheres x 1 view", sc, frusta_for([W.many_cubes_camera(0)]), None
    n10 = 10_000_000
    radius = 500.0 * (n10 / 1_000_000.0) ** (1.0 / 3.0)
    cams = [W.many_cubes_camera(0, yaw=v * np.pi / 2) for v in range(4)]
    for start in (0, n10 // 2 - 32_768, n10 - 65_536):
        yield f"configs[3] 10M x 4 views, rows [{start}, {start + 65536})", W.many_cubes(n10, radius=radius, start=start, count=65_536), frusta_for(cams), None
    sc = W.many_cubes(60_000, radius=60.0, ragged_flags=True)
    sc["flags"] = sc["flags"] | np.uint8(0x80)  # shadow casters
    lights = [W.many_cubes_camera(0, yaw=0.4 + 0.1 * k, position=(0.0, 30.0, 0.0)) for k in range(3)]
    yield "ragged 60k x 3 cascades (OBB only, planes 0-3 + far)", sc, frusta_for(lights, far=200.0), [1, 1, 1]


def test_numpy_emulation_reproduces_the_oracle():
    """With the orders the oracle uses, the numpy evaluation gives the oracle's GlobalTransforms and flags bit for bit -- on a
    ragged scene (every branch of the closure) and on the first 200k rows of configs[1]."""
    for sc, cams in ((W.many_cubes(40_000, radius=60.0, ragged_flags=True), [W.many_cubes_camera(0), W.many_cubes_camera(3, yaw=1.0, position=(10.0, -5.0, 20.0))]),
                     (W.many_cubes(1_000_000, start=0, count=200_000), [W.many_cubes_camera(0)])):
        fr = frusta_for(cams)
        g_exp, vis_exp = oracle_flags(sc, fr)
        g = BC.global_transform(sc["translation"], sc["rotation"], sc["scale"])
        assert np.stack(g, axis=1).astype(F).tobytes() == g_exp.tobytes()
        used = BC.visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], fr, BC.USED)
        assert np.array_equal(used, vis_exp)


def test_boundary_census_matches_the_committed_counts():
    got = {}
    for name, sc, fr, kinds in scenes():
        c, used = BC.census(sc, fr, kinds)
        got[name] = c
        if kinds is None and sc["n"] <= 200_000:  # the emulation against the oracle once more, on the scene itself
            assert np.array_equal(used, oracle_flags(sc, fr)[1]), name
    if os.environ.get("MI_UPDATE_GOLDEN") == "1" or not os.path.exists(GOLDEN):
        json.dump(got, open(GOLDEN, "w"), indent=1)
    want = json.load(open(GOLDEN))
    assert got == want
    # what the census is for: say it in the test log
    total_rows = sum(c["rows"] * c["views"] for c in got.values())
    worst = {k: max(c["flips"][k] for c in got.values()) for k in next(iter(got.values()))["flips"]}
    print(f"{total_rows} (row, view) decisions; flips per alternative order, worst scene: {worst}")
