"""What the unpinned last ulp could change (CPU only).

glam 0.33.2 -- the crate whose SSE2 lane orders decide the last bit of every product on this path -- is a crates.io dependency absent
from /root/reference, and neither box has a Rust toolchain: the oracle (and the kernels) restate its orders from memory.  This test
BOUNDS what a wrong memory could cost.  The oracle is rebuilt with one order swapped at a time (the ORC_* switches at the top of
oracle/bevy_oracle.c, `oracle_lib.PARITY_VARIANTS`) and once with every a*b+c fused (-ffp-contract=fast -mfma); every BASELINE
config is run end to end -- compute_frustum, propagate, check_visibility, the gather of the visible lights, cluster_view_setup,
assign_objects_to_clusters -- under the oracle proper and under each variant, at FULL size:

    configs[1]  1 M many_cubes x 1 camera                       ViewVisibility bits, per-view flags
    configs[2]  10 k meshes + 100 k point lights, 16 x 9 x 24    flags of the lights, (cluster, light) entries of the assignment
    configs[3]  10 M many_cubes x 4 cameras                     per-view flags (40 M decisions, in 1 M-row chunks)
    configs[4]  gen_tree(12, 4) cut at 1 M nodes                GlobalTransform lanes beyond 1e-5 (the north star's tolerance)

and reports (a) the histogram of |n . c + d + r| in ulps of its largest term over every (row, view, plane) value that decides
visibility (reference: crates/bevy_camera/src/primitives.rs:255-294, the `<= 0.0` at :263 and :289), (b) per variant the number of
visibility flags, cluster entries (crates/bevy_light/src/cluster/assign.rs:1046-1062 decides the z slice) and GlobalTransform lanes
that differ from the product order's.  The counts are committed as tests/golden/parity_margin.json (MI_UPDATE_GOLDEN=1 rewrites it),
quoted by DESIGN.md section 3.  A zero row means "bit-identical ViewVisibility" does not depend on that order on this config; a
non-zero row names exactly how many rows to look at when tools/golden_dump finally runs."""
import json
import os

import numpy as np
import pytest

from bevy_amd import api, workloads as W
import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_margin.json")
F = np.float32
VARIANTS = [None] + list(O.PARITY_VARIANTS)  # None = the oracle proper (the product's orders)
N10 = 10_000_000
CHUNK = 1_000_000


def _have_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


def frusta_for(cams):
    return np.concatenate([O.compute_frustum_perspective(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR, W.CAMERA_FAR, cam) for cam in cams])


def flat_flags(sc, frusta):
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    vv = O.reset_view_visibility(sc["flags"], np.zeros(sc["n"], np.uint8))
    vv, vis, _ = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], vv, frusta)
    return g, vv, vis


def run_config1(hist):
    sc = W.many_cubes(1_000_000)
    fr = frusta_for([W.many_cubes_camera(0)])
    g, vv, vis = flat_flags(sc, fr)
    if hist is not None:
        hist += O.visibility_margin_census(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], fr)
    return {"flags": np.packbits(vis), "view_visibility": vv}


def run_config2(hist):
    """The lights' and meshes' rows of the metric frame and the assignment of the visible lights (reference sequence: the gather of
    assign.rs:190-215, then :487-804)."""
    sc, first_light, pr = W.frame_scene(0, 100_000, 10_000)
    cam = W.many_cubes_camera(0)
    fr = frusta_for([cam])
    g, vv, vis = flat_flags(sc, fr)
    if hist is not None:
        hist += O.visibility_margin_census(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], fr)
    n_l = len(pr) // 4
    keep = np.nonzero((vv[first_light:first_light + n_l] & 1) != 0)[0]
    pr_g = np.asarray(pr, F).reshape(-1, 4)[keep].copy()
    pr_g[:, :3] = g.reshape(-1, 12)[first_light + keep, 9:12]
    cfv = O.perspective_infinite_reverse(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    ov = O.cluster_view_setup(cam, cfv, fr[:24], 1920, 1080, (16, 9, 24), 5.0, 1000.0)
    off, idx, counts, far, total = O.assign_objects_to_clusters(ov, pr_g.reshape(-1))
    cluster_of = np.repeat(np.arange(len(off) - 1, dtype=np.uint64), np.diff(off).astype(np.int64))
    entries = (cluster_of << np.uint64(32)) | keep[idx].astype(np.uint64)  # (cluster, light) pairs: what the lists say, order aside
    return {"flags": np.packbits(vis), "view_visibility": vv, "cluster_entries": np.sort(entries), "farthest_z": np.array([far], F)}


def run_config3(hist):
    radius = 500.0 * (N10 / 1_000_000.0) ** (1.0 / 3.0)
    fr = frusta_for([W.many_cubes_camera(0, yaw=v * np.pi / 2) for v in range(4)])
    packed = []
    for start in range(0, N10, CHUNK):
        sc = W.many_cubes(N10, radius=radius, start=start, count=CHUNK)
        g, vv, vis = flat_flags(sc, fr)
        if hist is not None:
            hist += O.visibility_margin_census(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], fr)
        packed.append(np.packbits(vis, axis=1))
    return {"flags": np.concatenate(packed, axis=1)}


def run_config4(hist):
    tr = W.gen_tree(12, 4, 1_000_000)
    rc, g, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert rc == 0
    return {"global_transform": g}


CONFIGS = [("configs[1] 1M many_cubes x 1 view", run_config1), ("configs[2] 10k meshes + 100k lights, 16x9x24 clusters", run_config2),
           ("configs[3] 10M many_cubes x 4 views", run_config3), ("configs[4] 1M-node tree, depth 12", run_config4)]


def differences(base, alt):
    out = {}
    for k, a in base.items():
        b = alt[k]
        if k == "flags":
            out["visibility_flags"] = int(np.unpackbits(a ^ b).sum())
        elif k == "view_visibility":
            out["view_visibility_bytes"] = int(np.count_nonzero(a != b))
        elif k == "cluster_entries":
            out["cluster_entries"] = int(len(np.setxor1d(a, b, assume_unique=True)))
        elif k == "farthest_z":
            out["farthest_z_bits"] = int(a.view(np.uint32)[0] != b.view(np.uint32)[0])
        elif k == "global_transform":
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            out["global_transform_lanes_differing"] = int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))
            out["global_transform_lanes_beyond_1e-5"] = int(np.count_nonzero(d > 1e-5))
            out["global_transform_max_abs_diff"] = float(d.max())
            # against the scene's own scale: a lane is a sum of terms of the magnitude of the tree's translations (hundreds of units at
            # depth 12), so one rounding step of such a TERM -- not of the lane -- is what another order can move a lane by; 1e-5 abs
            # is below that step for |term| >= 128 (ulp(128) = 1.5e-5)
            scale = float(np.abs(a).max())
            out["scene_scale_max_abs_lane"] = scale
            out["global_transform_max_abs_diff_in_ulps_of_scene_scale"] = float(d.max() / np.spacing(F(scale)))
    return out


@pytest.mark.timeout(900)
def test_parity_margin_of_the_baseline_configs():
    names = [v for v in VARIANTS if v != "fma" or _have_fma()]
    table = {}
    for cname, fn in CONFIGS:
        hist = np.zeros(6, np.uint64)
        base = fn(hist)
        row = {"deciding_values": int(hist[5]), "within_ulps": {str(k): int(h) for k, h in zip((1, 4, 16, 64, 1024), hist[:5])}} if hist[5] else {}
        if "cluster_entries" in base:
            row["cluster_entries_total"] = int(len(base["cluster_entries"]))
        row["differs_from_product_order"] = {}
        for v in names[1:]:
            with O.variant(v):
                alt = fn(None)
            row["differs_from_product_order"][v] = differences(base, alt)
        table[cname] = row
    if os.environ.get("MI_UPDATE_GOLDEN") == "1" or not os.path.exists(GOLDEN):
        json.dump(table, open(GOLDEN, "w"), indent=1)
    want = json.load(open(GOLDEN))
    if not _have_fma():
        for row in want.values():
            row["differs_from_product_order"].pop("fma", None)
    assert table == want
    # what the bound is for: "bit-identical ViewVisibility" must not hang on the lane orders nobody could check here
    for cname, row in table.items():
        for v, d in row["differs_from_product_order"].items():
            if v == "fma":
                continue  # (not an order glam could have: Bevy builds without FMA contraction; reported for scale)
            assert d.get("visibility_flags", 0) == 0 and d.get("view_visibility_bytes", 0) == 0, (cname, v, d)
            # GlobalTransform: another Mat3A * Vec3A order moves lanes by a rounding step of the scene's scale (reported, bounded)
            assert d.get("global_transform_max_abs_diff_in_ulps_of_scene_scale", 0.0) <= 8.0, (cname, v, d)
    print(json.dumps(table, indent=1))


def test_variant_builds_really_differ():
    """The switches do something: on values chosen to expose each order the variant's result differs from the oracle proper's in
    the last place (otherwise a zero count above would prove nothing)."""
    rng = np.random.default_rng(5)
    n = 20_000
    t = (rng.standard_normal((n, 3)) * 100).astype(F).reshape(-1)
    r = rng.standard_normal((n, 4)).astype(F)
    r = (r / np.linalg.norm(r, axis=1, keepdims=True)).astype(F).reshape(-1)
    s = (0.5 + rng.random((n, 3))).astype(F).reshape(-1)
    c = (rng.standard_normal((n, 3))).astype(F).reshape(-1)
    h = (0.5 + rng.random((n, 3))).astype(F).reshape(-1)
    flags = np.full(n, O.FLAG_INHERITED_VISIBLE | O.FLAG_HAS_AABB, np.uint8)
    layers = np.ones(n, np.uint32)
    cam = W.many_cubes_camera(0)

    ab = rng.standard_normal((512, 8)).astype(F)

    def probe():
        fr = frusta_for([cam])
        g, _ = O.sync_simple_transforms(t, r, s)
        parent = np.concatenate([[O.NO_PARENT], np.arange(63)]).astype(np.uint32)  # a chain: every level a Mat3A * Vec3A on top of the last
        rc, gt, _ = O.propagate_transforms(parent, t[:192], r[:256], s[:192])
        lanes = np.zeros((len(ab), 3), F)
        for i, row in enumerate(ab):
            O.lib().orc_probe_lane_orders(O.fp(row[:4].copy()), O.fp(row[4:].copy()), O.fp(lanes[i]))
        return fr, gt, lanes

    base = probe()
    for v in O.PARITY_VARIANTS:
        if v == "fma" and not _have_fma():
            continue
        with O.variant(v):
            alt = probe()
        same = all(np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)) for a, b in zip(base, alt) if a is not None)
        assert not same, f"variant {v} computed the same bits as the oracle proper on every probe"
