"""GPU parity of the per-wave row summary (kernels.h RowSummary, kernels_flat.hip k_row_summary): where the 64 rows of a wave agree
in Aabb / flags / RenderLayers the frame kernels take those from a 32-byte summary instead of the columns.

The summary is derived from the columns by the library (after mi_upload_bounds / mi_columns_resize / mi_visibility_propagate, for
the waves they touched), so results must be bit-identical to the oracle -- and to the same frames with the summary switched off --
for every mixture of uniform and ragged waves and across every kind of update: whole uploads, partial uploads that break or restore
a wave's uniformity, growth and shrinkage across wave boundaries, InheritedVisibility flips, and both frame kernels (k_frame,
k_frame_sph).  The reference reads the components of every entity (visibility/mod.rs:790-846): there is nothing to restate."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32


def assert_bits(a, b, what):
    bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first rows {bad[:8].tolist()}"


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


def oracle_frame(sc, g, vv, frusta):
    vv1 = O.reset_view_visibility(sc["flags"], vv)
    vv2, vis, chg = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], vv1, frusta)
    vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
    vv4, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
    return vv4, vis, chg | chg2 | chg3


def check_frame(ctx, vv_exp, vis_exp, chg_exp, what):
    for v in range(len(vis_exp)):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"{what}: view {v}")
    vv, chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, f"{what}: ViewVisibility")
    assert_bits(chg, chg_exp, f"{what}: ViewVisibility change ticks")


def mixed_scene(n, seed=5):
    """Blocks of rows that agree (whole waves and runs that straddle wave boundaries), blocks that do not, Sphere rows, hidden rows."""
    sc = W.many_cubes(n, radius=60.0, seed=seed, ragged_flags=True)
    c, h = sc["aabb_center"].reshape(n, 3), sc["aabb_half"].reshape(n, 3)
    fl, lay = sc["flags"], sc["layers"]
    rng = np.random.default_rng(seed)
    pos = 0
    while pos < n:
        run = int(rng.choice([1, 7, 64, 64, 128, 200, 1000]))
        end = min(n, pos + run)
        kind = int(rng.integers(0, 4))
        if kind <= 1:  # a uniform run: one mesh spawned `run` times
            c[pos:end] = c[pos]
            h[pos:end] = h[pos]
            fl[pos:end] = fl[pos]
            lay[pos:end] = lay[pos]
        elif kind == 2:  # Aabb uniform, flags / layers ragged
            c[pos:end] = c[pos]
            h[pos:end] = h[pos]
        # kind 3: as generated (ragged)
        pos = end
    return sc


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 4097, 70_001])
@pytest.mark.parametrize("fused", [False, True])
def test_summary_matches_the_oracle_and_the_plain_columns(n, fused):
    sc = mixed_scene(n)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    results = []
    for mode in (0, 1):
        with api.Context(0) as ctx:
            ctx.debug_set_row_summary(mode)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            vv = np.zeros(n, np.uint8)
            got = []
            for frame in range(3):
                frusta = frusta_for([W.many_cubes_camera(frame * 40), W.many_cubes_camera(frame * 40, yaw=2.1, position=(3.0, 1.0, -7.0))])
                if fused:
                    ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME)
                else:
                    ctx.propagate(B.PROPAGATE_ALL_DIRTY if frame == 0 else 0)
                    ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
                vv, vis, chg = oracle_frame(sc, g, vv, frusta)
                check_frame(ctx, vv, vis, chg, f"n={n} fused={fused} summary mode {mode} frame {frame}")
                got.append(ctx.download_view_visibility()[0].tobytes())
            results.append(got)
    assert results[0] == results[1]


@pytest.mark.parametrize("sphere_path", [1, 2])
def test_updates_keep_the_summary_current(sphere_path):
    """One context through a sequence of writes to the summarised columns; after each, a frame against the oracle."""
    n0 = 20_000
    sc = W.many_cubes(n0 + 5_000, radius=60.0)  # uniform: one Aabb, one flags byte, one layer mask -> every wave summarised
    n = n0
    cur = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    rng = np.random.default_rng(11)

    def sub(k, per, lo, hi):
        return np.ascontiguousarray(cur[k].reshape(-1, per)[lo:hi]).reshape(-1) if per > 1 else np.ascontiguousarray(cur[k][lo:hi])

    with api.Context(0) as ctx:
        ctx.debug_set_sphere_path(sphere_path)
        ctx.resize(n)
        ctx.upload_transforms(sub("translation", 3, 0, n), sub("rotation", 4, 0, n), sub("scale", 3, 0, n))
        ctx.upload_bounds(sub("aabb_center", 3, 0, n), sub("aabb_half", 3, 0, n), sub("flags", 1, 0, n), sub("layers", 1, 0, n))
        vv = np.zeros(n, np.uint8)
        frame = [0]

        def run(what, all_dirty=False):
            nonlocal vv
            f = frame[0]
            frame[0] += 1
            frusta = frusta_for([W.many_cubes_camera(f * 30), W.many_cubes_camera(f * 30, yaw=1.1)])
            ctx.propagate(B.PROPAGATE_ALL_DIRTY if all_dirty else 0)
            ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            view = dict(aabb_center=sub("aabb_center", 3, 0, n), aabb_half=sub("aabb_half", 3, 0, n), flags=sub("flags", 1, 0, n), layers=sub("layers", 1, 0, n))
            g, _ = O.sync_simple_transforms(sub("translation", 3, 0, n), sub("rotation", 4, 0, n), sub("scale", 3, 0, n))
            vv, vis, chg = oracle_frame(view, g, vv[:n] if len(vv) >= n else np.concatenate([vv, np.zeros(n - len(vv), np.uint8)]), frusta)
            check_frame(ctx, vv, vis, chg, f"sphere path {sphere_path}, {what}")

        def upload_bounds(lo, hi):
            ctx.upload_bounds(sub("aabb_center", 3, lo, hi), sub("aabb_half", 3, lo, hi), sub("flags", 1, lo, hi), sub("layers", 1, lo, hi), first_row=lo)

        run("uniform scene", all_dirty=True)
        run("uniform scene, second frame")
        # one row in the middle of a wave gets another Aabb: that wave reads its columns again
        cur["aabb_half"].reshape(-1, 3)[7000 + 13] = (2.0, 0.25, 1.5)
        upload_bounds(7000 + 13, 7000 + 14)
        run("one odd Aabb")
        # a run that straddles three waves gets another layer mask and the NoFrustumCulling flag
        cur["layers"][9990:10130] = 2
        cur["flags"][10000:10100] |= 0x02
        upload_bounds(9990, 10130)
        run("odd layers / flags over a wave boundary")
        # the odd row goes back: the wave is uniform again
        cur["aabb_half"].reshape(-1, 3)[7000 + 13] = (0.5, 0.5, 0.5)
        upload_bounds(7000, 7064)
        run("restored")
        # growth across a wave boundary: the new rows have default bounds until uploaded (zero Aabb, layer 0, InheritedVisibility)
        old_n = n
        n = n0 + 3_333
        ctx.resize(n)
        cur["aabb_center"].reshape(-1, 3)[old_n:n] = 0
        cur["aabb_half"].reshape(-1, 3)[old_n:n] = 0
        cur["flags"][old_n:n] = 0x01
        cur["layers"][old_n:n] = 1
        ctx.upload_transforms(sub("translation", 3, old_n, n), sub("rotation", 4, old_n, n), sub("scale", 3, old_n, n), first_row=old_n)
        run("grown, bounds of the new rows not uploaded", all_dirty=True)
        cur["aabb_half"].reshape(-1, 3)[old_n:n] = 0.5
        cur["flags"][old_n:n] = 0x05
        upload_bounds(old_n, n)
        run("grown, bounds uploaded")
        # shrink into the middle of a wave
        n = 12_345
        ctx.resize(n)
        vv = vv[:n]
        run("shrunk")
        # random rows hidden / shown through their flags
        rows = np.sort(rng.choice(n, 50, replace=False))
        cur["flags"][rows] &= ~np.uint8(0x01)
        for r in rows:
            upload_bounds(int(r), int(r) + 1)
        run("50 rows hidden")


def test_visibility_propagate_invalidates_the_flags_part():
    """mi_visibility_propagate rewrites bit 0 of the flags column on the device: the summary's flags must follow."""
    n = 10_000
    sc = W.many_cubes(n, radius=60.0)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        vv = np.zeros(n, np.uint8)
        flags = sc["flags"].copy()
        for frame in range(4):
            vis_comp = np.full(n, B.VISIBILITY_INHERITED, np.uint8)
            if frame % 2 == 1:
                vis_comp[100:4321] = B.VISIBILITY_HIDDEN
            ctx.upload_visibility(vis_comp)
            ctx.visibility_propagate()
            flags = (flags & ~np.uint8(1)) | np.where(vis_comp == B.VISIBILITY_HIDDEN, 0, 1).astype(np.uint8)
            frusta = frusta_for([W.many_cubes_camera(frame * 25)])
            ctx.propagate(B.PROPAGATE_ALL_DIRTY if frame == 0 else 0)
            ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME)
            view = dict(sc, flags=flags)
            vv, vis, chg = oracle_frame(view, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame}")
