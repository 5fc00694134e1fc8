"""GPU parity of the static cull order (kernels_cells.hip builds it, k_frame_cells in kernels_flat.hip runs the frame over it).

A scene that has gone static is culled over a cell-ordered copy: 64 spatial neighbours per wave, a bounding sphere per wave that is
tested against every view first.  That test may only REJECT (with an explicit f32 margin); whatever is not rejected runs the
reference's per-entity rule (check_visibility_cpu_culling, crates/bevy_camera/src/visibility/mod.rs:788-858) unchanged, and results
are written by row.  So every output must stay bit-identical to the oracle -- and to the other frame kernels -- through any sequence of
frames, and the order must be dropped (and rebuilt) whenever anything it mirrors changes behind its back."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32
WHOLE = B.CULL_BEGIN_FRAME | B.CULL_END_FRAME


def assert_bits(a, b, what):
    bad = np.nonzero(np.asarray(a) != np.asarray(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first rows {bad[:8].tolist()}"


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


def oracle_cull(sc, g, vv, frusta, view_flags=None):
    vv1 = O.reset_view_visibility(sc["flags"], vv)
    vv2, vis, chg = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"], vv1, frusta, view_flags=view_flags)
    vv3, chg2 = O.check_visibility_gpu_culling(sc["flags"], vv2)
    vv4, chg3 = O.mark_newly_hidden(sc["flags"], vv3)
    return vv4, vis, chg | chg2 | chg3


def check_frame(ctx, vv_exp, vis_exp, chg_exp, what):
    for v in range(len(vis_exp)):
        assert_bits(ctx.download_visibility(v), vis_exp[v], f"{what}: view {v}")
        rows = ctx.download_visible_entities(v, 0)[1]
        assert np.array_equal(rows, np.nonzero(vis_exp[v])[0].astype(np.uint32)), f"{what}: VisibleEntities of view {v}"
    vv, chg = ctx.download_view_visibility()
    assert_bits(vv, vv_exp, f"{what}: ViewVisibility")
    assert_bits(chg, chg_exp, f"{what}: ViewVisibility change ticks")


def setup(ctx, sc, mode=2):
    n = sc["n"]
    ctx.debug_set_static_cull_order(mode)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    ctx.upload_changed(np.ones(n, np.uint8))
    ctx.propagate(0)


def cams(frame, k=2):
    return [W.many_cubes_camera(frame * 40, yaw=v * 1.7, position=(3.0 * v, -2.0 * v, 7.0 * v)) for v in range(k)]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 4097, 50_003])
@pytest.mark.parametrize("more", [0, B.CULL_MORE_FRAMES])
def test_static_frames_over_the_cell_order(n, more):
    """Nothing moves, the cameras do.  The first cull frames bring the world-sphere column up (k_frame, k_frame_sph); with mode 2 the
    first frame that finds it current builds the order.  ragged_flags: hidden rows, rows without bounds, NoFrustumCulling, NoCpuCulling
    rows and other RenderLayers -- the cells that hold them must not be rejected."""
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        setup(ctx, sc)
        vv = np.zeros(n, np.uint8)
        for frame in range(8):
            ctx.propagate(0)
            frusta = frusta_for(cams(frame))
            ctx.cull(frusta, flags=WHOLE | more)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"n={n} frame {frame}")
        builds, frames = ctx.debug_static_cull_counts()
        assert builds == 1 and frames >= 5, (builds, frames)


def test_long_list_runs_like_a_table_beyond_16_million_rows():
    """k_cells_blocks cuts a view's 4 096-row blocks into at most 256 runs of at least 16 blocks: up to 16.7 M rows a run is 16 blocks,
    beyond that it grows.  mi_debug_set_static_cull_order(3) caps the runs at three, so a 700 k-row table (171 blocks: runs of 57)
    walks the long-run path -- several scan rounds per run -- that only such tables reach otherwise."""
    n = 700_001
    sc = W.many_cubes(n, radius=250.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        setup(ctx, sc, mode=3)
        vv = np.zeros(n, np.uint8)
        for frame in range(5):
            ctx.propagate(0)
            frusta = frusta_for(cams(frame, 3))
            ctx.cull(frusta, flags=WHOLE | B.CULL_MORE_FRAMES)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame}")
        assert ctx.debug_static_cull_counts()[1] >= 2


def test_default_rule_builds_on_the_second_quiet_frame_and_only_for_big_tables():
    n = 3_000_000
    sc = W.many_cubes(n, radius=700.0)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        ctx.debug_set_sphere_path(0)  # (the suite also runs with the sphere column forced on at once: this test counts the default rule's frames)
        setup(ctx, sc, mode=0)
        vv = np.zeros(n, np.uint8)
        seen = []
        for frame in range(7):
            ctx.propagate(0)
            frusta = frusta_for(cams(frame, 3))
            ctx.cull(frusta, flags=WHOLE | B.CULL_MORE_FRAMES)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame}")
            seen.append(ctx.debug_static_cull_counts())
        # frame 0: k_frame (column invalid), frame 1: k_frame_sph rebuilds the column, frame 2: eligible once, frame 3: builds
        assert seen[2] == (0, 0) and seen[3] == (1, 1) and seen[6] == (1, 4), seen
    with api.Context(0) as ctx:  # a small table never takes it on its own
        sc = W.many_cubes(20_000, radius=60.0)
        setup(ctx, sc, mode=0)
        for frame in range(6):
            ctx.cull(frusta_for(cams(frame)), flags=WHOLE)
        assert ctx.debug_static_cull_counts() == (0, 0)


def test_whatever_the_order_mirrors_drops_it():
    """Moved rows, a frame of another kind, ViewVisibility written from outside, flags / bounds uploads, a resize: the order is dropped,
    the frames in between take the other kernels, and a new order is built once the scene is quiet again -- same bits throughout."""
    n = 40_009
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    t = sc["translation"].reshape(n, 3).copy()
    r4, s3 = sc["rotation"].reshape(n, 4), sc["scale"].reshape(n, 3)
    rng = np.random.default_rng(11)
    with api.Context(0) as ctx:
        setup(ctx, sc)
        g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
        vv = np.zeros(n, np.uint8)
        script = ["still", "still", "still", "move", "still", "still", "vv", "still", "still", "flags", "still", "still", "bounds", "still", "still",
                  "separate", "still", "still", "all", "still", "still", "still"]
        builds_before = frames_before = 0
        for frame, what in enumerate(script):
            if what == "move":
                moved = np.sort(rng.choice(n, 777, replace=False)).astype(np.uint32)
                t[moved] += rng.normal(0.0, 6.0, (777, 3)).astype(F)
                ctx.upload_transforms_indexed(moved, t[moved].reshape(-1), r4[moved].reshape(-1), s3[moved].reshape(-1))
                g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
            if what == "vv":  # ViewVisibility from outside (what the shim does after a structural change)
                vv = rng.integers(0, 4, n).astype(np.uint8) & np.uint8(1)
                vv = (vv | (vv << 1)).astype(np.uint8)  # 0 or 3: states a frame can leave
                ctx.upload_view_visibility(vv)
            if what == "flags":
                sc["flags"] = sc["flags"].copy()
                sc["flags"][::7] ^= np.uint8(0x01)  # InheritedVisibility toggles
                ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            if what == "bounds":
                sc["aabb_half"] = (sc["aabb_half"] * F(1.5)).astype(F)
                ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            if what == "all":
                ctx.upload_changed(np.ones(n, np.uint8))
            frusta = frusta_for(cams(frame))
            ctx.propagate(0)
            if what == "separate":  # the three visibility systems as calls of their own
                ctx.visibility_begin_frame()
                ctx.cull(frusta, flags=0)
                ctx.visibility_end_frame()
            else:
                ctx.cull(frusta, flags=WHOLE)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame} ({what})")
            builds, frames = ctx.debug_static_cull_counts()
            if what != "still":  # never over the order that was there before: another kernel, or (what left the sphere column current) a new order
                assert frames == frames_before or builds == builds_before + 1, f"frame {frame} ({what}) ran over a stale order"
            if what in ("move", "bounds", "separate", "all"):
                assert frames == frames_before, f"frame {frame} ({what}) cannot run over any order"
            builds_before, frames_before = builds, frames
        assert builds_before >= 7, builds_before  # rebuilt after every disturbance


def test_resize_and_new_rows():
    n0, n1 = 9_000, 12_345
    sc = W.many_cubes(n1, radius=60.0, ragged_flags=True)
    part = {k: (v[:n0 * (len(v) // n1)] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    part["n"] = n0
    with api.Context(0) as ctx:
        setup(ctx, part)
        g0, _ = O.sync_simple_transforms(part["translation"], part["rotation"], part["scale"])
        vv = np.zeros(n0, np.uint8)
        for frame in range(4):
            frusta = frusta_for(cams(frame))
            ctx.cull(frusta, flags=WHOLE)
            vv, vis, chg = oracle_cull(part, g0, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"before, frame {frame}")
        assert ctx.debug_static_cull_counts()[0] == 1
        ctx.resize(n1)
        ctx.upload_transforms(sc["translation"][3 * n0:], sc["rotation"][4 * n0:], sc["scale"][3 * n0:], first_row=n0)
        ctx.upload_bounds(sc["aabb_center"][3 * n0:], sc["aabb_half"][3 * n0:], sc["flags"][n0:], sc["layers"][n0:], first_row=n0)
        g1, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
        vv = np.concatenate([vv, np.zeros(n1 - n0, np.uint8)])
        for frame in range(4, 9):
            ctx.propagate(0)
            frusta = frusta_for(cams(frame))
            ctx.cull(frusta, flags=WHOLE)
            vv, vis, chg = oracle_cull(sc, g1, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"after, frame {frame}")
        assert ctx.debug_static_cull_counts()[0] == 2


def test_views_that_do_not_cull_and_many_views():
    """A camera with NoCpuCulling never rejects a cell; nine views go through the device array (no kernarg copy)."""
    n = 20_011
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    for k, vflags in ((2, np.array([0, B.VIEW_FLAG_NO_CPU_CULLING], np.uint8)), (9, None)):
        with api.Context(0) as ctx:
            setup(ctx, sc)
            vv = np.zeros(n, np.uint8)
            for frame in range(5):
                frusta = frusta_for(cams(frame, k))
                ctx.cull(frusta, view_flags=vflags, flags=WHOLE)
                vv, vis, chg = oracle_cull(sc, g, vv, frusta, view_flags=vflags)
                check_frame(ctx, vv, vis, chg, f"{k} views, frame {frame}")
            assert ctx.debug_static_cull_counts()[1] >= 2


def test_cells_at_the_frustum_boundary_are_not_rejected():
    """Rows laid exactly on and around a frustum plane (the far side of the margin included): whatever the wave test decides, the rows'
    own f32 test decides the bits."""
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(0)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    planes = fr.reshape(6, 4)
    rng = np.random.default_rng(5)
    pts = []
    for p in planes[:5]:
        nrm, d = p[:3].astype(np.float64), float(p[3])
        for _ in range(3000):
            q = rng.normal(0.0, 200.0, 3)
            q -= (q @ nrm + d) * nrm            # onto the plane
            r = 0.8660254                        # |half extents| of the unit cube: the row's sphere radius
            off = rng.choice([-r, -r * (1 + 1e-6), -r * (1 - 1e-6), -r + 1e-4, -r - 1e-4, 0.0, -2 * r])
            pts.append(q + off * nrm)
    t = np.asarray(pts, F)
    n = len(t)
    sc = W.many_cubes(n, radius=60.0)
    sc["translation"] = t.reshape(-1).copy()
    sc["rotation"] = np.tile(np.array([0, 0, 0, 1], F), n)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        setup(ctx, sc)
        vv = np.zeros(n, np.uint8)
        for frame in range(4):
            ctx.cull(fr, flags=WHOLE)
            vv, vis, chg = oracle_cull(sc, g, vv, fr)
            check_frame(ctx, vv, vis, chg, f"frame {frame}")
        assert ctx.debug_static_cull_counts()[1] >= 2


def test_static_frames_of_a_hierarchy():
    """Rows of a hierarchy are rows like any other once their GlobalTransforms stand still."""
    tr = W.gen_tree(8, 4)
    n = tr["n"]
    sc = dict(n=n, aabb_center=np.zeros(3 * n, F), aabb_half=np.full(3 * n, 0.5, F), flags=np.full(n, 0x05, np.uint8), layers=np.ones(n, np.uint32))
    rc, g, _ = O.propagate_transforms(tr["parent"], tr["translation"], tr["rotation"], tr["scale"])
    assert rc == 0
    with api.Context(0) as ctx:
        ctx.debug_set_static_cull_order(2)
        ctx.resize(n)
        ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
        ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.zeros(n, np.uint8))  # the change column exists from here on: only marked rows are recomputed
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        vv = np.zeros(n, np.uint8)
        for frame in range(6):
            ctx.propagate(B.PROPAGATE_STATIC_OPT)
            frusta = frusta_for([W.many_cubes_camera(frame * 40, position=(0.0, 0.0, 150.0)), W.many_cubes_camera(frame * 40, yaw=2.0)])
            ctx.cull(frusta, flags=WHOLE)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame}")
        assert ctx.debug_static_cull_counts()[1] >= 2


def test_ten_million_rows_four_views():
    """BASELINE.json configs[3]'s scene on one GPU, static: the order's frames against the world-sphere path's (bit-identical masks,
    ViewVisibility and change ticks), and the all-dirty frame in between drops the order."""
    n = 10_000_000
    sc = W.many_cubes(n, radius=500.0 * 10.0 ** (1.0 / 3.0))
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)

    def frusta(frame):
        return np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(frame, yaw=v * np.pi / 2), W.CAMERA_FAR) for v in range(4)])
    out = {}
    for mode in (1, 0):
        with api.Context(0) as ctx:
            ctx.debug_set_static_cull_order(mode)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            res = []
            for frame in range(7):
                if frame == 1:
                    ctx.upload_changed(np.zeros(n, np.uint8))  # the change column exists from here on: propagate(0) finds nothing to do
                if frame == 5:
                    ctx.propagate_and_cull(frusta(frame * 100), flags=B.CULL_END_FRAME)
                else:
                    ctx.propagate(0)
                    ctx.cull(frusta(frame * 100), flags=WHOLE | B.CULL_MORE_FRAMES)
                vvb, chg = ctx.download_view_visibility()
                res.append(([ctx.download_visibility(v).copy() for v in range(4)], vvb.copy(), chg.copy(),
                            [ctx.download_visible_entities(v, 0)[1].copy() for v in range(4)]))
            out[mode] = res
            if mode == 0:
                builds, frames = ctx.debug_static_cull_counts()  # built once before frame 5 (the all-rows frame drops it; frame 6 is quiet frame one)
                assert builds == 1 and frames >= 1, (builds, frames)
    for frame in range(7):
        a, b = out[1][frame], out[0][frame]
        for v in range(4):
            assert_bits(a[0][v], b[0][v], f"frame {frame}, view {v}")
            assert np.array_equal(a[3][v], b[3][v]), f"frame {frame}: VisibleEntities of view {v}"
        assert_bits(a[1], b[1], f"frame {frame}: ViewVisibility")
        assert_bits(a[2], b[2], f"frame {frame}: change ticks")
    assert sum(int(m.sum()) for m in out[0][4][0]) > 100_000


def test_lists_that_rode_in_the_next_frames_launch_are_that_frames_lists():
    """MI_CULL_MORE_FRAMES: frame f's lists are not launched behind it, they ride at the head of frame f + 1's k_frame_cells launch.
    Nothing the API exposes reads them afterwards (every read joins the LATEST frame's), so this test goes behind it: the three list
    buffers of the rotating sets are captured through mi_device_buffer during three warm frames, two more frames run with nothing read
    in between, and the buffer frame 6 wrote -- by the riders of frame 7's launch -- is fetched with hipMemcpy and compared with the
    oracle's lists for frame 6's cameras."""
    import ctypes as C
    n = 400_001
    n_views = 3
    sc = W.many_cubes(n, radius=160.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    hip = C.CDLL("libamdhip64.so")
    with api.Context(0) as ctx:
        setup(ctx, sc)
        vv = np.zeros(n, np.uint8)
        ptrs, expected = [], {}
        for frame in range(8):  # 0 - 2: the sphere column and the order come up; 3 - 5: the buffers are captured; 6, 7: nothing is read
            ctx.propagate(0)
            frusta = frusta_for(cams(frame, n_views))
            ctx.cull(frusta, flags=WHOLE | B.CULL_MORE_FRAMES)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            expected[frame] = [np.nonzero(vis[v])[0].astype(np.uint32) for v in range(n_views)]
            if frame < 3:
                ctx.synchronize()
            elif frame < 6:
                ptrs.append(ctx.device_buffer(3))  # MI_BUF_VISIBLE_ROWS of the set this frame wrote (the call joins: these frames do not ride)
        builds, frames_over_order = ctx.debug_static_cull_counts()
        assert builds == 1 and frames_over_order >= 5, (builds, frames_over_order)
        ctx.synchronize()  # (joins frame 7's lists; frame 6's came out of frame 7's launch)
        assert len({p for p, _ in ptrs}) == 3, ptrs
        p, nbytes = ptrs[0]  # the sets rotate by three: frame 6 wrote the set of frame 3
        stride = nbytes // (4 * n_views)
        host = np.zeros(nbytes // 4, np.uint32)
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(p), C.c_size_t(nbytes), 2) == 0
        for v in range(n_views):
            want = expected[6][v]
            assert want.size > 100 and not np.array_equal(want, expected[3][v][:want.size])  # (the cameras move: not frame 3's leftovers)
            assert np.array_equal(host[v * stride:v * stride + want.size], want), f"frame 6, view {v}: the riders' list"
        # and what the API shows is frame 7's
        for v in range(n_views):
            assert np.array_equal(ctx.download_visible_entities(v, 0)[1], expected[7][v])


def test_alternating_view_counts_do_not_rebuild_the_order():
    """A caller that alternates between view counts on a static scene (a picture-in-picture camera every other frame): each such frame
    cannot continue the masks of the frame before -- it starts from zeroed masks and zeroed contributions (CellsWork::fresh) over the SAME
    order.  Round 4 rebuilt the order (bounds, keys, radix sort, gather) on every one of those frames; results are identical either way."""
    n = 30_007
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    with api.Context(0) as ctx:
        setup(ctx, sc)
        vv = np.zeros(n, np.uint8)
        builds = []
        for frame in range(9):
            k = (1, 3, 2)[frame % 3] if frame < 6 else 2  # 1, 3, 2, 1, 3, 2 views, then three chained frames of 2
            frusta = frusta_for(cams(frame, k))
            ctx.cull(frusta, flags=WHOLE)
            vv, vis, chg = oracle_cull(sc, g, vv, frusta)
            check_frame(ctx, vv, vis, chg, f"frame {frame} with {k} view(s)")
            builds.append(ctx.debug_static_cull_counts()[0])
        # (the first frames run over the world-sphere column until it is current; from the build on every frame runs over the order)
        assert builds[-1] == 1, f"the order was built {builds} times over nine frames of one static scene"
        assert ctx.debug_static_cull_counts()[1] == 9 - builds.index(1)
