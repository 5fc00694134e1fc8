"""GPU parity of dense uploads in pieces with the GlobalTransforms fetched ahead of the frame (context.cpp: mi_commit_upload_window
-- a sequence of dense windows that carries the whole flat table goes out piece by piece on a stream of its own, each piece's
GlobalTransforms are computed at once and start back to the host while the later pieces still arrive -- and
mi_download_frame_results, which hands them out when the frame in between was an all-rows one over the same Transforms): two
contexts through the same frames, one with the pieces, one without (mi_debug_set_chunked_frames(1)), and the oracle for the
GlobalTransforms -- changed rows, their GlobalTransforms, VisibleEntities, cluster lists, ViewVisibility, in place and copied out;
one window for the whole table and several windows committed one after the other; sequences that are broken off, windows out of
order, Transforms written again between the upload and the frame (what was fetched ahead no longer applies), frames that are
not all-rows frames.  The reference has no counterpart (sync_simple_transforms writes in place, systems.rs:45-50): what is pinned
is that the library's two ways deliver the same bytes."""
import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32


def frame_inputs(frame):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    cam = W.many_cubes_camera(frame * 20)
    fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
    view, keep = api.cluster_view_build(cam, cfv, fr, 1920, 1080, (16, 9, 24), 5.0, 1000.0, with_spheres=False)
    return fr, view, keep


@pytest.mark.parametrize("binding,mode", [("range", 0), ("list", 0), ("range", 2)])
@pytest.mark.parametrize("in_place", [True, False])
def test_all_dirty_frames_in_pieces(binding, mode, in_place):
    n_cubes, n_lights = 300_000, 3_000
    sc, first_light, pr = W.frame_scene(n_cubes, n_lights, 500, light_range=3.0)
    n = sc["n"]
    c_dev, h_dev = sc["aabb_center"].reshape(n, 3).copy(), sc["aabb_half"].reshape(n, 3).copy()
    c_dev[first_light:] = 0.0
    h_dev[first_light:, 1] = np.frombuffer(np.uint32(0x7FC0A11D).tobytes(), F)[0]
    t = sc["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(4)
    ctxs = [api.Context(0), api.Context(0)]
    ctxs[0].debug_set_chunked_frames(mode)  # 0: fetches ahead once the caller has fetched every GlobalTransform of such a frame; 2: always
    ctxs[1].debug_set_chunked_frames(1)
    try:
        for ctx in ctxs:
            ctx.resize(n)
            ctx.upload_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
            ctx.upload_bounds(c_dev.reshape(-1), h_dev.reshape(-1), sc["flags"], sc["layers"])
            ctx.cluster_upload_objects(pr)
            if binding == "range":
                ctx.cluster_bind_objects_to_rows(first_light, n_lights)
            else:
                ctx.cluster_bind_objects_to_row_list(np.arange(first_light, n, dtype=np.uint32))
        for frame in range(5):
            t += rng.normal(0.0, 0.5, (n, 3)).astype(F)  # everything moves
            fr, view, keep = frame_inputs(frame)
            outs = []
            for ctx in ctxs:
                w, _, wt, wr, ws = ctx.map_upload_window(n, dense=True)
                wt[:] = t.reshape(-1)
                wr[:] = sc["rotation"]
                ws[:] = sc["scale"]
                ctx.commit_upload_window(w, n)
                ctx.cluster_upload_view(view)
                if frame == 3:  # something else between the upload and the frame: it must see the whole upload
                    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
                    g_direct = ctx.download_global_transforms(want_changed=False)
                    w, _, wt, wr, ws = ctx.map_upload_window(n, dense=True)
                    wt[:] = t.reshape(-1)
                    wr[:] = sc["rotation"]
                    ws[:] = sc["scale"]
                    ctx.commit_upload_window(w, n)
                ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS | (B.CULL_MORE_FRAMES if frame % 2 else 0))
                bufs = api.FrameResultBuffers(n, n, view.n_clusters, 1 << 20, in_place=in_place)
                got = ctx.download_frame_results(bufs)
                outs.append({k: (np.array(v).copy() if isinstance(v, np.ndarray) else v) for k, v in got.items() if k != "lists"})
                outs[-1]["vv"] = ctx.download_view_visibility()[0].copy()
                if frame == 3:
                    outs[-1]["g_direct"] = g_direct
            a, b = outs
            for key in a:
                if isinstance(a[key], np.ndarray):
                    assert a[key].tobytes() == b[key].tobytes(), f"{binding} in_place={in_place} frame {frame}: {key} differs between the two forms"
                else:
                    assert a[key] == b[key], f"{binding} frame {frame}: {key}"
            g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
            assert len(a["changed_rows"]) == n and np.array_equal(a["changed_rows"], np.arange(n, dtype=np.uint32))
            assert a["changed_global"].tobytes() == g.tobytes(), f"frame {frame}: GlobalTransforms against the oracle"
            assert a["cluster_total"] > 0 and len(a["visible_rows"]) > 0
            if frame == 3:
                assert a["g_direct"].tobytes() == g.tobytes()
        # the pieces were taken where they should have been: six dense windows went out in pieces; with the default rule the first
        # frame fetches the usual way (and shows that the caller wants them all), frame 3's first upload was fetched ahead for nothing
        # (mi_propagate + a direct download came between), so its second upload did not fetch ahead and its download asked again
        assert ctxs[0].debug_chunked_counts()[:2] == ((6, 3) if mode == 0 else (6, 5)) and ctxs[1].debug_chunked_counts() == (0, 0, 0)
    finally:
        for ctx in ctxs:
            ctx.close()


def test_an_indexed_upload_after_a_dense_one():
    """A dense window for every row, then an indexed upload on top of it (the scatter kernel must run behind the whole dense upload),
    then a changed-rows frame."""
    n = 270_000
    sc = W.many_cubes(n, radius=80.0)
    t = sc["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(9)
    with api.Context(0) as ctx:
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        fr, _, _ = frame_inputs(0)
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
        for frame in range(3):
            t += F(0.25)
            w, _, wt, wr, ws = ctx.map_upload_window(n, dense=True)
            wt[:] = t.reshape(-1)
            wr[:] = sc["rotation"]
            ws[:] = sc["scale"]
            ctx.commit_upload_window(w, n)
            rows = np.sort(rng.choice(n, 100, replace=False)).astype(np.uint32)
            t[rows] += F(3.0)
            ctx.upload_transforms_indexed(rows, t[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1), sc["scale"].reshape(n, 3)[rows].reshape(-1))
            ctx.upload_changed(np.ones(n, np.uint8))  # (the dense window carries no change marks: every row counts)
            ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
            g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
            assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes(), f"frame {frame}"


def run_frames(ctx, sc, n, frames, upload, frame_flags=0, view=None):
    """frames x (upload(ctx, frame, t) -> an all-rows frame -> results in place); returns per frame (rows, G, visible rows)."""
    out = []
    for frame in range(frames):
        t = upload(ctx, frame)
        fr, _, _ = frame_inputs(frame)
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | frame_flags)
        got = ctx.download_frame_results(api.FrameResultBuffers(n, n, 0, 0, in_place=True))
        out.append((np.array(got["changed_rows"]).copy(), np.array(got["changed_global"]).copy(), np.array(got["visible_rows"]).copy(), t.copy()))
    return out


def dense(ctx, lo, hi, t, sc):
    w, _, wt, wr, ws = ctx.map_upload_window(hi - lo, dense=True)
    wt[:] = t[lo:hi].reshape(-1)
    wr[:] = sc["rotation"].reshape(-1, 4)[lo:hi].reshape(-1)
    ws[:] = sc["scale"].reshape(-1, 3)[lo:hi].reshape(-1)
    ctx.commit_upload_window(w, hi - lo, first_row=lo)


@pytest.mark.parametrize("pattern", ["eight windows", "ragged windows", "broken off", "out of order", "twenty windows", "written again"])
def test_windows_committed_one_after_the_other(pattern):
    """The caller fills and commits the table a window at a time (the upload of one crosses PCIe while it fills the next)."""
    n = 270_001
    sc = W.many_cubes(n, radius=80.0)
    t0 = sc["translation"].reshape(n, 3).copy()
    cuts = {"eight windows": [n * k // 8 for k in range(9)], "ragged windows": [0, 1, 100_001, 100_002, 250_000, n],
            "broken off": [0, n // 3, 2 * n // 3, n], "out of order": [0, n // 2, n], "twenty windows": [n * k // 20 for k in range(21)],
            "written again": [0, n // 2, n]}[pattern]
    results = []
    for mode in (2, 1):
        with api.Context(0) as ctx:
            ctx.debug_set_chunked_frames(mode)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            t = t0.copy()
            rng = np.random.default_rng(21)

            def upload(ctx, frame):
                t[:] += rng.normal(0.0, 0.4, (n, 3)).astype(F)
                pieces = list(zip(cuts[:-1], cuts[1:]))
                if pattern == "out of order":
                    pieces = pieces[::-1]
                for k, (lo, hi) in enumerate(pieces):
                    dense(ctx, lo, hi, t, sc)
                    if pattern == "broken off" and k == 0:  # another call between two windows: the next one starts over
                        ctx.upload_visibility_classes(np.ones(n, np.uint32))
                if pattern == "written again":  # some rows move once more: what was fetched ahead is not this frame's
                    rows = np.arange(7, n, 1009, dtype=np.uint32)
                    t[rows] += F(2.0)
                    ctx.upload_transforms_indexed(rows, t[rows].reshape(-1), sc["rotation"].reshape(n, 4)[rows].reshape(-1), sc["scale"].reshape(n, 3)[rows].reshape(-1))
                return t

            results.append(run_frames(ctx, sc, n, 3, upload))
            if mode == 2:
                ahead = ctx.debug_chunked_counts()[1]
                assert ahead == (3 if pattern in ("eight windows", "ragged windows") else 0), f"{pattern}: {ahead} downloads handed out what was fetched ahead"
    for frame, (a, b) in enumerate(zip(*results)):
        g, _ = O.sync_simple_transforms(a[3].reshape(-1), sc["rotation"], sc["scale"])
        assert np.array_equal(a[0], np.arange(n, dtype=np.uint32)) and np.array_equal(a[0], b[0])
        assert a[1].tobytes() == g.tobytes(), f"{pattern}, frame {frame}: GlobalTransforms against the oracle"
        assert a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]), f"{pattern}, frame {frame}: the two forms differ"


def test_a_changed_rows_frame_behind_a_dense_upload_fetches_the_usual_way():
    """Every Transform arrives, but the frame propagates the marked rows only: what was fetched ahead is not what the frame wrote."""
    n = 270_000
    sc = W.many_cubes(n, radius=80.0)
    t = sc["translation"].reshape(n, 3).copy()
    with api.Context(0) as ctx:
        ctx.debug_set_chunked_frames(2)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        fr, _, _ = frame_inputs(0)
        ctx.upload_changed(np.zeros(n, np.uint8))
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
        g_before = ctx.download_global_transforms(want_changed=False).reshape(n, 12).copy()
        t += F(0.5)
        dense(ctx, 0, n, t, sc)
        marks = np.zeros(n, np.uint8)
        marks[::3] = 1
        ctx.upload_changed(marks)
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
        got = ctx.download_frame_results(api.FrameResultBuffers(n, n, 0, 0, in_place=True))
        g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
        rows = np.arange(0, n, 3, dtype=np.uint32)
        assert np.array_equal(np.array(got["changed_rows"]), rows)
        assert np.array(got["changed_global"]).reshape(-1, 12).tobytes() == g.reshape(n, 12)[rows].tobytes()
        g_now = ctx.download_global_transforms(want_changed=False).reshape(n, 12)
        keep = np.ones(n, bool)
        keep[rows] = False
        assert g_now[keep].tobytes() == g_before[keep].tobytes()  # the rows that were not marked keep their GlobalTransform
        assert ctx.debug_chunked_counts()[1:] == (0, 0)


# ---- changed-rows frames: the GlobalTransforms of an indexed upload window, written ahead by the scatter launch -------------------------
def window(ctx, rows, t, sc, n):
    k = len(rows)
    w, wrows, wt, wr, ws = ctx.map_upload_window(k)
    wrows[:] = rows
    wt[:] = t[rows].reshape(-1)
    wr[:] = sc["rotation"].reshape(n, 4)[rows].reshape(-1)
    ws[:] = sc["scale"].reshape(n, 3)[rows].reshape(-1)
    ctx.commit_upload_window(w, k)


SPARSE_CASES = ["plain", "descending", "not monotonic", "a mark from elsewhere", "two windows", "a dense write in between", "an all-rows frame",
                "a second frame before the results", "results twice", "capacity too small", "default rule", "windows mapped before the results"]


@pytest.mark.parametrize("case", SPARSE_CASES)
@pytest.mark.parametrize("in_place", [True, False])
def test_changed_rows_frames_with_globals_written_ahead(case, in_place):
    n = 60_000
    sc = W.many_cubes(n, radius=60.0)
    outs, counts = [], []
    for mode in ((0 if case == "default rule" else 2), 1):
        t = sc["translation"].reshape(n, 3).copy()
        rng = np.random.default_rng(77)
        got_all = []
        with api.Context(0) as ctx:
            ctx.debug_set_chunked_frames(mode)
            ctx.resize(n)
            ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
            ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
            ctx.upload_changed(np.zeros(n, np.uint8))
            fr, _, _ = frame_inputs(0)
            ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)  # every GlobalTransform once; the change column is clean from here on
            for frame in range(4):
                k = int(rng.integers(1, 9000))
                rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
                expect = rows
                t[rows] += rng.normal(0.0, 1.0, (k, 3)).astype(F)
                if case == "descending":  # (a query in spawn order over rows numbered by Entity key: the index is stored inverted)
                    rows = rows[::-1].copy()
                if case == "not monotonic" and k > 2:
                    rows = np.roll(rows, 1)
                    if k == 3:
                        rows = rows[[1, 0, 2]]
                window(ctx, rows, t, sc, n)
                if case == "a mark from elsewhere" and frame % 2 == 1:
                    marks = np.zeros(n, np.uint8)
                    extra = int(np.setdiff1d(np.arange(n, dtype=np.uint32), expect)[5])
                    marks[extra] = 1
                    marks[expect] = 1  # (an uploaded byte column replaces the marks)
                    ctx.upload_changed(marks)
                    expect = np.sort(np.append(expect, np.uint32(extra))).astype(np.uint32)
                if case == "two windows":
                    rows2 = np.sort(rng.choice(n, 50, replace=False)).astype(np.uint32)
                    t[rows2] += F(1.5)
                    window(ctx, rows2, t, sc, n)
                    expect = np.union1d(expect, rows2).astype(np.uint32)
                if case == "a dense write in between" and frame % 2 == 0:
                    lo = int(expect[0])
                    t[lo:lo + 3] += F(0.75)  # (rows at and behind the first uploaded one move again; only marked rows will be propagated)
                    ctx.upload_transforms(t[lo:lo + 3].reshape(-1), sc["rotation"].reshape(n, 4)[lo:lo + 3].reshape(-1), sc["scale"].reshape(n, 3)[lo:lo + 3].reshape(-1), first_row=lo)
                flags = B.CULL_END_FRAME | (0 if case == "an all-rows frame" and frame % 2 == 1 else B.CULL_CHANGED_ROWS)
                fr, _, _ = frame_inputs(frame)
                ctx.propagate_and_cull(fr, flags=flags)
                if case == "an all-rows frame" and frame % 2 == 1:
                    expect = np.arange(n, dtype=np.uint32)
                if case == "a second frame before the results" and frame % 2 == 1:
                    ctx.propagate_and_cull(fr, flags=flags)  # nothing is marked any more: no GlobalTransform changes in this one
                    expect = np.zeros(0, np.uint32)
                if case == "windows mapped before the results":  # the caller already fills the next frame's windows: the memory of the
                    held = []                                     # committed one is handed out again (several maps force the recycling)
                    for _ in range(3):
                        w2, r2, t2, q2, s2 = ctx.map_upload_window(400_000)
                        r2[:] = 0xDEADBEEF
                        t2[:] = -1.0
                        held.append(w2)
                    for w2 in held:
                        ctx.commit_upload_window(w2, 0)
                    w2, r2, t2, q2, s2 = ctx.map_upload_window(400_000)  # (nothing is mapped, two chunks exist: this one starts over at the front)
                    r2[:] = 0xDEADBEEF
                    t2[:] = -1.0
                    ctx.commit_upload_window(w2, 0)
                cap = 10 if case == "capacity too small" and frame == 2 else n
                bufs = api.FrameResultBuffers(cap, n, 0, 0, in_place=in_place)
                if cap < len(expect):
                    with pytest.raises(api.MiError):
                        ctx.download_frame_results(bufs)
                    bufs = api.FrameResultBuffers(n, n, 0, 0, in_place=in_place)
                for rep in range(2 if case == "results twice" else 1):
                    got = ctx.download_frame_results(bufs)
                    r_, g_ = np.array(got["changed_rows"]).copy(), np.array(got["changed_global"]).reshape(-1, 12).copy()
                    assert np.array_equal(r_, expect), f"{case}, frame {frame}, mode {mode}: changed rows"
                    g_all = ctx.download_global_transforms(want_changed=False).reshape(n, 12)
                    assert g_.tobytes() == g_all[expect].tobytes(), f"{case}, frame {frame}, mode {mode}: changed GlobalTransforms against the column"
                    got_all.append((r_, g_, np.array(got["visible_rows"]).copy()))
                if case not in ("a dense write in between",):  # (there the rows the dense write moved keep their old GlobalTransform: not marked)
                    g, _ = O.sync_simple_transforms(t.reshape(-1), sc["rotation"], sc["scale"])
                    assert g_all.tobytes() == g.tobytes(), f"{case}, frame {frame}: the column against the oracle"
            counts.append(ctx.debug_chunked_counts()[2])
        outs.append(got_all)
    for x, y in zip(*outs):
        assert all(np.array_equal(p, q) for p, q in zip(x, y)), f"{case}: the two forms differ"
    want = {"plain": 4, "descending": 4, "not monotonic": 0, "a mark from elsewhere": 2, "two windows": 0, "a dense write in between": 2, "an all-rows frame": 2,
            "a second frame before the results": 2, "results twice": 8, "capacity too small": 4, "default rule": 3, "windows mapped before the results": 4}[case]
    assert counts == [want, 0], f"{case}: {counts} downloads handed out GlobalTransforms written ahead"


@pytest.mark.parametrize("mode", [2, 1])
def test_component_granular_windows(mode):
    """MI_UPLOAD_TRANSLATION / _ROTATION / _SCALE: a window carries only the named components (the columns of the others keep what they
    hold).  Dense sequences in pieces with the GlobalTransforms fetched ahead (every cube rotates: 16 B per row go up), several
    windows one after the other with different components each frame, and indexed windows whose scatter launch writes the
    GlobalTransforms ahead from the window's components + the resident ones -- all against the oracle on the host's mirror."""
    n = 270_001
    sc = W.many_cubes(n, radius=80.0)
    t, r, s = sc["translation"].reshape(n, 3).copy(), sc["rotation"].reshape(n, 4).copy(), sc["scale"].reshape(n, 3).copy()
    rng = np.random.default_rng(21)
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)

    def spin(q, k):
        a = F(0.01 * (k + 1))
        d = np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], F)
        out = W.quat_mul(q.T.astype(np.float64), np.broadcast_to(d[:, None].astype(np.float64), (4, len(q)))).T
        return (out / np.linalg.norm(out, axis=1, keepdims=True)).astype(F)

    with api.Context(0) as ctx:
        ctx.debug_set_chunked_frames(mode)
        ctx.resize(n)
        ctx.upload_transforms(t.reshape(-1), r.reshape(-1), s.reshape(-1))
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        for frame, comps in enumerate(["r", "r", "t", "ts", "rs", "trs", "r"]):
            if "t" in comps:
                t += rng.normal(0.0, 0.3, (n, 3)).astype(F)
            if "r" in comps:
                r = spin(r, frame)
            if "s" in comps:
                s = (s * F(1.01)).astype(F)
            fr = api.compute_frustum(cfv, W.many_cubes_camera(frame * 20), W.CAMERA_FAR)
            cuts = [0, n] if frame % 2 == 0 else [0, n // 3, n // 3 + 70_000, n]
            for lo, hi in zip(cuts, cuts[1:]):
                w, _, wt, wr, ws = ctx.map_upload_window(hi - lo, dense=True, components=comps)
                assert (wt is None) == ("t" not in comps) and (wr is None) == ("r" not in comps) and (ws is None) == ("s" not in comps)
                if wt is not None:
                    wt[:] = t[lo:hi].reshape(-1)
                if wr is not None:
                    wr[:] = r[lo:hi].reshape(-1)
                if ws is not None:
                    ws[:] = s[lo:hi].reshape(-1)
                ctx.commit_upload_window(w, hi - lo, first_row=lo)
            ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
            got = ctx.download_frame_results(api.FrameResultBuffers(n, n, 0, 0, in_place=bool(frame % 2)))
            g, _ = O.sync_simple_transforms(t.reshape(-1), r.reshape(-1), s.reshape(-1))
            assert np.array_equal(got["changed_rows"], np.arange(n, dtype=np.uint32)), f"frame {frame} ({comps})"
            assert np.array(got["changed_global"]).tobytes() == g.tobytes(), f"frame {frame} ({comps}): GlobalTransforms against the oracle"
            assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes(), f"frame {frame} ({comps}): the column"
        if mode == 2:
            assert ctx.debug_chunked_counts()[1] >= 5  # handed out from what was fetched ahead
        # indexed windows: some rows move / turn, the window carries that component only
        ctx.upload_changed(np.zeros(n, np.uint8))
        ctx.propagate(0)
        for frame, comps in enumerate(["t", "r", "t", "rs", "t"]):
            k = 20_000 + 17 * frame
            rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
            if frame == 3:
                rows = rows[::-1].copy()  # a descending window (spawn order over inverted indices)
            if "t" in comps:
                t[rows] += rng.normal(0.0, 2.0, (k, 3)).astype(F)
            if "r" in comps:
                r[rows] = spin(r[rows], frame)
            if "s" in comps:
                s[rows] = (s[rows] * F(0.97)).astype(F)
            w, wrows, wt, wr, ws = ctx.map_upload_window(k, components=comps)
            wrows[:] = rows
            if wt is not None:
                wt[:] = t[rows].reshape(-1)
            if wr is not None:
                wr[:] = r[rows].reshape(-1)
            if ws is not None:
                ws[:] = s[rows].reshape(-1)
            ctx.commit_upload_window(w, k)
            fr = api.compute_frustum(cfv, W.many_cubes_camera(frame * 20), W.CAMERA_FAR)
            ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_CHANGED_ROWS)
            got = ctx.download_frame_results(api.FrameResultBuffers(n, n, 0, 0, in_place=bool(frame % 2)))
            g, _ = O.sync_simple_transforms(t.reshape(-1), r.reshape(-1), s.reshape(-1))
            srt = np.sort(rows)
            assert np.array_equal(got["changed_rows"], srt), f"indexed frame {frame} ({comps})"
            assert np.array(got["changed_global"]).tobytes() == g.reshape(n, 12)[srt].tobytes(), f"indexed frame {frame} ({comps}): changed GlobalTransforms"
            assert ctx.download_global_transforms(want_changed=False).tobytes() == g.tobytes(), f"indexed frame {frame} ({comps}): the column"
        if mode == 2:
            assert ctx.debug_chunked_counts()[2] >= 3  # written ahead by the scatter launch
