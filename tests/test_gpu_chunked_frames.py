"""GPU parity of all-dirty end-to-end frames in pieces (context.cpp: mi_commit_upload_window with a dense window that carries
every row, cull_frame running behind the pieces of the upload, mi_download_frame_results starting the GlobalTransforms back piece
by piece): two contexts through the same frames, one with the pieces (default), one without (mi_debug_set_chunked_frames(1)),
and the oracle for the GlobalTransforms -- changed rows, their GlobalTransforms, VisibleEntities, cluster lists, ViewVisibility,
in place and copied out, with the cluster walk inside the rows' workgroups (row-range binding) and in workgroups of its own (row
list), and with other calls between the upload and the frame (which must wait for the whole upload)."""
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


@pytest.mark.parametrize("binding", ["range", "list"])
@pytest.mark.parametrize("in_place", [True, False])
def test_all_dirty_frames_in_pieces(binding, in_place):
    n_cubes, n_lights = 300_000, 3_000
    sc, first_light, pr = W.frame_scene(n_cubes, n_lights, 500, light_range=3.0)
    n = sc["n"]
    c_dev, h_dev = sc["aabb_center"].reshape(n, 3).copy(), sc["aabb_half"].reshape(n, 3).copy()
    c_dev[first_light:] = 0.0
    h_dev[first_light:, 1] = np.frombuffer(np.uint32(0x7FC0A11D).tobytes(), F)[0]
    t = sc["translation"].reshape(n, 3).copy()
    rng = np.random.default_rng(4)
    ctxs = [api.Context(0), api.Context(0)]
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
        # the pieces were taken where they should have been: every frame of the first context (one more upload in frame 3 was joined
        # by the mi_propagate behind it), none of the second
        assert ctxs[0].debug_chunked_counts() == (5, 5) and ctxs[1].debug_chunked_counts() == (0, 0)
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
