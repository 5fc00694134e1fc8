"""What the riders of a frame launch wrote, read behind the API's back.

With MI_CULL_MORE_FRAMES the VisibleEntities lists of frame f are built by extra workgroups of frame f + 1's kernel.  Every entry
point that exposes lists joins the LATEST frame's first, so nothing in the API ever shows lists that riders wrote -- parity tests
can only see that nothing else broke.  These tests capture the list buffers of the three rotating output sets through
mi_device_buffer(MI_BUF_VISIBLE_ROWS) during three frames, run two more frames with nothing read in between, and fetch with
hipMemcpy the buffer the first of them wrote: the work of the riders in the second one's launch.  (The static-cull-order form of the
same check is tests/test_gpu_cells.py::test_lists_that_rode_in_the_next_frames_launch_are_that_frames_lists.)"""
import ctypes as C

import numpy as np
import pytest

import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

pytestmark = pytest.mark.gpu
MI_BUF_VISIBLE_ROWS = 3


def frusta_for(cams):
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    return np.concatenate([api.compute_frustum(cfv, cam, W.CAMERA_FAR) for cam in cams])


@pytest.mark.parametrize("all_dirty", [True, False])
def test_the_compaction_that_rode_in_the_next_frames_kernel(all_dirty):
    """k_frame (every Transform changed: mi_propagate_and_cull) and k_frame<0> / k_frame_sph (nothing changed: mi_cull) carrying the
    previous frame's k_compact_fast work in their first workgroups."""
    n, n_views = 300_007, 2
    sc = W.many_cubes(n, radius=140.0, ragged_flags=True)
    hip = C.CDLL("libamdhip64.so")
    with api.Context(0) as ctx:
        ctx.debug_set_static_cull_order(1)  # (its own riders: the test named above)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.zeros(n, np.uint8))
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        vv = np.zeros(n, np.uint8)
        ptrs, expected = [], {}
        for frame in range(8):
            frusta = frusta_for([W.many_cubes_camera(frame * 25, yaw=v * 1.9) for v in range(n_views)])
            if all_dirty:
                ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES)
            else:
                ctx.propagate(0)
                ctx.cull(frusta, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME | B.CULL_MORE_FRAMES)
            g, vv, vis, _ = O.full_frame(sc["translation"], sc["rotation"], sc["scale"], sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"],
                                         vv, frusta)
            expected[frame] = [np.nonzero(vis[v])[0].astype(np.uint32) for v in range(n_views)]
            if frame < 3:
                ctx.synchronize()
            elif frame < 6:
                ptrs.append(ctx.device_buffer(MI_BUF_VISIBLE_ROWS))  # (joins: these three frames' lists are launches of their own)
        ctx.synchronize()  # frame 7's compaction on its own; frame 6's rode in frame 7's kernel
        assert len({p for p, _ in ptrs}) == 3, ptrs
        p, nbytes = ptrs[0]  # the sets rotate by three: frame 6 wrote the set of frame 3
        stride = nbytes // (4 * n_views)
        host = np.zeros(nbytes // 4, np.uint32)
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(p), C.c_size_t(nbytes), 2) == 0
        for v in range(n_views):
            want = expected[6][v]
            assert want.size > 100 and not np.array_equal(want, expected[3][v][:want.size])
            assert np.array_equal(host[v * stride:v * stride + want.size], want), f"frame 6, view {v}: the riders' list"
        for v in range(n_views):
            assert np.array_equal(ctx.download_visible_entities(v, 0)[1], expected[7][v])


def test_the_cluster_fill_that_rode_in_the_next_frames_kernel():
    """MI_CULL_WITH_CLUSTERS | MI_CULL_MORE_FRAMES (the metric frame): frame f's walk rides in frame f's launch, its FILL in frame
    f + 1's.  mi_cluster_download would launch frame f + 1's pending fill first and show that; mi_debug_cluster_download_unjoined
    shows the lists as the last fill that ran left them -- frame f's, written by riders -- which must be the reference's sequence for
    frame f's camera."""
    from test_gpu_cluster import reference_sequence, upload_scene
    sc, first_light, pr = W.frame_scene(60_000, 30_000, 3_000, light_range=1.5)
    n_l = len(pr) // 4
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    with api.Context(0) as ctx:
        upload_scene(ctx, sc)
        ctx.cluster_upload_objects(pr)
        ctx.cluster_bind_objects_to_rows(first_light, n_l)
        want = {}
        for frame in range(4):
            cam = W.many_cubes_camera(frame * 30, yaw=0.35 * frame)
            frusta = frusta_for([cam])
            view, keep = api.cluster_view_build(cam, cfv, frusta, 1920, 1080, (16, 9, 24), 5.0, 1000.0)
            ctx.upload_view_visibility(np.zeros(sc["n"], np.uint8))
            ctx.cluster_upload_view(view)
            ctx.propagate_and_cull(frusta, flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES | B.CULL_WITH_CLUSTERS)
            want[frame] = reference_sequence(sc, first_light, pr, frusta, cam)
        # frame 3's launch carried frame 2's fill; frame 3's own fill is still pending
        off, idx = ctx.debug_cluster_download_unjoined(view.n_clusters, 1 << 22)
        eoff, eidx, _, _, etotal, *_ = want[2]
        assert etotal > 100 and etotal != want[3][4]
        assert np.array_equal(off, eoff) and np.array_equal(idx, eidx), "frame 2's lists, written by the riders of frame 3's launch"
        got = ctx.cluster_download(view.n_clusters)  # (launches frame 3's fill)
        assert np.array_equal(got[0], want[3][0]) and np.array_equal(got[1], want[3][1])
