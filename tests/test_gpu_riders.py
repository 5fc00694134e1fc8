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
