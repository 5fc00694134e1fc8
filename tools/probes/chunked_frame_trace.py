"""One context, a few all-dirty frames through ONE dense window (the library splits it in eight): for `rocprofv3 --kernel-trace
--memory-copy-trace` -- tools/probes/summarize_copy_trace.py turns the trace of the last frame into a timeline."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import bevy_amd as B
from bevy_amd import api, workloads as W
n_c, n_l = 1_000_000, 100_000
sc, first_light, pr = W.frame_scene(n_c, n_l, 10_000)
n = sc["n"]
cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
fr = api.compute_frustum(cfv, W.many_cubes_camera(0), W.CAMERA_FAR)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with api.Context(0) as ctx:
    ctx.debug_set_chunked_frames(mode)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    bufs = api.FrameResultBuffers(n, n, 0, 0, in_place=True)
    for f in range(5):
        w, _, wt, wr, ws = ctx.map_upload_window(n, dense=True)
        wt[:] = sc["translation"]; wr[:] = sc["rotation"]; ws[:] = sc["scale"]
        ctx.synchronize()
        ctx.commit_upload_window(w, n)
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
        got = ctx.download_frame_results(bufs)
        assert len(got["changed_rows"]) == n
