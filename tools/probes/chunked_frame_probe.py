import time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import bevy_amd as B
from bevy_amd import api, workloads as W
n_c, n_l = 1_000_000, 100_000
sc, first_light, pr = W.frame_scene(n_c, n_l, 10_000)
n = sc["n"]
cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
cam = W.many_cubes_camera(0)
fr = api.compute_frustum(cfv, cam, W.CAMERA_FAR)
for mode in (1, 0, 1, 0):
    ctx = api.Context(0)
    ctx.debug_set_chunked_frames(mode)
    ctx.resize(n)
    ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
    ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
    bufs = api.FrameResultBuffers(n, n, 0, 0, in_place=True)
    ts = []
    for f in range(8):
        w, _, wt, wr, ws = ctx.map_upload_window(n, dense=True)
        wt[:] = sc["translation"]; wr[:] = sc["rotation"]; ws[:] = sc["scale"]
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.commit_upload_window(w, n)
        t1 = time.perf_counter()
        ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME)
        t2 = time.perf_counter()
        got = ctx.download_frame_results(bufs)
        t3 = time.perf_counter()
        assert len(got["changed_rows"]) == n
        ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
    a = np.median(np.array(ts[2:]), axis=0)
    print("chunked" if mode == 0 else "one piece", "commit %.0f  frame %.0f  download %.0f  total %.0f us" % tuple(a))
    ctx.close()
