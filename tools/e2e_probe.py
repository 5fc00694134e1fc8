"""Where an end-to-end frame's time goes (run on the GPU box): the metric scene with the host on both sides, per call, and the two
ways of running a partially dirty frame side by side -- mi_propagate(0) + mi_cull against mi_propagate_and_cull(MI_CULL_CHANGED_ROWS)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevy_amd as B
from bevy_amd import api
import bench
class A: pass
args = A(); args.entities = 0; args.lights = 100000; args.meshes = 10000; args.separate_cluster_calls = False; args.concurrent_clusters = False; args.inline_compaction = False
ctx = api.Context(0)
wl = bench.build_frame(ctx, args)
sc = wl.scene; n = sc["n"]; views = wl.keep[0]
t3 = sc["translation"].reshape(n, 3); r4 = sc["rotation"].reshape(n, 4); s3 = sc["scale"].reshape(n, 3)
rng = np.random.default_rng(0)
ctx.upload_changed(np.zeros(n, np.uint8)); ctx.propagate(B.PROPAGATE_ALL_DIRTY); ctx.synchronize()
bufs = api.FrameResultBuffers(n, n, views[0].n_clusters, 1 << 20)
CL = B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS
for pct in (1, 10):
    k = n * pct // 100
    rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
    tt, rr, ss = np.ascontiguousarray(t3[rows]).reshape(-1), np.ascontiguousarray(r4[rows]).reshape(-1), np.ascontiguousarray(s3[rows]).reshape(-1)
    for mode in ("two_calls", "fused", "two_calls", "fused"):
        acc = {}
        def T(name, f):
            t0 = time.perf_counter(); r = f(); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
        for f in range(40):
            fr = api.PreparedFrusta(bench.camera_frusta(1, f))
            ctx.synchronize()
            t_all = time.perf_counter()
            T("cluster_view", lambda: ctx.cluster_upload_view(views[f % 256]))
            T("upload_indexed", lambda: ctx.upload_transforms_indexed(rows, tt, rr, ss))
            if mode == "fused":
                T("frame", lambda: ctx.propagate_and_cull(fr, flags=CL | B.CULL_CHANGED_ROWS))
            else:
                T("propagate", lambda: ctx.propagate(0))
                T("cull", lambda: ctx.cull(fr, flags=B.CULL_BEGIN_FRAME | CL))
            t_sub = time.perf_counter()
            T("sync (kernels)", lambda: ctx.synchronize())
            acc.setdefault("submit -> kernels done", []).append(time.perf_counter() - t_sub + 0.0)
            T("frame_results", lambda: ctx.download_frame_results(bufs))
            acc.setdefault("TOTAL", []).append(time.perf_counter() - t_all)
        print(f"--- {pct} % dirty, {mode}")
        for kk, v in acc.items():
            print("  %-24s %.1f us" % (kk, 1e6 * np.median(v[4:])))
