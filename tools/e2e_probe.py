import sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bevy_amd as B
from bevy_amd import api, workloads as W
import bench
class A: pass
args = A(); args.entities = 0; args.lights = 100000; args.meshes = 10000; args.separate_cluster_calls = False; args.concurrent_clusters = False; args.inline_compaction = False
ctx = api.Context(0)
wl = bench.build_frame(ctx, args)
sc = wl.scene; n = sc["n"]; views = wl.keep[0]
t3 = sc["translation"].reshape(n, 3); r4 = sc["rotation"].reshape(n, 4); s3 = sc["scale"].reshape(n, 3)
rng = np.random.default_rng(0)
ctx.upload_changed(np.zeros(n, np.uint8)); ctx.propagate(B.PROPAGATE_ALL_DIRTY); ctx.synchronize()
k = n // 100
rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32)
tt, rr, ss = np.ascontiguousarray(t3[rows]).reshape(-1), np.ascontiguousarray(r4[rows]).reshape(-1), np.ascontiguousarray(s3[rows]).reshape(-1)
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
for f in range(14):
    fr = api.PreparedFrusta(bench.camera_frusta(1, f))
    ctx.synchronize()
    T("upload_indexed", lambda: ctx.upload_transforms_indexed(rows, tt, rr, ss))
    T("propagate", lambda: ctx.propagate(0))
    T("cull", lambda: ctx.cull(fr, flags=B.CULL_BEGIN_FRAME | B.CULL_END_FRAME))
    T("cluster_view", lambda: ctx.cluster_upload_view(views[f % 256]))
    T("cluster_assign", lambda: ctx.cluster_assign_resident())
    T("sync", lambda: ctx.synchronize())
    T("dl_changed", lambda: ctx.download_changed_global_transforms())
    T("dl_visible", lambda: ctx.download_visible_entities(0, 0))
    T("dl_cluster", lambda: ctx.cluster_download(views[0].n_clusters))
for kk, v in acc.items(): print("%-16s %.1f us" % (kk, 1e6 * np.median(v[2:])))
