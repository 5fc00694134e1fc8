"""Development aid (GPU box): per-tile phase timeline of the light tile kernel on the bench's 1 M-node tree.

    python tools/tree_trace.py [tile_mode] [cull]     (cull: the fused hierarchy frame, k_propagate_fans<true, true>)

Stamps (s_memrealtime, 10 ns): 0 start, 1 loads issued (descriptor landed), 2 loads consumed + barrier, 3 chain done,
4 levels done, 5 upper rows written back, 6 last level computed + stores issued, 7 stores drained."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bevy_amd as B
from bevy_amd import api, workloads as W

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cull = len(sys.argv) > 2 and sys.argv[2] == "cull"
tr = W.gen_tree(12, 4, 1_000_000)
ctx = api.Context(0)
ctx.resize(tr["n"])
ctx.upload_transforms(tr["translation"], tr["rotation"], tr["scale"])
ctx.debug_set_tile_mode(mode)
ctx.upload_hierarchy(tr["parent"], tr["level_offsets"])
plan = ctx.debug_tile_plan()
print("plan", plan)
root_t = [tr["translation"][:3].copy(), tr["translation"][:3] + np.float32(1.0)]
if cull:
    from benchlib.common import camera_frusta
    n = tr["n"]
    ctx.debug_set_tree_cull(2)
    ctx.upload_bounds(np.zeros(3 * n, np.float32), np.full(3 * n, 0.5, np.float32), np.full(n, 0x05, np.uint8), np.ones(n, np.uint32))
    frusta = [api.PreparedFrusta(camera_frusta(1, f)) for f in range(4)]


def frame(f):
    ctx.upload_transforms(root_t[f & 1], tr["rotation"][:4], tr["scale"][:3], first_row=0)
    if cull:
        ctx.propagate_and_cull(frusta[f % 4], flags=B.CULL_END_FRAME | B.CULL_MORE_FRAMES)
    else:
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)


for f in range(20):
    frame(f)
ctx.synchronize()
ctx.debug_tree_trace(0)
for f in range(3):
    frame(f)
ctx.synchronize()
t = ctx.debug_tree_trace(plan["tiles"]).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) * 0.01  # us
d = np.diff(rel, axis=1)
names = ["desc", "loads", "chain", "levels", "flush", "last", "drain"]
print("kernel span (first start -> last drain): %.2f us" % rel[:, 7].max())
print("tile start   p0 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(rel[:, 0], [0, 50, 90, 100])))
print("tile end     p0 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(rel[:, 7], [0, 50, 90, 100])))
life = rel[:, 7] - rel[:, 0]
print("tile life    p10 %.2f p50 %.2f p90 %.2f max %.2f mean %.2f" % (*np.percentile(life, [10, 50, 90, 100]), life.mean()))
for i, nme in enumerate(names):
    print("  %-7s mean %.2f  p50 %.2f  p90 %.2f" % (nme, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
# concurrency over time
ev = np.concatenate([np.stack([rel[:, 0], np.ones(len(rel))], 1), np.stack([rel[:, 7], -np.ones(len(rel))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
for tt in range(0, int(rel[:, 7].max()) + 1, 2):
    i = np.searchsorted(ev[:, 0], tt)
    print("  t=%2d us resident tiles %d" % (tt, conc[min(i, len(conc) - 1)]))
