"""Development aid (GPU box): the tile plan of a hierarchy stress shape launch by launch, and the per-tile phase timeline of an
all-dirty frame (stamps of k_propagate_fans: 0 start, 1 loads issued, 2 loads consumed + barrier, 3 chain done, 4 levels done,
5 upper rows written back, 6 last level computed, 7 stores drained; 10 ns ticks).
    python tools/shape_trace.py <shape> [tile_mode]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bevy_amd as B
from bevy_amd import api, workloads as W

name = sys.argv[1]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sh = W.hierarchy_shape(name)
n = sh["n"]
ctx = api.Context(0)
ctx.resize(n)
ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
ctx.debug_set_tile_mode(mode)
ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
print(name, n, "nodes; level widths", np.diff(sh["level_offsets"].astype(np.int64)).tolist())
print("plan", ctx.debug_tile_plan())
groups, tiles = ctx.debug_tile_groups()
roots = np.nonzero(sh["parent"] == W.NO_PARENT)[0].astype(np.uint32)
rows_of = lambda idx, col, w: np.ascontiguousarray(sh[col].reshape(n, w)[idx]).reshape(-1)
rt, rr, rs = rows_of(roots, "translation", 3), rows_of(roots, "rotation", 4), rows_of(roots, "scale", 3)
sets = [rt, (rt.reshape(-1, 3) + np.float32(1.0)).reshape(-1).copy()]


def frame(f):
    ctx.upload_transforms_indexed(roots, sets[f & 1], rr, rs)
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)


for f in range(10):
    frame(f)
ctx.synchronize()
ctx.debug_tree_trace(0)
for f in range(3):
    frame(f)
ctx.synchronize()
t = ctx.debug_tree_trace(len(tiles)).astype(np.int64)
t0 = t[t[:, 0] > 0, 0].min() if (t[:, 0] > 0).any() else 0
names = ["desc", "loads", "chain", "levels", "flush", "last", "drain"]
for gi, (first, count, n_chain, deep) in enumerate(groups.tolist()):
    tt, td = t[first:first + count], tiles[first:first + count]
    ok = tt[:, 0] > 0
    rel = (tt[ok] - t0) * 0.01
    kinds = td[:, 2]
    print(f"launch {gi}: {count} tiles ({n_chain} chain tiles, {'16' if deep else '8'}-level instantiation); levels per tile p50 {np.median(td[:, 0]):.0f} max {td[:, 0].max()}, "
          f"rows per tile mean {td[:, 1].mean():.0f} p10 {np.percentile(td[:, 1], 10):.0f} max {td[:, 1].max()}; chain length mean {(kinds & 0xFF).mean():.1f} max {(kinds & 0xFF).max()}; "
          f"roots tiles {(kinds & 0x100).astype(bool).sum()}; rows in launch {td[:, 1].sum()}")
    if ok.any():
        life = rel[:, 7] - rel[:, 0]
        print(f"   span: first start {rel[:, 0].min():.2f} us, last start {rel[:, 0].max():.2f}, last drain {rel[:, 7].max():.2f}; tile life p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f}")
        d = np.diff(rel, axis=1)
        print("   phases mean: " + ", ".join(f"{nm} {d[:, i].mean():.2f}" for i, nm in enumerate(names)))
