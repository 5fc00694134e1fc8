"""Development aid (GPU box): mid-size hierarchies through the workgroup tiles (tile mode 4) and through strips (mode 5), us per all-dirty
frame back to back -- what the planner's default rule (ctx_hierarchy.cpp) is calibrated on.   python tools/probes/mid_shapes_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bevy_amd as B
from bevy_amd import api, workloads as W
F = np.float32
rng = np.random.default_rng(3)
def tree(depth, branch):
    par = W._parent_map_tree(depth, branch)
    return np.concatenate([[W.NO_PARENT], np.asarray(par, np.int64)])
def lopsided(depth, maxc, p_leaf, cap):
    parent = [W.NO_PARENT]; level = [0]
    for _ in range(depth):
        nxt = []
        for p in level:
            if rng.random() < p_leaf and len(level) > 1: continue
            for _c in range(int(rng.integers(1, maxc + 1))):
                nxt.append(len(parent)); parent.append(p)
        if not nxt or len(parent) > cap: break
        level = nxt
    return np.array(parent, np.int64)
shapes = {"binary depth 14": tree(14, 2), "3-ary depth 9": tree(9, 3), "4-ary depth 8": tree(8, 4), "4-ary depth 9": tree(9, 4), "lopsided 12 levels": lopsided(12, 4, 0.3, 150000),
          "lopsided 9 levels": lopsided(9, 6, 0.3, 200000), "binary depth 16": tree(16, 2), "binary depth 17": tree(17, 2), "binary depth 18": tree(18, 2)}
for name, parent in shapes.items():
    _, p_new, offs = W.level_order(parent)
    n = len(parent)
    t = (rng.random((n, 3)) * 4 - 2).astype(F); q = rng.normal(size=(n, 4)); q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F); s3 = np.ones((n, 3), F)
    res = []
    for mode in (4, 5, 0):
        with api.Context(0) as ctx:
            ctx.debug_set_tile_mode(mode)
            ctx.resize(n); ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s3.reshape(-1)); ctx.upload_hierarchy(p_new, offs)
            plan = ctx.debug_tile_plan()
            for _ in range(20): ctx.propagate(B.PROPAGATE_ALL_DIRTY)
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(300): ctx.propagate(B.PROPAGATE_ALL_DIRTY)
            ctx.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e6
            res.append((plan["launches"], plan["tiles"], round(dt, 1)))
    print(f"{name}: {n} nodes, {len(offs)-1} levels; tiles (launches, tiles, us/frame) {res[0]}  strips {res[1]}  default {res[2]}")
