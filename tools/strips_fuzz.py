"""GPU fuzz of the strips (development aid): random forests of random shape at random strip widths, a few frames each (all dirty, movers
under the static-scene rule with the flags-first exit forced or not, a quiet frame, movers without the rule), GlobalTransform bits and
change ticks against the oracle.   python tools/strips_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevy_amd as B
from bevy_amd import api, workloads as W
import oracle_lib as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F = np.float32
O.build()


def forest():
    parent = []
    n_trees = int(rng.choice([1, 1, 2, 5, 40, 400]))
    depth = int(rng.integers(1, 40))
    maxc = int(rng.choice([1, 2, 2, 3, 5, 9]))
    p_leaf = float(rng.choice([0.0, 0.2, 0.5]))
    fan_every = int(rng.choice([0, 0, 25, 80]))
    fan = int(rng.integers(17, 500))
    cap = int(rng.choice([300, 3000, 30000]))
    for _ in range(n_trees):
        parent.append(W.NO_PARENT)
        level = [len(parent) - 1]
        for _d in range(depth):
            nxt = []
            for p in level:
                if rng.random() < p_leaf and len(level) > 1:
                    continue
                k = int(rng.integers(1, maxc + 1))
                if fan_every and rng.integers(0, fan_every) == 0:
                    k = fan
                for _c in range(k):
                    nxt.append(len(parent))
                    parent.append(p)
            if not nxt or len(parent) > cap:
                break
            level = nxt
    return np.array(parent, np.int64)


bad = 0
planned = 0
for case in range(cases):
    parent = forest()
    _, p_new, offs = W.level_order(parent)
    n = len(parent)
    width = int(rng.choice([1, 2, 5, 16, 17, 33, 64, 65, 100, 128]))
    os.environ["MI_STRIP_W"] = str(width)
    t = (rng.random((n, 3)) * 4 - 2).astype(F)
    q = rng.normal(size=(n, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    s3 = (0.9 + 0.2 * rng.random((n, 3))).astype(F)
    pretest = int(rng.choice([0, 1, 2]))
    with api.Context(0) as ctx:
        ctx.debug_set_tile_mode(5)
        ctx.debug_set_tile_pretest(pretest)
        ctx.resize(n)
        ctx.upload_transforms(t.reshape(-1), q.reshape(-1), s3.reshape(-1))
        ctx.upload_hierarchy(p_new, offs)
        strips = len(ctx.debug_strip_plan()[0])
        planned += strips > 0
        ctx.propagate(B.PROPAGATE_ALL_DIRTY)
        g, chg = ctx.download_global_transforms()
        rc, g_exp, chg_exp = O.propagate_transforms(p_new, t.reshape(-1), q.reshape(-1), s3.reshape(-1))
        ok = (g.view(np.uint32) == g_exp.view(np.uint32)).all() and (np.asarray(chg) == np.asarray(chg_exp)).all()
        tt = t.copy()
        for frac, static_opt in ((0.3, True), (0.01, True), (0.0, True), (0.2, False), (1.0, True)):
            moved = np.nonzero(rng.random(n) < frac)[0].astype(np.uint32)
            changed = np.zeros(n, np.uint8)
            changed[moved] = 1
            if len(moved):
                tt[moved] += F(0.25)
                ctx.upload_transforms_indexed(moved, np.ascontiguousarray(tt[moved]).reshape(-1), np.ascontiguousarray(q[moved]).reshape(-1), np.ascontiguousarray(s3[moved]).reshape(-1))
            else:
                ctx.upload_changed(changed)
            ctx.propagate(B.PROPAGATE_STATIC_OPT if static_opt else 0)
            g, chg = ctx.download_global_transforms()
            rc, g_exp, chg_exp = O.propagate_transforms(p_new, tt.reshape(-1), q.reshape(-1), s3.reshape(-1), global_in=g_exp, static_opt=static_opt,
                                                        tree_changed=O.mark_dirty_trees(p_new, changed), transform_changed=changed)
            ok = ok and (g.view(np.uint32) == g_exp.view(np.uint32)).all() and (np.asarray(chg) == np.asarray(chg_exp)).all()
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {n} nodes, {len(offs) - 1} levels, width {width}, pretest {pretest}, strips {strips}", flush=True)
print(f"{cases} cases, {planned} planned as strips, {bad} mismatches")
sys.exit(1 if bad else 0)
