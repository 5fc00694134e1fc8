"""Development aid (GPU box): the strips of a hierarchy stress shape (k_propagate_strips) -- rounds per strip, and the per-strip phase
timeline of an all-dirty frame (stamps: 0 start, 1 table in LDS and flags tested, 4 rounds done, 7 stores drained; with a
-DMI_EXP_STRIP_STAMPS build (MI_LIB_VARIANT=sx_stamps...) also how long consumer wave 0 and the producer work and wait per round).
    python tools/strip_trace.py <shape> [tile_mode, default 5]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bevy_amd as B
from bevy_amd import api, workloads as W

name = sys.argv[1]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sh = W.hierarchy_shape(name)
n = sh["n"]
ctx = api.Context(0)
ctx.resize(n)
ctx.upload_transforms(sh["translation"], sh["rotation"], sh["scale"])
ctx.debug_set_tile_mode(mode)
ctx.upload_hierarchy(sh["parent"], sh["level_offsets"])
print(name, n, "nodes; level widths", np.diff(sh["level_offsets"].astype(np.int64)).tolist())
print("plan", ctx.debug_tile_plan(), "MI_STRIP_W", os.environ.get("MI_STRIP_W"))
rounds, cone, total = ctx.debug_strip_plan()
ns = len(rounds)
if not ns:
    print("the plan has no strips")
    sys.exit(0)
print(f"{ns} strips, {total} rounds in the table ({n / max(1, 64 * total):.3f} of the lanes carry a row); rounds per strip p10 {np.percentile(rounds, 10):.0f} p50 {np.median(rounds):.0f} "
      f"max {rounds.max()}; cone rounds per strip mean {cone.mean():.1f} max {cone.max()}")
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
t = ctx.debug_tree_trace(ns).astype(np.int64)
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
rel = (t[ok] - t0) * 0.01
life = rel[:, 7] - rel[:, 0]
print(f"span: first start {rel[:, 0].min():.2f} us, last start {rel[:, 0].max():.2f}, p50 start {np.median(rel[:, 0]):.2f}; last drain {rel[:, 7].max():.2f}; "
      f"strip life p10 {np.percentile(life, 10):.2f} p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f}")
r_ok = rounds[ok].astype(np.float64)
print(f"table copied and flags tested: mean {np.mean(rel[:, 1] - rel[:, 0]):.2f} us; us per round (life / rounds): p10 {np.percentile(life / r_ok, 10):.3f} p50 {np.median(life / r_ok):.3f} p90 {np.percentile(life / r_ok, 90):.3f}")
if os.environ.get("MI_LIB_VARIANT", "").startswith("sx_stamps"):  # (-DMI_EXP_STRIP_STAMPS: slots 2, 3 = consumer wave 0's wait / work, 5, 6 = the producer's; 10 ns ticks)
    tt = t[ok].astype(np.float64) * 0.01
    for nm, col in (("consumer wait at the barrier", 2), ("consumer work", 3), ("producer wait at the barrier", 5), ("producer work", 6)):
        print(f"{nm}: per strip p50 {np.median(tt[:, col]):.2f} us, per round p50 {np.median(tt[:, col] / r_ok):.3f}")
if os.environ.get("STRIP_TRACE_SLOWEST"):
    idx = np.nonzero(ok)[0]
    order = np.argsort(-life)[:12]
    print("slowest strips: index, start us, life us, table entries, cone entries")
    for o in order:
        print(f"  {idx[o]:5d} {rel[o, 0]:6.2f} {life[o]:6.2f} {rounds[idx[o]]:3d} {cone[idx[o]]:3d}")
    for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 100)):
        sel = (life >= np.percentile(life, lo)) & (life <= np.percentile(life, hi))
        print(f"  life quartile {lo}-{hi}: life {life[sel].mean():.2f}, entries {rounds[idx[sel]].mean():.1f}, cone {cone[idx[sel]].mean():.1f}")
    print(f"  corr(life, entries) {np.corrcoef(life, rounds[idx])[0, 1]:.2f}, corr(life, cone) {np.corrcoef(life, cone[idx])[0, 1]:.2f}, corr(life, strip index) {np.corrcoef(life, idx)[0, 1]:.2f}")
