"""How fast does the scatter launch of an indexed upload window move bytes over PCIe -- reading the window (44 B per row) and, with the
GlobalTransforms written ahead, writing 48 B per row back?  commit + synchronize, median of 10, for k rows of a 1.11 M-row table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bevy_amd import api, workloads as W
n = 1_110_000
sc = W.many_cubes(n, radius=80.0)
t3, r4, s3 = sc["translation"].reshape(n, 3), sc["rotation"].reshape(n, 4), sc["scale"].reshape(n, 3)
for mode in (1, 2):
    with api.Context(0) as ctx:
        ctx.debug_set_chunked_frames(mode)
        ctx.resize(n)
        ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
        ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
        ctx.upload_changed(np.zeros(n, np.uint8))
        ctx.propagate(0)
        for k in (11_100, 111_000, 555_000, 1_110_000):
            rows = np.arange(0, n, n // k, dtype=np.uint32)[:k]
            ts = []
            for rep in range(12):
                w, wrows, wt, wr, ws = ctx.map_upload_window(k)
                wrows[:] = rows
                wt[:] = t3[rows].reshape(-1); wr[:] = r4[rows].reshape(-1); ws[:] = s3[rows].reshape(-1)
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.commit_upload_window(w, k)
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
                ctx.propagate(0)  # consumes the marks: the change column is clean for the next window
                ctx.synchronize()
            us = 1e6 * float(np.median(ts[2:]))
            mb_in, mb_out = k * 44 / 1e6, (k * 48 / 1e6 if mode == 2 else 0.0)
            print(f"mode {mode} k={k}: {us:8.1f} us  in {mb_in:5.1f} MB ({mb_in * 1e3 / us:5.1f} GB/s)  out {mb_out:5.1f} MB ({mb_out * 1e3 / us:5.1f} GB/s)")
