"""Experiment (GPU box, variant build -DMI_EXP_TIMELINE): when do the kinds of workgroups of the metric frame's launch start and end?
    MI_LIB_VARIANT=timeline python tools/exp_timeline.py [--row-summary 0|1]
Per frame: earliest start / latest end of the compaction, fill, walk and row workgroups relative to the launch's first start, and
their mean lifetime; median over the frames."""
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bevy_amd import api

args = bench.parse()
lib = C.CDLL(api.lib_path())
lib.mi_exp_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.mi_exp_walk_marks.argtypes = [C.POINTER(C.c_ulonglong)]
stream = torch.cuda.Stream()
ctx = api.Context(0, stream.cuda_stream)
with torch.cuda.stream(stream):
    wl = bench.build_frame(ctx, args)
    for f in range(10):
        wl.step(f)
    ctx.synchronize()
    rows, hist = [], []
    N = 16384
    buf = (C.c_ulonglong * (3 * N))()
    for f in range(10, 60):
        lib.mi_exp_timeline(None, 1)
        wl.step(f)
        ctx.synchronize()
        lib.mi_exp_timeline(buf, 0)
        v = np.frombuffer(buf, dtype=np.uint64).reshape(N, 3).astype(np.float64)
        v = v[v[:, 2] > 0]
        t0 = v[:, 0].min()
        fr = []
        for k in range(4):
            w = v[v[:, 2] == k + 1]
            fr.append([0, 0, 0, 0, 0] if len(w) == 0 else [(w[:, 0].min() - t0) / 100.0, (w[:, 0].max() - t0) / 100.0, (w[:, 1].max() - t0) / 100.0, (w[:, 1] - w[:, 0]).mean() / 100.0, len(w)])
        rows.append(fr)
        if f == 59:  # phases of the walking workgroups: 0 entry, 1 visibility known, 2 walked (first chunk), 3 slots reserved, 4 pairs written, 7 exit
            mb = (C.c_ulonglong * (16 * 4096))()
            lib.mi_exp_walk_marks(mb)
            m = np.frombuffer(mb, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
            m = m[m[:, 0] > 0]
            ph = {}
            if len(m) == 0:  # (the walk ran inside the row workgroups: no walking workgroups of their own)
                m = np.zeros((1, 16))
            for name, a, b in (("derive", 0, 1), ("to_planes_barrier", 1, 8), ("z_extent_to_barrier", 8, 9), ("clear_to_barrier", 9, 10), ("setup_walk_to_barrier", 10, 2), ("setup_walk", 1, 2), ("reserve", 2, 3), ("pairs", 3, 4), ("whole", 0, 7)):
                ok = (m[:, a] > 0) & (m[:, b] > 0)
                d = (m[ok, b] - m[ok, a]) / 100.0
                ph[name] = [round(float(x), 2) for x in np.percentile(d, [10, 50, 90, 100])] if len(d) else []
            ph["touched_rows_last_chunk_p50_p100"] = [float(np.percentile(m[:, 5], 50)), float(m[:, 5].max())]
            ph["chunks_p50_p100"] = [float(np.percentile(m[:, 6], 50)), float(m[:, 6].max())]
            walk_phases = ph
        if f == 59:  # end times of the walk workgroups and the row workgroups of one frame, as deciles
            for k in (2, 3):
                w = v[v[:, 2] == k + 1]
                hist.append([round(float(x), 2) for x in np.percentile((w[:, 1] - t0) / 100.0, [0, 10, 25, 50, 75, 90, 100])] if len(w) else [])
                if k == 3 and len(w):  # lifetimes of the row workgroups: the tiles that go on into the cluster walk live longest
                    hist.append([round(float(x), 2) for x in np.percentile((w[:, 1] - w[:, 0]) / 100.0, [50, 90, 95, 99, 100])])
    r = np.median(np.array(rows), axis=0)
    out = {name: {"first_start_us": round(r[k, 0], 2), "last_start_us": round(r[k, 1], 2), "last_end_us": round(r[k, 2], 2), "mean_lifetime_us": round(r[k, 3], 2), "workgroups": int(r[k, 4])}
           for k, name in enumerate(["compaction", "fill", "walk", "rows"])}
    out["end_time_percentiles_0_10_25_50_75_90_100"] = {"walk": hist[0], "rows": hist[1]}
    out["row_workgroup_lifetime_us_p50_p90_p95_p99_p100"] = hist[2] if len(hist) > 2 else []
    out["walk_phase_us_p10_p50_p90_p100"] = walk_phases
    print(json.dumps(out))
