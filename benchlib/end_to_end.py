"""End to end: the metric frame with the host on both sides of it (PCIe-inclusive; never `value`)."""
import json
import time

import numpy as np

from .common import N_FRAMES, camera_frusta


def pcie_peak():
    """What the link gives: hipMemcpyAsync between pinned host memory and the device, 64 MiB, both directions (GB/s)."""
    import torch
    n = 64 << 20
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = {}
    for name, (dst, src) in (("h2d", (dev, host)), ("d2h", (host, dev))):
        best = 0.0
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src, non_blocking=True)
            b.record()
            b.synchronize()
            best = max(best, n / (a.elapsed_time(b) * 1e-3) / 1e9)
        out[name] = round(best, 2)
    return out


_GATHER = None


def build_gather(force=False):
    """benchlib/ecs_gather.c -> benchlib/libecs_gather.so (gcc -O2 -pthread): the harness's stand-in for the ECS side's par_iter."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, "ecs_gather.c"), os.path.join(here, "libecs_gather.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-pthread", "-shared", "-fPIC", src, "-o", so], check=True, capture_output=True)
    return so


def gather_lib():
    global _GATHER
    if _GATHER is None:
        import ctypes as C
        _GATHER = C.CDLL(build_gather())
        _GATHER.ecs_gather_rows.restype = None
        _GATHER.ecs_copy_rows.restype = None
    return _GATHER


def end_to_end(ctx, wl, frames=12, cpu_frame_ms=None):
    """The same frame with the host on both sides of it.  Per frame: the rows a Changed<Transform> query yields go in -- written
    straight into the library's pinned upload window (mi_map_upload_window / mi_commit_upload_window; dense at 100 %) --, ONE frame
    call runs propagate + cull + cluster (MI_CULL_CHANGED_ROWS below 100 %), and what the ECS needs comes back with ONE
    mi_download_frame_results delivered in place: the changed GlobalTransforms, the camera's VisibleEntities list, the cluster
    offsets / counts / index list.  Wall clock, synchronised every frame."""
    import bevy_amd as B
    from bevy_amd import api
    sc = wl.scene
    n = sc["n"]
    views = wl.keep[0]
    t3 = sc["translation"].reshape(n, 3)
    r4, s3 = sc["rotation"].reshape(n, 4), sc["scale"].reshape(n, 3)
    link = pcie_peak()
    out = {"pcie_peak_GBps": link}
    import ctypes as C
    import os
    gl = gather_lib()
    # a par_iter's stand-in: up to 32 threads, a quarter of the hardware threads at most (the CPU port it is compared with takes the best of
    # a sweep up to all of them); the helper itself keeps >= 8 192 rows per thread
    threads = max(1, min(32, (os.cpu_count() or 2) // 4))
    fpt, u32t = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    fp = lambda a: None if a is None else a.ctypes.data_as(fpt)
    t3c, r4c, s3c = np.ascontiguousarray(t3), np.ascontiguousarray(r4), np.ascontiguousarray(s3)
    out["gather_threads"] = threads
    rng = np.random.default_rng(0)
    ctx.upload_changed(np.zeros(n, np.uint8))
    ctx.propagate(B.PROPAGATE_ALL_DIRTY)
    ctx.synchronize()
    bufs = api.FrameResultBuffers(n, n, views[0].n_clusters, 1 << 20, in_place=True)
    for pct, comps in ((1, "trs"), (10, "trs"), (100, "trs"), (100, "r")):  # "r": every cube rotates (many_cubes --rotate-cubes): rotations only go up
        k = n * pct // 100
        rows = np.sort(rng.choice(n, k, replace=False)).astype(np.uint32) if pct < 100 else None
        times, t_in, t_commit, t_run, t_out, h2d, d2h = [], [], [], [], [], 0, 0
        for f in range(frames + 2):
            fr = api.PreparedFrusta(camera_frusta(1, f))
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.cluster_upload_view(views[f % N_FRAMES])
            if rows is not None:  # the ECS side's gather loop, writing into the window
                w, wrows, wt, wr, ws = ctx.map_upload_window(k)
                gl.ecs_gather_rows(k, rows.ctypes.data_as(u32t), fp(t3c), fp(r4c), fp(s3c), wrows.ctypes.data_as(u32t), fp(wt), fp(wr), fp(ws), threads)
                tc = time.perf_counter()
                ctx.commit_upload_window(w, k)
                commit_s = time.perf_counter() - tc
            else:  # every row: dense windows, a chunk at a time -- chunk i crosses PCIe (DMA straight from the window) while the host
                   # fills chunk i + 1, and (the library's doing: a sequence of dense windows that carries the whole table) chunk i's
                   # GlobalTransforms are computed and start back at once, under the upload of the chunks behind it.
                   # (Four Python threads filling eight windows at once were SLOWER: 3.4 against 1.5 ms.)
                chunk = (n + 7) // 8
                commit_s = 0.0
                for lo in range(0, n, chunk):
                    m = min(chunk, n - lo)
                    w, _, wt, wr, ws = ctx.map_upload_window(m, dense=True, components=comps)
                    gl.ecs_copy_rows(lo, m, fp(t3c), fp(r4c), fp(s3c), fp(wt), fp(wr), fp(ws), threads)
                    tc = time.perf_counter()
                    ctx.commit_upload_window(w, m, first_row=lo)
                    commit_s += time.perf_counter() - tc
            t1 = time.perf_counter()
            ctx.propagate_and_cull(fr, flags=B.CULL_END_FRAME | B.CULL_WITH_CLUSTERS | (B.CULL_CHANGED_ROWS if rows is not None else 0))
            t2 = time.perf_counter()
            res = ctx.download_frame_results(bufs)
            got_g, vis_rows, off, counts, total = len(res["changed_rows"]), res["visible_rows"], res["cluster_offsets"], res["cluster_counts"], res["cluster_total"]
            t3_ = time.perf_counter()
            if f >= 2:
                times.append(t3_ - t0)
                t_in.append(t1 - t0)
                t_commit.append(commit_s)
                t_run.append(t2 - t1)
                t_out.append(t3_ - t2)
            h2d = k * 44 if rows is not None else n * (40 if comps == "trs" else 16)
            d2h = got_g * 52 + len(vis_rows) * 4 + len(off) * 4 + counts.size * 4 + total * 4
        med = float(np.median(times))
        eff = (h2d + d2h) / med / 1e9
        # 1.0 = the time both directions would take one after the other at their peaks (a frame whose results depend on its whole input);
        # where the library overlaps them (100 % dirty: results ahead of the frame) the figure can pass 1.0, up to 2.0 for equal halves
        link_s = h2d / (link["h2d"] * 1e9) + d2h / (link["d2h"] * 1e9)
        out[f"{pct}pct_dirty" + ("" if comps == "trs" else "_rotations_only")] = {"dirty_rows": int(k), "us_per_frame": round(1e6 * med, 1), "entities_per_s": round(wl.units / med, 1),
                                  "h2d_bytes": int(h2d), "d2h_bytes": int(d2h), "pcie_GBps_effective": round(eff, 2),
                                  "pcie_frac": round(link_s / med, 3),
                                  "stage_us": {"gather_into_window_and_commit": round(1e6 * float(np.median(t_in)), 1),
                                               "of_which_commit_calls": round(1e6 * float(np.median(t_commit)), 1),
                                               "frame_call": round(1e6 * float(np.median(t_run)), 1),
                                               "results_in_place": round(1e6 * float(np.median(t_out)), 1)},
                                  "library_us": round(1e6 * float(np.median(np.array(t_commit) + np.array(t_run) + np.array(t_out))), 1),
                                  "changed_global_transforms_read_back": int(got_g), "visible_entities": int(len(vis_rows)),
                                  "cluster_index_entries": int(total)}
    if cpu_frame_ms:
        # THE number a Bevy user would feel: how many times faster the whole frame is -- PCIe both ways included -- than the CPU port's
        # frame (cpu_baseline: every Transform dirty, best thread count; its visibility passes do not get cheaper when fewer rows move,
        # its propagate does: at 1 % / 10 % dirty the CPU figure is an upper bound of its cost, so these ratios are upper bounds too)
        out["x_cpu_port"] = {k: round(1e3 * cpu_frame_ms / v["us_per_frame"], 2) for k, v in out.items() if isinstance(v, dict) and "us_per_frame" in v}
        # the same ratio over the library's calls alone (commit + frame + results): what is left when the ECS-side gather into the window
        # (benchlib/ecs_gather.c, a few threads) is taken out
        out["x_cpu_port_library_calls"] = {k: round(1e3 * cpu_frame_ms / v["library_us"], 2) for k, v in out.items() if isinstance(v, dict) and "library_us" in v}
        out["cpu_port_frame_us_all_dirty"] = round(1e3 * cpu_frame_ms, 1)
    out["note"] = ("same frame as `value` with the host on both sides, through ctypes: dirty Transforms written into the library's pinned upload "
                   "window (no staging copy; the ECS side's gather loop is benchlib/ecs_gather.c on `gather_threads` threads -- a par_iter's stand-in; "
                   "until round 4 it was ONE Python thread of numpy.take, 1.9 of the 2.7 ms of the 10 % frame) and committed, ONE frame call (propagate + cull + "
                   "cluster, MI_CULL_CHANGED_ROWS), ONE mi_download_frame_results delivered in place (one packing launch into pinned memory, one "
                   f"device wait, no copy out); median wall time of {frames} frames, each synchronised.  library_us = the library's calls alone "
                   "(commit + frame + results; the rest of us_per_frame is the gather loop filling the window); "
                   "at 100 % the table goes in as eight dense windows in a row, which the library sends piece by piece with each piece's "
                   "GlobalTransforms computed at once and on their way back under the rest of the upload (PCIe full duplex): the results call "
                   "finds them on the host.  pcie_frac = (h2d / peak_h2d + d2h / "
                   "peak_d2h) / frame time, peaks measured in this run with pinned hipMemcpyAsync (pcie_peak_GBps): 1.0 = both directions one "
                   "after the other at their peaks, more than that only where they overlap")
    return out


def end_to_end_host_layer(n_entities):
    """The same frames through the C++ host layer (bevy_amd/host/bevy_mi355x_host.hpp: a World with the path's components and
    change flags, Mi355xPlugin) -- the code a maintainer would ship as the plugin's systems, not ctypes: the three systems of
    round 2 (a device wait each) next to the fused frame (Mi355xPlugin::frame: one upload window, one frame call, one in-place
    results call).  tests/cpp/host_systems_test --bench prints the JSON."""
    import subprocess
    from bevy_amd import build as mi_build
    exe = mi_build.build_host_tests()
    res = subprocess.run([exe, "--bench", str(n_entities), "8"], capture_output=True, text=True, timeout=600)
    if res.returncode != 0:
        return {"error": (res.stderr or res.stdout)[-500:]}
    d = json.loads(res.stdout.strip().splitlines()[-1])
    d["note"] = ("tests/cpp/host_systems_test --bench: median wall time of 8 frames per dirty fraction, World::clear_trackers outside the timed "
                 "region.  us_per_frame is the whole system -- the World's change scan, gather, library calls, ECS writes incl. the stock "
                 "reset_view_visibility / mark_newly_hidden passes; library_calls_us is upload commit .. results returned (what the ctypes block above times)")
    return d
