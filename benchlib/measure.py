"""Timing of a workload (blocks of exactly K steps between barrier + synchronize pairs) and its roofline record."""
import json
import os
import time

import numpy as np

from .common import HBM_PEAK_GBPS, PROFILED_BLOCKS, ROOT


def measure(ctx, wl, steps, warmup, n_blocks=0, profile_all=False, barrier=None, reduce_max=None, agree=None, target_s=0.4):
    """W untimed frames, then B blocks of exactly `steps` frames, each between barrier + synchronize pairs (MAX over ranks
    per block), then PROFILED_BLOCKS blocks with per-dispatch events.  Returns (block seconds [B], per-kernel profile, info)."""
    import gc
    import torch

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if barrier:
            barrier()

    # A generation-2 pass of Python's cyclic GC over everything torch imported takes ~50 ms -- a thousand frames of
    # this workload -- and fires after a fixed number of allocations, i.e. at a random frame: collect now, and keep the
    # collector off while frames are being enqueued (what timeit does).
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    frame = [0]

    def run_block(k):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k):
            wl.step(frame[0])
            frame[0] += 1
        t_enq = time.perf_counter()
        ctx.synchronize()
        torch.cuda.synchronize()
        if barrier:
            barrier()
        t1 = time.perf_counter()
        return t1 - t0, t_enq - t0

    if warmup:
        run_block(warmup)
    est, _ = run_block(steps)                       # untimed: sizes the number of blocks
    if not n_blocks:
        n_blocks = int(min(600, max(15, target_s / max(est, 1e-6))))
        if agree:
            n_blocks = agree(n_blocks)              # every rank runs the same number of blocks
    times, enq = [], []
    try:
        ctx.debug_exchange_times(reset=True)  # (the multi-GPU exchange's share of the calling thread's time over the timed blocks)
    except Exception:  # noqa: BLE001 -- an older library: no such hook
        pass
    for _ in range(n_blocks):
        t, e = run_block(steps)
        times.append(t)
        enq.append(e)
    xt = None
    try:
        xt = ctx.debug_exchange_times()
    except Exception:  # noqa: BLE001
        pass
    if reduce_max:
        times = reduce_max(times)
    # profiled blocks: same frames, every launch of the workload's kernels timed (not part of the statistics)
    ctx.profile_filter(None if profile_all else wl.kernels)
    ctx.profile_sample(1)
    ctx.profile_burst(0)
    ctx.profile_enable(True)
    prof_t = [run_block(steps)[0] for _ in range(PROFILED_BLOCKS)]
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    if gc_was_enabled:
        gc.enable()
    info = {"host_enqueue_ms_per_step": round(1e3 * float(np.median(enq)) / steps, 5),
            "profiled_blocks_ms_per_step": round(1e3 * float(np.median(prof_t)) / steps, 5)}
    if xt and xt["wait_ns"] > 0:
        # MI_EXCHANGE_PIPELINED paces the caller n_bufs - 1 frames ahead of the exchange: a device-bound frame shows up in the caller's
        # time as WAITING for a gathered buffer's previous all-gather, not as work.  host_busy = the enqueue time without that wait.
        wait_ms = 1e-6 * xt["wait_ns"] / (n_blocks * steps)
        info["host_backpressure_ms_per_step"] = round(wait_ms, 5)
        info["host_busy_ms_per_step"] = round(max(0.0, info["host_enqueue_ms_per_step"] - wait_ms), 5)
    return np.array(times), prof, info


def block_stats(times, steps):
    ms = 1e3 * times / steps
    return {"n": int(len(ms)), "steps_per_block": steps, "median_ms_per_step": round(float(np.median(ms)), 5),
            "p10_ms_per_step": round(float(np.percentile(ms, 10)), 5), "p90_ms_per_step": round(float(np.percentile(ms, 90)), 5),
            "min_ms_per_step": round(float(ms.min()), 5), "max_ms_per_step": round(float(ms.max()), 5)}


def load_profiles():
    """Committed rocprofv3 evidence of the same commands (profiles/rocprof_summary.json, written by tools/summarize_profiles.py):
    kernel-trace average durations and the HBM traffic of the separate --pmc passes, per workload and kernel."""
    p = os.path.join(ROOT, "profiles", "rocprof_summary.json")
    try:
        return json.load(open(p))
    except Exception:
        return {}


def roofline_of(wl, prof, steps, live_traffic=None):
    """The roofline record of the workload's dominant kernel.

    `achieved` / `frac` price a launch at the bytes the kernel is LAID OUT TO MOVE (`moved_bytes_per_launch`): with the row
    summary a wave whose 64 rows agree in Aabb / flags / RenderLayers reads 32 bytes instead of 64 x 29, so those bytes are not
    counted -- a `frac` near 1.0 can never come from reads that were skipped.  `frac_algorithmic` is the same launch priced at SURVEY
    8(d)'s algorithmic bytes (what the stage costs without the summary).  `traffic` = HBM bytes per launch from the PMC counters
    (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction): measured in this run (`traffic_source` "live ...": two short
    rocprofv3 --pmc passes of the same command, benchlib/traffic.py) or, failing that, replayed from profiles/ and labelled so."""
    dk = prof.get(wl.dominant)
    if not dk or not dk["launches"]:
        return None
    avg_s = dk["avg_us"] * 1e-6
    alg_bytes = wl.bytes_per_row * wl.rows
    lay = getattr(wl, "layout_bytes_per_row", None)
    moved = alg_bytes if lay is None else min(alg_bytes, lay * wl.rows)
    achieved = moved / avg_s / 1e9
    out = {"bound": "hbm", "kernel": getattr(wl, "kernel_name", wl.dominant), "timer_slot": wl.dominant, "achieved": round(achieved, 1),
           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
           "avg_kernel_us": round(dk["avg_us"], 3), "launches": dk["launches"], "moved_bytes_per_launch": int(moved),
           "algorithmic_bytes_per_launch": int(alg_bytes), "frac_algorithmic": round(alg_bytes / avg_s / 1e9 / HBM_PEAK_GBPS, 4),
           "timing": f"per-dispatch start/stop events (hipExtLaunchKernelGGL) on every launch of {PROFILED_BLOCKS} profiled blocks of "
                     f"{steps} steps that follow the timed blocks"}
    # `bound` keeps the contract's enum; what it means at this size is said beside it: a frame whose columns fit the 256 MiB Infinity
    # Cache is served from it from the second frame on (FETCH_SIZE counts those hits), so "hbm" is the roofline it is PRICED against,
    # not where the bytes came from -- the 10 M-row workloads in bench_full.json are the HBM-proper evidence
    ws_mib = moved / (1 << 20)
    out["working_set_mib"] = round(ws_mib, 1)
    out["bound_note"] = ("hbm (working set MALL-resident: %.0f MiB per launch < the 256 MiB Infinity Cache)" % ws_mib) if ws_mib < 256.0 else "hbm (working set beyond the Infinity Cache)"
    if lay is not None:
        out["layout_note"] = ("waves whose 64 rows agree in Aabb / flags / RenderLayers read a 32-byte summary instead of 64 x 29 B of "
                              "columns (bit-identical results; --row-summary 1 switches it off): moved < algorithmic")
    ev = load_profiles().get(getattr(wl, "profile_key", wl.name), {}).get(wl.dominant)
    if live_traffic and live_traffic.get("hbm_bytes_per_launch"):
        out["traffic"] = live_traffic["hbm_bytes_per_launch"]
        out["traffic_source"] = live_traffic["source"]
        if live_traffic.get("avg_us"):
            out["rocprof_avg_kernel_us"] = live_traffic["avg_us"]
            out["rocprof_source"] = live_traffic["source"]
    elif ev and ev.get("hbm_bytes_per_launch"):
        out["traffic"] = ev["hbm_bytes_per_launch"]
        out["traffic_source"] = f"replayed from {ev.get('source', 'profiles/')} (rocprofv3 --pmc passes of this command, NOT measured in this run)"
    if ev and ev.get("avg_us") and "rocprof_avg_kernel_us" not in out:
        out["rocprof_avg_kernel_us"] = ev["avg_us"]
        out["rocprof_source"] = ev.get("source", "profiles/") + " (replayed)"
    if "rocprof_avg_kernel_us" in out:
        out["rocprof_frac"] = round(moved / (out["rocprof_avg_kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
    return out
