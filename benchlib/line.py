"""The result line.  bench.py prints ONE compact JSON object as the LAST line of stdout -- the contract's keys only, under
MAX_LINE_BYTES -- and writes everything else it measured (every other workload, the thread sweep of the CPU baseline, the
end-to-end stages, block statistics) to bench_full.json (repo root, and gpurun_out/ when that exists).

Round 3's single 40 KB line was not parsed by the driver; tests/test_bench_line.py pins the size and the key set."""
import json
import os

from .common import ROOT

MAX_LINE_BYTES = 4096
FULL_NAME = "bench_full.json"

# keys of the compact line, in order; `scaling` only at N > 1
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "bound_note", "working_set_mib", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_kernel_us", "launches",
                 "moved_bytes_per_launch", "algorithmic_bytes_per_launch", "frac_algorithmic", "rocprof_avg_kernel_us")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "host_cores")
CONFIG_KEYS = ("workload", "baseline_config", "entities", "entities_total", "entities_this_rank", "rows_per_frame", "nodes", "lights",
               "meshes", "views", "items", "parallelism", "row_summary", "rccl_ranks", "exchange_mode")


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact(out):
    """The contract's object from the full one.  Strings are clipped, nothing else is dropped silently: what is left out is in
    bench_full.json, and `full` says where."""
    line = {}
    for k in TOP_KEYS:
        if k in out:
            line[k] = out[k]
    cfg = out.get("config") or {}
    line["config"] = {k: (_clip(cfg[k], 420) if isinstance(cfg[k], str) else cfg[k]) for k in CONFIG_KEYS if k in cfg}
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: (_clip(rf[k], 160) if isinstance(rf[k], str) else rf[k]) for k in ROOFLINE_KEYS if k in rf}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: (_clip(cb[k], 300) if isinstance(cb[k], str) else cb[k]) for k in CPU_KEYS if k in cb}
    else:
        line["cpu_baseline"] = None
    bl = out.get("blocks")
    if bl:
        line["blocks"] = {"n": bl["n"], "p10_ms_per_step": bl["p10_ms_per_step"], "p90_ms_per_step": bl["p90_ms_per_step"]}
    e2e = out.get("end_to_end")
    if isinstance(e2e, dict) and "x_cpu_port" in e2e:
        # PCIe-inclusive frames (never `value`): us per frame and the ratio to the CPU port's frame, per dirty fraction
        # (us_per_frame includes the harness's gather loop into the upload window, benchlib/ecs_gather.c; library_us = the library's calls alone)
        line["end_to_end"] = {"us_per_frame": {k: v["us_per_frame"] for k, v in e2e.items() if isinstance(v, dict) and "us_per_frame" in v},
                              "library_us": {k: v["library_us"] for k, v in e2e.items() if isinstance(v, dict) and "library_us" in v},
                              "x_cpu_port": e2e["x_cpu_port"], "x_cpu_port_library_calls": e2e.get("x_cpu_port_library_calls")}
    pc = (out.get("other_workloads") or {}).get("frame_plain_columns")
    if pc:
        # the metric frame's scene is the row summary's best case (every cube shares one Aabb / flags / RenderLayers): the same frame
        # with every row reading its own columns, beside it
        line["plain_columns"] = {"ms_per_step": pc["ms_per_step"], "value": pc["value"], "frac": (pc.get("roofline") or {}).get("frac"),
                                 "note": "the same frame, every row reading its own Aabb / flags / RenderLayers (no row summary)"}
    sg = out.get("single_gpu_same_workload")
    if sg:
        line["single_gpu_same_workload"] = {"value": sg["value"], "ms_per_step": sg["ms_per_step"]}
    # N > 1: value / (n_gpus x the same workload on ONE GPU); CPU time per frame call; the gathered masks against the masks of the whole
    # scene in one context, bit for bit; the collective alone (events on its stream); every rank's frame kernel
    for k in ("scaling_efficiency", "host_enqueue_ms_per_step", "host_busy_ms_per_step", "host_backpressure_ms_per_step", "gathered_masks_match_single_gpu", "all_gather_us",
              "kernel_us_per_rank"):
        if k in out:
            line[k] = out[k]
    line["full"] = FULL_NAME
    s = json.dumps(line)
    if len(s) > MAX_LINE_BYTES:  # cannot happen with the clips above; a guard, not a code path
        line["config"] = {"workload": _clip(cfg.get("workload", ""), 200)}
        line.pop("end_to_end", None)
        s = json.dumps(line)
    assert len(s) <= MAX_LINE_BYTES, len(s)
    return s


def write_full(out):
    paths = [os.path.join(ROOT, FULL_NAME)]
    scratch = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(scratch):
        paths.append(os.path.join(scratch, FULL_NAME))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
            written.append(p)
        except OSError:
            pass
    return written
