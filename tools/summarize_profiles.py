"""gpurun_out/prof_<tag>/ (raw rocprofv3 CSVs) -> profiles/<tag>/ (committed summaries) + profiles/rocprof_summary.json.

    python tools/summarize_profiles.py <tag> [workload ...]

Per workload: the kernel-trace statistics CSV (average duration per kernel) and, from the separate --pmc passes, HBM bytes per
launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes (MI355X_MICROARCH.md, section HBM), so the read side is doubled.  rocprof_summary.json keeps, per
workload and kernel, {avg_us, calls, hbm_bytes_per_launch, source}: bench.py quotes it next to its live event timing."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
workloads = sys.argv[2:] or ["frame", "frame_plain_columns", "flat", "flat_plain_columns", "tree_frame", "tree_frame_two_launches", "flat_10m_1view", "flat_10m_4views", "tree", "tree_subtree", "tree_leaves", "lights", "flat_static", "flat_static_no_sphere",
                             "flat_static_10m_4views", "flat_static_10m_4views_no_cull_order", "flat_static_4m_4views", "tree_by_levels", "batching",
                             "batching_sorted_1k", "batching_sorted_4k", "batching_sorted_64k", "batching_sorted_1m"]
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
ALIAS = {"k_frame_sph<true": "k_flat_propagate_cull", "k_frame_sph<false": "k_cull", "k_frame_sph_pairs<true": "k_flat_propagate_cull", "k_frame_sph_pairs<false": "k_cull",  # the world-sphere frame kernel: PARTIAL = the changed-rows frame, else cull only
         "k_frame<1": "k_flat_propagate_cull", "k_frame<2": "k_flat_propagate_cull", "k_frame<0": "k_cull",
         "k_frame_pairs<1": "k_flat_propagate_cull", "k_frame_pairs<2": "k_flat_propagate_cull", "k_frame_pairs<0": "k_cull",  # 2 .. 4 camera views  # PROP: 1 all rows, 2 changed rows, 0 resident G (any INLINE_VIEWS / WITH_WALK variant)
         "k_frame<true": "k_flat_propagate_cull", "k_frame<false": "k_cull",  # (profiles from before PROP was an int)
         "k_propagate_level": "k_propagate_stream", "k_propagate_narrow": "k_propagate_stream",
         "k_propagate_wave_tiles": "k_propagate_fans",  # a forest of small trees: a wave per tile (round 6), timed in the tile launch's slot
         "k_propagate_strips": "k_propagate_fans",  # a deep or lopsided tree: strips (round 6), timed in the tile launch's slot
         "k_frame_cells": "k_cull",  # the frame over the static cull order (its cell test: k_cells_test, its lists: k_cells_blocks / k_cells_lists)
         "k_sorted_walk<512u, 16u, true>": "k_batch_scan", "k_sorted_walk<512, 16, true>": "k_batch_scan",  # the tiles' records
         "k_sorted_walk": "k_batch_sorted"}


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("mi::", "")
    for k, v in ALIAS.items():
        if n.startswith(k):
            return v
    return n.split("(")[0].split("<")[0]


summary_path = os.path.join("profiles", "rocprof_summary.json")
try:
    summary = json.load(open(summary_path))
except Exception:
    summary = {}
for wl in workloads:
    entry = {}
    stats = os.path.join(src, wl, f"{wl}_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, f"{wl}_kernel_stats.csv"))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(stats)):
            k = short(r["Name"])
            agg[k][0] += int(r["Calls"])
            agg[k][1] += float(r["TotalDurationNs"])
        for k, (calls, tot) in agg.items():
            if calls and not k.startswith("__amd"):
                entry[k] = {"avg_us": round(tot / calls / 1e3, 3), "calls": calls}
    cmd = os.path.join(src, f"{wl}.cmd")
    if os.path.exists(cmd):
        shutil.copy(cmd, os.path.join(dst, f"{wl}.cmd"))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"{wl}_{ctr}", f"{wl}_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = {}
    for k, ctrs in pmc.items():
        if k.startswith("__amd"):
            continue
        e = {c: {"avg_KiB_per_dispatch": round(sum(v) / len(v), 3), "dispatches": len(v)} for c, v in ctrs.items()}
        if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            fetch = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"]) * 1024.0
            write = sum(ctrs["WRITE_SIZE"]) / len(ctrs["WRITE_SIZE"]) * 1024.0
            e["hbm_bytes_per_launch"] = int(2.0 * fetch + write)
            e["note"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts wide reads at half)"
            entry.setdefault(k, {})["hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
        counters[k] = e
    if counters:
        json.dump(counters, open(os.path.join(dst, f"{wl}_pmc_counters.json"), "w"), indent=1)
    for k in entry:
        entry[k]["source"] = f"profiles/{tag}/{wl}_kernel_stats.csv" + (f" + {wl}_pmc_counters.json" if k in counters else "")
    for extra in ("tree_frame_sq_counters.txt", "tree_sq_counters.txt"):  # (tools/profile.sh: SQ counters of the fused hierarchy frame / the tile launch)
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))
    if entry:
        summary[wl] = entry
json.dump(summary, open(summary_path, "w"), indent=1, sort_keys=True)
print(json.dumps({w: summary.get(w) for w in workloads}, indent=1))
