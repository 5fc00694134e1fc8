"""gpurun_out/prof_<tag>/ (raw rocprofv3 CSVs) -> profiles/<tag>/ (committed summaries) + profiles/pmc_traffic.json.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, section HBM), so the read side is doubled."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
ALIAS = {"k_frame<true": "k_flat_propagate_cull", "k_frame<false": "k_cull"}


def short(name):
    n = name.replace("void ", "").replace("mi::", "")
    for k, v in ALIAS.items():
        if n.startswith(k):
            return v
    return n.split("(")[0].split("<")[0]


pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for wl in ("flat", "tree", "lights", "flat_static", "batching"):
    stats = os.path.join(src, wl, f"{wl}_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, f"{wl}_kernel_stats.csv"))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"{wl}_{ctr}", f"{wl}_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, ctrs in pmc.items():
    if k.startswith("__amd"):
        continue
    e = {c: {"avg_KiB_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in ctrs.items()}
    if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
        fetch = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"]) * 1024.0
        write = sum(ctrs["WRITE_SIZE"]) / len(ctrs["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_per_launch"] = int(2.0 * fetch + write)
        e["note"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts wide reads at half)"
    summary[k] = e
json.dump(summary, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
json.dump({k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for k, v in summary.items() if "hbm_bytes_per_launch" in v},
          open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
