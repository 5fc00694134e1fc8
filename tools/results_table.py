"""Markdown rows of DESIGN.md's result tables from a bench_full.json (the bench's own output).

    python tools/results_table.py profiles/<tag>/bench_full.json"""
import json
import sys

d = json.load(open(sys.argv[1]))


def mb(x):
    return "-" if not x else f"{x / 1e6:.1f}"


def row(name, w):
    rf = w.get("roofline") or {}
    ks = " + ".join(f"{k} {v:.1f}" for k, v in (w.get("kernels") or {}).items())
    print(f"| {name} | {1e3 * w['ms_per_step']:.1f} | {w['value'] / 1e9:.2f} G {w.get('unit', '').split('/')[0]}/s | {rf.get('kernel')}: live {rf.get('avg_kernel_us')}"
          f" / rocprof {rf.get('rocprof_avg_kernel_us')} ({ks}) | {mb(rf.get('moved_bytes_per_launch'))}, {mb(rf.get('algorithmic_bytes_per_launch'))},"
          f" {mb(rf.get('traffic'))} | {rf.get('frac')} / {rf.get('frac_algorithmic')} |")


print("| workload | us / frame | throughput | dominant kernel | MB per launch: moved, 8d, PMC | frac (moved) / frac_algorithmic |")
print("|---|---|---|---|---|---|")
row("frame (the metric)", d)
for k, w in (d.get("other_workloads") or {}).items():
    if isinstance(w, dict) and "ms_per_step" in w:
        row(k, w)
cb = d.get("cpu_baseline") or {}
print("\ncpu_baseline:", {k: cb.get(k) for k in ("value", "cores", "frame_ms", "stage_ms")})
e = d.get("end_to_end") or {}
print("\n| dirty | us per frame = fill + frame call + results | library_us | x_cpu_port whole / library calls | pcie_frac |")
print("|---|---|---|---|---|")
for k, v in e.items():
    if isinstance(v, dict) and "us_per_frame" in v:
        s = v["stage_us"]
        print(f"| {k} | {v['us_per_frame']} = {s['gather_into_window_and_commit']} + {s['frame_call']} + {s['results_in_place']} | {v['library_us']} |"
              f" {e.get('x_cpu_port', {}).get(k)} / {(e.get('x_cpu_port_library_calls') or {}).get(k)} | {v['pcie_frac']} |")
h = d.get("end_to_end_host_layer") or {}
for k, v in (h.get("fused_frame") or {}).items():
    print(f"host layer fused {k}: {v['us_per_frame']} = gather {v['gather_us']} + library {v['library_calls_us']} + ECS writes {v['ecs_writes_us']}")
