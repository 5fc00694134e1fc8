"""bench.py's result line: the LAST stdout line is a compact JSON object the driver can parse -- under 4 KB, the contract's keys -- at
N = 1 (`frame`) and at N > 1 (`sharded`); everything else goes to bench_full.json.  (BENCH_r03: the 40 KB line was not parsed.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib import line as L  # noqa: E402
from benchlib import traffic as T  # noqa: E402


def canned(n_gpus=1):
    long = "x" * 5000
    roof = {"bound": "hbm", "kernel": "k_frame<1,true,true>", "timer_slot": "k_flat_propagate_cull", "achieved": 5100.0, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.6375, "traffic": 106_600_000, "traffic_source": "live: " + long, "avg_kernel_us": 20.1, "launches": 60,
            "moved_bytes_per_launch": 100_900_000, "algorithmic_bytes_per_launch": 132_500_000, "frac_algorithmic": 0.82, "timing": long,
            "layout_note": long, "rocprof_avg_kernel_us": 20.6, "rocprof_frac": 0.61, "rocprof_source": long,
            "bound_note": "hbm (working set MALL-resident: 96 MiB per launch < the 256 MiB Infinity Cache)", "working_set_mib": 96.2}
    out = {"metric": "entities/sec through propagate+cull+cluster at 1M entities", "value": 4.9e10, "unit": "entities/s", "n_gpus": n_gpus, "steps": 20,
           "warmup": 5, "ms_per_step": 0.0204, "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": long, "baseline_config": "BASELINE.json configs[1] + configs[2]", "entities": 1_000_000, "rows_per_frame": 1_110_000,
                      "lights": 100_000, "meshes": 10_000, "views": 1, "parallelism": "1 GPU", "row_summary": True, "tile_plan": {"a": list(range(500))}},
           "timing": long, "blocks": {"n": 30, "steps_per_block": 20, "median_ms_per_step": 0.02, "p10_ms_per_step": 0.0199, "p90_ms_per_step": 0.0209,
                                      "min_ms_per_step": 0.019, "max_ms_per_step": 0.03},
           "roofline": roof, "kernels": {f"k{i}": 1.0 for i in range(40)},
           "cpu_baseline": {"value": 3.5e8, "unit": "entities/s", "cores": 32, "kind": "port", "sample": long, "host_cores": 256,
                            "frame_ms": 2.88, "thread_sweep_ms_per_frame": {str(i): 1.0 for i in range(30)}, "stage_ms": {"a": 1.0}},
           "end_to_end": {"pcie_peak_GBps": {"h2d": 56.0, "d2h": 56.0},
                          "1pct_dirty": {"us_per_frame": 181.0, "stage_us": {"x": 1.0}}, "10pct_dirty": {"us_per_frame": 1800.0},
                          "100pct_dirty": {"us_per_frame": 2090.0}, "x_cpu_port": {"1pct_dirty": 15.9, "10pct_dirty": 1.6, "100pct_dirty": 1.38},
                          "note": long},
           "end_to_end_host_layer": {"note": long}, "other_workloads": {f"w{i}": {"config": {"workload": long}, "roofline": roof} for i in range(19)}}
    out["other_workloads"]["frame_plain_columns"] = {"ms_per_step": 0.0257, "value": 3.9e10, "roofline": dict(roof, frac=0.67), "config": {"workload": long}}
    if n_gpus > 1:
        out.update({"scaling": "strong", "metric": "entities/sec through propagate+cull (10M entities x 4 frusta, 1/2/4/8-GPU scaling)",
                    "cpu_baseline": None, "single_gpu_same_workload": {"value": 5.5e10, "unit": "entities/s", "ms_per_step": 0.18, "note": long}})
        out["config"].update({"entities_total": 10_000_000, "entities_this_rank": 1_250_000, "parallelism": "row-range shard x8", "rccl_ranks": 8,
                              "exchange_mode": "rccl-native, ncclAllGather enqueued by the library's exchange thread"})
        out.update({"scaling_efficiency": 0.81, "host_enqueue_ms_per_step": 0.0093, "gathered_masks_match_single_gpu": True, "all_gather_us": 14.2,
                    "kernel_us_per_rank": [22.1] * n_gpus})
        del out["end_to_end"], out["other_workloads"]
    return out


REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"}


def test_single_gpu_line_is_compact_and_complete():
    s = L.compact(canned(1))
    assert "\n" not in s and len(s) < L.MAX_LINE_BYTES == 4096
    d = json.loads(s)
    assert REQUIRED <= set(d) and "scaling" not in d
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_us", "algorithmic_bytes_per_launch",
            "moved_bytes_per_launch", "frac_algorithmic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["config"]["workload"].endswith("...") and "tile_plan" not in d["config"]
    assert d["end_to_end"]["x_cpu_port"]["100pct_dirty"] == 1.38 and d["end_to_end"]["us_per_frame"]["1pct_dirty"] == 181.0
    assert "other_workloads" not in d and d["full"] == L.FULL_NAME
    # frac is the bytes-moved figure: never above the algorithmic one
    assert d["roofline"]["frac"] <= d["roofline"]["frac_algorithmic"]
    # the honest headline: the contract's enum stays, what "hbm" means at this size is said beside it, and the
    # same frame without the row summary's best case travels in the line
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["bound_note"].startswith("hbm (working set MALL-resident")
    assert d["plain_columns"]["ms_per_step"] == 0.0257 and d["plain_columns"]["frac"] == 0.67


def test_sharded_line_is_compact_and_complete():
    s = L.compact(canned(8))
    assert len(s) < 4096
    d = json.loads(s)
    assert REQUIRED <= set(d) and d["scaling"] == "strong" and d["n_gpus"] == 8 and d["cpu_baseline"] is None
    assert d["single_gpu_same_workload"]["value"] == 5.5e10
    # what the exchange was and what it cost the calling thread, next to the efficiency against the same scene on one GPU
    assert d["config"]["rccl_ranks"] == 8 and d["config"]["exchange_mode"].startswith("rccl-native")
    assert d["scaling_efficiency"] == 0.81 and d["host_enqueue_ms_per_step"] == 0.0093
    # the first N > 1 run certifies itself: gathered masks against the whole scene in one context, the collective alone, every rank's kernel
    assert d["gathered_masks_match_single_gpu"] is True and d["all_gather_us"] == 14.2 and len(d["kernel_us_per_rank"]) == 8


def test_roofline_says_where_the_bytes_live():
    from benchlib.measure import roofline_of

    class W:
        dominant, name, kernel_name = "k_flat_propagate_cull", "x_not_in_profiles", "k_frame<1,true,0>"
        bytes_per_row, rows, layout_bytes_per_row = 119.0, 1_000_000, None
    r = roofline_of(W, {"k_flat_propagate_cull": {"avg_us": 20.0, "launches": 60}}, 20)
    assert r["bound"] == "hbm" and "MALL-resident" in r["bound_note"] and 113 < r["working_set_mib"] < 114
    W.rows = 10_000_000
    r = roofline_of(W, {"k_flat_propagate_cull": {"avg_us": 150.0, "launches": 60}}, 20)
    assert r["bound"] == "hbm" and "beyond the Infinity Cache" in r["bound_note"]


def test_unpack_gathered_masks():
    """MaskGatherer.unpack_gathered: [world][views][w] words -> bool [views][rows], shard by shard (what bench.py --gpus N compares with
    the single-context masks)."""
    import numpy as np
    from bevy_amd import sharding
    n, world, views = 1000, 3, 2
    g = sharding.MaskGatherer(n, world, views, 0, device=None, direct=False)
    rng = np.random.default_rng(1)
    want = rng.random((views, n)) < 0.3
    words = np.zeros((world, views, g.w), np.uint64)
    for r in range(world):
        lo, hi = sharding.shard_rows(n, world, r)
        for v in range(views):
            bits = np.zeros(g.w * 64, np.uint8)
            bits[:hi - lo] = want[v, lo:hi]
            words[r, v] = np.packbits(bits, bitorder="little").view(np.uint64)
    assert np.array_equal(g.unpack_gathered(words.reshape(-1)), want)


def test_the_stage_list_leads_the_workload_string():
    """The driver keeps the first ~100 characters of config.workload: they must say which stages ran (the clipped string
    did not say that clustering is in the metric frame)."""
    import re
    for name in ("wl_frame.py", "wl_flat.py"):
        src = open(os.path.join(ROOT, "benchlib", name)).read()
        m = re.search(r'config = \{"workload": f"([^"]*)"', src)
        assert m and m.group(1).startswith("propagate+cull"), (name, m and m.group(1)[:60])
    assert "propagate+cull+cluster 16x9x24" in open(os.path.join(ROOT, "benchlib", "wl_frame.py")).read()


def test_roofline_prices_the_launch_at_the_bytes_it_moves():
    from benchlib.measure import roofline_of

    class W:
        dominant, name, kernel_name = "k_flat_propagate_cull", "frame_not_in_profiles", "k_frame<1,true,true>"
        bytes_per_row, rows, layout_bytes_per_row = 119.0, 1_000_000, 90.5
    prof = {"k_flat_propagate_cull": {"avg_us": 20.0, "launches": 60}}
    r = roofline_of(W, prof, 20)
    assert r["moved_bytes_per_launch"] == 90_500_000 and r["algorithmic_bytes_per_launch"] == 119_000_000
    assert abs(r["frac"] - 90.5e6 / 20e-6 / 1e9 / 8000.0) < 1e-3 and abs(r["frac_algorithmic"] - 119e6 / 20e-6 / 1e9 / 8000.0) < 1e-3
    assert r["traffic"] is None
    r = roofline_of(W, prof, 20, {"hbm_bytes_per_launch": 93_000_000, "source": "live: test"})
    assert r["traffic"] == 93_000_000 and r["traffic_source"].startswith("live")
    W.layout_bytes_per_row = None
    r = roofline_of(W, prof, 20)
    assert r["moved_bytes_per_launch"] == r["algorithmic_bytes_per_launch"] and r["frac"] == r["frac_algorithmic"]


def test_counter_csv_parsing(tmp_path):
    p = tmp_path / "pmc_counter_collection.csv"
    p.write_text('"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n'
                 '1,1,"void mi::(anonymous namespace)::k_frame<1, true, true>(mi::FrameArgs)","FETCH_SIZE",1000.0\n'
                 '2,2,"void mi::(anonymous namespace)::k_frame<1, true, true>(mi::FrameArgs)","FETCH_SIZE",3000.0\n'
                 '3,3,"void mi::(anonymous namespace)::k_frame<1, true, false>(mi::FrameArgs)","FETCH_SIZE",9.0\n'
                 '4,4,"mi::k_row_summary(void*)","FETCH_SIZE",5.0\n')
    assert T.parse_counter_csv(str(p), "k_frame<1,true,true>", "FETCH_SIZE") == [1000.0, 3000.0]
    assert T.parse_counter_csv(str(p), "k_frame<1,true,true>", "WRITE_SIZE") == []


def test_bench_modules_keep_the_oracle_out_of_the_timed_path():
    """Only benchlib/cpu_baseline.py may import the oracle (the checker timed as a baseline)."""
    for name in os.listdir(os.path.join(ROOT, "benchlib")):
        if name.endswith(".py") and name != "cpu_baseline.py":
            assert "oracle_lib" not in open(os.path.join(ROOT, "benchlib", name)).read(), name
    assert "oracle_lib" not in open(os.path.join(ROOT, "bench.py")).read()


def test_a_frame_of_several_launches_is_priced_per_frame():
    """benchlib.wl_tree.reprice_per_frame: `roofline` divides the frame's bytes by ONE launch's average -- four launches of 7.5 us read
    0.61 where the frame reaches 0.15."""
    from benchlib.wl_tree import reprice_per_frame
    per_launch = {"achieved": 4847.7, "frac": 0.606, "frac_algorithmic": 0.606, "avg_kernel_us": 7.476, "rocprof_frac": 0.57}
    frame = {"launches_per_frame": 4.0, "achieved": 1211.9, "frac": 0.1515, "kernels_us_per_frame": 29.9}
    r = reprice_per_frame(per_launch, frame)
    assert r["frac"] == 0.1515 and r["frac_algorithmic"] == 0.1515 and r["achieved"] == 1211.9 and r["avg_kernel_us"] == 29.9
    assert r["per_launch_avg_kernel_us"] == 7.476 and r["launches_per_frame"] == 4.0 and "rocprof_frac" not in r
    assert reprice_per_frame(per_launch, {"launches_per_frame": 1.0}) == per_launch and per_launch["frac"] == 0.606  # (one launch: untouched)
