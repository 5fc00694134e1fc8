"""Tracing hooks (the reference wraps these systems in info_span!, crates/bevy_transform/src/systems.rs:169-283):
every entry point of libbevy_mi355x.so is a roctx range named after itself when MI_ROCTX=1 (or under rocprofv3 --marker-trace).  The
library looks roctxRangePushA / roctxRangePop up at run time, so a stand-in that logs the calls is enough to see the ranges: pushes
and pops balance, nest (an entry point that calls another), and carry the entry points' names."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = r"""
#include <stdio.h>
#include <stdlib.h>
static FILE* f;
static int depth;
static void open_log(void) { if (!f) f = fopen(getenv("MI_TEST_ROCTX_LOG"), "a"); }
int roctxRangePushA(const char* name) { open_log(); fprintf(f, "push %d %s\n", depth, name); fflush(f); return depth++; }
int roctxRangePop(void) { open_log(); --depth; fprintf(f, "pop %d\n", depth); fflush(f); return depth; }
"""

SCRIPT = """
import numpy as np, sys
sys.path.insert(0, %r)
import bevy_amd as B
from bevy_amd import api, workloads as W
ctx = api.Context(0)
sc = W.many_cubes(5000)
cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
fr = api.compute_frustum(cfv, W.many_cubes_camera(0), W.CAMERA_FAR)
ctx.resize(sc["n"])
ctx.upload_transforms(sc["translation"], sc["rotation"], sc["scale"])
ctx.upload_bounds(sc["aabb_center"], sc["aabb_half"], sc["flags"], sc["layers"])
ctx.propagate_and_cull(fr)
ctx.download_visibility(0)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("on", [True, False])
def test_entry_points_are_roctx_ranges(tmp_path, on):
    src = tmp_path / "fake_roctx.c"
    src.write_text(FAKE)
    lib = tmp_path / "libfake_roctx.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", str(src), "-o", str(lib)], check=True)
    log = tmp_path / "roctx.log"
    log.write_text("")
    env = dict(os.environ, LD_PRELOAD=str(lib), MI_TEST_ROCTX_LOG=str(log))
    env.pop("ROCPROF_MARKER_API_TRACE", None)
    if on:
        env["MI_ROCTX"] = "1"
    else:
        env.pop("MI_ROCTX", None)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(SCRIPT % ROOT)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = log.read_text().splitlines()
    if not on:
        assert lines == [], "ranges without MI_ROCTX / a profiler's marker tracing"
        return
    pushes = [l.split(" ", 2) for l in lines if l.startswith("push")]
    pops = [l for l in lines if l.startswith("pop")]
    assert len(pushes) == len(pops) and len(pushes) >= 5
    names = {p[2] for p in pushes}
    for want in ("mi_columns_resize", "mi_upload_transforms", "mi_upload_bounds", "mi_propagate_and_cull", "mi_download_visibility"):
        assert any(n.startswith(want) for n in names), (want, sorted(names))
    depth = 0
    for l in lines:  # well nested, never below zero
        depth += 1 if l.startswith("push") else -1
        assert depth >= 0
    assert depth == 0
