"""HBM traffic of the dominant kernel, measured in THIS run: two short rocprofv3 passes of the same bench command in child
processes -- `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with --kernel-trace only (never a sys / hip / hsa trace domain next
to counters) -- and bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, the correction MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes; both counters are in KiB).

Any failure (no rocprofv3, a timeout, an unexpected CSV) returns None and the roofline record falls back to the committed
profiles/, labelled as replayed."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile
import time

from .common import ROOT


def _rocprof():
    for cand in (shutil.which("rocprofv3"), "/opt/rocm/bin/rocprofv3"):
        if cand and os.path.exists(cand):
            return cand
    return None


def kernel_matches(symbol, prefix):
    """rocprofv3 prints 'void mi::(anonymous namespace)::k_frame<1, true, true>(...)': compare without blanks / namespaces."""
    n = symbol.replace("void ", "").replace("(anonymous namespace)::", "").replace("mi::", "").replace(" ", "")
    return n.startswith(prefix.replace(" ", ""))


def parse_counter_csv(path, kernel_prefix, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter and kernel_matches(r.get("Kernel_Name", ""), kernel_prefix):
                vals.append(float(r["Counter_Value"]))
    return vals


def measure_live(bench_args, kernel_prefix, budget_s=150.0):
    """bench_args: the argv tail that selects the workload (e.g. ['--workload', 'frame']).  Returns
    {'hbm_bytes_per_launch', 'fetch_KiB', 'write_KiB', 'dispatches', 'source', 'seconds'} or None."""
    exe = _rocprof()
    if exe is None:
        return None
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", MI_BENCH_CHILD="1")
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.time() - t0)
            if left < 20.0:
                return None
            out_dir = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + list(bench_args) + \
                  ["--steps", "10", "--warmup", "2", "--blocks", "2", "--no-cpu-baseline", "--no-other-workloads", "--no-end-to-end",
                   "--no-live-traffic"]
            res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            if res.returncode != 0:
                return None
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in files:
                vals += parse_counter_csv(f, kernel_prefix, ctr)
            if not vals:
                return None
            got[ctr] = (sum(vals) / len(vals), len(vals))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"][0] * 1024.0, got["WRITE_SIZE"][0] * 1024.0
    return {"hbm_bytes_per_launch": int(2.0 * fetch + write), "fetch_KiB": round(got["FETCH_SIZE"][0], 1),
            "write_KiB": round(got["WRITE_SIZE"][0], 1), "dispatches": got["FETCH_SIZE"][1],
            "source": "live: 2 x FETCH_SIZE + WRITE_SIZE over two rocprofv3 --pmc passes of this command in this run",
            "seconds": round(time.time() - t0, 1)}
