"""The C++ host layer above the C ABI (bevy_amd/host/bevy_mi355x_host.hpp) and the reference's own system tests
restated against it (tests/cpp/host_systems_test.cpp).
CPU: the test program compiles and links against the library.  GPU: it runs and every test passes."""
import os
import subprocess

import pytest

from bevy_amd import build as mi_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_compiles_and_links():
    mi_build.build()
    exe = mi_build.build_host_tests(force=True)
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


def test_task_pool_runs_every_chunk_exactly_once(tmp_path):
    """The host layer's TaskPool (gather / write-back loops in chunks on a few threads), standalone: tests/cpp/task_pool_test.cpp, plain and
    -- where the compiler has it -- under ThreadSanitizer, which is what found that a worker leaving work() late could draw a chunk of
    the NEXT job from counters that were still being set up (the job then ran that chunk twice)."""
    src = os.path.join(ROOT, "tests", "cpp", "task_pool_test.cpp")
    exe = str(tmp_path / "task_pool_test")
    res = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", src, "-o", exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    res = subprocess.run([exe, "100000"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "task pool: ok" in res.stdout, res.stdout[-1000:]
    tsan = str(tmp_path / "task_pool_tsan")
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread", src, "-o", tsan], capture_output=True, text=True)
    if res.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime here")
    res = subprocess.run([tsan, "1500"], capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in res.stderr:  # (the sanitizer itself could not start here: address space layout, ptrace limits ...)
        pytest.skip("ThreadSanitizer cannot run in this environment")
    assert res.returncode == 0 and "task pool: ok" in res.stdout and "ThreadSanitizer" not in res.stderr, (res.stdout + res.stderr)[-3000:]


@pytest.mark.gpu
def test_reference_system_tests_against_the_host_layer():
    mi_build.build()
    exe = mi_build.build_host_tests()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    failed = [line for line in res.stdout.splitlines() if line.startswith("FAILED") or "EXCEPTION" in line]
    assert res.returncode == 0 and not failed, res.stdout[-3000:] + res.stderr[-1000:]
    assert "36 tests, 0 failed" in res.stdout  # 20 tests against the three systems (four of them drive both forms / both plugins themselves), 16 of them again against the fused frame


def test_host_visibility_test_compiles_and_links():
    """tests/cpp/host_visibility_test.cpp (VisibilityRange + shadow views behind the host layer, against the oracle)."""
    import oracle_lib
    mi_build.build()
    oracle_lib.build()
    import cpp_build
    exe = cpp_build.build_host_visibility_test(force=True)
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_visibility_ranges_and_shadow_views_behind_the_host_layer():
    """VisibilityRange (visibility/range.rs:159-284, mod.rs:814-820) and check_dir_light_mesh_visibility / check_point_light_mesh_visibility
    (bevy_light/src/lib.rs:342-757) through the C++ plugin in both boundary forms: known answers, then random Worlds over four frames
    against the oracle (lists of every cascade / cube face / spot light, ViewVisibility bytes, change ticks)."""
    import oracle_lib
    mi_build.build()
    oracle_lib.build()
    import cpp_build
    exe = cpp_build.build_host_visibility_test()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout)
    failed = [line for line in res.stdout.splitlines() if line.startswith("FAILED") or "EXCEPTION" in line]
    assert res.returncode == 0 and not failed, res.stdout[-3000:] + res.stderr[-1000:]
    assert "6 tests, 0 failed" in res.stdout


def test_single_process_multi_gpu_driver_compiles():
    mi_build.build()
    exe = mi_build.build_multi_gpu_test(force=True)
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_one_thread_drives_every_gpu_of_the_node():
    """tests/cpp/multi_gpu_single_process.cpp: a context per device, ncclCommInitAll, MI_EXCHANGE_GROUPED -- the sharded frame of
    configs[3] from ONE thread of ONE process (what a Bevy App is); every rank's gathered buffer must equal the masks of one
    unsharded context.  On a one-GPU box that is a 1-rank communicator: everything but the wire."""
    import json
    mi_build.build()
    exe = mi_build.build_multi_gpu_test()
    res = subprocess.run([exe, "300000", "4"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-1000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["mismatches"] == 0 and out["devices"] >= 1 and out["frames"] == 4
