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


@pytest.mark.gpu
def test_reference_system_tests_against_the_host_layer():
    mi_build.build()
    exe = mi_build.build_host_tests()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    failed = [line for line in res.stdout.splitlines() if line.startswith("FAILED") or "EXCEPTION" in line]
    assert res.returncode == 0 and not failed, res.stdout[-3000:] + res.stderr[-1000:]
    assert "27 tests, 0 failed" in res.stdout  # 14 tests against the three systems, 13 of them again against the fused frame
