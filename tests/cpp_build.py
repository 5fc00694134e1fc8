"""Build recipe of tests/cpp/host_visibility_test.cpp -- the one C++ test program that links the CPU oracle (oracle/libbevy_oracle.so,
the checker) next to the product library.  It lives under tests/ because nothing under bevy_amd/ may name the oracle
(tests/test_abi_and_host.py::test_product_does_not_import_oracle); host_systems_test, whose --bench mode bench.py runs, is built by
bevy_amd/build.py and does not link it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_visibility_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_visibility_test")


def build_host_visibility_test(force=False):
    """VisibilityRange and the shadow-view systems behind the C++ host layer, checked against the oracle.  Host code only:
    -ffp-contract=off keeps the value constructors in glam order."""
    lib_dir, oracle_dir = os.path.join(ROOT, "bevy_amd"), os.path.join(ROOT, "oracle")
    deps = [SRC, os.path.join(lib_dir, "host", "bevy_mi355x_host.hpp"), os.path.join(lib_dir, "libbevy_mi355x.so"),
            os.path.join(lib_dir, "csrc", "glam_math.h"), os.path.join(ROOT, "include", "bevy_mi355x.h"),
            os.path.join(oracle_dir, "libbevy_oracle.so"), os.path.join(oracle_dir, "bevy_oracle.h")]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-Wall", SRC, "-o", EXE, "-L", lib_dir, "-lbevy_mi355x",
           "-L", oracle_dir, "-lbevy_oracle", "-Wl,-rpath,$ORIGIN/../../bevy_amd", "-Wl,-rpath,$ORIGIN/../../oracle", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed building tests/cpp/host_visibility_test")
    return EXE
