"""Builds libbevy_mi355x.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python -m bevy_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
-ffp-contract=off is REQUIRED: bit-exact parity with the reference's SSE2 arithmetic depends on
every multiply and add being rounded separately (see csrc/glam_math.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbevy_mi355x.so")
SOURCES = ["kernels_flat.hip", "kernels_tree.hip", "kernels_cluster.hip", "kernels_batch.hip", "kernels_sorted.hip", "kernels_cells.hip", "context.cpp", "ctx_hierarchy.cpp", "ctx_exchange.cpp",
           "ctx_batch.cpp", "ctx_cluster.cpp", "host_helpers.cpp"]
HEADERS = ["kernels.h", "ctx.h", "glam_math.h", "visibility_rule.h", "cluster_walk.h", "cluster_fill.h", "compact_fast.h", "strip_plan.h", os.path.join("..", "..", "include", "bevy_mi355x.h"), os.path.join("..", "..", "include", "bevy_mi355x_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
# Per-file flags.  -fno-slp-vectorize on the row kernels: the SLP vectorizer pairs their independent f32 multiplies / adds into
# v_pk_mul_f32 / v_pk_add_f32 (same IEEE results), and the operand pairs it has to assemble for them (v_pk_mov_b32 copies of
# values that stay live in their scalar form as well) cost 8 - 12 VGPRs per kernel: k_frame 70 -> 62 (7 -> 8 waves per SIMD),
# k_propagate_fans<false> 72 -> 60 (7 -> 8)
# (tools/kernel_resources.py; profiles/r03h/kernel_resources.md).  The batching kernels are integer code and keep it.
FILE_FLAGS = {"kernels_flat.hip": ["-fno-slp-vectorize"], "kernels_tree.hip": ["-fno-slp-vectorize"],
              "kernels_cluster.hip": ["-fno-slp-vectorize"]}
# (kernels_cells.hip: the build of the static cull order -- a radix sort from rocPRIM's headers and three small kernels; default flags)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


OBJ_DIR = os.path.join(HERE, "build")
CFLAGS = [f for f in FLAGS if f != "-shared"]


def _deps():
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.abspath(__file__)]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _deps()
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, variant=None, file_flags=None, defines=()):
    """One object per translation unit (only the stale ones are recompiled, all of them in parallel), then one link.
    variant: build bevy_amd/libbevy_mi355x_<variant>.so with `file_flags` (instead of FILE_FLAGS) and extra -D `defines` -- A/B
    experiments on the GPU box (the test harness loads it when MI_LIB_VARIANT=<variant>); the product is the default build."""
    lib, obj_dir, fflags = LIB, OBJ_DIR, FILE_FLAGS
    if variant:
        lib = os.path.join(HERE, f"libbevy_mi355x_{variant}.so")
        obj_dir = os.path.join(HERE, "build", variant)
        fflags = FILE_FLAGS if file_flags is None else file_flags
        force = True
    if not force and up_to_date():
        return LIB
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = max(os.path.getmtime(d) for d in _deps())
    procs, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_t):
            continue
        cmd = [hipcc()] + CFLAGS + fflags.get(src, []) + [f"-D{d}" for d in defines] + ["-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(out)
        elif verbose and out:
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed building libbevy_mi355x.so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libbevy_mi355x.so")
    return lib


HOST_TEST_SRC = os.path.join(HERE, "..", "tests", "cpp", "host_systems_test.cpp")
HOST_TEST_EXE = os.path.join(HERE, "..", "tests", "cpp", "host_systems_test")
HOST_HEADER = os.path.join(HERE, "host", "bevy_mi355x_host.hpp")


def build_host_tests(force=False):
    """Compiles tests/cpp/host_systems_test.cpp (the C++ host layer + the reference's system tests) with g++ and links
    it against libbevy_mi355x.so.  Host code only: -ffp-contract=off keeps the value constructors in glam order."""
    exe, src = os.path.abspath(HOST_TEST_EXE), os.path.abspath(HOST_TEST_SRC)
    deps = [src, HOST_HEADER, os.path.join(os.path.dirname(HOST_HEADER), "bevy_mi355x_sharded.hpp"), LIB, os.path.join(CSRC, "glam_math.h"),
            os.path.join(CSRC, "..", "..", "include", "bevy_mi355x.h")]
    if not force and os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
        return exe
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-Wall", src, "-o", exe, "-L", HERE,
           "-lbevy_mi355x", "-ldl", "-Wl,-rpath,$ORIGIN/../../bevy_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed building tests/cpp/host_systems_test")
    return exe


MULTI_GPU_SRC = os.path.join(HERE, "..", "tests", "cpp", "multi_gpu_single_process.cpp")
MULTI_GPU_EXE = os.path.join(HERE, "..", "tests", "cpp", "multi_gpu_single_process")


def build_multi_gpu_test(force=False):
    """tests/cpp/multi_gpu_single_process.cpp: one process, one thread, a context per GPU, RCCL through dlopen (host code only)."""
    exe, src = os.path.abspath(MULTI_GPU_EXE), os.path.abspath(MULTI_GPU_SRC)
    deps = [src, LIB, os.path.join(CSRC, "..", "..", "include", "bevy_mi355x.h")]
    if not force and os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
        return exe
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", exe, "-L", HERE, "-lbevy_mi355x",
           "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,$ORIGIN/../../bevy_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("g++ failed building tests/cpp/multi_gpu_single_process")
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
