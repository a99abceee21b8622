"""Build libpww_hip.so (gfx950 code objects + C ABI) in-tree with hipcc.

No torch headers are involved: the library is plain HIP behind the C ABI of include/pww_hip.h, and
is loaded from Python with ctypes (pww_hip/_lib.py). hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "pww_hip", "libpww_hip.so")
SOURCES = ["pww_api.hip", "pww_attn.hip", "pww_cross.hip", "pww_cross_lean.hip", "pww_cross_out.hip", "pww_reduce.hip", "pww_mask.hip", "pww_qproj.hip", "pww_norm.hip", "pww_blocks.hip"]
HEADERS = ["pww_common.h", "pww_tile.h", "pww_attn_core.h", "pww_cross_tile.h", os.path.join(REPO, "include", "pww_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
         "-Wno-unused-function"]


# pww_mask.hip restates fp32 formulas that must match the CPU oracle bit for bit: no FMA contraction.
PER_FILE_FLAGS = {"pww_mask.hip": ["-ffp-contract=off"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [__file__]
    if not force and not _newer(LIB, deps):
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:  # compile translation units in parallel
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


def build_native_check(force=False):
    """tests/native/attn_check: C++ harness that drives the C ABI directly (test infrastructure)."""
    src = os.path.join(REPO, "tests", "native", "attn_check.cpp")
    exe = os.path.join(REPO, "tests", "native", "attn_check")
    lib = build_lib()
    if not force and not _newer(exe, [src, lib, os.path.join(REPO, "include", "pww_hip.h")]):
        return exe
    libdir = os.path.dirname(lib)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", exe, "-L" + libdir, "-lpww_hip",
           "-Wl,-rpath,$ORIGIN/../../paint-with-words-sd_amd/pww_hip"]
    subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_lib(force=force, verbose=True))
    print(build_native_check(force=force))
