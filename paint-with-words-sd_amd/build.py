"""Build libpww_hip.so (gfx950 code objects + C ABI) in-tree with hipcc.

No torch headers are involved: the library is plain HIP behind the C ABI of include/pww_hip.h, and
is loaded from Python with ctypes (pww_hip/_lib.py). hipcc cross-compiles without a GPU.

Two libraries come out of the same sources:
  libpww_hip.so               the product: what the default routes and the documented switches call. `build_lib()`, and all that
                              __graft_entry__.build() compiles.
  libpww_hip_experiments.so   the same sources with -DPWW_EXPERIMENTS=1 (+ pww_cross_out.hip): the product plus the forms that were built,
                              measured and not made a default (include/pww_hip.h, section "experiments"). Built by the tests / tools that
                              need it (`build_experiments()`, `python build.py --experiments`), never loaded by the product.
The large kernel families are instantiated in slices (one translation unit per storage type, the general cross-attention kernel also per
workgroup width) so that the compile runs side by side on the build box's cores.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "pww_hip", "libpww_hip.so")
LIB_EXPERIMENTS = os.path.join(HERE, "pww_hip", "libpww_hip_experiments.so")
# (source, extra defines, object suffix): the instantiation units are compiled once per slice
UNITS = [("pww_api.hip", [], ""), ("pww_attn.hip", [], ""), ("pww_cross.hip", [], ""), ("pww_cross_lean.hip", [], ""), ("pww_reduce.hip", [], ""),
         ("pww_mask.hip", [], ""), ("pww_qproj.hip", [], ""), ("pww_norm.hip", [], ""), ("pww_blocks.hip", [], ""),
         ("pww_attn_inst.hip", ["-DPWW_INST_F16"], ".f16"), ("pww_attn_inst.hip", ["-DPWW_INST_BF16"], ".bf16"),
         ("pww_cross_inst.hip", ["-DPWW_INST_F16", "-DPWW_INST_NW=2"], ".f16.nw2"), ("pww_cross_inst.hip", ["-DPWW_INST_F16", "-DPWW_INST_NW=4"], ".f16.nw4"),
         ("pww_cross_inst.hip", ["-DPWW_INST_BF16", "-DPWW_INST_NW=2"], ".bf16.nw2"), ("pww_cross_inst.hip", ["-DPWW_INST_BF16", "-DPWW_INST_NW=4"], ".bf16.nw4")]
EXPERIMENT_UNITS = [("pww_cross_out.hip", [], "")]      # sources only the experiments library has
SOURCES = sorted({u[0] for u in UNITS + EXPERIMENT_UNITS})
HEADERS = ["pww_common.h", "pww_tile.h", "pww_attn_core.h", "pww_attn_kernel.h", "pww_cross_tile.h", "pww_cross_kernel.h", os.path.join(REPO, "include", "pww_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count=16: the leading SCALAR arguments of a kernel (<= 14 dwords: 16 user SGPRs less the kernarg pointer) are preloaded
# into SGPRs by the dispatcher instead of fetched by the wave's first s_load (gfx950 supports kernarg preload; the compiler keeps a compatibility
# prologue for firmware that does not). Kernels that take one by-value struct are unaffected; the small kernels pass their hot arguments first.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


# pww_mask.hip restates fp32 formulas that must match the CPU oracle bit for bit: no FMA contraction.
PER_FILE_FLAGS = {"pww_mask.hip": ["-ffp-contract=off"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    return ([os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [__file__])


def _build(lib, units, defines, objdir, verbose, jobs=None):
    os.makedirs(objdir, exist_ok=True)
    # longest compiles first (the general cross-attention slices), at most `jobs` at a time
    order = sorted(units, key=lambda u: {"pww_cross_inst.hip": 0, "pww_attn_inst.hip": 1, "pww_cross_lean.hip": 2}.get(u[0], 3))
    jobs = jobs or max(2, min(len(order), (os.cpu_count() or 4)))
    pending, running, objs = list(order), [], []
    while pending or running:
        while pending and len(running) < jobs:
            src, defs, suffix = pending.pop(0)
            o = os.path.join(objdir, src + suffix + ".o")
            objs.append(o)
            cmd = [HIPCC] + FLAGS + defines + defs + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", o]
            if verbose:
                print(" ".join(cmd))
            running.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        cmd, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    # (-s: the host-side symbol table goes -- 100 KB; the exported entry points live in .dynsym, kernel names in the embedded code objects)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-s"] + sorted(objs) + ["-o", lib], check=True)
    return lib


def build_lib(force=False, verbose=False):
    """The product library (what __graft_entry__.build() compiles)."""
    if not force and not _newer(LIB, _deps()):
        return LIB
    return _build(LIB, UNITS, [], os.path.join(HERE, "build"), verbose)


def build_experiments(force=False, verbose=False):
    """libpww_hip_experiments.so: the product + the measured-and-not-default forms (tests and tools only)."""
    if not force and not _newer(LIB_EXPERIMENTS, _deps()):
        return LIB_EXPERIMENTS
    return _build(LIB_EXPERIMENTS, UNITS + EXPERIMENT_UNITS, ["-DPWW_EXPERIMENTS=1"], os.path.join(HERE, "build", "experiments"), verbose)


def _build_check(exe, lib, libname, defines, force):
    src = os.path.join(REPO, "tests", "native", "attn_check.cpp")
    if not force and not _newer(exe, [src, lib, os.path.join(REPO, "include", "pww_hip.h")]):
        return exe
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17"] + defines + [src, "-o", exe, "-L" + os.path.dirname(lib), "-l" + libname,
                                                                              "-Wl,-rpath,$ORIGIN/../../paint-with-words-sd_amd/pww_hip"]
    subprocess.run(cmd, check=True)
    return exe


def build_native_check(force=False):
    """tests/native/attn_check: C++ harness that drives the C ABI of the PRODUCT library directly (test infrastructure)."""
    return _build_check(os.path.join(REPO, "tests", "native", "attn_check"), build_lib(), "pww_hip", [], force)


def build_native_check_experiments(force=False):
    """tests/native/attn_check_experiments: the same harness over libpww_hip_experiments.so, with the cases of the moved entry points."""
    return _build_check(os.path.join(REPO, "tests", "native", "attn_check_experiments"), build_experiments(), "pww_hip_experiments", ["-DPWW_EXPERIMENTS=1"], force)


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_lib(force=force, verbose=True))
    print(build_native_check(force=force))
    if "--experiments" in sys.argv:
        print(build_experiments(force=force, verbose=True))
        print(build_native_check_experiments(force=force))
