#!/usr/bin/env python
"""Static checks of compiler-dependent properties the measured kernel times rest on (no GPU needed: hipcc cross-compiles gfx950).

Each of these was found by measurement in rounds 2 / 3 and can be undone silently by a source change or a compiler
update, with correct results and slower kernels:

 1. register budget: the hot kernels stay inside the occupancy they were tuned for, without scratch in their hot variants
    (hipcc -Rpass-analysis=kernel-resource-usage);
 2. cross_fused_kernel, several query blocks per workgroup: the Q loads are unconditional buffer loads (a load under an `if` is waited
    for at the join: the prefetch becomes synchronous), the pass-1 loop waits with a COUNTED vmcnt (the ring of three blocks in flight
    survives), and the LDS-direct bias copies are issued back to back (inline asm: no wait between two copies).

    python tools/check_kernel_invariants.py            # prints one line per check, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "paint-with-words-sd_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(REPO, "include")]


def resource_usage(src, extra=()):
    out = subprocess.run([HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = rows.setdefault(t.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def kernel_isa(src, symbol_regex, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC] + FLAGS + list(extra) + ["-S", "--cuda-device-only", src, "-o", asm], capture_output=True, text=True, check=True)
        text = open(asm).read()
    out = {}
    for m in re.finditer(r"^(%s):.*?s_endpgm" % symbol_regex, text, re.M | re.S):
        out[m.group(1)] = m.group(0).splitlines()
    return out


def main():
    bad = []

    def check(ok, what):
        print(("ok   " if ok else "FAIL ") + what)
        if not ok:
            bad.append(what)

    # ---- 1. register budgets -------------------------------------------------------------------------------------------------
    # (round 6: the kernel families are instantiated per storage type -- and the general cross kernel per workgroup width -- in
    # pww_attn_inst.hip / pww_cross_inst.hip; the hand-off form of section 2 is an experiments-library kernel: -DPWW_EXPERIMENTS=1)
    attn = {}
    for t in ("-DPWW_INST_F16", "-DPWW_INST_BF16"):
        attn.update(resource_usage(os.path.join(CSRC, "pww_attn_inst.hip"), [t]))
    for name, r in attn.items():
        if "attn_fwd_fold_kernel" in name and not name.endswith("ELi2EEEvNS_10AttnParamsE"):      # d = 40 folded kernels, 8- and 4-wave workgroups
            # (the dominant launch; the 2-wave variant serves under-filled launches at one wave per SIMD): two waves per SIMD, no scratch
            check(int(r["Occupancy [waves/SIMD]"]) >= 2 and int(r["ScratchSize [bytes/lane]"]) == 0,
                  "%s: %s VGPRs, occupancy %s, %s B scratch (>= 2 waves per SIMD, no scratch)" % (name, r["VGPRs"], r["Occupancy [waves/SIMD]"], r["ScratchSize [bytes/lane]"]))
    cross = {}
    for t in ("-DPWW_INST_F16", "-DPWW_INST_BF16"):
        cross.update(resource_usage(os.path.join(CSRC, "pww_cross_inst.hip"), [t, "-DPWW_INST_NW=4"]))
    for name, r in cross.items():
        m = re.match(r"_ZN3pww18cross_fused_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])EEE", name)
        if not m:
            continue
        ks, single, compact = int(m.group(2)), m.group(5) == "1", m.group(6) == "1"
        if ks <= 3 and not compact:              # the d = 40 dense kernels (SD1.5 at N = 4096: the batched launch of configs 3 / 4): no scratch
            check(int(r["ScratchSize [bytes/lane]"]) == 0 and int(r["Occupancy [waves/SIMD]"]) >= 2,
                  "%s (d <= 48, dense, %s): %s VGPRs, %s B scratch, occupancy %s" % (name, "one block" if single else "several blocks", r["VGPRs"],
                                                                                      r["ScratchSize [bytes/lane]"], r["Occupancy [waves/SIMD]"]))

    # ---- 2. the several-blocks-per-workgroup cross kernel (bf16, d = 40, dense) -----------------------------------------------
    isa = kernel_isa(os.path.join(CSRC, "pww_cross_inst.hip"), r"_ZN3pww18cross_fused_kernelIDF16bLi3ELi2ELi4ELb0ELb0EEEvNS_11CrossParamsE",
                     ["-DPWW_INST_BF16", "-DPWW_INST_NW=4", "-DPWW_EXPERIMENTS=1"])
    lines = next(iter(isa.values()))
    body = "\n".join(lines)
    # (round 4: the entry fold of the projection's fp64 partials is a small loop of two global 16-byte loads next to v_max_f64 / v_add_f64 --
    # the only global loads allowed)
    glob = [i for i, line in enumerate(lines) if line.strip().startswith(("global_load_dwordx4", "flat_load_dwordx4"))]
    in_fold = [i for i in glob if any("v_max_f64" in l or "v_add_f64" in l for l in lines[i:i + 12])]
    check(len(glob) == len(in_fold) <= 2, "Q fragments: no flat / global (conditional) 16-byte loads outside the partial fold (%d, %d of them in the fold), buffer loads only" % (len(glob), len(in_fold)))
    # LDS-direct copies come in runs (2 / 4 / 8 per block): no s_waitcnt between the copies of a run
    runs, cur, waits_inside = [], 0, 0
    pending_wait = 0
    for line in lines:
        s = line.strip()
        if s.startswith("buffer_load_dwordx4") and s.endswith("lds"):
            if cur:
                waits_inside += pending_wait
            cur += 1
            pending_wait = 0
        elif s.startswith("s_waitcnt") and "vmcnt" in s and cur:
            pending_wait += 1
        elif s.startswith("s_cbranch") or s.startswith(".LBB") or s.startswith("s_branch"):
            if cur:
                runs.append(cur)
            cur, pending_wait = 0, 0
    check(len(runs) >= 3 and waits_inside == 0, "LDS-direct bias copies: %d runs of %s copies, %d vmcnt waits between copies of a run" % (len(runs), sorted(set(runs)), waits_inside))
    # the pass-1 loop: the first inner loop that holds MFMAs and requests the ring's next block; its header must not drain vmcnt
    loop_heads = [i for i, line in enumerate(lines) if "Loop Header" in line] + [len(lines)]
    found = False
    for h, nxt_h in zip(loop_heads, loop_heads[1:]):
        loop = "\n".join(lines[h:nxt_h])
        if "v_mfma" in loop and "sc1" in loop and "v_exp_f32" not in loop:      # MFMAs and the published partials, no softmax: pass 1
            waits = [s.strip() for s in lines[h:h + 12] if "s_waitcnt" in s and "vmcnt" in s]
            found = True
            check(bool(waits) and all("vmcnt(0)" not in w for w in waits), "pass-1 loop header waits with a counted vmcnt (ring of 3 blocks in flight): %s" % waits)
            break
    check(found, "pass-1 loop located")
    # ---- 3. (round 5) the small cross-attention kernels: what their short chains rest on -------------------------------------------------
    lean_src = os.path.join(CSRC, "pww_cross_lean.hip")
    lean = resource_usage(lean_src)
    for name, r in lean.items():
        if "cross_lean_kernel" in name or "qk_parts_kernel" in name:
            check(int(r["ScratchSize [bytes/lane]"]) == 0, "%s: %s VGPRs, %s B scratch (no scratch)" % (name[:60], r["VGPRs"], r["ScratchSize [bytes/lane]"]))
    isa = kernel_isa(lean_src, r"_ZN3pww17cross_lean_kernelIDF16bLi10ELi5ELi4ELb0EEEvNS_10LeanParamsE")
    lines = [l.strip() for l in next(iter(isa.values()))]
    nbar = sum(1 for l in lines if l.startswith("s_barrier"))
    check(nbar == 1, "cross_lean_kernel (bf16, d = 160): %d barrier(s) in the kernel (one)" % nbar)
    # every buffer load of the prologue sits in front of the barrier and none of them is behind a waterfall loop (a descriptor built from a
    # value hipcc cannot prove uniform is loaded through v_readfirstlane + s_cbranch_execnz per load: 185 readfirstlanes in the first version)
    bar = next(i for i, l in enumerate(lines) if l.startswith("s_barrier"))
    loads = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dwordx4")]
    rfl = sum(1 for l in lines[:bar] if l.startswith("v_readfirstlane"))
    check(loads and max(loads) < bar and rfl <= 16, "cross_lean_kernel: %d buffer loads, all in front of the barrier; %d v_readfirstlane in the prologue (no waterfall loops)" % (len(loads), rfl))
    # the last prologue load is issued within ~600 instructions of the kernel's entry (1500 in the first version: per-chunk index arithmetic)
    check(max(loads) < 700, "cross_lean_kernel: last prologue load at instruction %d (< 700)" % max(loads))
    # the score MFMAs of a tile run behind counted LDS waits (pww_tile.h score_tile requests all K fragments first)
    mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    gated = sum(1 for i in mf if any(lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j] for j in range(max(0, i - 1), i)))
    check(gated * 4 <= len(mf), "cross_lean_kernel: %d of %d MFMAs directly behind a full LDS wait (at most a quarter)" % (gated, len(mf)))
    isa = kernel_isa(lean_src, r"_ZN3pww15qk_parts_kernelIDF16bLi10ELb0EEEv\w+")
    lines = [l.strip() for l in next(iter(isa.values()))]
    check(not any(l.startswith(("s_barrier", "ds_write", "ds_read")) for l in lines), "qk_parts_kernel (fine form): no barrier, no LDS traffic besides the shuffles' ds_bpermute")
    # ---- 4. (round 5) cross-attention + to_out in one launch: the two-phase form must stay inside the 256 architectural registers (past them
    # hipcc selects the accumulator-file form of every MFMA and copies each score / O tile to and from it: 1136 v_accvgpr moves and 40 us per
    # workgroup in the one-phase version), with no scratch and no waterfall loop around a buffer load
    out_src = os.path.join(CSRC, "pww_cross_out.hip")
    for name, r in resource_usage(out_src, ["-DPWW_EXPERIMENTS=1"]).items():
        if "cross_out_kernel" in name:
            check(int(r["ScratchSize [bytes/lane]"]) == 0 and int(r["VGPRs"]) <= 256, "%s: %s VGPRs (<= 256), %s B scratch (none)" % (name[:60], r["VGPRs"], r["ScratchSize [bytes/lane]"]))
    isa = kernel_isa(out_src, r"_ZN3pww16cross_out_kernelIDF16bLi3ELi10EEEvNS_9OutParamsE", ["-DPWW_EXPERIMENTS=1"])
    lines = [l.strip() for l in next(iter(isa.values()))]
    acc = sum(1 for l in lines if l.startswith("v_accvgpr"))
    wf = sum(1 for i, l in enumerate(lines) if l.startswith("buffer_load") and any(x.startswith("s_cbranch_execnz") for x in lines[i + 1:i + 4]))      # (load; s_xor exec; s_cbranch_execnz = a waterfall loop)
    check(acc == 0 and wf == 0, "cross_out_kernel (bf16, d = 40): %d accumulator-file moves, %d buffer loads inside a waterfall loop (none)" % (acc, wf))
    print("%d violation(s)" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
