#!/usr/bin/env python
"""Summarise kernels of a gfx950 assembly file (hipcc -S --cuda-device-only): instruction counts by class, code bytes, and the
wait structure around the MFMAs (how many MFMAs sit directly behind an `s_waitcnt lgkmcnt(0)`: an exposed LDS latency at one
wave per SIMD). Usage: isa_summary.py file.s [regex on the demangled name] [--dump]"""
import re
import subprocess
import sys


def kernels(path):
    text = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.M | re.S):
        yield m.group(1), m.group(2).splitlines()


def demangle(n):
    """c++filt of this toolchain does not know DF16_ / DF16b: decode `_ZN3pww<len><name>I<template args>E...` by hand"""
    m = re.match(r"_ZN3pww(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    base = n[m.end():m.end() + ln]
    rest = n[m.end() + ln:]
    args = []
    if rest.startswith("I"):
        for t in re.finditer(r"DF16_|DF16b|Li(\d+)E|Lb([01])E", rest.split("EEv")[0] + "E"):
            args.append("f16" if t.group(0) == "DF16_" else "bf16" if t.group(0) == "DF16b" else t.group(1) if t.group(1) is not None else ("true" if t.group(2) == "1" else "false"))
    return "%s<%s>" % (base, ", ".join(args))


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    dump = "--dump" in sys.argv
    for name, lines in kernels(path):
        dn = demangle(name)
        dn = re.sub(r"\((AttnParams|CrossParams|QprojParams)\)$", "", dn)
        if pat and not pat.search(dn):
            continue
        ins = [l.strip() for l in lines if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        cnt = {}
        for i in ins:
            op = i.split()[0]
            key = ("mfma" if "mfma" in op else "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write") else
                   "buffer_load" if op.startswith("buffer_load") else "global" if op.startswith("global_") else "waitcnt" if op == "s_waitcnt" else
                   "barrier" if op == "s_barrier" else "exp" if op.startswith("v_exp") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
            cnt[key] = cnt.get(key, 0) + 1
        # MFMAs directly gated by a full lgkm wait (no other mfma since that wait)
        gated, since = 0, None
        for i in ins:
            if i.startswith("s_waitcnt") and "lgkmcnt(0)" in i:
                since = 0
            elif "mfma" in i.split()[0]:
                if since == 0:
                    gated += 1
                since = 1 if since is not None else None
        print("%-90s insts %6d  ~%6.1f KB  %s  mfma-behind-lgkm0 %d" % (dn[:90], len(ins), len(ins) * 6.0 / 1024, " ".join("%s=%d" % kv for kv in sorted(cnt.items())), gated))
        if dump:
            print("\n".join(ins))


if __name__ == "__main__":
    main()
