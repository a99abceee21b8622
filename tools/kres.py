#!/usr/bin/env python
"""Per-kernel register / spill / LDS summary of one .hip translation unit (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
    name = re.sub(r"\(pww::\w+\)$", "", name).replace("pww::", "")
    print("%-70s vgpr %-4s agpr %-4s sgpr %-4s spillV %-4s scratch %-6s occ %-3s lds %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"),
          r.get("VGPR Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
