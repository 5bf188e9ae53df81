#!/usr/bin/env python3
"""Kernel resource table: hipcc -Rpass-analysis=kernel-resource-usage for one csrc file, filtered by a name substring."""
import re, subprocess, sys
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Iinclude", *(["-fno-slp-vectorize"] if any(n in src for n in ("k_gemv4", "k_qkvattn", "k_gemvb", "k_attn", "k_gemm4k")) else []),
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"], capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    if pat in r["name"]:
        print(f'{r["name"][:90]:90s} vgpr {r.get("VGPRs"):>4s} agpr {r.get("AGPRs"):>3s} spill {r.get("VGPRs Spill"):>3s} scratch {r.get("ScratchSize [bytes/lane]"):>4s} occ {r.get("Occupancy [waves/SIMD]")} lds {r.get("LDS Size [bytes/block]")}')
