#!/usr/bin/env python3
"""Per-kernel average duration by call order (rocprofv3 rocpd database): the calls of each kernel are split into `groups` equal
runs in start order -- with tools/g4k_exp.py (2048-token prompt in 128-token chunks) group i of the last pass is chunk i.
usage: prof_by_chunk.py results.db substring[,substring...] [calls_per_group=32] [last_n_groups=16] [--mod N]
--mod N: instead of groups, average the calls by (call index % N) (e.g. 3: the QKV / O / down launches of gemm4k_kernel<0>)"""
import sqlite3, sys
mod = 0
if "--mod" in sys.argv:
    i = sys.argv.index("--mod"); mod = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = sorted(cur.execute(f"select {namecol}, start, end from kernels"), key=lambda r: r[1])
per = int(sys.argv[3]) if len(sys.argv) > 3 else 32
last = int(sys.argv[4]) if len(sys.argv) > 4 else 16
for pat in sys.argv[2].split(","):
    d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
    if mod:
        print(f"{pat:28s} n={len(d):5d} avg us by call index % {mod}:", " ".join(f"{sum(d[k::mod]) / max(len(d[k::mod]), 1):7.1f}" for k in range(mod)))
        continue
    d = d[-per * last:]
    g = [sum(d[i:i + per]) / per for i in range(0, len(d), per)]
    print(f"{pat:28s} n={len(d):5d} avg us by group:", " ".join(f"{x:6.1f}" for x in g))
