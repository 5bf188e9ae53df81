#!/usr/bin/env python3
"""Per-kernel time of the prefill passes in a rocprofv3 kernel trace of bench.py (rocpd database): the launches between the first
chunk mat-mul and the first single-token mat-vec = the bench's first + warm prefill pass.  usage: prefill_breakdown.py results.db"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select name, start, end from kernels"), key=lambda r: r[1])
i0 = next(i for i, r in enumerate(rows) if "gemm4k_kernel" in r[0] or "gemm8m" in r[0])
i1 = next(i for i, r in enumerate(rows) if i > i0 and ("gemv4_kernel" in r[0] or "gemvb_kernel" in r[0] or "gemvk_kernel" in r[0]))
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0][:56]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows[i0:i1]:
    a = agg[short(n)]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"prefill passes before the first decode step: {i1 - i0} launches, kernel time {tot / 1e3:.1f} ms, first start -> last end {(rows[i1 - 1][2] - rows[i0][1]) / 1e6:.1f} ms")
print(f"{'kernel':58s} {'calls':>6s} {'ms':>9s} {'%':>6s} {'avg us':>8s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:58s} {v[0]:6d} {v[1] / 1e3:9.2f} {100 * v[1] / tot:6.1f} {v[1] / v[0]:8.1f}")
