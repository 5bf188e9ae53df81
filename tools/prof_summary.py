#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table: calls, total, avg, min, max, %.
usage: tools/prof_summary.py results.db [substring-filter]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone[^\]]*\])?$", "", n)
    return n[:110]


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute(f"select {namecol}, start, end from kernels"))
    if "--decode" in sys.argv:  # keep the greedy-decode window: from the first arg-max to the roofline replay's first empty launch
        sys.argv.remove("--decode")
        t0 = min((s for n, s, e in rows if "argmax_partial" in n), default=0)
        t1 = min((s for n, s, e in rows if "null_kernel" in n), default=1 << 62)
        rows = [r for r in rows if t0 <= r[1] < t1]
        nsteps = sum(1 for n, s, e in rows if "argmax_final" in n)
        span = (max(e for n, s, e in rows) - t0) / 1e6
        print(f"decode window: {nsteps} steps, {span:.3f} ms wall -> {span / max(nsteps, 1):.3f} ms/step (eager)")
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':110s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        print(f"{k:110s} {a[0]:8d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:6.2f}")
    print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
