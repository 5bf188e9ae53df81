#!/usr/bin/env python3
"""Mean of every counter of a `rocprofv3 --pmc ... --output-format csv` pass per kernel (filtered by substrings).
usage: pmc_generic.py <counter_collection.csv> substring[,substring...]"""
import collections, csv, statistics, sys
pats = sys.argv[2].split(",") if len(sys.argv) > 2 else [""]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0] if "<" not in k else k[:k.index(">") + 1]
        if any(p in k for p in pats):
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print(f"{k[:70]:70s} launches {n:6d}  " + "  ".join(f"{c} {statistics.mean(v):.4g}" for c, v in sorted(cs.items())))
