#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from <dir>/<prefix>_counter_collection.csv."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
want = sys.argv[2:] or None
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].replace("void ", "")[:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if want and not any(w in k for w in want):
        continue
    n = max(len(v) for v in cs.values())
    print(f"{k}  (dispatches {n})")
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {sum(v) / len(v):16.1f}")
