#!/usr/bin/env python3
"""Average PMC counter values per dispatch of the kernels whose name starts with argv[2], from a rocprofv3 counter_collection.csv."""
import csv
import sys
from collections import defaultdict

acc, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith(sys.argv[2]):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k] / n[k]:16.1f}   ({n[k]} dispatches)")
