#!/usr/bin/env python3
"""Print avg/min us of selected kernels from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "")
    if any(k in n for k in sys.argv[2:]):
        print(f"  {n[:46]:46s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
