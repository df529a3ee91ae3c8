#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV pair into a short markdown table
(kernel names truncated) for profiles/.  usage: summarize_rocprof.py <dir> <prefix> <out.md> [note]"""
import csv
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    d, prefix, out = sys.argv[1], sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    rows = list(csv.DictReader(open(f"{d}/{prefix}_kernel_stats.csv")))
    trace = list(csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")))
    res = {}
    for t in trace:
        res.setdefault(t["Kernel_Name"], (t["VGPR_Count"], t["Accum_VGPR_Count"], t["SGPR_Count"], t["LDS_Block_Size"],
                                          t["Workgroup_Size_X"], int(t["Grid_Size_X"]) // max(1, int(t["Workgroup_Size_X"])),
                                          t["Grid_Size_Y"], t["Grid_Size_Z"]))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({prefix})\n\n{note}\n\n")
        f.write("| kernel | calls | avg us | min us | max us | % | VGPR | AGPR | SGPR | LDS B | WG | grid (WGs x,y,z) |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:14]:
            v = res.get(r["Name"], ("?",) * 8)
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                    f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | {v[0]} | {v[1]} | {v[2]} | {v[3]} | {v[4]} | "
                    f"{v[5]},{v[6]},{v[7]} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
