#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd (.db) kernel trace: scripts/rocpd_stats.py <dir-or-db> [steps-marker-kernel [calls-per-step]]."""
import glob
import os
import sqlite3
import sys


def main():
    src = sys.argv[1]
    f = src if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0]
    marker = sys.argv[2] if len(sys.argv) > 2 else None
    per = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    c = sqlite3.connect(f)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
    steps = 1.0
    if marker:
        steps = sum(r[1] for r in rows if marker in r[0]) / per
    tot, n = sum(r[2] for r in rows), sum(r[1] for r in rows)
    print(f"steps {steps:g}  kernel time {tot / 1e6 / steps:.3f} ms/step  launches {n / steps:.1f}/step")
    for r in rows[:int(os.environ.get('TOP', 30))]:
        print("%-64s %7.1f/step  avg %8.2f us  %7.3f ms/step" % (r[0][:64], r[1] / steps, r[3] / 1e3, r[2] / 1e6 / steps))
    if os.environ.get("GRIDS"):
        q = "select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), avg(end-start), min(end-start) from kernels where name like ? group by 1,2,3,4 order by 1, 5 desc"
        for r in c.execute(q, ("%" + os.environ["GRIDS"] + "%",)):
            print("%-40s grid %4d x %3d x %3d  %6.1f/step avg %7.2f us min %7.2f" % (r[0][:40], r[1], r[2], r[3], r[4] / steps, r[5] / 1e3, r[6] / 1e3))


def timeline(db, per, iters=10):
    """Average duration of each of the last `per`-kernel periods, in dispatch order (SEQ=<kernels per iteration>)."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, start, end from kernels order by start").fetchall()
    seq = rows[len(rows) - per * iters:]
    tot = 0.0
    for i in range(per):
        d = sum(seq[it * per + i][5] - seq[it * per + i][4] for it in range(iters)) / iters
        r = seq[i]
        print("%2d %-60s %4dx%3dx%3d %7.2f us" % (i, r[0][:60], r[1], r[2], r[3], d / 1e3))
        tot += d
    print("sum %.2f us" % (tot / 1e3))


if __name__ == "__main__":
    if os.environ.get("SEQ"):
        src = sys.argv[1]
        timeline(src if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0], int(os.environ["SEQ"]))
        sys.exit(0)
    main()
