#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite output) kernel trace as text:  tools/rocprof_summary.py <results.db>
Prints per-kernel calls / total / average / share from the database's `top_kernels` view, plus VGPR/SGPR/LDS and grid
of the first dispatch of each kernel.  Used to produce the committed summaries under profiles/."""
import sqlite3
import sys


def short(name):
    if "segmented_radix_sort" in name:
        return "rocprim::segmented_radix_sort (line keys)"
    if "radix_sort" in name or "onesweep" in name:
        return "rocprim::radix_sort:" + name.split("detail::")[-1][:60]
    return name.split("(")[0].replace("void ", "")


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    info = {}
    for r in cur.execute("select name,grid_x,grid_y,grid_z,workgroup_x,lds_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels group by name"):
        info[r[0]] = r[1:]
    print("%-58s %6s %12s %12s %7s  %s" % ("kernel", "calls", "total_us", "avg_us", "%", "grid(x,y,z) wg lds vgpr agpr sgpr"))
    for name, calls, total, avg, pct in rows:
        i = info.get(name, ())
        print("%-58s %6d %12.1f %12.2f %7.2f  %s" % (short(name)[:58], calls, total, avg, pct, " ".join(str(x) for x in i)))


if __name__ == "__main__":
    main(sys.argv[1])
