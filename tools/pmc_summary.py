#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection.csv files: per kernel, mean counter value per dispatch.
usage: tools/pmc_summary.py <dir-with-counter_collection.csv> [...]"""
import csv, glob, os, sys
from collections import defaultdict

def short(name):
    name = name.replace("void ", "")
    if "radix" in name or "onesweep" in name: return "rocprim::radix_sort"
    return name.split("(")[0]

def main(dirs):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                a = agg[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    for k in sorted(agg):
        if not k.startswith("compvhip") and "rocprim" not in k: continue
        print(k)
        for c in sorted(agg[k]):
            tot, n = agg[k][c]
            print("   %-26s mean/dispatch %16.1f   (n=%d)" % (c, tot / n, n))

if __name__ == "__main__":
    main(sys.argv[1:])
