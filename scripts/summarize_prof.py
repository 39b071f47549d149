#!/usr/bin/env python3
"""Summarise rocprofv3 csv output dirs (kernel stats + PMC means per kernel) into one text block.
usage: summarize_prof.py gpurun_out/prof_<tag>   (prefix; reads <prefix>_stats, _fetch, _l2, _sq, _sq2)"""
import collections
import csv
import glob
import os
import sys


def newest(pattern):  # gpurun_out/ keeps the csv files of earlier calls: take the most recent one per pass
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


prefix = sys.argv[1]
for f in newest(prefix + "_stats/*/*_kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    print(open(f).read().strip())
for suffix in ("fetch", "l2", "sq", "sq2"):
    for f in newest(f"{prefix}_{suffix}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        print(f"== PMC pass '{suffix}' (mean per dispatch)")
        for (k, c), v in sorted(agg.items()):
            print(f"{k:62s} {c:24s} n={len(v):3d} mean={sum(v) / len(v):.5g}")
