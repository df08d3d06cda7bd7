#!/usr/bin/env python3
"""Averages rocprofv3 counter_collection.csv files per (kernel, counter): python tools/pmc_summary.py dir [filter]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        agg[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k[:48]:48s} {c:28s} n={len(v):3d} avg={sum(v) / len(v):18.1f}")
