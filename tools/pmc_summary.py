#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import csv, sys, collections, glob
for path in sys.argv[2:]:
    for f in glob.glob(path):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
