#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 PMC counters (counter_collection.csv), averaged per launch.
usage: pmc_kernel_table.py <counter_collection.csv> [kernel-name-substring ...]"""
import csv
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if sys.argv[2:] and not any(s in k for s in sys.argv[2:]):
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    n = len(launches[k])
    print(f"{k}  ({n} launches; per launch)")
    for c, v in sorted(tot[k].items()):
        print(f"    {c:28s} {v / n:16.0f}")
