#!/usr/bin/env python3
"""Average duration of a kernel per distinct launch grid (rocprofv3 kernel_trace.csv): separates the launches of
one kernel that differ in size (k_agg_lds runs with Cu = 24 channels twice a step and with Cu = 1 once).
usage: trace_by_grid.py <kernel_trace.csv> <kernel-name-substring>"""
import csv
import sys
from collections import defaultdict

tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        grid = "x".join(r.get(f"Grid_Size_{a}", "?") for a in "XYZ")
        wg = "x".join(r.get(f"Workgroup_Size_{a}", "?") for a in "XYZ")
        key = (r["Kernel_Name"].split("(")[0], grid, wg)
        tot[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[key] += 1
for k in sorted(tot, key=lambda k: -cnt[k]):
    print(f"{k[0]}  grid {k[1]} (work-items), workgroup {k[2]}:  {cnt[k]} launches, avg {tot[k] / cnt[k]:.1f} us")
