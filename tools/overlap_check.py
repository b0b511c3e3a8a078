#!/usr/bin/env python3
"""Do kernels of the two streams overlap?  usage: overlap_check.py <kernel_trace.csv> <nameA> <nameB>
Prints, for the last launches of A, the B launches whose [start, end) intersect it."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
A = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if sys.argv[2] in r["Kernel_Name"]]
B = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows if sys.argv[3] in r["Kernel_Name"]]
for a0, a1 in A[-6:]:
    hits = [(max(a0, b0), min(a1, b1), n) for b0, b1, n in B if b0 < a1 and b1 > a0]
    print(f"A [{(a1 - a0) / 1e3:.1f} us]:", ", ".join(f"{n} overlaps {(e - s) / 1e3:.1f} us" for s, e, n in hits) or "no overlap")
