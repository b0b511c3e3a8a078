#!/usr/bin/env python3
"""Per-kernel totals over the steady-state tail of a rocprofv3 kernel_trace.csv.

usage: trace_summary.py <kernel_trace.csv> <anchor> <per_step> [steps=3] [top=45]

The window is [start of the (steps*per_step+1)-th from last launch whose name contains `anchor`,
start of the last such launch): a whole number of steps at the same phase, after warm-up and
library auto-tuning.  Times are printed per step."""
import csv
import sys
from collections import defaultdict

path, anchor, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
top = int(sys.argv[5]) if len(sys.argv) > 5 else 45
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if anchor in r["Kernel_Name"]]
lo, hi = marks[-1 - steps * per_step], marks[-1]
tot, cnt = defaultdict(float), defaultdict(int)
for r in rows:
    if lo <= int(r["Start_Timestamp"]) < hi:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[r["Kernel_Name"]] += d
        cnt[r["Kernel_Name"]] += 1
busy = sum(tot.values())
print(f"{steps} steps: wall {(hi - lo) / 1e6 / steps:.2f} ms/step, busy {busy / 1e3 / steps:.2f} ms/step, "
      f"{sum(cnt.values()) // steps} launches/step")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{v / 1e3 / steps:8.3f} ms/step {100 * v / busy:5.1f}%  n/step={cnt[k] / steps:6.1f}  avg {v / cnt[k]:8.1f} us  {k[:120]}")
