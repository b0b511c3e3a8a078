#!/usr/bin/env python3
"""Separates the launches of the attention-aggregate kernel in a rocprofv3 kernel trace of `bench.py --no-baselines`:
the launches inside the timed steps, the roofline loop on four operand sets (cold operands: the figure `roofline.frac`
is computed from) and the loop on one operand set (cache-resident operands).

usage: roofline_trace_table.py <kernel_trace.csv> [kernel-substring=k_agg_ring] [loop-launches=44]

bench.py's roofline_object() runs 4 warm-up + 40 timed launches per loop, cold loop first; they are the last
2 x 44 launches of the kernel with the second-depth grid (the largest grid of that kernel in the trace)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_agg_ring"
per_loop = int(sys.argv[3]) if len(sys.argv) > 3 else 44
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # noqa: E731
grid = lambda r: "x".join(r[f"Grid_Size_{a}"] for a in "XYZ")  # noqa: E731
by_grid = defaultdict(list)
for r in rows:
    by_grid[grid(r)].append(r)
big = max(by_grid, key=lambda g: eval(g.replace("x", "*")))
main = by_grid[big]
cached, cold, in_step = main[-per_loop:], main[-2 * per_loop:-per_loop], main[:-2 * per_loop]
warm = per_loop - 40


def line(label, rs):
    d = [dur(r) for r in rs]
    if d:
        print(f"{label:58s} {len(d):4d} launches   avg {sum(d) / len(d):8.2f} us   min {min(d):8.2f}   max {max(d):8.2f}")


print(f"{name}: grid {big} work-items (second GACN depth), workgroup {main[0]['Workgroup_Size_X']}")
line("inside the steps (input just written by the projection)", in_step)
line("roofline loop, four operand sets (cold), 40 timed", cold[warm:])
line("roofline loop, one operand set (cache-resident), 40 timed", cached[warm:])
for g, rs in by_grid.items():
    if g != big:
        line(f"other grid {g}", rs)
