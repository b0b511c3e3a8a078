#!/usr/bin/env python3
"""Prints the launches of ONE step (the last complete one) of a rocprofv3 kernel trace in launch order.

    python tools/trace_one_step.py <dir with *_kernel_trace.csv> --per-step 30 [--skip K]
    python tools/trace_one_step.py <dir with *_kernel_trace.csv> --anchor k_qonly     # a kernel launched once per step
"""
import argparse
import csv
import glob
import os
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--per-step", type=int, default=0, help="launches per step")
    ap.add_argument("--anchor", default="", help="name of a kernel launched once per step: the step between its last two launches")
    ap.add_argument("--skip", type=int, default=0, help="launches after the last whole step (baselines etc.)")
    a = ap.parse_args()
    f = max(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows = [r for r in rows if "msgat::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if a.anchor:
        marks = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
        step = rows[marks[-2]:marks[-1]]
    else:
        end = len(rows) - a.skip
        step = rows[end - a.per_step:end]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    tot = 0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void msgat::", "").replace("msgat::", "")
        grid = "x".join(str(int(r[k]) // max(1, int(r[k.replace("Grid", "Workgroup")]))) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:7.1f}  {name:44s} grid {grid} wg {r['Workgroup_Size_X']} lds {r.get('LDS_Block_Size', '?')} vgpr {r.get('VGPR_Count', '?')}")
        prev_end = e
        tot += e - s
    print(f"busy {tot / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
