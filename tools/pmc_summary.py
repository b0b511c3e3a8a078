#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: both count kilobytes; on gfx950 FETCH_SIZE tallies 32-B units for
the coalesced 16-B-per-lane streams and must be doubled, WRITE_SIZE and 64-B request streams are exact).

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [exact_fetch_kernel_substring ...]"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
exact = sys.argv[4:]
out = {}
for k in sorted(fetch):
    if "msgat::" not in k:
        continue
    name = k.split("(")[0].replace("void ", "")
    f = 1 if any(e in k for e in exact) else 2
    out[name] = {"fetch_kb_raw": round(fetch[k], 1), "write_kb_raw": round(write.get(k, 0.0), 1), "fetch_factor": f,
                 "hbm_bytes_per_launch": int((fetch[k] * f + write.get(k, 0.0)) * 1024)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python3 tools/kbench.py --eager --reps 5 "
                     "(pemsd7 workload: G=96, N=883, T=12, C=72, Co=24)",
           "correction": "FETCH_SIZE x2 for 16-B/lane coalesced streaming reads (gfx950); x1 for kernels listed as exact "
                         "(64-B row-piece requests); WRITE_SIZE x1", "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
