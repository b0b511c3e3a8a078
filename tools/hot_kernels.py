#!/usr/bin/env python3
"""Per-kernel GPU time of the hot-path step (bench.HotPath) from torch's profiler, for A/B runs of another build:
    python tools/hot_kernels.py [--lib build/lab/x.so] [--filter substring]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--filter", default="")
ap.add_argument("--workload", default="pemsd7", help="pemsd7 | stress (bench.WORKLOADS)")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
from ms_gat_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS[a.workload], dev, 0)
for _ in range(max(a.steps // 2, 2)):
    hp.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(a.steps):
        hp.step()
    torch.cuda.synchronize()
tot = 0.0
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
    if e.device_time_total <= 0:
        continue
    tot += e.device_time_total / a.steps
    if a.filter in e.key:
        print(f"{e.device_time_total / a.steps:9.1f} us/step  n={e.count / a.steps:4.1f}  avg {e.device_time_total / max(e.count, 1):8.1f} us  {e.key[:90]}")
print(f"{a.lib or 'in-tree'}: busy {tot:.1f} us/step")
