#!/usr/bin/env python3
"""Per-kernel GPU time of the hot path's FORWARD alone (no_grad: the inference variants of the kernels):
    python tools/forward_kernels.py [--workload pemsd4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pemsd4")
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS[a.workload], dev, 0)
for _ in range(20):
    hp.forward_only()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(a.steps):
        hp.forward_only()
    torch.cuda.synchronize()
tot = 0.0
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
    if e.device_time_total > 0:
        tot += e.device_time_total / a.steps
        print(f"{e.device_time_total / a.steps:8.1f} us  n={e.count / a.steps:3.1f}  {e.key[:100]}")
print(f"forward busy {tot:.1f} us")
