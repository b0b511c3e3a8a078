#!/usr/bin/env python3
"""Where the HOST time of an eager hot-path step goes (cProfile over bench.HotPath.step at a launch-bound size):
    python tools/host_overhead.py [--workload pemsd4] [--steps 300] [--dropin]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pemsd4")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--dropin", action="store_true", help="the un-stacked loop of GACN module calls (bench.DropInLoop)")
a = ap.parse_args()
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS[a.workload], dev, 0)
if a.dropin:
    hp = bench.DropInLoop(hp)
for _ in range(50):
    hp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    hp.step()
host = time.perf_counter() - t0          # time to ENQUEUE the steps (the GPU may lag behind)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{a.workload}: host enqueue {host / a.steps * 1e6:.1f} us/step, wall {wall / a.steps * 1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    hp.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
