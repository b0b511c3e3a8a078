#!/usr/bin/env python3
"""Hot-path step (bench.HotPath) as eager launches vs one HIP-graph replay per step."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "pemsd7"
hp = bench.HotPath(bench.WORKLOADS[WL], dev, 0)
sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
bench.settle(hp.step, dev)
for k, w in ((20, 5), (50, 10)):
    wall, per = bench.timed_steps(hp.step, k, w, dev, sync)
    print(f"eager  K={k}: wall {wall / k * 1e3:.4f} ms  median {statistics.median(per):.4f}")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        hp.step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    hp.step()
for k, w in ((20, 5), (50, 10)):
    wall, per = bench.timed_steps(g.replay, k, w, dev, sync)
    print(f"graph  K={k}: wall {wall / k * 1e3:.4f} ms  median {statistics.median(per):.4f}")
