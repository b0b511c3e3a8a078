#!/usr/bin/env python3
"""Where the HOST time of an eager msgat72 training step goes (bench.TrainStep through engine.Trainer):
    python tools/host_overhead_train.py [--R 3] [--steps 40]
Prints the host time to enqueue a step (no synchronisation inside the loop), the wall time, and a cProfile table of the
main thread (the autograd engine's device thread shows up as `run_backward`)."""
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
ap.add_argument("--R", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--small", action="store_true", help="N = 64, B = 2: the GPU work is negligible, so the wall time per step IS "
                "the host time to enqueue it (same launch sequence)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG4, R=a.R)
if a.small:
    cfg.update(N=64, E=60, B=2)
ts = bench.TrainStep(cfg, dev)
ts.run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
ts.run(a.steps)          # ends with one host read of the epoch's totals
wall = time.perf_counter() - t0
print(f"R={a.R}: wall {wall / a.steps * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
ts.run(a.steps)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
