#!/usr/bin/env python3
"""One whole msgat72 TRAINING step (engine.Trainer: forward, fused loss + metrics, backward, Adam) at the sizes of the five
datasets of the reference's registry (data/meta.yaml) with the reference's default flags (main.py:28-29: five components,
batch 64), launched eagerly and replayed as a HIP graph (`Trainer(hip_graph=True)`).  Synthetic data and graphs (E = N
undirected edges + self loops: the PeMS graphs have ~1 edge per node); GPU box only:
    python tools/datasets_sweep.py [--steps 12]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

DATASETS = {"pemsd3": (358, 1), "pemsd4": (307, 3), "pemsd7": (883, 1), "pemsd8": (170, 3), "pemsd-bay": (325, 1)}  # meta.yaml

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--R", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
print(f"# msgat72 training step, R = {a.R} components, B = {a.B}, T = 12; median of {a.steps} HIP-event-timed steps (ms) and samples/s")
print(f"{'dataset':10s} {'N':>5s} {'C':>2s} {'eager ms':>9s} {'graph ms':>9s} {'eager/graph':>11s} {'samples/s (best)':>17s}")
for name, (N, C) in DATASETS.items():
    cfg = dict(N=N, E=N, B=a.B, R=a.R, Cin=C, T=12)
    row = []
    for graph in (False, True):
        ts = bench.TrainStep(cfg, dev, hip_graph=graph)
        ts.run(6)          # warm-up epoch (allocator, graph capture)
        ts.run(6)
        _, per = bench.time_train_step(ts, a.steps, 3, sync)
        row.append(statistics.median(per))
        del ts
        torch.cuda.empty_cache()
    best = min(row)
    print(f"{name:10s} {N:5d} {C:2d} {row[0]:9.3f} {row[1]:9.3f} {row[0] / row[1]:11.2f} {a.B / best * 1e3:17.0f}", flush=True)
