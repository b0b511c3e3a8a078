#!/usr/bin/env python3
"""Runs a few whole msgat72 training steps through engine.Trainer (bench.TrainStep) so rocprofv3 can show what a step
costs kernel by kernel:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/full_model_profile.py [--R 5] [--steps 6] [--graph]
then  python3 tools/trace_summary.py out/*/*kernel_trace.csv k_adam 1 3"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=5)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--lib", default="", help="another build of libmsgat_hip.so (A/B runs)")
a = ap.parse_args()
if a.lib:
    from ms_gat_amd import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = torch.device("cuda:0")
ts = bench.TrainStep(dict(bench.CFG4, R=a.R), dev, hip_graph=a.graph)
wall, per = bench.time_train_step(ts, a.steps, 4, lambda: torch.cuda.synchronize(dev))
print(f"R={a.R} graph={a.graph}: {wall / a.steps * 1e3:.3f} ms/step wall, median {sorted(per)[len(per) // 2]:.3f} ms by HIP events")
