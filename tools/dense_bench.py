#!/usr/bin/env python3
"""Times k_scores / k_bwd_dense_col alone (bench.time_dense_kernels) for a given build of the library:
    python tools/dense_bench.py [--lib build/lab/x.so] [--workload pemsd7|stress]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--workload", default="pemsd7")
a = ap.parse_args()
from ms_gat_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import bench  # noqa: E402

hp = bench.HotPath(bench.WORKLOADS[a.workload], torch.device("cuda:0"), seed=0)
print(a.lib or "in-tree", json.dumps({k: (v["us_per_launch"], v["frac"]) for k, v in bench.time_dense_kernels(hp, reps=40)["kernels"].items()}))
