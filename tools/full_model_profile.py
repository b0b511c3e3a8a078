#!/usr/bin/env python3
"""Runs a few full msgat72 training steps (HIP graph branch) so rocprofv3 can show what the eager
PyTorch ops around the hot path cost:  rocprofv3 --kernel-trace --stats -- python3 tools/full_model_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

wl = bench.WORKLOADS["pemsd7"]
dev = torch.device("cuda:0")
print("full model step ms:", bench.full_model_step_ms(wl, dev, dense=False, steps=4, warmup=4))
