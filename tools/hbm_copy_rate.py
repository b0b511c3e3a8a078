#!/usr/bin/env python3
"""Device-to-device copy rate versus footprint (torch copy_, HIP events): what a streaming kernel can expect
from HBM once its operands no longer fit the 256 MB infinity cache."""
import torch

dev = torch.device("cuda:0")
for mb in (64, 98, 196, 400, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x, y = torch.randn(n, device=dev), torch.empty(n, device=dev)
    for _ in range(3):
        y.copy_(x)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        y.copy_(x)
    b.record()
    b.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"copy of {mb:5d} MB (footprint {2 * mb} MB): {us:8.1f} us  {2 * mb * 1.048576 / us * 1e3 / 1e3:6.2f} TB/s read+write", flush=True)
