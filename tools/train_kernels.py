#!/usr/bin/env python3
"""Per-kernel GPU time of the msgat72 TRAINING step (bench.TrainStep, engine.Trainer) from torch's profiler:
    python tools/train_kernels.py [--R 3] [--unstacked] [--hidden 72] [--lib build/lab/x.so] [--top 40]
--unstacked evaluates the components one by one (the reference's loop, msgat.py:204) instead of the stacked schedule."""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=3)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--unstacked", action="store_true")
ap.add_argument("--hidden", type=int, default=72, help="48 | 72 | 96: the model of the registry (msgat.py:220-229)")
ap.add_argument("--lib", default="")
ap.add_argument("--ops", action="store_true", help="list the torch operators (with input shapes) that still launch kernels")
a = ap.parse_args()
from ms_gat_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
ts = bench.TrainStep(dict(bench.CFG4, R=a.R, hidden=a.hidden), dev, stacked=not a.unstacked)
ts.run(4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    ts.run(a.steps)
    torch.cuda.synchronize()
if a.ops:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
        ts.run(a.steps)
        torch.cuda.synchronize()
    print("torch operators that launch kernels (self device time; per step) and the package line that issued them:")
    evs = [e for e in prof2.key_averages(group_by_input_shape=True, group_by_stack_n=30)
           if e.self_device_time_total > 0 and e.key.startswith("aten::")]
    for e in sorted(evs, key=lambda e: -e.self_device_time_total):
        where = [f for f in (e.stack or []) if "ms_gat_amd" in f or "bench.py" in f]
        where = re.sub(r".*/(ms_gat_amd/|bench)", r"\1", where[0]) if where else "(autograd engine)"
        print(f"{e.self_device_time_total / a.steps:9.1f} us/step  n={e.count / a.steps:5.1f}  {e.key:26s} {str(e.input_shapes)[:60]:60s} {where[:70]}")
rows = [e for e in prof.key_averages() if e.device_time_total > 0]
tot = sum(e.device_time_total for e in rows) / a.steps
n = sum(e.count for e in rows) / a.steps
print(f"R={a.R} {'un-stacked loop' if a.unstacked else 'stacked'}: busy {tot / 1e3:.3f} ms/step, {n:.0f} launches/step")
for e in sorted(rows, key=lambda e: -e.device_time_total)[: a.top]:
    name = re.sub(r"\(.*", "", e.key).replace("void ", "").replace("msgat::", "")
    print(f"{e.device_time_total / a.steps:9.1f} us/step  n={e.count / a.steps:5.1f}  avg {e.device_time_total / max(e.count, 1):8.1f} us  {name[:70]}")
