#!/usr/bin/env python3
"""Which lines of ms_gat_amd launch the small PyTorch ops (copies, fills, sums ...) of a training step?  One msgat72
step (R = 3) under a TorchDispatchMode that records, per aten op, the innermost ms_gat_amd / bench.py frame."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

WATCH = ("copy_", "fill_", "zero_", "sum", "add", "mul", "cat", "clone", "index_select", "embedding", "zeros", "stack",
         "empty_like", "new_empty", "expand", "where", "sub", "div", "neg", "masked_fill", "mean", "select_backward",
         "slice_backward", "_to_copy", "threshold_backward", "native_layer_norm")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        if any(name == w or name == w + "_" for w in WATCH):
            big = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:1]
            frame = "?"
            for f in reversed(traceback.extract_stack()):
                if ("ms_gat_amd" in f.filename or f.filename.endswith("bench.py")) and "glue_profile" not in f.filename:
                    frame = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
                    break
            self.count[(name, frame, str(big))] += 1
        return out


dev = torch.device("cuda:0")
ts = bench.TrainStep(dict(bench.CFG4, R=3), dev, hip_graph=False)
ts.run(3)
torch.cuda.synchronize()
spy = Spy()
with spy:
    ts.run(1)
    torch.cuda.synchronize()
for (name, frame, shp), n in sorted(spy.count.items(), key=lambda kv: (kv[0][1], -kv[1])):
    print(f"{n:4d}  {name:22s} {frame:48s} {shp}")
