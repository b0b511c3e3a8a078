"""Times LayerNorm over T: libmsgat_hip.so against torch's kernel, forward and backward, PEMSD7 sizes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ms_gat_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


dev = torch.device("cuda:0")
B = int(os.environ.get("LN_BENCH_B", "32"))   # 96 = three stacked relations: 293 MB per tensor, beyond the 256 MB cache
for C in (3, 72):
    x = torch.randn(B, C, 883, 12, device=dev, requires_grad=True)
    w = torch.rand(12, device=dev, requires_grad=True)
    b = torch.rand(12, device=dev, requires_grad=True)
    dy = torch.randn_like(x)
    nbytes = x.numel() * 4
    for name, f in (("hip", lambda: ops.layer_norm_t(x, w, b)), ("torch", lambda: F.layer_norm(x, [12], w, b))):
        with torch.no_grad():
            tf = timeit(f)
        y = f()
        tb = timeit(lambda: torch.autograd.grad(y, (x, w, b), dy, retain_graph=True))
        print(f"C={C:3d} {name:5s} fwd {tf:8.1f} us ({2 * nbytes / tf / 1e6:6.2f} TB/s)   "
              f"bwd {tb:8.1f} us ({3 * nbytes / tb / 1e6:6.2f} TB/s)", flush=True)
