#!/usr/bin/env python3
"""What a write-heavy stream can expect: fill (write only), copy (1:1) and a 1:3 read:write elementwise pass, on buffers
rotated through 1.2+ GB so that nothing stays in the 256 MB infinity cache.  HIP events around 20 launches."""
import torch

dev = torch.device("cuda:0")
n = 293 * 1000 * 1000 // 4
bufs = [torch.empty(n, device=dev) for _ in range(4)]
srcs = [torch.randn(n // 3, device=dev) for _ in range(4)]


def timeit(fn, reps=20):
    for i in range(4):
        fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


us = timeit(lambda i: bufs[i % 4].fill_(1.0))
print(f"fill 293 MB: {us:.1f} us  {293e6 / us / 1e6:.2f} TB/s written")
us = timeit(lambda i: bufs[i % 4].copy_(bufs[(i + 2) % 4]))
print(f"copy 293 MB: {us:.1f} us  {2 * 293e6 / us / 1e6:.2f} TB/s read+write")
out3 = [b.view(3, -1) for b in bufs]
us = timeit(lambda i: torch.mul(srcs[i % 4].unsqueeze(0), 2.0, out=out3[i % 4][0:1]) if False else out3[i % 4].copy_(srcs[i % 4].unsqueeze(0).expand(3, -1)))
print(f"broadcast 98 MB -> 293 MB: {us:.1f} us  {(293e6 + 98e6) / us / 1e6:.2f} TB/s read+write")
