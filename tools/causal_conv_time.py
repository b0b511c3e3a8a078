#!/usr/bin/env python3
"""Times ops.causal_conv (forward and its input gradient) at the msgat72 training shapes with HIP events:
    python tools/causal_conv_time.py [--lib build/lab/libmsgat_lab.so]   (lab builds read MSGAT_LAB_CCPB = tiles per block)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--warm", action="store_true", help="do not spoil the infinity cache between launches (in the training step the "
                "producer has just written the input)")
a = ap.parse_args()
from ms_gat_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import torch  # noqa: E402
from ms_gat_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (G, R, Cr, Co, N, T, d) in ((96, 3, 24, 24, 883, 12, 2), (96, 3, 48, 48, 883, 12, 2), (96, 3, 32, 32, 883, 12, 4), (96, 3, 24, 24, 307, 12, 2),
                             # lab (round-5 review, item 5): fewer output rows per block -- 24 -> 16 and 24 -> 8 channels as the two / three
                             # passes of a form whose blocks write 16 / 8 row streams instead of 24 (each pass reads all 24 input rows)
                             (96, 3, 24, 16, 883, 12, 2), (96, 3, 24, 8, 883, 12, 2)):
    x = torch.randn(G, Cr, N, T, device=dev, requires_grad=True)
    w = torch.randn(R, 2 * Co, Cr, device=dev) * 0.1
    b = torch.randn(R, Co, device=dev)
    spoil = torch.empty(1 << 28, device=dev)   # 1 GiB: the operands come from HBM, not the infinity cache
    def timed(fn):
        fn(); torch.cuda.synchronize()
        tot = 0.0
        for _ in range(a.reps):
            if not a.warm:
                spoil.zero_()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); fn(); t1.record(); t1.synchronize()
            tot += t0.elapsed_time(t1)
        return tot / a.reps * 1e3
    with torch.no_grad():
        us = timed(lambda: ops.causal_conv(x, w, b, d))
    mb = 4 * G * (Cr + Co) * N * T / 1e6
    print(f"G={G} Cr={Cr} Co={Co} N={N} d={d}: forward {us:7.1f} us  {mb / us:6.2f} TB/s  ({mb:.0f} MB)", flush=True)
