#!/usr/bin/env python3
"""Times the edge kernels of the N = 8192 stress graph (BASELINE.json configs[4]: degree 16, R = 4, B = 64,
C = 72 -> 24) with HIP events: forward aggregate, backward (transposed) aggregate and SDDMM through the fused
backward is not separable, so the stages are called directly.

    python tools/stress_kernels.py [--B 64] [--reps 5] [--sell auto|never]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ms_gat_amd  # noqa: E402
from ms_gat_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--R", type=int, default=4)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--E", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sell", default="auto")
    ap.add_argument("--lab", type=int, default=-1, help="time tools/agg_sell_lab.hip's k_agg_sell with one phase removed "
                    "(0..4; needs `python -m ms_gat_amd.build --lab`)")
    ap.add_argument("--fused", type=int, default=0, help="time tools/bwd_sell_lab.hip's fused du + dE column pass with this "
                    "many waves per block (16 | 12 | 8), variants 0..2 (needs `python -m ms_gat_amd.build --lab`)")
    a = ap.parse_args()
    if a.lab >= 0 or a.fused:
        _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "lab", "libmsgat_lab.so")
    N, T, R, B, Cc, Co = a.N, 12, a.R, a.B, 72, 24
    G = R * B
    dev = torch.device("cuda:0")
    L = _lib.lib()
    graph = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(N, a.E, 0), sell=a.sell)
    gs, _keep = graph.on(dev)
    nnz = graph.nnz
    shape = _lib.Shape(R, B, Cc, Co, N, T)
    sp, gp = C.byref(shape), C.byref(gs)
    u = torch.randn(G, Co, N, T, device=dev)
    v = torch.empty_like(u)
    E = torch.rand(G, nnz, device=dev)
    nscr = int(L.msgat_edge_scratch_floats(sp, gp))
    scr = torch.empty(max(nscr, 1), device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.reps):
            fn()
        t1.record()
        t1.synchronize()
        return t0.elapsed_time(t1) / a.reps

    alg = 2 * 4 * G * Co * N * T + 4 * G * nnz + 8 * nnz + 4 * (N + 1)
    ms = timed(lambda: _lib.check(L.msgat_stage_aggregate(sp, gp, Co, u.data_ptr(), E.data_ptr(), v.data_ptr(),
                                                          scr.data_ptr() if nscr else None, st), "agg"))
    print(f"aggregate (incl. edge permute)  {ms:8.3f} ms   {alg / ms / 1e6:8.1f} GB/s algorithmic  sell={graph.has_sell} nnz={nnz}",
          flush=True)

    if a.fused:
        fn = L.msgat_lab_bwd_sell_fused
        fn.restype, fn.argtypes = C.c_int, [C.POINTER(_lib.Shape), C.POINTER(_lib.Graph), C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p]
        npos = gs.sell_cols.n_pos
        Es = torch.rand(G * npos + _lib.SELL_SLACK, device=dev)      # coefficients in sell_cols position order
        dz, du, dE = torch.randn(G, Co, N, T, device=dev), torch.empty(G, Co, N, T, device=dev), torch.empty(G * npos + _lib.SELL_SLACK, device=dev)
        print(f"sell_cols: {gs.sell_cols.n_slices} slices, n_pos {npos}, pair_trips {gs.sell_cols.pair_trips}", flush=True)
        for lab, what in ((0, "scattered du stores"), (1, "no du stores"), (2, "du through LDS, N/Q consecutive rows per block")):
            ms = timed(lambda: _lib.check(fn(sp, gp, Co, dz.data_ptr(), u.data_ptr(), Es.data_ptr(), du.data_ptr(), dE.data_ptr(),
                                             a.fused, lab, st), "fused lab"))
            print(f"fused du + dE, {a.fused} waves/block, {what:48s} {ms:8.3f} ms", flush=True)
        return

    if a.lab >= 0:
        fn = L.msgat_lab_aggregate_sell
        fn.restype, fn.argtypes = C.c_int, [C.POINTER(_lib.Shape), C.POINTER(_lib.Graph), C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_void_p]
        Es = torch.rand(G, gs.sell_rows.n_pos + _lib.SELL_SLACK, device=dev)
        ms = timed(lambda: _lib.check(fn(sp, gp, Co, u.data_ptr(), Es.data_ptr(), v.data_ptr(), a.lab, st), "lab"))
        print(f"k_agg_sell lab variant {a.lab}       {ms:8.3f} ms", flush=True)
        return

    # attention backward on the projected features: SDDMM + edge/row passes + dense column pass + transposed aggregate
    shp = _lib.Shape(R, B, Co, 0, N, T)
    q, kW, pq = (torch.randn(G, N, T, device=dev) * 0.3 for _ in range(3))
    lse = torch.randn(G, N, device=dev) + 12.0
    Wg = torch.randn(R, T, T, device=dev) * 0.3
    dv = torch.randn(G, Co, N, T, device=dev)
    du, dq, dWg = torch.empty_like(u), torch.empty(G, N, T, device=dev), torch.empty(R, T, T, device=dev)
    nb = int(L.msgat_attention_bwd_workspace_bytes(C.byref(shp), gp))
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    ms = timed(lambda: _lib.check(L.msgat_attention_backward(C.byref(shp), gp, u.data_ptr(), dv.data_ptr(), 0, q.data_ptr(),
                                                             kW.data_ptr(), lse.data_ptr(), pq.data_ptr(), E.data_ptr(), None,
                                                             Wg.data_ptr(), du.data_ptr(), dq.data_ptr(), dWg.data_ptr(),
                                                             ws.data_ptr(), ws.numel(), st), "bwd"))
    print(f"attention backward (all stages) {ms:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
