#!/usr/bin/env python3
"""Times the individual stage kernels of libmsgat_hip.so with HIP events (GPU box only).

    python tools/kbench.py [--workload pemsd7] [--reps 20] [--only mix,contract,...]

Prints one line per stage: microseconds per launch and algorithmic GB/s.  Used to iterate on a
single kernel without the autograd plumbing around it.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ms_gat_amd  # noqa: E402
from ms_gat_amd import _lib  # noqa: E402

WL = {"pemsd7": dict(N=883, E=866, B=32, R=3, C=72, Co=24, T=12),
      "aligned": dict(N=880, E=866, B=32, R=3, C=72, Co=24, T=12),   # N*T*4 a multiple of 128: rows start on cache lines
      "pemsd4": dict(N=307, E=340, B=64, R=1, C=72, Co=24, T=12),
      "stress": dict(N=8192, E=65536, B=8, R=4, C=72, Co=24, T=12)}


def timeit(fn, reps):
    """Per-launch GPU time: `reps` launches captured in one HIP graph and replayed, so host-side launch
    cost (20+ us per launch from Python on the GPU box) cannot hide in the figure."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    graph.replay()
    s = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(s)
    for _ in range(3):
        graph.replay()
    t1.record(s)
    t1.synchronize()
    return t0.elapsed_time(t1) * 1e-3 / (3 * reps)


def timeit_eager(fn, reps):
    for _ in range(3):
        fn()
    s = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(s)
    for _ in range(reps):
        fn()
    t1.record(s)
    t1.synchronize()
    return t0.elapsed_time(t1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pemsd7")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--sets", type=int, default=1, help="copies of the big operands to rotate through (cold-cache timing)")
    ap.add_argument("--eager", action="store_true", help="plain launches instead of graph replay (counter collection)")
    ap.add_argument("--lib", default="", help="another build of libmsgat_hip.so to time (A/B runs of one kernel)")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    if a.eager:
        globals()["timeit"] = timeit_eager
    w = WL[a.workload]
    N, T, R, B, Cc, Co = w["N"], w["T"], w["R"], w["B"], w["C"], w["Co"]
    G, P = R * B, N * T
    dev = torch.device("cuda:0")
    L = _lib.lib()
    graph = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(N, w["E"], 0))
    gs, _keep = graph.on(dev)
    nnz = graph.nnz
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731  (read at call time: graph capture runs on a side stream)
    rnd = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    # --sets K: the streaming stages rotate through K copies of their big operands, so a replay does not find them
    # in the 256 MB last-level cache (inside a training step they never are)
    xs = [rnd(G, Cc, N, T) for _ in range(a.sets)]
    us = [rnd(G, Co, N, T) for _ in range(a.sets)]
    oxs = [torch.empty_like(xs[0]) for _ in range(a.sets)]
    ous = [torch.empty_like(us[0]) for _ in range(a.sets)]
    tick = [0]

    def rot(lst):
        tick[0] += 1
        return lst[tick[0] % len(lst)]
    x, u, dz = xs[0], us[0], rnd(G, Co, N, T)
    alpha, Wg, W = rnd(R, Cc) * 0.1, rnd(R, T, T) * 0.3, rnd(R, Co, Cc) * 0.1
    q, kW, pq, dq = rnd(G, N, T), rnd(G, N, T), rnd(G, N, T), rnd(G, N, T)
    lse, E = rnd(G, N), torch.rand(G, max(nnz, 1), device=dev)
    out_u, out_x = ous[0], oxs[0]
    shape = _lib.Shape(R, B, Cc, Co, N, T)
    sp = C.byref(shape)
    gp = C.byref(gs)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    nscr = int(L.msgat_edge_scratch_floats(sp, gp))
    escr = torch.empty(nscr, device=dev) if nscr else None
    ndense = int(L.msgat_dense_scratch_bytes(sp))
    dscr = torch.empty(ndense, device=dev, dtype=torch.uint8) if ndense else None
    stages = {}

    def reg(name, nbytes, fn):
        stages[name] = (nbytes, fn)

    reg("project_fwd  x->u,q", 4 * G * P * (Cc + Co + 1),
        lambda: _lib.check(L.msgat_stage_project(sp, ptr(rot(xs)), ptr(alpha), ptr(W), ptr(q), ptr(rot(ous)), st()), "p"))
    reg("scores       q->kW,lse,pq,E", 4 * G * P * 4,
        lambda: _lib.check(L.msgat_stage_scores(sp, gp, ptr(q), ptr(Wg), ptr(kW), ptr(lse), ptr(pq), ptr(E), None, ptr(dscr), st()), "s"))
    reg("scores_nopq  q->kW,lse,E", 4 * G * P * 3,
        lambda: _lib.check(L.msgat_stage_scores(sp, gp, ptr(q), ptr(Wg), ptr(kW), ptr(lse), None, ptr(E), None, ptr(dscr), st()), "s"))
    reg("aggregate    u->z (Cu=Co)", 8 * G * Co * P,
        lambda: _lib.check(L.msgat_stage_aggregate(sp, gp, Co, ptr(rot(us)), ptr(E), ptr(rot(ous)), ptr(escr), st()), "a"))
    reg("mix_bwd      du,dq->dx", 4 * G * P * (Co + 1 + Cc),
        lambda: _lib.check(L.msgat_stage_mix(sp, Co, Cc, ptr(rot(us)), ptr(W), 1, ptr(alpha), ptr(dq), ptr(rot(oxs)), st()), "m"))
    reg("project_bwd  du,dq,x->dW,dalpha,dx", 4 * G * P * (Co + 1 + 2 * Cc),
        lambda: _lib.check(L.msgat_stage_project_backward(sp, ptr(rot(us)), ptr(dq), ptr(rot(xs)), ptr(W), ptr(alpha),
                                                          ptr(part), ptr(dW), ptr(da), ptr(rot(oxs)), st()), "pb"))
    # the merged channel mixing of a MEAM block (stacked.py): 72 -> 98 channels forward, 98 -> 72 backward
    Cm = 98
    ys = [rnd(G, Cm, N, T) for _ in range(min(a.sets, 2))]
    Wm = rnd(R, Cm, Cc) * 0.1
    reg("mix_fwd98    x(72)->98", 4 * G * P * (Cc + Cm),
        lambda: _lib.check(L.msgat_stage_mix(sp, Cc, Cm, ptr(rot(xs)), ptr(Wm), 0, None, None, ptr(rot(ys)), st()), "m"))
    # lab (round-5 review, item 5): the 72 -> 98 mixing as TWO passes over x with half of the output rows each -- the time
    # of a two-z-block form whose second reader finds NOTHING in a cache (x is 293 MB): an upper bound for such a form
    for Ch in (48, 50):
        yh = [rnd(G, Ch, N, T) for _ in range(min(a.sets, 2))]
        Wh = rnd(R, Ch, Cc) * 0.1
        reg(f"mix_fwd{Ch}    x(72)->{Ch}", 4 * G * P * (Cc + Ch),
            lambda yh=yh, Wh=Wh, Ch=Ch: _lib.check(L.msgat_stage_mix(sp, Cc, Ch, ptr(rot(xs)), ptr(Wh), 0, None, None,
                                                                      ptr(rot(yh)), st()), "m"))
    reg("mix_bwd98    d98->dx(72)", 4 * G * P * (Cc + Cm),
        lambda: _lib.check(L.msgat_stage_mix(sp, Cm, Cc, ptr(rot(ys)), ptr(Wm), 1, None, None, ptr(rot(oxs)), st()), "m"))
    # the same passes as the model issues them (stacked.py / model.MEAM): channel axes assembled from several tensors
    def segs(*items):
        arr = (_lib.Seg * len(items))()
        for i, (t, ch, gstride) in enumerate(items):
            arr[i] = _lib.Seg(t.data_ptr(), ch, gstride)
        return arr, len(items)

    br = [[rnd(G, Co, N, T) for _ in range(3)] for _ in range(a.sets)]      # the three branch outputs
    Wres, bres = rnd(R, Cc, Cc) * 0.1, rnd(R, Cc) * 0.1

    def seg_tail():   # relu(cat(branches) + res(x) + bias): in x[72], add = 3 x 24 channels, out [72]
        xi, oi, bi = rot(xs), rot(oxs), rot(br)
        i_, ni = segs((xi, Cc, 0))
        a_, na = segs(*[(b, Co, 0) for b in bi])
        o_, no = segs((oi, Cc, 0))
        _lib.check(L.msgat_mix_segments(R, B, N, T, i_, ni, ptr(Wres), 0, ptr(bres), 1, a_, na, 1, o_, no, st()), "t")
    reg("seg_tail     x(72)+3x24->72 relu", 4 * G * P * (Cc + 3 * Co + Cc), seg_tail)

    o98 = [[rnd(G, Co, N, T), rnd(G, 2 * Co, N, T), rnd(G, Co, N, T), rnd(G, 1, N, T), rnd(G, 1, N, T)]
           for _ in range(min(a.sets, 2))]

    def seg_fwd98():  # all channel mixings of one LayerNorm output: 72 -> 24 | 48 | 24 | 1 | 1
        xi, oi = rot(xs), rot(o98)
        i_, ni = segs((xi, Cc, 0))
        o_, no = segs(*[(t, t.shape[1], 0) for t in oi])
        _lib.check(L.msgat_mix_segments(R, B, N, T, i_, ni, ptr(Wm), 0, None, 0, None, 0, 0, o_, no, st()), "f")
    reg("seg_fwd98    x(72)->24|48|24|1|1", 4 * G * P * (Cc + Cm), seg_fwd98)

    def seg_bwd98():  # its backward: the five gradients -> dx[72]
        gi, oi = rot(o98), rot(oxs)
        i_, ni = segs(*[(t, t.shape[1], 0) for t in gi])
        o_, no = segs((oi, Cc, 0))
        _lib.check(L.msgat_mix_segments(R, B, N, T, i_, ni, ptr(Wm), 1, None, 0, None, 0, 0, o_, no, st()), "b")
    reg("seg_bwd98    24|48|24|1|1->dx(72)", 4 * G * P * (Cc + Cm), seg_bwd98)

    def mix_bias72():  # a 72 -> 72 1x1 convolution with bias (unsegmented, epilogue only)
        _lib.check(L.msgat_stage_mix_epilogue(sp, Cc, Cc, ptr(rot(xs)), ptr(Wres), 0, ptr(bres), 1, None, 0, ptr(rot(oxs)), st()), "e")
    reg("mix_bias72   x(72)->72 +bias", 4 * G * P * 2 * Cc, mix_bias72)

    nfl = L.msgat_contract_partial_floats(sp, Co + 1, Cc)
    part = torch.empty(nfl, device=dev)
    dW, da = torch.empty(R, Co, Cc, device=dev), torch.empty(R, Cc, device=dev)
    reg("contract     du,dq,x->dW,dalpha", 4 * G * P * (Co + 1 + Cc),
        lambda: _lib.check(L.msgat_stage_contract(sp, Co + 1, Cc, ptr(rot(us)), ptr(dq), ptr(rot(xs)), ptr(part), ptr(dW),
                                                  Co * Cc, ptr(da), Cc, st()), "c"))
    nfl98 = L.msgat_contract_partial_floats(sp, Cm, Cc)
    part98, dW98 = torch.empty(nfl98, device=dev), torch.empty(R, Cm, Cc, device=dev)
    reg("contract98   d98,x->dW[98,72]", 4 * G * P * (Cm + Cc),
        lambda: _lib.check(L.msgat_stage_contract(sp, Cm, Cc, ptr(rot(ys)), None, ptr(rot(xs)), ptr(part98), ptr(dW98),
                                                  Cm * Cc, None, 0, st()), "c"))
    # the whole backward of a 1x1 convolution in one pass (msgat_contract_mix_segments): dM, dbias and dx
    def cmix(Ca, srcs, outs):   # gradient from `srcs`, input from xs, dx into `outs`: three distinct pools
        nf = L.msgat_contract_segments_partial_floats(R, Ca, Cc + 1)
        pt, dM = torch.empty(nf, device=dev), torch.empty(R, Ca, Cc + 1, device=dev)
        Mx = rnd(R, Ca, Cc) * 0.1

        def run():
            t = rot(srcs)
            arr = (_lib.Seg * 1)(_lib.Seg(t.data_ptr(), Ca, 0))
            _lib.check(L.msgat_contract_mix_segments(R, B, N, T, arr, 1, ptr(rot(xs)), Cc, 1, ptr(Mx), ptr(pt), ptr(dM),
                                                     ptr(rot(outs)), st()), "cm")
        return run
    reg("cmix98       d98,x->dW[98,73],dx", 4 * G * P * (Cm + 2 * Cc), cmix(Cm, ys, oxs))
    reg("cmix72       d72,x->dW[72,73],dx", 4 * G * P * (3 * Cc), cmix(Cc, oxs, ys))

    def cmix96():   # the first 96 of the 98 gradient rows (a channel slice): a six-tile block where a lab build has one
        Ca = Cm - 2
        nf = L.msgat_contract_segments_partial_floats(R, Ca, Cc + 1)
        pt, dM = torch.empty(nf, device=dev), torch.empty(R, Ca, Cc + 1, device=dev)
        Mx = rnd(R, Ca, Cc) * 0.1

        def run():
            t = rot(ys)
            arr = (_lib.Seg * 1)(_lib.Seg(t.data_ptr(), Ca, Cm))
            _lib.check(L.msgat_contract_mix_segments(R, B, N, T, arr, 1, ptr(rot(xs)), Cc, 1, ptr(Mx), ptr(pt), ptr(dM),
                                                     ptr(rot(oxs)), st()), "cm96")
        return run
    reg("cmix96       d96,x->dW[96,73],dx", 4 * G * P * (Cm - 2 + 2 * Cc), cmix96())
    only = [s for s in a.only.split(",") if s]
    for name, (nbytes, fn) in stages.items():
        if only and not any(o in name for o in only):
            continue
        sec = timeit(fn, a.reps)
        print(f"{name:36s} {sec * 1e6:9.1f} us   {nbytes / sec / 1e9:8.1f} GB/s  ({nbytes / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
