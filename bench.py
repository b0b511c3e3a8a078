#!/usr/bin/env python3
"""Throughput of the MS-GAT graph-attention hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pemsd7|pemsd4|stress]

N > 1 is launched by the driver as one process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one forward + backward pass of the hot path over one synthetic batch:
both GACN depths of every MS-GAT component (reference msgat.py:25-28 called from
msgat.py:127, twice per TPC, R components), i.e. for msgat72 on PEMSD7
GACN(1->24) and GACN(72->24), R = 3 relations stacked into one launch sequence,
B = 32 samples per GPU.  With N > 1 the batch axis is sharded (weak scaling: B per GPU is
fixed) and the parameter gradients are all-reduced over RCCL once per step.

Rank 0 prints ONE JSON line (see the repo prompt for the contract) carrying
`roofline` (attention-aggregate kernel, HIP-event timed on the launch stream) and
`cpu_baseline` (oracle/dense_torch.py -- the reference's op sequence -- on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "MS-GAT fwd+bwd samples/sec (B×T node-updates/s), PEMSD7 N=883 T=12"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves

# rehearsal switch: run the RCCL path (process group, gradient all-reduce, MAX over ranks) even with one rank,
# so the multi-GPU code is exercised on a one-GPU box:  MSGAT_BENCH_FORCE_DIST=1 python -m torch.distributed.run ...
FORCE_DIST = os.environ.get("MSGAT_BENCH_FORCE_DIST") == "1"

WORKLOADS = {
    # name: (N nodes, E undirected edges, B per GPU, R relations, in_channels of the first MEAM, hidden, Co)
    "pemsd7": dict(N=883, E=866, B=32, R=3, Cin=1, hidden=72, Co=24, T=12),
    "pemsd4": dict(N=307, E=340, B=64, R=1, Cin=3, hidden=72, Co=24, T=12),
    "stress": dict(N=8192, E=65536, B=64, R=4, Cin=1, hidden=72, Co=24, T=12),
}


def layer_norm_t(x):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],))


class HotPath:
    """The two stacked GACN depths of R components with synthetic, seeded inputs."""

    def __init__(self, wl, device, seed):
        import ms_gat_amd
        self.wl, self.device = wl, device
        R, B, N, T = wl["R"], wl["B"], wl["N"], wl["T"]
        g = torch.Generator().manual_seed(2)
        self.adj = ms_gat_amd.synthetic_adjacency(N, wl["E"], seed=0)
        self.graph = ms_gat_amd.SparseGraph(self.adj)
        self.layers = []
        for cin in (wl["Cin"], wl["hidden"]):
            m = ms_gat_amd.StackedGACN(R, cin, wl["Co"], T)
            with torch.no_grad():  # msgat.py:206-217: xavier_normal_ for >=2-D, U(+-size0^-1/2) for 1-D (per relation)
                for r in range(R):
                    torch.nn.init.xavier_normal_(m.Wg[r], generator=g)
                    torch.nn.init.xavier_normal_(m.W[r], generator=g)
                    m.alpha[r].uniform_(-cin ** -0.5, cin ** -0.5, generator=g)
            self.layers.append(m.to(device))
        gx = torch.Generator().manual_seed(1000 + seed)
        self.xs, self.dzs = [], []
        for cin in (wl["Cin"], wl["hidden"]):
            x = layer_norm_t(torch.randn(R, B, cin, N, T, generator=gx))  # msgat.py:122: GACN sees LayerNorm output
            self.xs.append(x.to(device).requires_grad_(True))
            self.dzs.append(torch.randn(R, B, wl["Co"], N, T, generator=gx).to(device))
        from ms_gat_amd import parallel
        self.params = [p for m in self.layers for p in m.parameters()]
        self.sync = parallel.FlatGradAllReduce(self.params)

    def step(self, world):
        for m, x, dz in zip(self.layers, self.xs, self.dzs):
            x.grad = None
            for p in m.parameters():
                p.grad = None
            z = m(x, self.graph)
            z.backward(dz)
        if world > 1 or FORCE_DIST:  # one flat bucket: the payload is KBs, the collective is latency-bound
            self.sync(weight=float(self.wl["B"]))

    def forward_only(self):
        with torch.no_grad():
            return [m(x, self.graph) for m, x in zip(self.layers, self.xs)]


def dense_reference_step(hp, device, B, backward=True):
    """The reference's dense op sequence (oracle/dense_torch.py) on `device` for B samples/relation."""
    from oracle import dense_torch
    adj = hp.adj.to(device)
    outs = []
    for m, x, dz in zip(hp.layers, hp.xs, hp.dzs):
        for r in range(hp.wl["R"]):
            xr = x[r, :B].detach().to(device).requires_grad_(backward)
            Wg, al, W = (p[r].detach().to(device).requires_grad_(backward) for p in (m.Wg, m.alpha, m.W))
            z = dense_torch.gacn_dense(xr, adj, Wg, al, W)
            if backward:
                z.backward(dz[r, :B].to(device))
            outs.append(z)
    return outs


def time_aggregate_kernel(hp, reps=40):
    """HIP-event timing of the attention-aggregate kernel alone (second depth: Cu = Co channels of
    the projected features), on torch's current stream -- the stream the kernel is launched on."""
    from ms_gat_amd import _lib
    wl, dev = hp.wl, hp.device
    G, Cu, N, T = wl["R"] * wl["B"], wl["Co"], wl["N"], wl["T"]
    L = _lib.lib()
    gs, _keep = hp.graph.on(dev)
    shape = _lib.Shape(wl["R"], wl["B"], wl["hidden"], wl["Co"], N, T)
    # Timed on operands that are NOT cache-resident: one launch touches 196 MB, which would sit in the 256 MB
    # infinity cache from one repetition to the next if the same buffers were re-used (40 us, 4.9 TB/s -- reported
    # beside it as `us_per_launch_cached_operands`).  Four operand sets used in turn (784 MB) keep every launch on
    # HBM: the conservative figure, and the one the roofline fraction is computed from.  In the step the kernel
    # sits between the two (its input was just written by the projection): 43.8 us in the rocprofv3 per-grid
    # average of the in-application launches (profiles/r01/c_final_agg_lds_by_grid.txt).
    us = [torch.randn(G, Cu, N, T, device=dev) for _ in range(4)]
    vs = [torch.empty_like(u) for u in us]
    E = torch.rand(G, max(hp.graph.nnz, 1), device=dev)
    stream = torch.cuda.current_stream(dev)

    nscratch = int(L.msgat_edge_scratch_floats(C.byref(shape), C.byref(gs)))
    scratch = torch.empty(nscratch, device=dev) if nscratch else None

    def launch(i):
        _lib.check(L.msgat_stage_aggregate(C.byref(shape), C.byref(gs), Cu, us[i].data_ptr(), E.data_ptr(),
                                           vs[i].data_ptr(), None if scratch is None else scratch.data_ptr(),
                                           stream.cuda_stream), "msgat_stage_aggregate")

    def timed(nsets):
        for i in range(4):
            launch(i % nsets)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for i in range(reps):
            launch(i % nsets)
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3 / reps

    sec = timed(4)
    time_aggregate_kernel.cached_sec = timed(1)
    # algorithmic bytes per launch: read u once + write v once + E + CSR
    nnz = hp.graph.nnz
    bytes_ = 2 * 4 * G * Cu * N * T + 4 * G * nnz + 8 * nnz + 4 * (N + 1)
    return sec, bytes_


class _EagerMEAM(torch.nn.Module):
    """The reference's op sequence for a whole MEAM block (oracle/dense_torch.py: LayerNorm, CACN, TACN, dense
    GACN, residual tail -- all PyTorch-ROCm eager ops) behind the MEAM interface, sharing the parameters of the
    module it replaces: the eager baseline of the full-model comparison."""

    def __init__(self, meam):
        super().__init__()
        self.inner = meam

    def forward(self, signals, adjacency):
        from oracle import dense_torch
        return dense_torch.meam_dense(signals, adjacency, dict(self.inner.named_parameters()), self.inner.dilations,
                                      self.inner.ln.eps)


class _EagerLayerNorm(torch.nn.Module):
    def __init__(self, ln):
        super().__init__()
        self.inner = ln

    def forward(self, x):
        return torch.nn.functional.layer_norm(x, self.inner.normalized_shape, self.inner.weight, self.inner.bias,
                                              self.inner.eps)


def full_model_step_ms(wl, dev, dense, steps=6, warmup=5):
    """One training step (forward, Huber loss, backward, Adam) of the whole msgat72 model: MEAM blocks in the
    library (LayerNorm, the three branches, the tail), or -- `dense=True` -- the reference's eager op sequence."""
    from ms_gat_amd import engine, model
    import ms_gat_amd
    torch.manual_seed(0)
    adj = ms_gat_amd.synthetic_adjacency(wl["N"], wl["E"], seed=0)
    net = model.msgat72(n_components=wl["R"], in_channels=wl["Cin"], in_timesteps=wl["T"], out_timesteps=wl["T"],
                        use_te=True, adj=adj).to(dev)
    if dense:
        net.stack_components = False   # the reference's loop over components (msgat.py:204)
        for tpc in net.tpcs:
            tpc.tgacns = torch.nn.ModuleList(_EagerMEAM(m) for m in tpc.tgacns)
            tpc.ln = _EagerLayerNorm(tpc.ln)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
    loss_fn = engine.HuberLoss(50.0)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(wl["B"], wl["R"], wl["Cin"], wl["N"], wl["T"], generator=g).to(dev)
    H = torch.randint(0, 24, (wl["B"],), generator=g).to(dev)
    D = torch.randint(0, 7, (wl["B"],), generator=g).to(dev)
    Y = (torch.randn(wl["B"], wl["N"], wl["T"], generator=g) * 30).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss_fn(net(X, H, D), Y).backward()
        opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def full_model_graph_step_ms(wl, dev, steps=10):
    """The same training step through `engine.Trainer(hip_graph=True)`: one HIP-graph replay per batch."""
    import tempfile
    import ms_gat_amd
    from ms_gat_amd import engine, model
    torch.manual_seed(0)
    adj = ms_gat_amd.synthetic_adjacency(wl["N"], wl["E"], seed=0)
    net = model.msgat72(n_components=wl["R"], in_channels=wl["Cin"], in_timesteps=wl["T"], out_timesteps=wl["T"],
                        use_te=True, adj=adj).to(dev)
    g = torch.Generator().manual_seed(3)
    batch = [torch.randn(wl["B"], wl["R"], wl["Cin"], wl["N"], wl["T"], generator=g).to(dev),
             torch.randint(0, 24, (wl["B"],), generator=g).to(dev), torch.randint(0, 7, (wl["B"],), generator=g).to(dev),
             (torch.randn(wl["B"], wl["N"], wl["T"], generator=g) * 30).to(dev)]
    with tempfile.TemporaryDirectory() as tmp:
        tr = engine.Trainer(net, 50.0, tmp, hip_graph=True)
        tr.run_epoch([batch] * 2, gpu_id=dev.index, epoch=0, mode="train")   # captures
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        tr.run_epoch([batch] * steps, gpu_id=dev.index, epoch=1, mode="train")
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3


def recorded_traffic(workload):
    """HBM bytes per launch of k_agg_lds from the PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected separately, gfx950 x2 correction on the fetch side; see profiles/*/hbm_traffic.json).
    PMC counters cannot be collected from inside this process, so the newest committed record is quoted."""
    import glob
    if workload != "pemsd7":
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            for name, v in k.items():
                if "k_agg_lds" in name:
                    return int(v["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="pemsd7", choices=sorted(WORKLOADS))
    ap.add_argument("--no-baselines", action="store_true", help="skip the CPU / eager baselines")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or FORCE_DIST
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    wl = WORKLOADS[args.workload]
    hp = HotPath(wl, dev, seed=rank)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        hp.step(world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.step(world)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    samples = wl["B"] * world  # samples per step over all ranks
    value = samples / (elapsed / args.steps)

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"{args.workload}: N={wl['N']} nodes, {wl['E']} undirected edges (+self loops, sym-normalised), "
                         f"T={wl['T']}, B={wl['B']}/GPU, R={wl['R']} relations, GACN {wl['Cin']}->{wl['Co']} and "
                         f"{wl['hidden']}->{wl['Co']} (msgat72 widths), forward+backward of the hot path"),
            "global_batch": samples, "parallelism": f"batch-sharded x{world}, RCCL all-reduce of parameter grads",
        },
        "node_updates_per_s": round(value * wl["R"] * wl["T"] * wl["N"], 1),
    }

    if rank == 0:
        sec, nbytes = time_aggregate_kernel(hp)
        out["roofline"] = {
            # the library picks the variant by slab size: whole [N,T] slab in LDS, one 4-timestep column of it,
            # or gather from L2 (aggregate.hip)
            "kernel": ("k_agg_lds" if wl["N"] * wl["T"] * 4 <= 159 * 1024 else
                       "k_agg_cols" if wl["N"] * 16 <= 159 * 1024 else "k_agg_glb")
                      + " (attention-aggregate, second GACN depth)", "bound": "hbm",
            "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": recorded_traffic(args.workload),
            "us_per_launch": round(sec * 1e6, 2),
            "us_per_launch_cached_operands": round(time_aggregate_kernel.cached_sec * 1e6, 2), "algorithmic_bytes": nbytes,
        }
    if rank == 0 and world == 1 and not args.no_baselines:
        # PyTorch-ROCm eager on the same GPU: the reference's dense op sequence
        for bw, key in ((False, "eager_rocm_forward"), (True, "eager_rocm_fwd_bwd")):
            for _ in range(2):
                dense_reference_step(hp, dev, wl["B"], backward=bw)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                dense_reference_step(hp, dev, wl["B"], backward=bw)
            torch.cuda.synchronize(dev)
            out[key + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        for _ in range(3):
            hp.forward_only()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            hp.forward_only()
        torch.cuda.synchronize(dev)
        out["forward_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        out["speedup_vs_eager_rocm_forward"] = round(out["eager_rocm_forward_ms"] / out["forward_ms"], 2)
        out["speedup_vs_eager_rocm_fwd_bwd"] = round(out["eager_rocm_fwd_bwd_ms"] / ms_per_step, 2)

        # secondary: the whole msgat72 training step, HIP graph branch vs the reference's dense eager one
        try:
            out["full_model_step_ms"] = round(full_model_step_ms(wl, dev, dense=False), 3)
            out["full_model_hip_graph_step_ms"] = round(full_model_graph_step_ms(wl, dev), 3)
            out["full_model_eager_rocm_step_ms"] = round(full_model_step_ms(wl, dev, dense=True), 3)
        except RuntimeError as e:  # e.g. the dense [B,N,N] tensors of the stress graph do not fit
            out["full_model_error"] = str(e).splitlines()[0][:120]

        # CPU baseline: same op sequence on the host cores, bounded sample of the same workload
        cores = min(len(os.sched_getaffinity(0)), 16)  # the GPU box gives one GPU a 16-core share
        torch.set_num_threads(cores)
        cpu = torch.device("cpu")
        Bs = wl["B"] if wl["N"] <= 1024 else max(1, wl["B"] // 8)  # bounded: ~10-20 s of host time
        dense_reference_step(hp, cpu, Bs)  # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < 12.0 and n < 40):
            dense_reference_step(hp, cpu, Bs)
            n += 1
        dt = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {
            "value": round(Bs / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": (f"{n} fwd+bwd passes of oracle/dense_torch.py (the reference's dense op sequence) on "
                       f"{Bs} of the {wl['B']} samples per relation, all {wl['R']} relations and both GACN depths, "
                       f"same graph and widths, {cores} torch threads"),
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
