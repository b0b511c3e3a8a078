#!/usr/bin/env python3
"""Throughput of the MS-GAT graph-attention hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pemsd7|pemsd4|stress]

One JSON line (rank 0 prints it; see the repo prompt for the contract).  `value` means the SAME thing at every N
(`value_kind` = "hot_path"), so any two lines of a scaling curve compare like for like:

* one *step* = forward + backward of the hot path over one synthetic batch -- BASELINE.json's headline configuration,
  configs[2]: both GACN depths of every MS-GAT component (reference msgat.py:25-28 called from msgat.py:127, twice
  per TPC), i.e. for msgat72 on PEMSD7 GACN(1->24) and GACN(72->24), R = 3 relations stacked into one launch
  sequence, B = 32 samples PER GPU (weak scaling); with several ranks the step ends with the flat all-reduce of the
  hot path's parameter gradients (one bucket).  `value` = B * N / t_step.
* `full_step_cfg4` (every N, both launch modes) -- configs[3]: one whole msgat72 TRAINING step through
  `engine.Trainer`: forward of all R = 5 components, Huber loss + metrics, backward, ONE flat RCCL all-reduce of every
  gradient (1.96 M parameters, 7.8 MB) and Adam, B = 32 per GPU; `full_step_cfg4.value` is ITS samples/s over all ranks,
  timed with the same barrier + MAX-over-ranks clock.  It replaces the reference's `nn.DataParallel` loop
  (main.py:52-55, engine.py:49-63).
* rank 0 adds `roofline` (attention-aggregate kernel, HIP-event timed on the launch stream), `roofline_dense` (the
  score / column passes against the fp32 matrix-core peak) and -- single GPU only -- `cpu_baseline`
  (oracle/dense_torch.py, the reference's op sequence, on the host cores), the eager PyTorch-ROCm baselines,
  `full_step_cfg3`, `pemsd4` (configs[1]: N = 307, B = 64, one relation, against eager) and `stress` (configs[4]: N = 8192).

Launch: `python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` by itself
when N > 1 and no process group environment is present (before anything touches the GPU) and exits with its code;
under torch.distributed.run it joins the group it finds (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
Rehearsal on a one-GPU box: MSGAT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 puts every rank on cuda:0 with the
`gloo` transport (RCCL refuses two ranks on one device); MSGAT_BENCH_FORCE_DIST=1 runs the process-group path (RCCL)
with a single rank.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

# dmabuf IPC only on this driver (RCCL / device-tensor sharing across processes): must be in the environment before HSA
# comes up, i.e. before the first HIP call of this process -- not only before init_process_group
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "MS-GAT fwd+bwd samples/sec (B×T node-updates/s), PEMSD7 N=883 T=12"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix-core peak (MI355X_MICROARCH.md; tools/mfma_rate.hip measures 156)

# rehearsal switch: the multi-GPU mode (process group, flat gradient all-reduce, MAX over ranks) with one rank on a
# one-GPU box:  MSGAT_BENCH_FORCE_DIST=1 python bench.py
FORCE_DIST = os.environ.get("MSGAT_BENCH_FORCE_DIST") == "1"
# rehearsal switch: N ranks on ONE GPU (all on cuda:0, gloo transport) -- everything of the N > 1 path but RCCL
SHARE_GPU = os.environ.get("MSGAT_BENCH_SHARE_GPU") == "1"

WORKLOADS = {
    # name: (N nodes, E undirected edges, B per GPU, R relations, in_channels of the first MEAM, hidden, Co)
    "pemsd7": dict(N=883, E=866, B=32, R=3, Cin=1, hidden=72, Co=24, T=12),
    "pemsd4": dict(N=307, E=340, B=64, R=1, Cin=3, hidden=72, Co=24, T=12),
    "stress": dict(N=8192, E=65536, B=64, R=4, Cin=1, hidden=72, Co=24, T=12),
}
CFG4 = dict(N=883, E=866, B=32, R=5, Cin=1, T=12)   # configs[3]: per-GPU workload of the 1/2/4/8 scaling curve


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
        big = R * B * wl["hidden"] * N * T > (1 << 28)     # the stress inputs are generated on the device
        gdev = device if big else torch.device("cpu")
        gx = torch.Generator(device=gdev).manual_seed(1000 + seed)
        self.xs, self.dzs = [], []
        for cin in (wl["Cin"], wl["hidden"]):
            x = layer_norm_t(torch.randn(R, B, cin, N, T, generator=gx, device=gdev))  # msgat.py:122: GACN sees LayerNorm output
            self.xs.append(x.to(device).requires_grad_(True))
            self.dzs.append(torch.randn(R, B, wl["Co"], N, T, generator=gx, device=gdev).to(device))
        from ms_gat_amd import parallel
        self.params = [p for m in self.layers for p in m.parameters()]
        self.sync = parallel.FlatGradAllReduce(self.params)

    def step(self, allreduce=False):
        # the order of a training step of the model that holds these layers (msgat.py:143-150: the two MEAMs of a TPC
        # in sequence): forward of the first depth, forward of the second, then backward second depth first
        for m, x in zip(self.layers, self.xs):
            x.grad = None
            for p in m.parameters():
                p.grad = None
        # (each depth's forward and backward back to back measures the same: 0.737 vs 0.730-0.740 ms)
        zs = [m(x, self.graph) for m, x in zip(self.layers, self.xs)]
        # ONE backward call, as a training step makes one: the engine runs the second depth's node first (it was
        # created last).  Two calls cost a second start of the autograd engine (~150 us of host time each at this
        # size, tools/host_overhead.py), which a launch-bound step (PEMSD4) would be charged for.
        torch.autograd.backward(zs, self.dzs)
        if allreduce:  # one flat bucket: the payload is KBs, the collective is latency-bound
            self.sync(weight=float(self.wl["B"]))

    def forward_only(self):
        with torch.no_grad():
            return [m(x, self.graph) for m, x in zip(self.layers, self.xs)]


SETTLE_MS = 60.0   # see settle()


def settle(fn, device, ms=SETTLE_MS, agree=None):
    """Runs `fn` untimed for about `ms` milliseconds of GPU time.  Coming out of set-up (input generation on the host,
    graph build) the first ~25 hot-path steps are up to 10 % slower than the rate the GPU then holds (0.81-0.86 ms
    falling to 0.775 ms over 20 ms: tools/ramp.py) -- power state, cache and allocator warm-up.  The contract's W
    warm-up steps follow this; with W = 5 (4 ms) alone the K timed steps would sit inside that ramp.

    With several ranks `fn` may hold a collective, so every rank must leave the loop on the same trip: `agree(x)`
    (MAX over ranks) makes the elapsed time the ranks compare identical -- a rank must never issue a batch of
    all-reduces its peers do not."""
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(device)
    done = 0.0
    while done < ms:
        t0.record(stream)
        for _ in range(8):
            fn()
        t1.record(stream)
        t1.synchronize()
        batch = t0.elapsed_time(t1)
        done += batch if agree is None else agree(batch)


def timed_steps(fn, steps, warmup, device, barrier):
    """`warmup` untimed calls, then EXACTLY `steps` timed ones bracketed by barrier + synchronize (the contract's
    wall clock), with a HIP event between consecutive steps on the launch stream for the per-step distribution."""
    for _ in range(warmup):
        fn()
    barrier()
    stream = torch.cuda.current_stream(device)
    events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    events[0].record(stream)
    for i in range(steps):
        fn()
        events[i + 1].record(stream)
    barrier()
    wall = time.perf_counter() - t0
    per_step = [events[i].elapsed_time(events[i + 1]) for i in range(steps)]
    return wall, per_step


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


def aggregate_kernel_name(wl):
    """The library picks the aggregate variant by slab size (aggregate.hip): three [N,T] slabs in an LDS-DMA ring
    (one persistent block per CU); one whole slab in LDS; one 4-timestep column of it with the edges in the SELL layout;
    or gather from L2."""
    if wl["N"] * wl["T"] // 4 <= 3 * 1024 and wl["R"] * wl["B"] * wl["Co"] >= 512:
        return "k_agg_ring"
    if wl["N"] * wl["T"] * 4 <= 159 * 1024:
        return "k_agg_lds"
    return "k_agg_sell" if wl["N"] * 16 <= 159 * 1024 else "k_agg_glb"


def time_aggregate_kernel(hp, reps=40):
    """HIP-event timing of the attention-aggregate kernel alone (second depth: Cu = Co channels of the projected
    features), on torch's current stream -- the stream the kernel is launched on.  Returns (seconds per launch on
    operands that are NOT cache-resident, seconds per launch re-using one operand set, algorithmic bytes)."""
    from ms_gat_amd import _lib
    wl, dev = hp.wl, hp.device
    G, Cu, N, T = wl["R"] * wl["B"], wl["Co"], wl["N"], wl["T"]
    L = _lib.lib()
    gs, _keep = hp.graph.on(dev)
    shape = _lib.Shape(wl["R"], wl["B"], wl["hidden"], wl["Co"], N, T)
    # One launch touches 196 MB at the PEMSD7 workload, which would sit in the 256 MB infinity cache from one
    # repetition to the next if the same buffers were re-used (`us_per_launch_cached_operands`).  Four operand sets
    # used in turn (784 MB) keep every launch on HBM: the conservative figure, and the one the roofline fraction is
    # computed from.  (At the stress workload one set is 4.8 GB: two sets, far beyond any cache.)
    small = G * Cu * N * T * 4 < (1 << 30)
    nsets = 4 if small else 2
    us = [torch.randn(G, Cu, N, T, device=dev) for _ in range(nsets)]
    vs = [torch.empty_like(u) for u in us]
    E = torch.rand(G, max(hp.graph.nnz, 1), device=dev)
    stream = torch.cuda.current_stream(dev)
    nscratch = int(L.msgat_edge_scratch_floats(C.byref(shape), C.byref(gs)))
    scratch = torch.empty(nscratch, device=dev) if nscratch else None

    def launch(i):
        _lib.check(L.msgat_stage_aggregate(C.byref(shape), C.byref(gs), Cu, us[i].data_ptr(), E.data_ptr(),
                                           vs[i].data_ptr(), None if scratch is None else scratch.data_ptr(),
                                           stream.cuda_stream), "msgat_stage_aggregate")

    def timed(sets, n):
        for i in range(4):
            launch(i % sets)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for i in range(n):
            launch(i % sets)
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3 / n

    reps = reps if small else 8
    cold = timed(nsets, reps)
    cached = timed(1, reps)
    # algorithmic bytes per launch: read u once + write v once + E + CSR
    nnz = hp.graph.nnz
    nbytes = 2 * 4 * G * Cu * N * T + 4 * G * nnz + 8 * nnz + 4 * (N + 1)
    return cold, cached, nbytes


def recorded_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE
    collected in separate rocprofv3 passes, gfx950 x2 correction on the fetch side; see profiles/*/hbm_traffic*.json).
    PMC counters cannot be collected from inside this process, so the newest committed record is quoted -- with its
    source file, so the figure is never mistaken for something measured in this run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            for name, v in k.items():
                if kernel in name:
                    return {"bytes_per_launch": int(v["hbm_bytes_per_launch"]), "source": os.path.relpath(path, ROOT)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def roofline_object(hp):
    cold, cached, nbytes = time_aggregate_kernel(hp)
    kernel = aggregate_kernel_name(hp.wl)
    rec = recorded_traffic(kernel)
    return {
        "kernel": kernel + " (attention-aggregate, second GACN depth)", "bound": "hbm",
        "achieved": round(nbytes / cold / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(nbytes / cold / 1e9 / HBM_PEAK_GBS, 4),
        "traffic": None if rec is None else rec["bytes_per_launch"],   # recorded, not measured in this run:
        "traffic_recorded": rec,                                       # ... this names the committed PMC summary
        "us_per_launch": round(cold * 1e6, 2), "us_per_launch_cached_operands": round(cached * 1e6, 2),
        "algorithmic_bytes": nbytes,
    }


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


class _Recording:
    """Batches that are already this rank's shard (`msgat_sharded`: `Engine.run_epoch` must not slice them again),
    with a HIP event recorded on the launch stream before each one and after the last."""
    msgat_sharded = True

    def __init__(self, batches, events, dev):
        self.batches, self.events, self.dev = batches, events, dev

    def _mark(self):
        if self.events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.dev))
            self.events.append(e)

    def __iter__(self):
        for b in self.batches:
            self._mark()
            yield b
        self._mark()


class TrainStep:
    """One whole msgat72 training step through `engine.Trainer` on synthetic data: forward, Huber loss + metrics,
    backward, (with several ranks) ONE flat all-reduce of all gradients, Adam.  `dense=True` swaps every block for
    the reference's eager op sequence (the PyTorch-ROCm baseline of the same step)."""

    def __init__(self, cfg, dev, seed=0, dense=False, hip_graph=False, stacked=True):
        import tempfile
        import ms_gat_amd
        from ms_gat_amd import engine, model
        torch.manual_seed(0)
        self.cfg, self.dev = cfg, dev
        adj = ms_gat_amd.synthetic_adjacency(cfg["N"], cfg["E"], seed=0)
        net = model.build_msgat(f"ms-gat{cfg.get('hidden', 72)}", n_components=cfg["R"], in_channels=cfg["Cin"],
                                in_timesteps=cfg["T"], out_timesteps=cfg["T"], use_te=True, adj=adj).to(dev)
        if not stacked:
            net.stack_components = False   # the reference's loop over components (msgat.py:204), this package's blocks
        if dense:
            net.stack_components = False   # the reference's loop over components (msgat.py:204)
            for tpc in net.tpcs:
                tpc.tgacns = torch.nn.ModuleList(_EagerMEAM(m) for m in tpc.tgacns)
                tpc.ln = _EagerLayerNorm(tpc.ln)
        self.net = net
        self._tmp = tempfile.TemporaryDirectory()
        self.trainer = engine.Trainer(net, 50.0, self._tmp.name, hip_graph=hip_graph)
        g = torch.Generator().manual_seed(3 + seed)
        B = cfg["B"]
        self.batch = [torch.randn(B, cfg["R"], cfg["Cin"], cfg["N"], cfg["T"], generator=g).to(dev),
                      torch.randint(0, 24, (B,), generator=g).to(dev), torch.randint(0, 7, (B,), generator=g).to(dev),
                      (torch.randn(B, cfg["N"], cfg["T"], generator=g) * 30).to(dev)]
        self.n_params = sum(p.numel() for p in net.parameters() if p.requires_grad)

    def run(self, steps, record=None):
        """`steps` training steps in one `run_epoch` (one host read at its end, like an epoch of the engine)."""
        return self.trainer.run_epoch(_Recording([self.batch] * steps, record, self.dev), gpu_id=self.dev.index,
                                      epoch=1, mode="train")

    @property
    def allreduce_bytes(self):
        opt = self.trainer.optimizer
        return int(opt.allreduce_bytes) if hasattr(opt, "allreduce_bytes") else 4 * (self.n_params + 1)


def time_train_step(ts, steps, warmup, barrier):
    ts.run(warmup)
    barrier()
    events = []
    t0 = time.perf_counter()
    ts.run(steps, record=events)
    barrier()
    wall = time.perf_counter() - t0
    per_step = [events[i].elapsed_time(events[i + 1]) for i in range(steps)]
    return wall, per_step


def cpu_baseline(hp, wl):
    """oracle/dense_torch.py on the host cores: all of them (torch threads = the process's CPU share, at most 16 --
    a GPU box gives one GPU a 16-core share) and ONE thread, on a bounded sample of the same workload."""
    out = {}
    cpu = torch.device("cpu")
    cores = min(len(os.sched_getaffinity(0)), 16)
    for key, threads, budget, frac in (("cpu_baseline", cores, 12.0, 1), ("cpu_baseline_1thread", 1, 10.0, 4)):
        torch.set_num_threads(threads)
        Bs = max(1, (wl["B"] if wl["N"] <= 1024 else max(1, wl["B"] // 8)) // frac)
        dense_reference_step(hp, cpu, Bs)  # warm-up
        t0, n = time.perf_counter(), 0
        while n < 2 or (time.perf_counter() - t0 < budget and n < 40):
            dense_reference_step(hp, cpu, Bs)
            n += 1
        dt = (time.perf_counter() - t0) / n
        out[key] = {
            "value": round(Bs / dt, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": (f"{n} fwd+bwd passes of oracle/dense_torch.py (the reference's dense op sequence) on "
                       f"{Bs} of the {wl['B']} samples per relation, all {wl['R']} relations and both GACN depths, "
                       f"same graph and widths, {threads} torch thread{'s' if threads > 1 else ''}"),
        }
    torch.set_num_threads(cores)
    return out


def stress_object(dev):
    """configs[4] (N = 8192, degree 16, R = 4, B = 64, C = 72 -> 24): the hot-path step and the aggregate kernel's
    roofline at full size, for the default bench line (the full-size parity properties are tests/test_gpu_parity.py)."""
    wl = WORKLOADS["stress"]
    hp = HotPath(wl, dev, seed=0)
    wall, per_step = timed_steps(hp.step, 5, 2, dev, lambda: torch.cuda.synchronize(dev))
    obj = {
        "workload": (f"stress: N={wl['N']} nodes, {wl['E']} undirected edges (+self loops), T={wl['T']}, B={wl['B']}, "
                     f"R={wl['R']} relations, GACN {wl['Cin']}->{wl['Co']} and {wl['hidden']}->{wl['Co']}, forward+backward "
                     "of the hot path"),
        "ms_per_step": round(wall / 5 * 1e3, 3), "ms_per_step_median_hip_events": round(statistics.median(per_step), 3),
        "samples_per_s": round(wl["B"] / (wall / 5), 2),
        "roofline": roofline_object(hp),
        "roofline_dense": time_dense_kernels(hp),
    }
    del hp
    torch.cuda.empty_cache()
    return obj


def time_dense_kernels(hp, reps=20):
    """HIP-event timing of the two dense passes of the attention (second depth's q; both depths have the same [G,N,T]
    problem): `k_scores` (kW, row log-sum-exp over all N columns, pq = softmax @ q, edge coefficients; forward) and
    `k_bwd_dense_col` (the softmax's dense correction of dq; backward).  They read [G,N,T] and write [G,N,T]: their
    bound is the fp32 matrix core + exp issue, not HBM -- reported against the dense fp32 MFMA peak.
    Flops per launch: 2*G*N*N*T for the score tile + 2*G*N*N*T for the payload product (pq, resp. the column sums)."""
    from ms_gat_amd import _lib
    wl, dev = hp.wl, hp.device
    R, B, N, T = wl["R"], wl["B"], wl["N"], wl["T"]
    G = R * B
    L = _lib.lib()
    gs, _keep = hp.graph.on(dev)
    shape = _lib.Shape(R, B, wl["hidden"], wl["Co"], N, T)
    nnz = max(hp.graph.nnz, 1)
    g = torch.Generator(device=dev).manual_seed(5)
    q = layer_norm_t(torch.randn(G, N, T, device=dev, generator=g)) * 0.5
    Wg = hp.layers[1].Wg.detach().contiguous()
    kW, pq, dq = torch.empty_like(q), torch.empty_like(q), torch.zeros_like(q)
    lse = torch.empty(G, N, device=dev)
    E = torch.empty(G, nnz, device=dev)
    delta = torch.randn(G, N, device=dev, generator=g) * 1e-3
    gE = torch.randn(G, nnz, device=dev, generator=g) * 1e-3
    stream = torch.cuda.current_stream(dev)
    ndense = int(L.msgat_dense_scratch_bytes(C.byref(shape)))   # > 0: the passes run on split bf16 / fp16 operands (large N)
    dense = torch.empty(max(ndense, 1), device=dev, dtype=torch.uint8)
    dense_ptr = dense.data_ptr() if ndense else None

    def scores():
        _lib.check(L.msgat_stage_scores(C.byref(shape), C.byref(gs), q.data_ptr(), Wg.data_ptr(), kW.data_ptr(),
                                        lse.data_ptr(), pq.data_ptr(), E.data_ptr(), None, dense_ptr, stream.cuda_stream),
                   "msgat_stage_scores")

    def column():
        _lib.check(L.msgat_stage_dense_column_pass(C.byref(shape), C.byref(gs), q.data_ptr(), kW.data_ptr(), lse.data_ptr(),
                                                   delta.data_ptr(), gE.data_ptr(), dq.data_ptr(), dense_ptr, stream.cuda_stream),
                   "msgat_stage_dense_column_pass")

    n = reps if G * N * N < (1 << 32) else 3
    out = {}
    flops = 4.0 * G * N * N * T
    for name, fn in (("k_scores", scores), ("k_bwd_dense_col", column)):
        for _ in range(2):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for _ in range(n):
            fn()
        t1.record(stream)
        t1.synchronize()
        sec = t0.elapsed_time(t1) * 1e-3 / n
        out[name] = {"us_per_launch": round(sec * 1e6, 2), "flops": flops, "exps": float(G) * N * N,
                     "achieved": round(flops / sec / 1e12, 2), "frac": round(flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
        if ndense:
            # what the matrix core is actually given: per 16x16 tile three 16x16x32 bf16 MFMAs for the scores (72 of 96 slots
            # used: 3 terms x 3 terms, 6 products kept, K = 12) and two 16x16x32 fp16 MFMAs for the payload (2 x 2 terms)
            issued = float(G) * (-(-N // 16)) ** 2 * 5 * 2 * 16 * 16 * 32
            out[name].update(flops_issued=issued, achieved_issued=round(issued / sec / 1e12, 1),
                             frac_of_bf16_peak=round(issued / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                             includes="the pass's two operand-image launches (k_dense_absmax, k_dense_images)")
    if ndense:
        return {"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "matrix": "bf16 split x6 (scores: 3 x v_mfma_f32_16x16x32_bf16 per tile) + fp16 split x4 (payload: 2 x "
                          "v_mfma_f32_16x16x32_f16), fp32 accumulate",
                "dtype": "f32 operands as three bf16 / two scaled fp16 terms; `frac` = useful fp32-equivalent flops over the FP32 "
                         "matrix peak (what the round-5 kernels were priced against), `frac_of_bf16_peak` = issued flops over "
                         f"{MFMA_BF16_PEAK_TFLOPS:.0f} TFLOP/s dense bf16: the passes are bound by instruction issue (4 v_exp, 8 split "
                         "and ~10 other vector instructions per tile beside 5 MFMAs), not by the pipe",
                "kernels": out}
    return {"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "dtype": "f32; scores on v_mfma_f32_16x16x4_f32 (exact), payload product on two-term fp16 operands, fp32 accumulate",
            "matrix": "fp32 scores (3 x v_mfma_f32_16x16x4_f32 per tile) + fp16 split x4 payload (2 x v_mfma_f32_16x16x32_f16): below "
                      "N = 1536 the operand images of the all-split form cost what it gains (profiles/r06/dense_split_lab.txt)",
            "kernels": out}


def collective_us(last):
    """Median duration (us, HIP events on the launch stream, this rank) of the last `last` flat all-reduces recorded by
    `parallel.all_reduce_flat`; None when nothing was recorded (single process).  Clears the record."""
    from ms_gat_amd import parallel
    rec = parallel.collective_events
    if not rec:
        return None
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in rec[-last:]]
    rec.clear()
    return round(statistics.median(us), 1)


class _StdoutToStderr:
    """OS-level redirection of fd 1 to fd 2 (C libraries write there directly), undone on exit."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)      # what the libraries still hold in C stdio buffers goes to stderr too
        except OSError:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no process-group environment: become the launcher -- start one rank per
    GPU with torch.distributed.run as a CHILD process (nothing here has touched the GPU yet) and exit with its code.
    Rank 0 of the child prints the JSON line on the inherited stdout."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


class DropInLoop:
    """The hot path as INTEGRATION.md's Option A issues it: the reference's un-stacked loop over components
    (msgat.py:204) with `ms_gat_amd.GACN` swapped in for its `GACN` (msgat.py:127) -- R separate module calls per depth,
    B groups each, the same parameters / inputs / upstream gradients as `HotPath` (relation r of the stacked tensors)."""

    def __init__(self, hp):
        import ms_gat_amd
        wl = hp.wl
        self.wl, self.graph, self.adj = wl, hp.graph, hp.adj.to(hp.device)
        self.mods, self.xs, self.dzs = [], [], []
        for depth, (stacked_layer, x, dz) in enumerate(zip(hp.layers, hp.xs, hp.dzs)):
            cin = x.shape[2]
            for r in range(wl["R"]):
                m = ms_gat_amd.GACN(cin, wl["Co"], wl["T"]).to(hp.device)
                with torch.no_grad():
                    m.gatt.Wg.copy_(stacked_layer.Wg[r]); m.gatt.alpha.copy_(stacked_layer.alpha[r]); m.W.copy_(stacked_layer.W[r])
                self.mods.append(m)
                self.xs.append(x[r].detach().clone().requires_grad_(True))
                self.dzs.append(dz[r].contiguous())
        self.calls_per_step = 2 * len(self.mods)    # one library call forward, one backward per module

    def step(self, adjacency=None):
        adjacency = self.adj if adjacency is None else adjacency     # the dense [N,N] tensor, as the reference passes it
        for m, x in zip(self.mods, self.xs):
            x.grad = None
            for p in m.parameters():
                p.grad = None
        zs = [m(x, adjacency) for m, x in zip(self.mods, self.xs)]
        torch.autograd.backward(zs, self.dzs)
        return zs


def host_enqueue_us(fn, dev, steps=100, repeats=3):
    """Host time to ENQUEUE one call of `fn` (the GPU may lag behind), and the wall time per call: the best of
    `repeats` rounds (the host of a shared box is noisy; the GPU side is what the other figures are for)."""
    best = None
    for _ in range(repeats):
        for _ in range(10):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        if best is None or host < best[0]:
            best = (host, wall)
    return best[0] / steps * 1e6, best[1] / steps * 1e6


def dropin_object(hp, dev, steps=50):
    """What a maintainer gets who follows INTEGRATION.md section 3, Option A, and nothing else: (a) the hot path as R
    separate GACN module calls per depth, launched eagerly; (b) the msgat72 training step with the components evaluated
    one by one (`stack_components = False`), eager and replayed as a HIP graph -- next to the stacked schedule's figures."""
    wl = hp.wl
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    loop = DropInLoop(hp)
    settle(loop.step, dev, 20.0)
    wall, per = timed_steps(loop.step, steps, 10, dev, sync)
    host, _ = host_enqueue_us(loop.step, dev)            # (before the profiler runs: its hooks linger on some builds)
    host_stacked, _ = host_enqueue_us(hp.step, dev)
    busy, launches = gpu_busy_us(loop.step, dev)
    busy_stacked, launches_stacked = gpu_busy_us(hp.step, dev)
    obj = {
        "workload": (f"hot path as R={wl['R']} separate ms_gat_amd.GACN module calls per depth (reference loop msgat.py:204, "
                     f"module swap msgat.py:127), B={wl['B']} groups per call, dense [N,N] adjacency argument, eager launches"),
        "ms_per_step": round(wall / steps * 1e3, 4), "ms_per_step_median_hip_events": round(statistics.median(per), 4),
        "gpu_busy_ms_per_step": round(busy * 1e-3, 4), "launches_per_step": round(launches, 1),
        "library_calls_per_step": loop.calls_per_step,
        "host_enqueue_us_per_step": round(host, 1), "host_enqueue_us_per_library_call": round(host / loop.calls_per_step, 1),
        "stacked": {"gpu_busy_ms_per_step": round(busy_stacked * 1e-3, 4), "launches_per_step": round(launches_stacked, 1),
                    "host_enqueue_us_per_step": round(host_stacked, 1), "library_calls_per_step": 4,
                    "host_enqueue_us_per_library_call": round(host_stacked / 4, 1)},
    }
    del loop
    cfg3 = dict(CFG4, R=wl["R"])
    for key, kw in (("train_step_unstacked_eager_ms", dict(stacked=False)),
                    ("train_step_unstacked_hip_graph_ms", dict(stacked=False, hip_graph=True)),
                    ("train_step_stacked_eager_ms", dict())):
        ts = TrainStep(cfg3, dev, **kw)
        w, per = time_train_step(ts, 20, 5, sync)
        obj[key] = round(statistics.median(per), 3)
        obj[key.replace("_ms", "_wall_ms")] = round(w / 20 * 1e3, 3)
        del ts
        torch.cuda.empty_cache()
    obj["train_step_unstacked_over_stacked"] = round(obj["train_step_unstacked_eager_wall_ms"] / obj["train_step_stacked_eager_wall_ms"], 3)
    return obj


def capture(fn, dev, warm=3):
    """`fn` (launches only: no host synchronisation, no allocation outside the graph pool) as one HIP graph; returns the
    replay callable.  Warm-up runs on a side stream first, as torch requires before a capture."""
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay, g


def gpu_busy_us(fn, dev, steps=20):
    """Sum of the kernel durations of one call of `fn` (torch's profiler, device activity only): what the step costs when
    the host never makes the GPU wait."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize(dev)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
    ev = [e for e in prof.key_averages() if e.device_time_total > 0]
    return sum(e.device_time_total for e in ev) / steps, sum(e.count for e in ev) / steps


def hbm_kernel_table(hp, dev, steps=10):
    """Per-kernel GPU time of one hot-path step (torch's profiler) and, for the four HBM-bound slab / channel-mixing
    passes of the second depth, the fraction of the 8 TB/s roofline their algorithmic bytes reach:
      k_project_mfma   u = W x, q = alpha . x         4 G P (C + Co + 1)
      k_agg_ring/lds   v = E u                        2 . 4 G Co P
      k_agg_sddmm      du = E^T dz and the SDDMM      3 . 4 G Co P
      k_chanpair_*     dW | dalpha AND dx, one pass   4 G P (Co + 1 + 2 C)
    (in-step durations: the operands of one kernel are partly cache-resident from its producer)."""
    import re
    from torch.profiler import ProfilerActivity, profile
    wl = hp.wl
    G, P, C, Co = wl["R"] * wl["B"], wl["N"] * wl["T"], wl["hidden"], wl["Co"]
    algo = {"k_project_mfma": 4 * G * P * (C + Co + 1), "k_agg_ring": 8 * G * Co * P, "k_agg_sddmm": 12 * G * Co * P,
            "k_chanpair_glds": 4 * G * P * (Co + 1 + 2 * C), "k_chanpair_mfma": 4 * G * P * (Co + 1 + C)}
    torch.cuda.synchronize(dev)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            hp.step()
        torch.cuda.synchronize(dev)
    rows = {}
    for e in prof.key_averages():
        if e.device_time_total <= 0:
            continue
        name = re.sub(r"\(.*", "", e.key).replace("void ", "").replace("msgat::", "")
        us = e.device_time_total / max(e.count, 1)
        row = {"us": round(us, 1), "per_step": round(e.count / steps, 1)}
        for key, nbytes in algo.items():
            # the second depth's launch is the large one; the contraction forms appear once per step at these widths
            if name.startswith(key) and e.count / steps <= 1.0:
                row["algorithmic_bytes"] = nbytes
                row["frac_of_hbm_peak"] = round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)
        rows[name] = row
    return dict(sorted(rows.items(), key=lambda kv: -kv[1]["us"] * kv[1]["per_step"]))


def widths_object(hidden, dev, steps=20):
    """The hot-path step at the widths of the other two models of the reference's registry (main.py:17,
    msgat.py:220-229): msgat48 (48 -> 16) and msgat96 (96 -> 32) on the headline graph (PEMSD7-like, R = 3, B = 32)."""
    wl = dict(WORKLOADS["pemsd7"], hidden=hidden, Co=hidden // 3)
    hp = HotPath(wl, dev, seed=0)
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    settle(hp.step, dev, 30.0)
    wall, per = timed_steps(hp.step, steps, 5, dev, sync)
    obj = {"workload": (f"pemsd7 graph, msgat{hidden} widths: GACN {wl['Cin']}->{wl['Co']} and {hidden}->{wl['Co']}, "
                        f"R={wl['R']}, B={wl['B']}, forward+backward of the hot path"),
           "ms_per_step": round(wall / steps * 1e3, 4), "ms_per_step_median_hip_events": round(statistics.median(per), 4),
           "samples_per_s": round(wl["B"] / (wall / steps), 2), "kernels": hbm_kernel_table(hp, dev)}
    del hp
    torch.cuda.empty_cache()
    # the whole training step of that model (engine.Trainer, R = 3, B = 32): its wide one-pass backward forms (two
    # z-blocks over B at 96 channels, each staging all of A) are timed here, with their names as the library reports them
    from ms_gat_amd import _lib
    co, P = hidden // 3, wl["N"] * wl["T"]
    ts = TrainStep(dict(CFG4, R=wl["R"], hidden=hidden), dev)
    w, per = time_train_step(ts, 20, 5, sync)
    obj["train_step"] = {
        "workload": f"msgat{hidden} training step (engine.Trainer), N={wl['N']}, R={wl['R']}, B={wl['B']}, T={wl['T']}",
        "ms_per_step": round(w / 20 * 1e3, 3), "ms_per_step_median_hip_events": round(statistics.median(per), 3),
        "samples_per_s": round(wl["B"] / (w / 20), 2), "trainable_parameters": ts.n_params,
        "convolution_backward_forms": {
            "gacn_projection": _lib.contract_form_name(co + 1, hidden, False, P, True),
            "merged_channel_mixing": _lib.contract_form_name(4 * co + 2, hidden, True, P, True),
            "residual_convolution": _lib.contract_form_name(hidden, hidden, True, P, True)}}
    del ts
    torch.cuda.empty_cache()
    # the same step replayed as ONE HIP graph (engine.Trainer(hip_graph=True)): msgat48 is 2.8 ms of GPU work in ~125
    # launches, which a slow or shared host cannot enqueue in time -- the replay is the GPU-side figure
    try:
        tg = TrainStep(dict(CFG4, R=wl["R"], hidden=hidden), dev, hip_graph=True)
        wg, perg = time_train_step(tg, 20, 8, sync)
        obj["train_step"]["hip_graph_replay_ms_per_step"] = round(wg / 20 * 1e3, 3)
        obj["train_step"]["hip_graph_replay_ms_per_step_median_hip_events"] = round(statistics.median(perg), 3)
        del tg
    except Exception as exc:  # noqa: BLE001 -- an extra figure: the eager one above stands on its own
        obj["train_step"]["hip_graph_replay_error"] = f"{type(exc).__name__}: {exc}"[:200]
    torch.cuda.empty_cache()
    return obj


def small_graph_object(name, dev, steps=100):
    """A workload whose step is ~0.2 ms of GPU work in 17 launches (configs[1], PEMSD4: N = 307, 3 features, B = 64, ONE
    relation): launched kernel by kernel the host cannot keep the GPU busy, so the step is also captured once and
    replayed as ONE HIP graph (what `engine.Trainer(hip_graph=True)` does for the training step) -- `ms_per_step` is the
    replay; the eager-launch figure and the GPU-busy time (sum of kernel durations) stand next to it.  Forward alone
    likewise, against the same ops in PyTorch-ROCm eager (the dense [B,N,N] sequence of oracle/dense_torch.py)."""
    wl = WORKLOADS[name]
    hp = HotPath(wl, dev, seed=0)
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    settle(hp.step, dev, 20.0)
    wall_e, per_e = timed_steps(hp.step, steps, 10, dev, sync)
    _, fwd_e = timed_steps(hp.forward_only, 50, 5, dev, sync)
    busy, launches = gpu_busy_us(hp.step, dev)
    busy_f, launches_f = gpu_busy_us(hp.forward_only, dev)
    step_replay, _g1 = capture(hp.step, dev)
    fwd_replay, _g2 = capture(hp.forward_only, dev)
    settle(step_replay, dev, 20.0)
    wall_g, per_g = timed_steps(step_replay, steps, 10, dev, sync)
    _, fwd_g = timed_steps(fwd_replay, 50, 5, dev, sync)
    _, eager_fwd = timed_steps(lambda: dense_reference_step(hp, dev, wl["B"], backward=False), 10, 2, dev, sync)
    _, eager_all = timed_steps(lambda: dense_reference_step(hp, dev, wl["B"], backward=True), 10, 2, dev, sync)
    # the forward alone is 6 launches from two library calls, back to back on the stream: launched eagerly it is not
    # host-bound, and a replayed graph pays ~1 us of node scheduling per kernel -- the faster of the two is `forward_ms`
    fwd = min(statistics.median(fwd_g), statistics.median(fwd_e))
    # the whole msgat72 TRAINING step at this size (engine.Trainer: forward, loss, backward, Adam; ~150 launches for a
    # few ms of GPU work): launched eagerly and as the HIP graph `Trainer(hip_graph=True)` captures
    cfg = dict(N=wl["N"], E=wl["E"], B=wl["B"], R=wl["R"], Cin=wl["Cin"], T=wl["T"])
    train = {}
    for key, graph in (("train_step_eager_launch_ms", False), ("train_step_hip_graph_ms", True)):
        ts = TrainStep(cfg, dev, hip_graph=graph)
        w, per = time_train_step(ts, 20, 5, sync)
        train[key] = round(statistics.median(per), 4)
        del ts
    torch.cuda.empty_cache()
    return {
        **train,
        "workload": (f"{name}: N={wl['N']} nodes, {wl['E']} undirected edges (+self loops), T={wl['T']}, B={wl['B']}, "
                     f"R={wl['R']} relation{'s' if wl['R'] > 1 else ''}, GACN {wl['Cin']}->{wl['Co']} and "
                     f"{wl['hidden']}->{wl['Co']}, forward+backward"),
        "launch_mode": "hip_graph_replay (one graph = one step: both depths, forward + backward)",
        "ms_per_step": round(wall_g / steps * 1e3, 4), "ms_per_step_median_hip_events": round(statistics.median(per_g), 4),
        "hip_graph_replay_ms_per_step": round(wall_g / steps * 1e3, 4),
        "eager_launch_ms_per_step": round(wall_e / steps * 1e3, 4),
        "eager_launch_ms_per_step_median_hip_events": round(statistics.median(per_e), 4),
        "gpu_busy_ms_per_step": round(busy * 1e-3, 4), "launches_per_step": round(launches, 1),
        "samples_per_s": round(wl["B"] / (wall_g / steps), 2),
        "forward_ms": round(fwd, 4), "forward_launch_mode": ("eager launches" if fwd == statistics.median(fwd_e) else "hip_graph_replay"),
        "forward_eager_launch_ms": round(statistics.median(fwd_e), 4), "forward_hip_graph_replay_ms": round(statistics.median(fwd_g), 4),
        "forward_gpu_busy_ms": round(busy_f * 1e-3, 4), "forward_launches": round(launches_f, 1),
        "eager_rocm_forward_ms": round(statistics.median(eager_fwd), 3),
        "eager_rocm_fwd_bwd_ms": round(statistics.median(eager_all), 3),
        # like for like: both sides launched eagerly from Python (round-4 advisor finding: the replayed step against an
        # eager baseline mixed launch modes); the replayed figures carry their launch mode in the key
        "speedup_vs_eager_rocm_forward": round(statistics.median(eager_fwd) / statistics.median(fwd_e), 2),
        "speedup_vs_eager_rocm_forward_hip_graph": round(statistics.median(eager_fwd) / statistics.median(fwd_g), 2),
        "speedup_vs_eager_rocm_fwd_bwd": round(statistics.median(eager_all) / statistics.median(per_e), 2),
        "speedup_vs_eager_rocm_fwd_bwd_hip_graph": round(statistics.median(eager_all) / statistics.median(per_g), 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="pemsd7", choices=sorted(WORKLOADS))
    ap.add_argument("--no-baselines", action="store_true", help="skip the CPU / eager baselines and the secondary objects")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # under torch.distributed.run
    if not launched and args.gpus > 1:
        raise SystemExit(respawn_under_torchrun(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if SHARE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    n_dev = torch.cuda.device_count()
    if local >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but this box has {n_dev} GPU(s) -- one process per GPU "
                         "(RCCL refuses two ranks on one device).  To rehearse the N > 1 path on one GPU: "
                         "MSGAT_BENCH_SHARE_GPU=1 python bench.py --gpus N (all ranks on cuda:0, gloo transport).")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = launched or FORCE_DIST
    backend = "gloo" if SHARE_GPU else "nccl"
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner (and gloo its connection report) on fd 1 when a communicator comes up: stdout must
        # carry the ONE JSON line and nothing else, so fd 1 points at stderr until the first collective has run
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            first = torch.ones(1, device=dev)
            dist.all_reduce(first)
            torch.cuda.synchronize(dev)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not multi:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    wl = WORKLOADS[args.workload]
    hp = HotPath(wl, dev, seed=rank)
    settle(lambda: hp.step(allreduce=multi), dev, agree=max_over_ranks if multi else None)
    from ms_gat_amd import parallel
    if multi:
        parallel.collective_events = []     # HIP events around every flat all-reduce (launch stream)
    hot_wall, hot_steps = timed_steps(lambda: hp.step(allreduce=multi), args.steps, args.warmup, dev, barrier)
    hot_wall = max_over_ranks(hot_wall)
    hot_allreduce_us = collective_us(args.steps)
    workload = (f"{args.workload}: N={wl['N']} nodes, {wl['E']} undirected edges (+self loops, sym-normalised), "
                f"T={wl['T']}, B={wl['B']}/GPU, R={wl['R']} relations, GACN {wl['Cin']}->{wl['Co']} and "
                f"{wl['hidden']}->{wl['Co']} (msgat72 widths), forward+backward of the hot path"
                + (", flat all-reduce of its parameter gradients" if world > 1 else ""))
    value = round(wl["B"] * world / (hot_wall / args.steps), 2)
    out = {"metric": METRIC, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           # what `value` is, at EVERY N and in both launch modes: compare value with value, and
           # full_step_cfg4.value with full_step_cfg4.value
           "value_kind": "hot_path",
           "value": value, "ms_per_step": round(hot_wall / args.steps * 1e3, 4),
           "ms_per_step_median_hip_events": round(statistics.median(hot_steps), 4),
           "config": {"workload": workload, "global_batch": wl["B"] * world,
                      "parallelism": (f"batch-sharded x{world}, one flat all-reduce of the hot path's parameter gradients "
                                      f"per step ({hp.sync.nbytes} bytes)" if world > 1 else "single GPU (batch-sharded x1)")},
           "node_updates_per_s": round(value * wl["R"] * wl["T"] * wl["N"], 1),
           # the step's one collective, this rank's median over the timed steps (null without a process group): what
           # separates transport from compute in a scaling curve
           "allreduce_us": hot_allreduce_us,
           "settle_ms_before_warmup": SETTLE_MS,   # untimed steps in front of the W warm-up steps (clock / cache ramp)
           "launch": ("torch.distributed.run" if launched else "forced process group" if FORCE_DIST else "plain"),
           "transport": (("gloo, all ranks share cuda:0 (REHEARSAL, not a scaling measurement)" if SHARE_GPU else "rccl")
                         if multi else None)}
    # kept under its round-2 name as well: the same quantity as the top-level value
    out["hot_path"] = {"workload": workload, "value": value, "ms_per_step": out["ms_per_step"],
                       "ms_per_step_median_hip_events": out["ms_per_step_median_hip_events"]}

    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    with_secondary = not args.no_baselines
    if with_secondary:
        # configs[3]: the whole training step, B = 32 per GPU, all gradients in one flat all-reduce.  Every rank runs
        # it (the collective needs them all); timed like `value`: barrier on both sides, MAX over ranks.
        # (no try/except around it with several ranks: a rank that failed alone would leave its peers inside the next
        # collective until the process-group timeout -- the exception propagates and torchrun tears the job down)
        def cfg4_object():
            ts = TrainStep(CFG4, dev, seed=rank)
            steps = max(5, min(args.steps, 20))
            wall, per_step = time_train_step(ts, steps, max(2, min(args.warmup, 5)), barrier)
            wall = max_over_ranks(wall)
            allreduce_us = collective_us(steps)
            out["full_step_cfg4"] = {
                "workload": (f"configs[3] per-GPU workload: msgat72 TRAINING step through engine.Trainer -- forward of "
                             f"R={CFG4['R']} components, Huber loss + metrics, backward, one flat all-reduce of "
                             f"{ts.n_params} gradients ({ts.allreduce_bytes} bytes), Adam; PEMSD7-like N={CFG4['N']}, "
                             f"{CFG4['E']} edges, T={CFG4['T']}, B={CFG4['B']}/GPU"),
                "value_kind": "train_step_cfg4", "steps": steps,
                "ms_per_step": round(wall / steps * 1e3, 4),
                "ms_per_step_median_hip_events": round(statistics.median(per_step), 4),
                "value": round(CFG4["B"] * world / (wall / steps), 2), "unit": "samples/s",
                "global_batch": CFG4["B"] * world, "trainable_parameters": ts.n_params,
                "allreduce_bytes_per_step": ts.allreduce_bytes if world > 1 else 0, "allreduce_us": allreduce_us,
                "allreduce_bytes_per_step_when_sharded": ts.allreduce_bytes,
            }
            del ts
            torch.cuda.empty_cache()
            if not multi:   # the same step replayed as one HIP graph (Trainer(hip_graph=True)): independent of the host's speed
                tg = TrainStep(CFG4, dev, seed=rank, hip_graph=True)
                wg, perg = time_train_step(tg, steps, max(2, min(args.warmup, 5)), barrier)
                out["full_step_cfg4"]["hip_graph_replay_ms_per_step"] = round(wg / steps * 1e3, 4)
                out["full_step_cfg4"]["hip_graph_replay_ms_per_step_median_hip_events"] = round(statistics.median(perg), 4)
                del tg
                torch.cuda.empty_cache()

        if multi:
            cfg4_object()
        else:
            try:
                cfg4_object()
            except RuntimeError as e:
                out["full_step_cfg4"] = {"error": str(e).splitlines()[0][:160]}

    if rank == 0:
        out["roofline"] = roofline_object(hp)
        out["roofline_dense"] = time_dense_kernels(hp)
    if rank == 0 and not multi and with_secondary:
        ms_per_step = out["ms_per_step_median_hip_events"]
        # PyTorch-ROCm eager on the same GPU: the reference's dense op sequence, median of 10 event-timed passes
        for bw, key in ((False, "eager_rocm_forward"), (True, "eager_rocm_fwd_bwd")):
            _, per = timed_steps(lambda: dense_reference_step(hp, dev, wl["B"], backward=bw), 10, 2, dev, sync)
            out[key + "_ms"] = round(statistics.median(per), 3)
        _, per = timed_steps(hp.forward_only, 30, 5, dev, sync)
        out["forward_ms"] = round(statistics.median(per), 4)
        out["speedup_vs_eager_rocm_forward"] = round(out["eager_rocm_forward_ms"] / out["forward_ms"], 2)
        out["speedup_vs_eager_rocm_fwd_bwd"] = round(out["eager_rocm_fwd_bwd_ms"] / ms_per_step, 2)

        # secondary: the whole msgat72 training step at the headline config's R (engine.Trainer: fused loss + metrics,
        # FlatAdam), its HIP-graph replay and the eager PyTorch-ROCm op sequence of the same step
        try:
            cfg3 = dict(CFG4, R=wl["R"])
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            mem0 = torch.cuda.memory_allocated(dev)
            ts = TrainStep(cfg3, dev)
            wall, per = time_train_step(ts, 20, 5, sync)
            peak_lib = (torch.cuda.max_memory_allocated(dev) - mem0) / 2**20
            out["full_step_cfg3"] = {
                "workload": f"msgat72 training step (engine.Trainer), N={cfg3['N']}, R={cfg3['R']}, B={cfg3['B']}, T={cfg3['T']}",
                "ms_per_step": round(wall / 20 * 1e3, 3), "ms_per_step_median_hip_events": round(statistics.median(per), 3),
                "value": round(cfg3["B"] / (wall / 20), 2), "unit": "samples/s",
                "trainable_parameters": ts.n_params, "allreduce_bytes_per_step_when_sharded": ts.allreduce_bytes,
            }
            del ts
            out["full_model_samples_per_s"] = out["full_step_cfg3"]["value"]
            ts = TrainStep(cfg3, dev, hip_graph=True)
            wall, per = time_train_step(ts, 20, 5, sync)
            out["full_step_cfg3"]["hip_graph_replay_ms_per_step"] = round(wall / 20 * 1e3, 3)
            del ts
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
            mem0 = torch.cuda.memory_allocated(dev)
            ts = TrainStep(cfg3, dev, dense=True)
            wall, per = time_train_step(ts, 10, 3, sync)
            out["full_step_cfg3"]["eager_rocm_ms_per_step"] = round(wall / 10 * 1e3, 3)
            # peak device memory of a step above what was allocated before the model was built (parameters, optimizer state,
            # batch, saved activations, workspaces): the library's step and the reference op sequence
            out["full_step_cfg3"]["peak_device_memory_mib"] = round(peak_lib, 1)
            out["full_step_cfg3"]["eager_rocm_peak_device_memory_mib"] = round((torch.cuda.max_memory_allocated(dev) - mem0) / 2**20, 1)
            del ts
            torch.cuda.empty_cache()
        except RuntimeError as e:
            out["full_model_error"] = str(e).splitlines()[0][:160]
            import traceback
            traceback.print_exc()       # (stderr: stdout carries the JSON line only)

        if args.workload == "pemsd7":
            try:
                out["pemsd4"] = small_graph_object("pemsd4", dev)
            except RuntimeError as e:
                out["pemsd4"] = {"error": str(e).splitlines()[0][:160]}
            for hidden in (48, 96):
                try:
                    out[f"widths{hidden}"] = widths_object(hidden, dev)
                except RuntimeError as e:
                    out[f"widths{hidden}"] = {"error": str(e).splitlines()[0][:160]}
            out["widths72_kernels"] = hbm_kernel_table(hp, dev)
            try:
                out["dropin_loop"] = dropin_object(hp, dev)
            except RuntimeError as e:
                out["dropin_loop"] = {"error": str(e).splitlines()[0][:160]}
            try:
                out["stress"] = stress_object(dev)
            except RuntimeError as e:
                out["stress"] = {"error": str(e).splitlines()[0][:160]}
        out.update(cpu_baseline(hp, wl))
    if rank == 0:
        print(json.dumps(ordered_for_the_record(out)), flush=True)
    if multi:
        barrier()
        dist.destroy_process_group()


def ordered_for_the_record(out: dict) -> dict:
    """The same line, keys re-ordered: the driver's record keeps the TAIL of stdout, and the line is ~12 KB.  Large tables
    first; then a `summary` of the figures a reader looks for (every step time of the line in one place); the contract's own
    keys, `config`, `roofline` and `cpu_baseline` last -- all inside the final 2 KB."""
    contract = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"]
    last = ["roofline", "cpu_baseline"]
    def get(path, default=None):
        cur = out
        for k in path.split("."):
            if not isinstance(cur, dict) or k not in cur:
                return default
            cur = cur[k]
        return cur
    summary = {k: v for k, v in {
        "ms_per_step": out.get("ms_per_step"),
        "ms_per_step_median_hip_events": out.get("ms_per_step_median_hip_events"),
        "full_step_cfg3_ms": get("full_step_cfg3.ms_per_step"),
        "full_step_cfg3_hip_graph_ms": get("full_step_cfg3.hip_graph_replay_ms_per_step"),
        "full_step_cfg4_ms": get("full_step_cfg4.ms_per_step"),
        "full_step_cfg4_samples_per_s": get("full_step_cfg4.value"),
        "stress_ms_per_step": get("stress.ms_per_step"),
        "stress_dense_us": [get("stress.roofline_dense.kernels.k_scores.us_per_launch"),
                            get("stress.roofline_dense.kernels.k_bwd_dense_col.us_per_launch")],
        "stress_matrix": get("stress.roofline_dense.matrix"),
        "dense_us": [get("roofline_dense.kernels.k_scores.us_per_launch"), get("roofline_dense.kernels.k_bwd_dense_col.us_per_launch")],
        "dense_matrix": get("roofline_dense.matrix"),
        "pemsd4_ms_per_step": get("pemsd4.ms_per_step"),
        "dropin_loop_ms": get("dropin_loop.ms_per_step"),
        "allreduce_us": out.get("allreduce_us"),
    }.items() if v is not None and v != [None, None]}
    size = lambda k: len(json.dumps(out[k]))  # noqa: E731
    rest = [k for k in out if k not in contract and k not in last]
    rest.sort(key=lambda k: -size(k))          # the bulky tables first
    ordered = {k: out[k] for k in rest}
    ordered["summary"] = summary
    for k in contract + last:
        if k in out:
            ordered[k] = out[k]
    return ordered


if __name__ == "__main__":
    main()
