"""Train / validate / evaluate loop with the reference's API, one process per GPU.

Reference: /root/reference/src/engine.py -- `Engine.run_epoch(data, gpu_id, epoch, mode)` (:40),
`Trainer(model, loss_delta, out_dir).fit((train, val), gpu_id)` (:104-133), `.save/.load(ckpt)`
(:135-157), `Evaluator(model, delta, out_dir, ckpt).eval(loader, gpu_id)` (:160-168);
src/loss.py:51-52 (Huber), src/metrics.py:20-35 (MAE / MAPE / RMSE).

Differences that are the point of this build:
  * fp32 end to end (the reference wraps the forward in CUDA AMP, engine.py:54; the parity bar of
    the hot path is fp32).  No loss scaling is needed, but checkpoints carry the state dict of a
    default `GradScaler` under the reference's key so that either side loads the other's files.
  * the step tail runs in the library: ONE pass over the prediction gives the
    Huber loss and the metric sums (`ops.huber_metrics`), which stay on the device -- the host reads them
    once per epoch instead of four `.item()` syncs per batch (engine.py:66, metrics.py:24,30,34) -- and
    Adam updates every parameter in one launch over flat state buffers (`FlatAdam`).
  * under `torch.distributed` every rank runs its batch shard and the gradients are averaged with ONE
    flat all-reduce of the buffer Adam reads (`nn.DataParallel`, main.py:52-55, is not used).
  * `hip_graph="auto"` (the default on the GPU) captures a step in a HIP graph once its batch shape recurs and replays it.
CPU tensors (the host-logic tests run a small CPU `nn.Module` through this loop) take plain PyTorch
ops for loss and optimizer; the library has no CPU path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from time import localtime, strftime
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn, optim
from torch.optim import lr_scheduler

from . import parallel


def huber_loss(output: torch.Tensor, target: torch.Tensor, delta: float = 1.0) -> torch.Tensor:
    """Mean Huber loss (loss.py:51-52): quadratic within `delta`, linear beyond."""
    err = (output - target).abs()
    return torch.where(err <= delta, 0.5 * err * err, delta * err - 0.5 * delta * delta).mean()


class HuberLoss(nn.Module):
    def __init__(self, delta: float = 1.0):
        super().__init__()
        self.delta = delta

    def forward(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if output.is_cuda:
            from . import ops
            return ops.huber_metrics(output, target, self.delta)
        return huber_loss(output, target, self.delta)


class Metrics:
    """Running MAE / MAPE / RMSE with the reference's definitions (metrics.py:11-38): MAPE sums
    |err/y| over entries with y > mask_value but divides by ALL entries, like the reference.
    Totals [AE, APE, SE, sum of batch losses] accumulate on the device in float64; properties synchronise
    when read.  Under a process group a rank adds its share n_r / n_b of every global batch's loss
    (`loss_weight`), so the all-reduced total is the sum of the reference's per-batch mean losses
    (engine.py:66-67) -- also for uneven shards -- and `batches` counts GLOBAL batches (every rank sees them all)."""

    def __init__(self, mask_value: float = 0.0):
        self.mask_value = mask_value
        self.n = 0
        self.batches = 0
        self._sums: Optional[torch.Tensor] = None

    def reset(self) -> None:
        """Start a new epoch on the SAME totals buffer (captured loss kernels hold its address)."""
        if self._sums is not None:
            self._sums.zero_()
        self.n = self.batches = 0

    def totals(self, device) -> torch.Tensor:
        if self._sums is None:
            self._sums = torch.zeros(4, device=device, dtype=torch.float64)
        return self._sums

    def count(self, y_true: torch.Tensor) -> None:
        """A batch whose sums went into `totals` inside the fused loss kernel."""
        self.n += y_true.numel()
        self.batches += 1

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor, loss: Optional[torch.Tensor] = None,
               count: bool = True, loss_weight: float = 1.0) -> None:
        err = (y_pred.detach() - y_true).double()
        truth = y_true.double()
        mask = truth > self.mask_value
        ape = torch.where(mask, (err / torch.where(mask, truth, torch.ones_like(truth))).abs(), torch.zeros_like(err))
        zero = err.new_zeros(())
        s = torch.stack([err.abs().sum(), 100.0 * ape.sum(), (err * err).sum(),
                         zero if loss is None else loss.detach().double() * loss_weight])
        self.totals(y_pred.device).add_(s)
        if count:
            self.count(y_true)

    def _get(self, i: int) -> float:
        return 0.0 if self._sums is None else float(self._sums[i].item())

    def all_reduce(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self._sums is not None:
            buf = torch.cat([self._sums, self._sums.new_tensor([float(self.n)])])
            dist.all_reduce(buf)
            self._sums.copy_(buf[:4])      # in place: the kernels of a captured step keep adding to THIS buffer
            self.n = int(buf[4].item())

    @property
    def MAE(self) -> float:
        return self._get(0) / max(self.n, 1)

    @property
    def MAPE(self) -> float:
        return self._get(1) / max(self.n, 1)

    @property
    def RMSE(self) -> float:
        return (self._get(2) / max(self.n, 1)) ** 0.5

    @property
    def loss(self) -> float:
        """Mean over the (global) batches of the batch-mean loss, the reference's `loss_ave` (engine.py:66-67)."""
        return self._get(3) / max(self.batches, 1)

    def todict(self):
        return {"MAE": self.MAE, "MAPE": self.MAPE, "RMSE": self.RMSE}


class FlatAdam(optim.Optimizer):
    """`torch.optim.Adam` (engine.py:106; L2 weight decay on the gradient, bias correction, eps outside the
    square root) as ONE launch over flat state buffers (`msgat_adam_step`, csrc/tail.hip).

    The gradients are gathered into one flat fp32 buffer (the buffer a multi-rank step all-reduces, so the
    collective and the update share it), both moments are flat, and the parameters stay where they are -- views
    of the parameter bank (`stacked.ParamBank`) -- reached through a device table of 2048-element chunks.  The
    step count and the learning rate live in device memory: a captured launch follows `StepLR`.
    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (per-parameter `step`, `exp_avg`,
    `exp_avg_sq`), so checkpoints interchange with the reference's optimizer."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam takes one parameter group (the reference trains with one, engine.py:106)")
        self._params: List[nn.Parameter] = []
        self._loaded_steps: Dict[int, float] = {}    # steps that arrived through load_state_dict, by id(param)
        self._host_steps: List[int] = []             # host mirror of the per-parameter step counts
        self._captured_active: Optional[tuple] = None
        self._tables: Dict[tuple, tuple] = {}
        self._gather_tables: dict = {}
        self._lr_on_device = None
        self._last_active: Optional[tuple] = None
        self.flat_grad = None
        self.numel = -1

    # -- flat buffers ----------------------------------------------------------------------------------------
    def _build(self) -> None:
        self._params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not self._params:
            raise ValueError("no trainable parameters")
        dev = self._params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs in libmsgat_hip.so on the GPU; use torch.optim.Adam for CPU parameters")
        self._offsets, off = [], 0
        for p in self._params:
            self._offsets.append(off)
            off += p.numel()
        # A rebuild with the same layout (load_state_dict on a live optimizer) keeps every device buffer and writes
        # into it: step graphs captured earlier hold these addresses (and those of the chunk tables).
        reuse = (self.flat_grad is not None and self.numel == off and self.flat_grad.device == dev
                 and len(self._host_steps) == len(self._params))
        self.numel = off
        self._host_steps = [int(self._loaded_steps.get(id(p), 0)) for p in self._params]
        steps = torch.tensor([float(t) for t in self._host_steps])
        if reuse:
            self._dev_steps.copy_(steps)
            self._dev_lr.fill_(float(self.param_groups[0]["lr"]))
        else:
            self.flat_grad = torch.zeros(off + 1, device=dev, dtype=torch.float32)   # + the rank's weight (all-reduce)
            self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
            self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
            self._dev_steps = steps.to(dev)
            self._dev_lr = torch.tensor([float(self.param_groups[0]["lr"])], device=dev)
            self._tables.clear()
            self._gather_tables.clear()
        self._grad_views = [self.flat_grad[o:o + p.numel()].view_as(p) for p, o in zip(self._params, self._offsets)]
        self._lr_on_device = float(self.param_groups[0]["lr"])
        for i, (p, o) in enumerate(zip(self._params, self._offsets)):
            old = self.state.get(p, {})
            m, v = self.exp_avg[o:o + p.numel()].view_as(p), self.exp_avg_sq[o:o + p.numel()].view_as(p)
            if "exp_avg" in old:      # state that arrived through load_state_dict: move it into the flat buffers
                if old["exp_avg"].data_ptr() != m.data_ptr():
                    m.copy_(old["exp_avg"])
                    v.copy_(old["exp_avg_sq"])
            elif reuse:               # no state for this parameter in what was loaded: fresh moments
                m.zero_()
                v.zero_()
            self.state[p] = {"step": torch.tensor(float(self._host_steps[i])), "exp_avg": m, "exp_avg_sq": v}

    def buffer_token(self) -> tuple:
        """Addresses a captured step graph depends on; a change means such graphs must be dropped."""
        if self.flat_grad is None:
            return ()
        return (self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self._dev_steps.data_ptr(), self._dev_lr.data_ptr())

    def _table(self, active: tuple):
        """Device chunk table of the parameters that have a gradient (torch's Adam skips the others)."""
        key = (active, tuple(p.data_ptr() for p in self._params))
        hit = self._tables.get(key)
        if hit is None:
            from . import _lib
            chunk = int(_lib.lib().msgat_adam_chunk_elems())
            ptrs, offs, lens, tens, act = [], [], [], [], []
            for i, (p, o, on) in enumerate(zip(self._params, self._offsets, active)):
                if not on:
                    continue
                if not p.is_contiguous():
                    raise RuntimeError("FlatAdam needs contiguous parameters")
                act.append(i)
                for s in range(0, p.numel(), chunk):
                    ptrs.append(p.data_ptr() + 4 * s)
                    offs.append(o + s)
                    lens.append(min(chunk, p.numel() - s))
                    tens.append(i)
            dev = self.flat_grad.device
            hit = (torch.tensor(ptrs, dtype=torch.int64).to(dev), torch.tensor(offs, dtype=torch.int64).to(dev),
                   torch.tensor(lens, dtype=torch.int32).to(dev), torch.tensor(tens, dtype=torch.int32).to(dev),
                   torch.tensor(act, dtype=torch.int32).to(dev), len(ptrs), len(act))
            self._tables[key] = hit       # never evicted: a captured step may hold the addresses of an older table
        return hit

    def _gather(self, grads: List[torch.Tensor], active: tuple, weight: float) -> None:
        """flat = weight * gradients, flat[-1] = weight, in ONE launch (`parallel.gather_scaled`)."""
        offsets = [o for o, on in zip(self._offsets, active) if on]
        parallel.gather_scaled(self.flat_grad, grads, offsets, weight, self.numel, self._gather_tables)

    def sync_lr(self) -> None:
        """Write the group's learning rate (a host float the scheduler edits) into device memory when it changed."""
        lr = float(self.param_groups[0]["lr"])
        if self._lr_on_device is not None and lr != self._lr_on_device:
            self._dev_lr.fill_(lr)
            self._lr_on_device = lr

    def note_replayed_step(self) -> None:
        """A HIP-graph replay ran the captured update: advance the host-side mirror of the step counts."""
        for i, on in enumerate(self._captured_active or ()):
            self._host_steps[i] += int(on)

    @torch.no_grad()
    def step(self, closure=None, rank_weight: Optional[float] = None):
        """One update.  `rank_weight` (this rank's sample count) makes it a data-parallel step: the flat gradient
        buffer is all-reduced as sum(w g) / sum(w) -- the gradient of the mean loss over the global batch, also for
        uneven shards -- before the update, in the one collective of the step."""
        from . import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._params or self.flat_grad is None or self._params[0].device != self.flat_grad.device:
            self._build()
        active = tuple(p.grad is not None for p in self._params)
        grads = [p.grad for p in self._params if p.grad is not None]
        divisor = 0
        if rank_weight is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # one launch before the collective (gather, scaled by the rank's weight, which also rides in the last
            # element) and one after it (the update divides by the summed weight on the way in)
            if active != self._last_active:
                self.flat_grad.zero_()           # parameters without a gradient contribute zeros on every rank
            if grads:
                self._gather(grads, active, float(rank_weight))
            else:
                self.flat_grad[self.numel:].fill_(float(rank_weight))
            parallel.all_reduce_flat(self.flat_grad)
            divisor = self.flat_grad.data_ptr() + 4 * self.numel
        elif grads:
            torch._foreach_copy_([v for v, on in zip(self._grad_views, active) if on], grads)
        self._last_active = active
        self.sync_lr()
        ptrs, offs, lens, tens, act, n, n_act = self._table(active)
        g = self.param_groups[0]
        st = _lib.lib().msgat_adam_step(ptrs.data_ptr(), offs.data_ptr(), lens.data_ptr(), tens.data_ptr(), n, act.data_ptr(),
                                        n_act, self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                        self._dev_steps.data_ptr(), self._dev_lr.data_ptr(), float(g["betas"][0]),
                                        float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), divisor,
                                        torch.cuda.current_stream(self.flat_grad.device).cuda_stream)
        _lib.check(st, "msgat_adam_step")
        if torch.cuda.is_current_stream_capturing():
            self._captured_active = active
        else:
            for i, on in enumerate(active):
                self._host_steps[i] += int(on)
        return loss

    @property
    def allreduce_bytes(self) -> int:
        if not self._params:
            self._build()
        return self.flat_grad.numel() * 4

    # -- torch.optim.Adam's checkpoint format ------------------------------------------------------------------
    def state_dict(self):
        for p, t in zip(self._params, self._host_steps):
            self.state[p]["step"] = torch.tensor(float(t))
        sd = super().state_dict()
        for group in sd["param_groups"]:            # the keys torch.optim.Adam writes, so the reference's loader accepts it
            group.setdefault("amsgrad", False)
            group.setdefault("maximize", False)
            group.setdefault("foreach", None)
            group.setdefault("capturable", False)
            group.setdefault("differentiable", False)
            group.setdefault("fused", None)
            group["lr"] = float(group["lr"])
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._loaded_steps = {id(p): float(st["step"]) for p, st in self.state.items() if "step" in st}
        if self.param_groups[0]["params"] and self.param_groups[0]["params"][0].is_cuda:
            self._build()                             # moves the loaded moments and step counts into the flat buffers

    def set_steps(self, steps: List[int]) -> None:
        """Put the step counts (host mirror and device) back, e.g. after the warm-up iterations of a graph capture."""
        self._host_steps = list(steps)
        self._dev_steps.copy_(torch.tensor([float(t) for t in steps]))


class _GraphedStep:
    """One captured step for one batch shape: static input buffers, `replay()` per batch.

    Training steps are captured whole (forward, loss, backward and, on a single GPU, the optimizer
    step); with several ranks the graph ends after backward and the all-reduce and the optimizer run
    eagerly.  Capture follows PyTorch's whole-network recipe: warm-up iterations on a side stream
    (they initialise the optimizer's lazy state), then capture -- with parameters and optimizer state put back
    afterwards, so the captured run starts from exactly the state an eager run would."""

    def __init__(self, engine: "Engine", batch, training: bool, step_in_graph: bool, loss_weight: float = 1.0):
        model, opt = engine.model, engine.optimizer
        self.static = [t.clone() for t in batch]
        *inputs, truth = self.static
        self.training = training
        self.grads = None
        trained = saved_params = saved_state = None
        if training:
            trained = [p for group in opt.param_groups for p in group["params"] if p.requires_grad]
            saved_params = [p.detach().clone() for p in trained]
            saved_state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state.get(p, {}).items()}
                           for group in opt.param_groups for p in group["params"]}
            saved_steps = list(getattr(opt, "_host_steps", []))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                if training:
                    opt.zero_grad(set_to_none=True)
                    engine._loss(model(*inputs), truth, None).backward()
                    if step_in_graph:
                        opt.step()
                else:
                    with torch.no_grad():
                        model(*inputs)
        torch.cuda.current_stream().wait_stream(side)
        if training:   # undo the warm-up: parameters and optimizer state as before it (fresh state = zeros)
            with torch.no_grad():
                for p, old in zip(trained, saved_params):   # not the frozen adjacency: its version keys the CSR cache
                    p.copy_(old)
                for p, st in opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v) and v.is_cuda:
                            old = saved_state.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
                if isinstance(opt, FlatAdam) and step_in_graph:
                    opt.set_steps(saved_steps if saved_steps else [0] * len(opt._host_steps))
            opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if training:
                self.pred = model(*inputs)
                self.loss = engine._loss(self.pred, truth, engine._graph_metrics, loss_weight)
                self.loss.backward()
                if step_in_graph:
                    opt.step()
            else:
                with torch.no_grad():
                    self.pred = model(*inputs)
                    self.loss = engine._loss(self.pred, truth, engine._graph_metrics, loss_weight)
        if training:
            # the gradients of THIS capture live in its private pool: whoever reads p.grad after a replay (the eager
            # all-reduce + optimizer of a multi-rank step) must see these tensors, not those of a later capture
            self.grads = [(p, p.grad) for p in trained]

    def replay(self, batch):
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self.grads is not None:
            for p, g in self.grads:
                p.grad = g
        return self.pred, self.loss, self.static[-1]


class Engine:
    __labels__ = {"train": "[Train   ]", "validate": "[Validate]", "evaluate": "[Evaluate]"}

    def __init__(self, model: nn.Module, loss_delta: float, out_dir: str):
        self.model = model
        self.loss_fn, self.out_dir = HuberLoss(loss_delta), Path(out_dir)
        self.out_dir.mkdir(parents=True, exist_ok=True)
        self.log_file = self.out_dir / "run.log"
        self.optimizer = None
        self._grad_sync = None
        self.hip_graph = False       # False | True | "auto" (see Trainer)
        self.graph_after = 3         # "auto": a batch shape is captured once it has been seen more often than this
        self._graph_seen = {}
        self._graphs = {}
        self._graph_metrics = None   # the Metrics whose device totals the captured loss kernels add to

    # -- helpers -------------------------------------------------------------------------
    def _device(self, gpu_id):
        if gpu_id is not None:
            return torch.device("cuda", gpu_id)
        return next(self.model.parameters()).device

    @staticmethod
    def _world() -> Tuple[int, int]:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _loss(self, pred: torch.Tensor, truth: torch.Tensor, metrics: Optional[Metrics],
              loss_weight: float = 1.0) -> torch.Tensor:
        """Huber loss of a batch; feeds `metrics` (engine.py:56,66-70).  On the GPU one library pass does both.
        `loss_weight` = this rank's share of the global batch (1 without a process group)."""
        if pred.is_cuda:
            from . import ops
            sums = None if metrics is None else metrics.totals(pred.device)
            return ops.huber_metrics(pred, truth, self.loss_fn.delta, 0.0 if metrics is None else metrics.mask_value, sums,
                                     loss_weight)
        loss = huber_loss(pred, truth, self.loss_fn.delta)
        if metrics is not None:
            metrics.update(pred, truth, loss, count=False, loss_weight=loss_weight)   # counted by the caller
        return loss

    def _optimizer_step(self, n_samples: int, world: int) -> None:
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.step(rank_weight=float(n_samples) if world > 1 else None)
            return
        if world > 1:
            if self._grad_sync is None:
                self._grad_sync = parallel.FlatGradAllReduce(self.model.parameters())
            self._grad_sync(weight=float(n_samples))
        self.optimizer.step()

    def run_epoch(self, data, gpu_id=None, epoch=None, mode: str = "train") -> float:
        """One pass over `data` (an iterable of (X, H, D, Y) batches).  Returns the mean batch loss.

        Under torch.distributed every rank must see the SAME sequence of global batches and takes its dim-0 shard
        (`parallel.shard_batch`) -- loaders built by `data.make_loaders` under a process group are already sharded
        (`msgat_sharded`), each rank loading only its own samples of a shared per-epoch permutation.  Global batches
        with fewer samples than ranks are skipped on every rank alike."""
        training = mode == "train"
        self.model.train(training)
        device = self._device(gpu_id)
        rank, world = self._world()
        presharded = bool(getattr(data, "msgat_sharded", False))
        sampler = getattr(data, "batch_sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(0 if epoch is None else int(epoch))
        if self._graph_metrics is None or not self.hip_graph or device.type != "cuda":
            metrics = Metrics()
        else:
            metrics = self._graph_metrics       # captured kernels hold its totals buffer: re-use it, zeroed
            metrics.reset()
        if self.hip_graph and device.type == "cuda":
            self._graph_metrics = metrics
            metrics.totals(device)
        # sizes of the GLOBAL batches behind pre-sharded ones (the sampler knows them); without a sampler the
        # shards are taken to be even
        global_sizes = iter(sampler.global_batch_sizes()) if presharded and hasattr(sampler, "global_batch_sizes") else None
        guard = torch.cuda.device(device) if device.type == "cuda" else _NullContext()
        pred = loss = None
        with guard, torch.set_grad_enabled(training):
            for batch in data:
                n_global = batch[0].shape[0]
                if world > 1 and not presharded:
                    if n_global < world:
                        continue
                    batch = parallel.shard_batch(batch, rank, world)
                elif world > 1:
                    n_global = next(global_sizes) if global_sizes is not None else n_global * world
                loss_weight = batch[0].shape[0] / n_global
                batch = [t.to(device, non_blocking=True) for t in batch]
                *inputs, truth = batch
                graphed = None
                if self.hip_graph and device.type == "cuda":
                    key = (training, tuple(tuple(t.shape) for t in batch), n_global)
                    graphed = self._graphs.get(key)
                    capture = graphed is None
                    if capture and self.hip_graph == "auto":
                        # a shape earns its graph by recurring: the ragged last batch of an epoch (or a one-off evaluation)
                        # is launched eagerly instead of paying two warm-up steps and a capture for a single replay
                        seen = self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
                        capture = seen > self.graph_after
                    if capture:
                        # nothing may keep the autograd graph of an eager step alive across a capture: its AccumulateGrad nodes
                        # belong to the default stream, and a backward that meets them on the capturing stream breaks the
                        # capture (hipStreamEndCapture crashes) -- the previous iteration's `loss` / `pred` do exactly that
                        pred = loss = None
                        with torch.enable_grad():
                            graphed = self._graphs[key] = _GraphedStep(self, batch, training, step_in_graph=world == 1,
                                                                       loss_weight=loss_weight)
                if graphed is not None:
                    pred, loss, truth = graphed.replay(batch)
                    if training and world == 1 and isinstance(self.optimizer, FlatAdam):
                        self.optimizer.note_replayed_step()
                else:
                    pred = self.model(*inputs)
                    loss = self._loss(pred, truth, metrics, loss_weight)
                    if training:
                        self.optimizer.zero_grad(set_to_none=True)
                        loss.backward()
                if training and (world > 1 or graphed is None):
                    self._optimizer_step(truth.shape[0], world)
                metrics.count(truth)
        if world > 1:
            if metrics._sums is None:
                metrics.totals(device)
            metrics.all_reduce()
        loss_ave = metrics.loss
        stats = {"loss": loss_ave, **metrics.todict()}
        if rank == 0:
            if mode == "evaluate":
                self.log_to_file(self.__labels__[mode], **stats)
            else:
                self.log_to_file(self.__labels__[mode], epoch=epoch, **stats)
        self.last_stats = stats
        return loss_ave

    def log_to_file(self, *args, **kwargs) -> None:
        """`date - label - k=v,...` lines appended to run.log (engine.py:85-92)."""
        with open(self.log_file, "a") as f:
            f.write(strftime("%Y/%m/%d %H:%M:%S", localtime()))
            f.write(" - " + " - ".join(f"{a}" for a in args))
            f.write(" - " + ",".join(f"{k}={v}" for k, v in kwargs.items()) + "\n")


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class Trainer(Engine):
    """Adam(lr 1e-3, wd 5e-4), StepLR(30, 0.1), early stopping after 20 stale epochs, best-val checkpoints
    after epoch 20 (engine.py:104-133).  GPU parameters train with `FlatAdam` (one launch), CPU parameters (host-logic
    tests) with torch.optim.Adam -- same update, same checkpoint format."""

    def __init__(self, model: nn.Module, loss_delta: float, out_dir: str, hip_graph="auto"):
        """`hip_graph`: "auto" (default) -- on the GPU a training / validation step is captured in a HIP graph once its batch
        shape has recurred `graph_after` (3) times and replayed from then on, other shapes are launched eagerly (a step of
        ~120 launches is host-bound wherever it is under ~3 ms of GPU work: msgat48 3.3 -> 2.8 ms, PEMSD4 2.7 -> 1.5 ms);
        True -- capture every shape at first sight; False -- eager launches only.  Graphs are keyed by (grad mode, batch
        shape, global batch size) and dropped when `load()` moves the optimizer's buffers."""
        super().__init__(model, loss_delta=loss_delta, out_dir=out_dir)
        on_gpu = next(model.parameters()).is_cuda
        if hip_graph is True and not on_gpu:
            raise ValueError("hip_graph=True needs the model on the GPU")
        if hip_graph not in (True, False, "auto"):
            raise ValueError(f"hip_graph must be True, False or 'auto', not {hip_graph!r}")
        self.hip_graph = hip_graph if on_gpu else False
        parallel.sync_parameters(model)     # data-parallel replicas start from rank 0's weights (no-op for one process)
        adam = FlatAdam if on_gpu else optim.Adam
        self.optimizer = adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
        self.scheduler = lr_scheduler.StepLR(self.optimizer, step_size=30, gamma=0.1)
        self.best = {"epoch": 0, "loss": float("inf"), "ckpt": ""}
        self.epoch = 1
        self.patience, self.min_delta = 20, 1e-4
        self.max_epochs, self.min_epochs = 100, 20

    def fit(self, data_loaders, gpu_id=None) -> None:
        train, val = data_loaders
        while self.epoch <= self.max_epochs:
            self.run_epoch(train, gpu_id=gpu_id, epoch=self.epoch, mode="train")
            loss = self.run_epoch(val, gpu_id=gpu_id, epoch=self.epoch, mode="validate")
            self.scheduler.step()
            self._sync_lr()
            if self.epoch > self.min_epochs:
                if loss < (1 - self.min_delta) * self.best["loss"]:
                    self.best = dict(epoch=self.epoch, loss=loss, ckpt=self.out_dir / f"{self.epoch}_{loss:.2f}.pkl")
                    if self._world()[0] == 0:
                        self.save(ckpt=self.best["ckpt"])
                elif self.epoch > self.best["epoch"] + self.patience:
                    break
            self.epoch += 1

    def _sync_lr(self) -> None:
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.sync_lr()   # a captured update reads the rate from device memory

    def save(self, ckpt) -> None:
        """The reference's five keys (engine.py:135-146).  `grad_scaler` holds the state of a default-constructed,
        enabled GradScaler -- this build trains in fp32 and never scales, but the reference's `load` feeds that entry
        to `GradScaler.load_state_dict`, which rejects an empty dict (engine.py:155)."""
        torch.save(dict(best=self.best, epoch=self.epoch, model=self.model.state_dict(),
                        optimizer=self.optimizer.state_dict(), scheduler=self.scheduler.state_dict(),
                        grad_scaler=default_grad_scaler_state()), ckpt)

    def load(self, ckpt) -> None:
        states = torch.load(ckpt, map_location=next(self.model.parameters()).device, weights_only=False)
        self.best = states["best"]
        self.epoch = states["epoch"] + 1
        token = self.optimizer.buffer_token() if isinstance(self.optimizer, FlatAdam) else ()
        self.model.load_state_dict(strip_data_parallel_prefix(states["model"]))
        self.optimizer.load_state_dict(states["optimizer"])
        self.scheduler.load_state_dict(states["scheduler"])
        self._sync_lr()
        if isinstance(self.optimizer, FlatAdam) and token and token != self.optimizer.buffer_token():
            self._graphs.clear()            # the optimizer's buffers moved: captured steps point at the old ones
            self._graph_metrics = None


def default_grad_scaler_state() -> dict:
    """`torch.cuda.amp.GradScaler().state_dict()` of an enabled, never-stepped scaler (its constructor defaults)."""
    return {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


class Evaluator(Engine):
    def __init__(self, model: nn.Module, delta: float, out_dir: str, ckpt, hip_graph="auto"):
        super().__init__(model, loss_delta=delta, out_dir=out_dir)
        states = torch.load(ckpt, map_location=next(model.parameters()).device, weights_only=False)
        model.load_state_dict(strip_data_parallel_prefix(states["model"]))
        self.hip_graph = hip_graph if next(model.parameters()).is_cuda else False   # see Trainer

    def eval(self, data_loader, gpu_id=None) -> float:
        return self.run_epoch(data_loader, gpu_id=gpu_id, mode="evaluate")


def strip_data_parallel_prefix(state_dict):
    """Checkpoints written through `nn.DataParallel` (main.py:54,60) prefix every key with `module.`."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict
