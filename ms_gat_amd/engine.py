"""Train / validate / evaluate loop with the reference's API, one process per GPU.

Reference: /root/reference/src/engine.py -- `Engine.run_epoch(data, gpu_id, epoch, mode)` (:40),
`Trainer(model, loss_delta, out_dir).fit((train, val), gpu_id)` (:104-133), `.save/.load(ckpt)`
(:135-157), `Evaluator(model, delta, out_dir, ckpt).eval(loader, gpu_id)` (:160-168);
src/loss.py:51-52 (Huber), src/metrics.py:20-35 (MAE / MAPE / RMSE).

Differences that are the point of this build:
  * fp32 end to end (the reference wraps the forward in CUDA AMP, engine.py:54; the parity bar of
    the hot path is fp32), so there is no GradScaler; checkpoints keep the reference's keys and
    carry an empty `grad_scaler` entry so either side can load the other's files.
  * loss and metric sums stay on the device; the host reads them once per epoch instead of four
    `.item()` syncs per batch (engine.py:66, metrics.py:24,30,34).
  * under `torch.distributed` every rank runs its batch shard and gradients are averaged with one
    flat all-reduce (`parallel.FlatGradAllReduce`); `nn.DataParallel` (main.py:52-55) is not used.
  * `hip_graph=True` captures one training step (forward, loss, backward, Adam) per batch shape in a
    HIP graph and replays it: a step is ~1100 kernel launches whose host-side issue cost otherwise
    exceeds the GPU time of the small kernels.  The library never allocates or synchronises, so its
    launches are capturable as they are.
"""
from __future__ import annotations

from pathlib import Path
from time import localtime, strftime
from typing import Optional, Tuple

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
        return huber_loss(output, target, self.delta)


class Metrics:
    """Running MAE / MAPE / RMSE with the reference's definitions (metrics.py:11-38): MAPE sums
    |err/y| over entries with y > mask_value but divides by ALL entries, like the reference.
    Sums accumulate on the device in float64; properties synchronise when read."""

    def __init__(self, mask_value: float = 0.0):
        self.mask_value = mask_value
        self.n = 0
        self._sums: Optional[torch.Tensor] = None  # [AE, APE, SE]

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> None:
        err = (y_pred.detach() - y_true).double()
        truth = y_true.double()
        mask = truth > self.mask_value
        ape = torch.where(mask, (err / torch.where(mask, truth, torch.ones_like(truth))).abs(), torch.zeros_like(err))
        s = torch.stack([err.abs().sum(), 100.0 * ape.sum(), (err * err).sum()])
        self._sums = s if self._sums is None else self._sums + s
        self.n += y_true.numel()

    def _get(self, i: int) -> float:
        return 0.0 if self._sums is None else float(self._sums[i].item())

    def all_reduce(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self._sums is not None:
            buf = torch.cat([self._sums, self._sums.new_tensor([float(self.n)])])
            dist.all_reduce(buf)
            self._sums, self.n = buf[:3], int(buf[3].item())

    @property
    def MAE(self) -> float:
        return self._get(0) / max(self.n, 1)

    @property
    def MAPE(self) -> float:
        return self._get(1) / max(self.n, 1)

    @property
    def RMSE(self) -> float:
        return (self._get(2) / max(self.n, 1)) ** 0.5

    def todict(self):
        return {"MAE": self.MAE, "MAPE": self.MAPE, "RMSE": self.RMSE}


class _GraphedStep:
    """One captured step for one batch shape: static input buffers, `replay()` per batch.

    Training steps are captured whole (forward, loss, backward and, on a single GPU, the optimizer
    step); with several ranks the graph ends after backward and the all-reduce and the optimizer run
    eagerly.  Capture follows PyTorch's whole-network recipe: warm-up iterations on a side stream
    (they initialise Adam's lazy state), then capture -- with parameters and optimizer state put back
    afterwards, so the captured run starts from exactly the state an eager run would."""

    def __init__(self, engine: "Engine", batch, training: bool, step_in_graph: bool):
        model, loss_fn, opt = engine.model, engine.loss_fn, engine.optimizer
        self.static = [t.clone() for t in batch]
        *inputs, truth = self.static
        self.training = training
        saved_params = saved_state = None
        if training:
            trained = [p for group in opt.param_groups for p in group["params"] if p.requires_grad]
            saved_params = [p.detach().clone() for p in trained]
            saved_state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state.get(p, {}).items()}
                           for group in opt.param_groups for p in group["params"]}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                if training:
                    opt.zero_grad(set_to_none=True)
                    loss_fn(model(*inputs), truth).backward()
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
                        if torch.is_tensor(v):
                            old = saved_state.get(p, {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
            opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if training:
                self.pred = model(*inputs)
                self.loss = loss_fn(self.pred, truth)
                self.loss.backward()
                if step_in_graph:
                    opt.step()
            else:
                with torch.no_grad():
                    self.pred = model(*inputs)
                    self.loss = loss_fn(self.pred, truth)

    def replay(self, batch):
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
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
        self.hip_graph = False
        self._graphs = {}

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

    def run_epoch(self, data, gpu_id=None, epoch=None, mode: str = "train") -> float:
        """One pass over `data` (an iterable of (X, H, D, Y) batches).  Returns the mean batch loss."""
        training = mode == "train"
        self.model.train(training)
        device = self._device(gpu_id)
        rank, world = self._world()
        metrics = Metrics()
        loss_sum = torch.zeros((), device=device, dtype=torch.float64)
        n_batches = 0
        with torch.set_grad_enabled(training):
            for batch in data:
                if world > 1:
                    batch = parallel.shard_batch(batch, rank, world)
                batch = [t.to(device, non_blocking=True) for t in batch]
                *inputs, truth = batch
                if self.hip_graph:
                    key = (training, tuple(tuple(t.shape) for t in batch))
                    graphed = self._graphs.get(key)
                    if graphed is None:
                        with torch.enable_grad():
                            graphed = self._graphs[key] = _GraphedStep(self, batch, training, step_in_graph=world == 1)
                    pred, loss, truth = graphed.replay(batch)
                else:
                    pred = self.model(*inputs)
                    loss = self.loss_fn(pred, truth)
                    if training:
                        self.optimizer.zero_grad(set_to_none=True)
                        loss.backward()
                if training and (world > 1 or not self.hip_graph):
                    if world > 1:
                        if self._grad_sync is None:
                            self._grad_sync = parallel.FlatGradAllReduce(self.model.parameters())
                        self._grad_sync(weight=float(truth.shape[0]))
                    self.optimizer.step()
                loss_sum += loss.detach().double()
                metrics.update(pred, truth)
                n_batches += 1
        if world > 1:
            dist.all_reduce(loss_sum)
            loss_sum /= world
            metrics.all_reduce()
        loss_ave = float(loss_sum.item()) / max(n_batches, 1)
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


class Trainer(Engine):
    """Adam(lr 1e-3, wd 5e-4), StepLR(30, 0.1), early stopping after 20 stale epochs, best-val checkpoints
    after epoch 20 (engine.py:104-133)."""

    def __init__(self, model: nn.Module, loss_delta: float, out_dir: str, hip_graph: bool = False):
        super().__init__(model, loss_delta=loss_delta, out_dir=out_dir)
        self.hip_graph = hip_graph
        if hip_graph:
            # a captured Adam reads its learning rate from device memory: keep ONE tensor alive and write
            # the scheduler's value into it (`_sync_lr`), or replays would keep the captured rate
            dev = next(model.parameters()).device
            self._lr = torch.tensor(1e-3, device=dev)
            self.optimizer = optim.Adam(model.parameters(), lr=self._lr, weight_decay=5e-4, capturable=True)
        else:
            self.optimizer = optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
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
        if self.hip_graph:
            for group in self.optimizer.param_groups:
                if group["lr"] is not self._lr:
                    self._lr.fill_(float(group["lr"]))
                    group["lr"] = self._lr

    def save(self, ckpt) -> None:
        torch.save(dict(best=self.best, epoch=self.epoch, model=self.model.state_dict(),
                        optimizer=self.optimizer.state_dict(), scheduler=self.scheduler.state_dict(),
                        grad_scaler={}), ckpt)

    def load(self, ckpt) -> None:
        states = torch.load(ckpt, map_location=next(self.model.parameters()).device, weights_only=False)
        self.best = states["best"]
        self.epoch = states["epoch"] + 1
        self.model.load_state_dict(strip_data_parallel_prefix(states["model"]))
        self.optimizer.load_state_dict(states["optimizer"])
        self.scheduler.load_state_dict(states["scheduler"])
        self._sync_lr()


class Evaluator(Engine):
    def __init__(self, model: nn.Module, delta: float, out_dir: str, ckpt):
        super().__init__(model, loss_delta=delta, out_dir=out_dir)
        states = torch.load(ckpt, map_location=next(model.parameters()).device, weights_only=False)
        model.load_state_dict(strip_data_parallel_prefix(states["model"]))

    def eval(self, data_loader, gpu_id=None) -> float:
        return self.run_epoch(data_loader, gpu_id=gpu_id, mode="evaluate")


def strip_data_parallel_prefix(state_dict):
    """Checkpoints written through `nn.DataParallel` (main.py:54,60) prefix every key with `module.`."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict
