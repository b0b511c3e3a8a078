"""Input side of the path: PEMS-style series -> (X, H, D, Y) batches, and a synthetic stand-in.

Reference: /root/reference/src/data_loader.py -- adjacency csv (:49-66, in `graph.py` here),
npz series `data` of shape [T_total, N, C] (:71), z-score over the training span (:79, :118-120),
60/20/20 split (:73-78), multi-period windows (:92-115).  On-disk formats are the reference's:
csv `from,to,cost` with a header line, npz with key `data`.

The reference ships no data files, so `SyntheticPEMS` generates a seeded series of the same
layout for benchmarks and tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader, Dataset, Sampler

from .graph import sym_norm_adjacency, synthetic_adjacency


def load_adjacency_csv(path: str, n_nodes: int) -> torch.Tensor:
    """`from,to,cost` edge list (header skipped, cost ignored) -> D^-1/2 (A+I) D^-1/2 (data_loader.py:59-66)."""
    edges = []
    with open(path) as f:
        next(f, None)
        for line in f:
            if line.strip():
                s, d = line.split(",")[:2]
                edges.append((int(s), int(d)))
    return sym_norm_adjacency(n_nodes, edges)


def zscore(series: torch.Tensor, split: int) -> torch.Tensor:
    """Per (channel, node) z-score with statistics of the first `split` steps only (data_loader.py:118-120)."""
    std, mean = torch.std_mean(series[..., :split], dim=-1, keepdim=True)
    return (series - mean) / std


def split_intervals(total_steps: int, in_hours: Sequence[int], out_timesteps: int, tau: int):
    """Training / validation / evaluation [start, stop) of forecast origins (data_loader.py:69-78)."""
    history = tau * max(in_hours)
    length = total_steps - history - out_timesteps + 1
    a, b = int(0.6 * length), int(0.8 * length)
    return [(history, history + a), (history + a, history + b), (history + b, history + length)], history + a


class PeriodicWindows(Dataset):
    """Forecast origin t -> (X [R,C,N,tau], hour-of-day, day-of-week, Y [N,q])  (data_loader.py:92-115).

    Component r looks `hours[r]` hours back: X[r] = inputs[..., t - hours[r]*tau : t - hours[r]*tau + tau].
    The target is channel 0 of the raw (un-normalised) series.
    """

    def __init__(self, inputs: torch.Tensor, target: torch.Tensor, interval: Tuple[int, int],
                 hours: Sequence[int], out_timesteps: int, timesteps_per_hour: int):
        self.inputs, self.target = inputs, target
        self.start, self.stop = interval
        self.hours, self.q, self.tau = list(hours), out_timesteps, timesteps_per_hour

    def __len__(self) -> int:
        return self.stop - self.start

    def __getitem__(self, i: int):
        t = i + self.start
        hour = t // self.tau
        x = torch.stack([self.inputs[..., t - h * self.tau: t - h * self.tau + self.tau] for h in self.hours])
        y = self.target[..., t: t + self.q]
        return x, torch.tensor(hour % 24), torch.tensor((hour // 24) % 7), y


class MSGATData:
    """Adjacency + the three loaders, as `DataLoaderForMSGAT` exposes them (data_loader.py:16-47).

    `meta` is the reference's `data/meta.yaml` entry (adj-file, data-file, num-nodes, num-channels,
    timesteps-per-hour) or a path to such a yaml plus `name`.
    """

    def __init__(self, name: str, in_hours: List[int], out_timesteps: int, batch_size: int, num_workers: int = 0,
                 meta_file: str = "data/meta.yaml", meta: dict | None = None):
        if meta is None:
            with open(meta_file) as f:
                meta = yaml.safe_load(f)[name]
        self.num_nodes, self.num_channels = meta["num-nodes"], meta["num-channels"]
        self.timesteps_per_hour = meta["timesteps-per-hour"]
        self.in_hours, self.out_timesteps = list(in_hours), out_timesteps
        self.adj = load_adjacency_csv(meta["adj-file"], self.num_nodes)
        raw = torch.from_numpy(np.load(meta["data-file"])["data"]).float().transpose(0, -1)  # [C,N,T_total]
        self.training, self.validation, self.evaluation = make_loaders(
            raw, self.in_hours, out_timesteps, self.timesteps_per_hour, batch_size, num_workers)


class ShardedBatchSampler(Sampler):
    """This rank's samples of every GLOBAL batch, for one-process-per-GPU data parallelism.

    `nn.DataParallel` (main.py:52-55) loads a batch in one process and scatters it on dim 0.  Here all ranks
    walk the same sequence of global batches of `batch_size` samples -- the same per-epoch permutation, drawn
    from `seed + epoch` on every rank -- and each yields only its contiguous shard (`parallel.shard_bounds`), so
    a rank loads 1/world of the data and the shards partition the epoch.  A last global batch with fewer samples
    than ranks is dropped on every rank alike (an empty shard would leave a rank out of the gradient collective).
    `set_epoch(e)` re-seeds the permutation; `Engine.run_epoch` calls it."""

    def __init__(self, n: int, batch_size: int, shuffle: bool, rank: int, world: int, seed: int = 0):
        if not 0 <= rank < world or batch_size < 1:
            raise ValueError(f"bad shard: rank {rank} of {world}, batch {batch_size}")
        self.n, self.batch_size, self.shuffle, self.rank, self.world, self.seed = n, batch_size, shuffle, rank, world, seed
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _starts(self):
        return [b for b in range(0, self.n, self.batch_size) if min(self.batch_size, self.n - b) >= self.world]

    def __len__(self) -> int:
        return len(self._starts())

    def global_batch_sizes(self) -> List[int]:
        """Sample count of every global batch this sampler yields a shard of, in order (the engine weights a shard's
        loss by its share of the global batch)."""
        return [min(self.batch_size, self.n - b) for b in self._starts()]

    def __iter__(self):
        from .parallel import shard_bounds
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        for b in self._starts():
            size = min(self.batch_size, self.n - b)
            lo, hi = shard_bounds(size, self.rank, self.world)
            yield order[b + lo: b + hi]


def make_loaders(raw: torch.Tensor, in_hours, out_timesteps, tau, batch_size, num_workers=0, pin_memory=True,
                 rank: int | None = None, world: int | None = None, seed: int = 0):
    """raw [C,N,T_total] -> (training, validation, evaluation) loaders; only training shuffles (data_loader.py:80-89).

    `batch_size` is the GLOBAL batch (the reference's `-b`).  Under an initialised process group (or explicit
    `rank` / `world`) the loaders are sharded (`ShardedBatchSampler`, attribute `msgat_sharded`): every rank loads
    only its part of each global batch."""
    import torch.distributed as dist
    if world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    intervals, train_end = split_intervals(raw.size(-1), in_hours, out_timesteps, tau)
    normed = zscore(raw, split=train_end)
    pin = pin_memory and torch.cuda.is_available()
    loaders = []
    for i, iv in enumerate(intervals):
        ds = PeriodicWindows(normed, raw[0], iv, in_hours, out_timesteps, tau)
        if world > 1:
            loader = DataLoader(ds, batch_sampler=ShardedBatchSampler(len(ds), batch_size, i == 0, rank, world, seed),
                                pin_memory=pin, num_workers=num_workers)
            loader.msgat_sharded = True
        else:
            loader = DataLoader(ds, batch_size, shuffle=(i == 0), pin_memory=pin, num_workers=num_workers)
        loaders.append(loader)
    return loaders


class SyntheticPEMS:
    """Seeded stand-in for a PEMS dataset: same tensors, shapes and loaders, no files."""

    def __init__(self, n_nodes: int, n_edges: int, n_channels: int, in_hours: List[int], out_timesteps: int = 12,
                 batch_size: int = 32, timesteps_per_hour: int = 12, days: int = 14, seed: int = 0,
                 num_workers: int = 0):
        self.num_nodes, self.num_channels, self.timesteps_per_hour = n_nodes, n_channels, timesteps_per_hour
        self.in_hours, self.out_timesteps = list(in_hours), out_timesteps
        self.adj = synthetic_adjacency(n_nodes, n_edges, seed)
        g = torch.Generator().manual_seed(seed + 1)
        total = days * 24 * timesteps_per_hour
        t = torch.arange(total, dtype=torch.float32)
        daily = torch.sin(2 * torch.pi * t / (24 * timesteps_per_hour))
        base = 200 + 80 * torch.rand(n_channels, n_nodes, 1, generator=g)
        raw = base * (1 + 0.4 * daily) + 15 * torch.randn(n_channels, n_nodes, total, generator=g)
        self.raw = raw.clamp_min(1.0)
        self.training, self.validation, self.evaluation = make_loaders(
            self.raw, self.in_hours, out_timesteps, timesteps_per_hour, batch_size, num_workers)
