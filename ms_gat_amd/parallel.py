"""Data parallelism for the hot path: one process per GPU, batch axis sharded, one flat
gradient all-reduce per step.

The reference's only parallelism is `nn.DataParallel` (/root/reference/src/main.py:52-55):
a single process that re-broadcasts the parameters every iteration, scatters the batch on
dim 0 and reduce-adds gradients onto GPU 0.  Here each rank owns one MI355X and a batch
shard; every (relation, sample) group is independent in forward and backward, so the only
exchange is the parameter-gradient sum -- a few KB for the GACN parameters, 7.8 MB for all
of msgat72 -- sent as ONE flat fp32 bucket over RCCL (ring over xGMI: latency-bound at this
size, so per-tensor buckets or overlap would only add launches).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world_size, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of `n` samples for `rank`; the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Dim-0 shard of every tensor of a batch (x, h, d, y), as `DataParallel` scatters it."""
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world)
    return [t[lo:hi] for t in tensors]


def sync_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Every rank starts from rank `src`'s parameters and buffers (what `nn.DataParallel` gets for free by replicating
    the module from device 0 every iteration, main.py:52-55).  One broadcast per dtype over a flat copy."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for group in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, src)
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


# HIP events around each flat all-reduce (on the launch stream), for the benchmark's `allreduce_us`: set to a list to
# collect (start, end) pairs, None (default) to record nothing.
collective_events = None


def all_reduce_flat(flat: torch.Tensor) -> None:
    """The step's ONE collective: sum of the flat fp32 buffer over all ranks (RCCL over xGMI for device tensors)."""
    rec = collective_events
    if rec is None or not flat.is_cuda:
        dist.all_reduce(flat)
        return
    stream = torch.cuda.current_stream(flat.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    dist.all_reduce(flat)
    e1.record(stream)
    rec.append((e0, e1))


class _GatherTable:
    """Device chunk table of one gradient layout (sizes + offsets): the offset and length columns never change; the
    pointer column is refreshed from a pinned host buffer with a stream-ordered copy when an allocator hands the
    gradients out at new addresses (eager steps with ragged batches do) -- no synchronising pageable upload on the
    path in front of the collective."""

    def __init__(self, sizes, offsets, dev):
        from . import _lib
        chunk = int(_lib.lib().msgat_adam_chunk_elems())
        self.rel, offs, lens = [], [], []       # rel: (gradient index, byte offset) of every chunk
        for i, (n, o) in enumerate(zip(sizes, offsets)):
            for s in range(0, n, chunk):
                self.rel.append((i, 4 * s))
                offs.append(o + s)
                lens.append(min(chunk, n - s))
        self.n = len(self.rel)
        self.offs = torch.tensor(offs, dtype=torch.int64).to(dev)
        self.lens = torch.tensor(lens, dtype=torch.int32).to(dev)
        self.ptrs = torch.zeros(max(self.n, 1), dtype=torch.int64, device=dev)
        # a ring of pinned staging buffers, each with the event behind its last copy: re-using one waits for THAT copy
        # only -- four steps back, i.e. never in practice -- so a step whose gradients moved does not stall the host
        # behind the device (waiting for the newest copy would: it sits at the end of the previous step)
        self.host = [torch.zeros(max(self.n, 1), dtype=torch.int64).pin_memory() for _ in range(4)]
        self.copied = [None] * len(self.host)
        self.turn = 0
        self.current = None

    def point_at(self, bases: tuple) -> None:
        if bases == self.current:
            return
        if torch.cuda.is_current_stream_capturing():
            # a captured upload would read a pinned ring buffer that later eager re-pointings overwrite, and after a
            # replay `current` would no longer describe the device table: the gather would read stale addresses
            raise RuntimeError(
                "gather_scaled: the gradient addresses changed inside a HIP-graph capture; the pointer table cannot be "
                "re-pointed there.  Keep the gather (and the collective behind it) outside the captured region, as "
                "engine.Trainer does with several ranks, or run one eager step with these gradients first.")
        k = self.turn
        self.turn = (k + 1) % len(self.host)
        if self.copied[k] is not None:
            self.copied[k].synchronize()
        h = self.host[k].numpy()
        for j, (i, b) in enumerate(self.rel):
            h[j] = bases[i] + b
        self.ptrs.copy_(self.host[k], non_blocking=True)
        self.copied[k] = torch.cuda.Event()
        self.copied[k].record(torch.cuda.current_stream(self.ptrs.device))
        self.current = bases


def gather_scaled(flat: torch.Tensor, grads: Sequence[torch.Tensor], offsets: Sequence[int], weight: float,
                  weight_index: int, cache: dict) -> None:
    """flat[offsets[i] : ...] = weight * grads[i] for every gradient and flat[weight_index] = weight, in ONE launch
    (`msgat_gather_scaled`, csrc/tail.hip).  `cache` keeps one persistent device table per gradient layout
    (`_GatherTable`): a replayed HIP graph or a warm caching allocator hands out the same pointers step after step and
    nothing is uploaded; new pointers cost one asynchronous copy from pinned memory."""
    from . import _lib
    keep = [g if g.is_contiguous() else g.contiguous() for g in grads]
    key = (tuple(g.numel() for g in keep), tuple(offsets))
    table = cache.get(key)
    if table is None:
        table = cache[key] = _GatherTable(key[0], key[1], flat.device)
    table.point_at(tuple(g.data_ptr() for g in keep))
    st = _lib.lib().msgat_gather_scaled(table.ptrs.data_ptr(), table.offs.data_ptr(), table.lens.data_ptr(), table.n,
                                        float(weight), flat.data_ptr(), int(weight_index),
                                        torch.cuda.current_stream(flat.device).cuda_stream)
    _lib.check(st, "msgat_gather_scaled")


class FlatGradAllReduce:
    """Averages the gradients of `params` across ranks through one contiguous fp32 buffer.

    The rank's weight (its sample count) rides in the last element of the same buffer, so a
    step costs exactly one collective and no host synchronisation.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel + 1, device=dev, dtype=torch.float32)
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            self.offsets.append(off)
            off += p.numel()
        self._tables: dict = {}

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def __call__(self, weight: float = 1.0) -> None:
        """grad <- sum_ranks(weight * grad) / sum_ranks(weight).  With `weight` = the rank's
        sample count the result is the gradient of the mean loss over the global batch,
        also when the shards are uneven."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        if self.flat.is_cuda and all(p.grad is not None for p in self.params):
            # device tensors: one launch in front of the collective, one behind it; the averaged gradients are
            # handed back as views of the flat buffer (no copy back)
            gather_scaled(self.flat, [p.grad for p in self.params], self.offsets, weight, self.numel, self._tables)
            all_reduce_flat(self.flat)
            self.flat[: self.numel].div_(self.flat[self.numel])
            for p, v in zip(self.params, self.views):
                p.grad = v
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        if weight != 1.0:
            self.flat[: self.numel].mul_(weight)
        self.flat[self.numel] = weight
        dist.all_reduce(self.flat)
        self.flat[: self.numel].div_(self.flat[self.numel])
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
