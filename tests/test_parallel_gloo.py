"""Data-parallel host logic on CPU: world_size-2 `gloo` processes must reproduce the
single-process full-batch gradients after the flat all-reduce."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from ms_gat_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 6, 1), nn.ReLU(), nn.Conv2d(6, 2, 1))


def _batch(n):
    g = torch.Generator().manual_seed(1)
    return torch.randn(n, 3, 5, 4, generator=g), torch.randn(n, 2, 5, 4, generator=g)


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    x, y = parallel.shard_batch(_batch(n), rank, world)
    loss = ((model(x) - y) ** 2).mean()
    loss.backward()
    sync = parallel.FlatGradAllReduce(model.parameters())
    sync(weight=float(x.shape[0]))
    if rank == 0:
        torch.save([p.grad.clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n, tmp_path):
    out = str(tmp_path / f"grads_{world}_{n}.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    return torch.load(out)


def test_two_rank_gradients_equal_single_process(tmp_path):
    for n in (8, 7):  # even and uneven shards
        model = _model()
        x, y = _batch(n)
        ((model(x) - y) ** 2).mean().backward()
        want = [p.grad for p in model.parameters()]
        got = _run(2, n, tmp_path)
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_shard_bounds_cover_the_batch_exactly():
    for n in (1, 7, 32, 255):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    model = _model()
    x, y = _batch(4)
    ((model(x) - y) ** 2).mean().backward()
    before = [p.grad.clone() for p in model.parameters()]
    parallel.FlatGradAllReduce(model.parameters())()
    for a, b in zip(before, model.parameters()):
        assert torch.equal(a, b.grad)


# ---- Engine.run_epoch under a process group (the loop main.py:52-55 runs through nn.DataParallel) ----------------
class _TinyMSGAT(nn.Module):
    """CPU stand-in with the MSGAT call signature model(X, H, D) -> [B,N,T]."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.mix = nn.Conv2d(4, 1, 1)
        self.h = nn.Embedding(24, 1)

    def forward(self, X, H, D):
        B, R, C, N, T = X.shape
        return self.mix(X.reshape(B, R * C, N, T)).squeeze(1) + self.h(H).view(B, 1, 1)


def _epoch_batches(sizes):
    g = torch.Generator().manual_seed(11)
    return [(torch.randn(b, 2, 2, 5, 12, generator=g), torch.randint(0, 24, (b,), generator=g),
             torch.randint(0, 7, (b,), generator=g), torch.randn(b, 5, 12, generator=g) * 30) for b in sizes]


def _epoch_worker(rank, world, port, sizes, out_dir):
    from ms_gat_amd import engine
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        parallel.init_from_env("gloo")
    model = _TinyMSGAT()
    tr = engine.Trainer(model, 50.0, os.path.join(out_dir, f"w{world}"))
    losses = [tr.run_epoch(_epoch_batches(sizes), epoch=e, mode="train") for e in (1, 2)]
    val = tr.run_epoch(_epoch_batches(sizes), epoch=2, mode="validate")
    if rank == 0:
        torch.save(dict(losses=losses, val=val, stats=tr.last_stats, params=[p.detach().clone() for p in model.parameters()]),
                   os.path.join(out_dir, f"result_w{world}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(8, 8, 8), (8, 7, 3, 1)])   # even shards; ragged ones incl. a batch smaller than the world
def test_run_epoch_on_two_ranks_equals_the_single_process_run(tmp_path, sizes):
    """Every rank takes its dim-0 shard of the same global batches; the weighted flat all-reduce makes the
    update the single-process one.  A global batch with fewer samples than ranks is skipped by all ranks."""
    out = str(tmp_path)
    _epoch_worker(0, 1, 0, [s for s in sizes if s >= 2], out)     # the single-process run skips nothing by itself
    mp.spawn(_epoch_worker, args=(2, _free_port(), sizes, out), nprocs=2, join=True)
    one, two = (torch.load(os.path.join(out, f"result_w{w}.pt"), weights_only=False) for w in (1, 2))
    for a, b in zip(one["params"], two["params"]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # a rank adds its share n_r / n_b of every batch's mean loss, so the logged loss is the reference's mean of
    # global batch means (engine.py:66-67) for uneven shards too
    assert one["losses"] == pytest.approx(two["losses"], rel=1e-5)
    assert one["val"] == pytest.approx(two["val"], rel=1e-5)
    for k in ("MAE", "MAPE", "RMSE"):                               # metric sums are exact totals over all samples
        assert one["stats"][k] == pytest.approx(two["stats"][k], rel=1e-6)


def test_sharded_batch_sampler_partitions_every_epoch():
    from ms_gat_amd import data
    for n, bs, world in ((50, 8, 2), (33, 16, 3), (17, 4, 4)):
        for shuffle in (False, True):
            per_rank = []
            for r in range(world):
                s = data.ShardedBatchSampler(n, bs, shuffle, r, world, seed=5)
                s.set_epoch(3)
                per_rank.append(list(s))
            assert len({len(b) for b in per_rank}) == 1                      # same number of steps on every rank
            kept = sorted(i for batches in per_rank for b in batches for i in b)
            tail = n % bs
            dropped = tail if 0 < tail < world else 0
            assert len(kept) == n - dropped and len(set(kept)) == len(kept)   # a partition: nothing twice
            for step in zip(*per_rank):                                       # shards of one global batch, sizes within 1
                sizes = [len(b) for b in step]
                assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
            assert [sum(len(b) for b in step) for step in zip(*per_rank)] == s.global_batch_sizes()
            again = data.ShardedBatchSampler(n, bs, shuffle, 0, world, seed=5)
            again.set_epoch(4)
            assert (list(again) != per_rank[0]) == shuffle                    # a new permutation per epoch


def _sync_worker(rank, world, port, out_dir):
    from ms_gat_amd import engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    model = _TinyMSGAT()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(float(rank))                       # the replicas disagree before the trainer is built
    engine.Trainer(model, 50.0, os.path.join(out_dir, f"s{rank}"))
    torch.save([p.detach().clone() for p in model.parameters()], os.path.join(out_dir, f"sync_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_starts_every_replica_from_rank_zero_weights(tmp_path):
    """nn.DataParallel replicates the module from device 0 (main.py:52-55); one process per GPU must do it itself."""
    out = str(tmp_path)
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(out, f"sync_r{r}.pt"), weights_only=False) for r in (0, 1))
    want = [p.detach() for p in _TinyMSGAT().parameters()]
    for p, q, w in zip(a, b, want):
        assert torch.equal(p, q) and torch.equal(p, w)


def _loader_worker(rank, world, port, out_dir):
    from ms_gat_amd import data, engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    ds = data.SyntheticPEMS(n_nodes=5, n_edges=5, n_channels=2, in_hours=[1, 2], batch_size=16, days=2)
    assert getattr(ds.training, "msgat_sharded", False)
    seen = []
    ds.training.batch_sampler.set_epoch(1)
    for x, h, d, y in ds.training:
        assert x.shape[0] in (8,) or x.shape[0] <= 8          # half of each global batch of 16
        seen.append(x.shape[0])
    model = _TinyMSGAT()
    tr = engine.Trainer(model, 50.0, os.path.join(out_dir, f"r{rank}"))
    loss = tr.run_epoch(ds.training, epoch=1, mode="train")
    torch.save(dict(loss=loss, n=sum(seen), params=[p.detach().clone() for p in model.parameters()]),
               os.path.join(out_dir, f"loader_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loaders_feed_run_epoch_without_resharding(tmp_path):
    out = str(tmp_path)
    mp.spawn(_loader_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(out, f"loader_r{r}.pt"), weights_only=False) for r in (0, 1))
    assert a["loss"] == pytest.approx(b["loss"], rel=1e-6)                  # the all-reduced epoch loss
    for p, q in zip(a["params"], b["params"]):
        assert torch.equal(p, q)                                             # replicas stay in lockstep
    assert abs(a["n"] - b["n"]) <= len(range(0, a["n"] + b["n"], 16))      # each rank loaded its half only
