"""Data-parallel host logic on CPU: world_size-2 `gloo` processes must reproduce the
single-process full-batch gradients after the flat all-reduce."""
import os
import socket

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
