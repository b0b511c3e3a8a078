"""GPU: the tail of a training step in the library -- fused Huber loss + metric sums and
flat Adam -- against PyTorch's own ops, plus the boundary behaviours around them (HIP-graph replays of several
batch shapes, autocast regions, checkpoints)."""
import copy

import numpy as np
import pytest
import torch
from torch import nn, optim

from conftest import record_err, rel_err

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape,delta", [((32, 883, 12), 50.0), ((3, 7, 12), 1.0), ((1, 1, 1), 50.0)])
def test_fused_huber_and_metric_sums_match_torch_in_float64(shape, delta):
    """loss.py:51-52 and metrics.py:20-35 in one pass: loss, its gradient, and the AE / APE / SE totals."""
    from ms_gat_amd import engine, ops
    g = torch.Generator().manual_seed(5)
    y = (torch.randn(shape, generator=g) * 40 + 60).to(_dev())          # some targets <= 0: the MAPE mask matters
    p = (y.cpu() + torch.randn(shape, generator=g) * 45).to(_dev()).requires_grad_(True)
    sums = torch.zeros(4, device=_dev(), dtype=torch.float64)
    loss = ops.huber_metrics(p, y, delta, 0.0, sums)
    loss.backward(torch.tensor(1.7, device=_dev()))
    pd = p.detach().double().requires_grad_(True)
    want = engine.huber_loss(pd, y.double(), delta)
    want.backward(torch.tensor(1.7, device=_dev(), dtype=torch.float64))
    assert abs(float(loss) - float(want)) < 1e-6 * abs(float(want))
    assert rel_err(p.grad, pd.grad) < 1e-6
    m = engine.Metrics()
    m.update(p.detach(), y)
    for k in range(3):
        e = abs(float(sums[k]) - float(m._sums[k])) / max(abs(float(m._sums[k])), 1e-30)
        record_err(f"fused huber+metrics {shape}", ("AE", "APE", "SE")[k], e, 1e-6)
        assert e < 1e-6
    assert abs(float(sums[3]) - float(want)) < 1e-6 * abs(float(want))
    # a second call accumulates
    ops.huber_metrics(p.detach(), y, delta, 0.0, sums)
    assert abs(float(sums[0]) - 2 * float(m._sums[0])) < 1e-6 * float(sums[0])
    # reproducible bit for bit
    a, b = (ops.huber_metrics(p.detach(), y, delta) for _ in range(2))
    assert torch.equal(a, b)


def _param_set(dev):
    torch.manual_seed(3)
    shapes = [(7,), (24, 72), (3, 12, 12), (5000,), (24, 4097), (1,)]   # below / across / exactly on 2048-element chunks
    return [nn.Parameter(torch.randn(s, device=dev)) for s in shapes]


def test_flat_adam_tracks_torch_adam_step_for_step():
    from ms_gat_amd import engine
    dev = _dev()
    ours, theirs = _param_set(dev), _param_set(dev)
    a = engine.FlatAdam(ours, lr=1e-3, weight_decay=5e-4)
    b = optim.Adam(theirs, lr=1e-3, weight_decay=5e-4)
    g = torch.Generator(device=dev).manual_seed(9)
    for step in range(12):
        if step == 6:                                   # StepLR's edit of the host-side rate reaches the device
            for opt in (a, b):
                opt.param_groups[0]["lr"] = 1e-4
        for p, q in zip(ours, theirs):
            grad = torch.randn(p.shape, device=dev, generator=g) * (0.1 + step)
            p.grad, q.grad = grad.clone(), grad.clone()
        if step == 3:                                   # torch's Adam skips parameters without a gradient
            ours[1].grad = theirs[1].grad = None
        a.step()
        b.step()
    for i, (p, q) in enumerate(zip(ours, theirs)):
        e = rel_err(p.detach(), q.detach())
        record_err("FlatAdam vs torch.optim.Adam, 12 steps", f"param{i}{tuple(p.shape)}", e, 1e-6)
        assert e < 1e-6, i
        assert rel_err(a.state[p]["exp_avg"], b.state[q]["exp_avg"]) < 1e-6
        assert rel_err(a.state[p]["exp_avg_sq"], b.state[q]["exp_avg_sq"]) < 1e-6
    assert a._host_steps == [12, 11, 12, 12, 12, 12] and a._dev_steps.tolist() == [12.0, 11.0, 12.0, 12.0, 12.0, 12.0]


def test_flat_adam_and_torch_adam_load_each_others_state():
    from ms_gat_amd import engine
    dev = _dev()

    def run(opt, params, steps, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        for _ in range(steps):
            for p in params:
                p.grad = torch.randn(p.shape, device=dev, generator=g)
            opt.step()

    ours, theirs = _param_set(dev), _param_set(dev)
    a, b = engine.FlatAdam(ours, lr=1e-3, weight_decay=5e-4), optim.Adam(theirs, lr=1e-3, weight_decay=5e-4)
    run(a, ours, 4, 1)
    run(b, theirs, 4, 1)
    # our state into a fresh torch Adam, torch's into a fresh FlatAdam; both continue identically
    ours2, theirs2 = [nn.Parameter(p.detach().clone()) for p in theirs], [nn.Parameter(p.detach().clone()) for p in ours]
    a2, b2 = engine.FlatAdam(ours2, lr=1e-3, weight_decay=5e-4), optim.Adam(theirs2, lr=1e-3, weight_decay=5e-4)
    a2.load_state_dict(copy.deepcopy(b.state_dict()))
    b2.load_state_dict(copy.deepcopy(a.state_dict()))
    assert set(a2._host_steps) == {4} and float(next(iter(b2.state.values()))["step"]) == 4.0
    run(a2, ours2, 3, 2)
    run(b2, theirs2, 3, 2)
    for p, q in zip(ours2, theirs2):
        assert rel_err(p.detach(), q.detach()) < 1e-6


def _small_model_and_batches():
    from ms_gat_amd import data, model
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=40, n_edges=50, n_channels=1, in_hours=[1, 2], batch_size=8, days=2)
    net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj)
    batches = [[t.to(_dev()) for t in b] for _, b in zip(range(3), ds.training)]
    return net.to(_dev()), batches


def test_replaying_an_older_graph_hands_its_own_gradients_to_the_eager_optimizer(tmp_path):
    """Multi-rank steps end the captured graph after backward; the all-reduce and the optimizer then read p.grad.
    Every capture binds p.grad to tensors of its own pool, so after a SECOND batch shape was captured a replay of the
    first graph must put its gradients back (round-1 advisor finding: stale gradients)."""
    from ms_gat_amd import engine
    net, batches = _small_model_and_batches()
    small = [t[:3] for t in batches[1]]
    tr = engine.Trainer(net, 50.0, str(tmp_path), hip_graph=True)
    with torch.enable_grad():
        g_full = engine._GraphedStep(tr, batches[0], training=True, step_in_graph=False)
        g_small = engine._GraphedStep(tr, small, training=True, step_in_graph=False)   # re-points p.grad
    g_full.replay(batches[2])
    got = [p.grad.clone() for p in net.parameters() if p.requires_grad]
    net.zero_grad(set_to_none=True)
    *inputs, truth = batches[2]
    tr._loss(net(*inputs), truth, None).backward()
    want = [p.grad for p in net.parameters() if p.requires_grad]
    for a, b in zip(got, want):
        assert rel_err(a, b) < 1e-5
    g_small.replay(small)
    assert all(p.grad.data_ptr() == g.data_ptr() for p, g in g_small.grads)


def test_forward_under_autocast_is_the_fp32_forward():
    """The reference wraps the forward in CUDA AMP (engine.py:54-56).  The library ops switch autocast off inside
    and take fp32, so the same call under autocast gives the fp32 result -- also when a neighbouring matmul handed
    them half-precision inputs."""
    import ms_gat_amd
    from ms_gat_amd import model
    dev = _dev()
    torch.manual_seed(2)
    adj = ms_gat_amd.synthetic_adjacency(30, 40, 0).to(dev)
    x = torch.randn(2, 6, 30, 12, device=dev)
    gacn = ms_gat_amd.GACN(6, 4, 12).to(dev)
    for p in gacn.parameters():
        nn.init.normal_(p, std=0.3)
    meam = model.MEAM(6, 12, n_nodes=30, n_timesteps=12, dilations=[1, 2]).to(dev)
    for p in meam.parameters():
        nn.init.normal_(p, std=0.2)
    for mod, tol in ((gacn, 0.0), (meam, 2e-3)):     # MEAM's tiny [C,C] / [T,T] attention matmuls do run in half
        want = mod(x, adj)
        with torch.autocast("cuda"):
            got = mod(x, adj)
            from_half = mod(x.half(), adj)               # what an autocast matmul upstream would hand over
        assert got.dtype == torch.float32 and from_half.dtype == torch.float32
        assert rel_err(got, want) <= tol
        assert rel_err(from_half, mod(x.half().float(), adj)) <= tol
    with pytest.raises(TypeError):
        gacn(x.half(), adj)                              # outside autocast the parity type is enforced


def test_msgat72_training_step_under_autocast_like_the_reference_engine(tmp_path):
    """engine.py:54-63: forward + loss under autocast, scaled backward, optimizer step -- runs, and stays close to fp32."""
    from ms_gat_amd import engine
    net, batches = _small_model_and_batches()
    *inputs, truth = batches[0]
    loss_fn = engine.HuberLoss(50.0)
    ref = loss_fn(net(*inputs), truth)
    scaler = torch.amp.GradScaler("cuda")
    opt = optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
    with torch.autocast("cuda"):
        pred = net(*inputs)
        loss = loss_fn(pred, truth)
    opt.zero_grad()
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    assert loss.dtype == torch.float32 and abs(float(loss) - float(ref)) < 2e-3 * abs(float(ref))
    assert all(torch.isfinite(p).all() for p in net.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("R,B,N,To,te", [(3, 32, 883, 12, True), (1, 2, 5, 3, True), (5, 7, 33, 12, True), (3, 4, 50, 12, False)])
def test_gate_sum_is_the_models_last_line(R, B, N, To, te):
    """ops.gate_sum = sum_r pred_r * (h_ebd(H) + d_ebd(D)).view(B,R,N,To)[:, r] (msgat.py:203-205, embeddings.py:36-39):
    the forward bit for bit against the same torch ops in the reference's order, the gradients (pred and both DENSE
    embedding-table gradients) against float64."""
    from ms_gat_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(R * 100 + B)
    pred = torch.randn(R, B, N, To, generator=gen).to(dev).requires_grad_(True)
    dout = torch.randn(B, N, To, generator=gen).to(dev)
    if te:
        h_w = torch.randn(24, R * N * To, generator=gen).to(dev).requires_grad_(True)
        d_w = torch.randn(7, R * N * To, generator=gen).to(dev).requires_grad_(True)
        H = torch.randint(0, 24, (B,), generator=gen).to(dev)
        D = torch.randint(0, 7, (B,), generator=gen).to(dev)
        out = ops.gate_sum(pred, H, D, h_w, d_w)
        params = [pred, h_w, d_w]
    else:
        W = torch.randn(R, N, To, generator=gen).to(dev).requires_grad_(True)
        out = ops.gate_sum(pred, None, None, W)
        params = [pred, W]
    out.backward(dout)
    got = [out.detach()] + [p.grad for p in params]

    def reference(dtype):
        ps = [p.detach().to(dtype).requires_grad_(True) for p in params]
        if te:
            gate = (torch.nn.functional.embedding(H, ps[1]) + torch.nn.functional.embedding(D, ps[2])).view(B, R, N, To)
            gates = gate.unbind(1)
        else:
            gates = ps[1].unbind(0)
        o = None
        for r in range(R):                                   # msgat.py:204-205: the generator summed left to right
            term = ps[0][r] * gates[r]
            o = term if o is None else o + term
        o.backward(dout.to(dtype))
        return [o.detach()] + [p.grad for p in ps]

    same = reference(torch.float32)
    assert torch.equal(got[0], same[0])
    ref = reference(torch.float64)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        err = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        assert err < 1e-5, err
    if te:     # rows no sample selected: exactly zero, like nn.Embedding's dense gradient
        unused = torch.ones(24, dtype=torch.bool, device=dev)
        unused[H] = False
        assert not got[2][unused].any()
