#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (it has /root/reference; the GPU box does not):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported from /root/reference/src, fed seeded synthetic
inputs, and its outputs / autograd gradients are stored as small .npz files.
Nothing of the reference's source travels: the fixtures hold data only.

Fixtures:
  gatt_*.npz   GraphAttention fwd+bwd   (attention.py:12-39)
  gacn_*.npz   GACN fwd+bwd             (msgat.py:17-31)
  meam_*.npz   MEAM fwd+bwd with its full state_dict (msgat.py:103-134)
  msgat72_n32.npz  msgat72 fwd + HuberLoss(50) + all grads (msgat.py:166-229, loss.py:51-52)
  adj_n12.npz  sym-normalised adjacency from a csv edge list (data_loader.py:49-66)
  slices_*.npz TimeSeriesSlice / normalize (data_loader.py:92-120)
  loader_tiny.npz  DataLoaderForMSGAT on csv + npz + meta.yaml files (data_loader.py:29-89)
  msgat72_cfg1_pemsd4.npz  BASELINE.json configs[0]: msgat72 forward, N=307, 3 features, B=4, five components
  gacn_headline_n883.npz   GACN(72 -> 24) fwd+bwd at the HEADLINE graph size (PEMSD7-like N=883, 866 edges, T=12), one
                           sample: the size bench.py reports on is pinned by the reference itself, not only by the oracle.
                           Inputs are stored as int8 multiples of 1/32 (exact in fp32), the adjacency as its non-zeros.

  meam_72to72_n64.npz      MEAM(72 -> 72) fwd+bwd at N = 64: N*T = 768 positions per channel slab, past the 512 below which
                           the library's one-pass convolution backward and segmented mixing forms are not selected -- the
                           reference's own MEAM gradients reach those kernels.  x and dout are fp16-exact (stored as fp16).

  gacn_w48_n64.npz, gacn_w96_n64.npz   GACN(48 -> 16) / GACN(96 -> 32) fwd+bwd at N = 64 (768 positions per slab): the
                           widths of the other two models of the registry (msgat.py:220-229) reach the library's LDS-DMA
                           one-pass backward forms <2,3,128,3,1> / <3,6,64,3,2> with the reference's own gradients.
                           Self-contained (x, adj, parameters, dz); x and dz fp16-exact.
  meam_96to96_n64.npz      MEAM(96 -> 96), dilations [4,4] (the second block of msgat96), fwd+bwd at N = 64: the forms
                           <9,4,64,3,2> (130 x 97 merged mixing) and <6,4,64,3,2> (96 x 97 residual convolution).
  meam_48to48_n64.npz      MEAM(48 -> 48), dilations [2,4] (second block of msgat48): <5,4,64,3,2> and <3,4,128,3,1>.
  msgat72_traj_n32.npz     msgat72 (three components, N = 32, B = 2) stepped FIVE times by the reference loop's own sequence
                           (engine.py:56-63,106: zero_grad, backward, Adam(lr 1e-3, weight_decay 5e-4); HuberLoss(50)), one batch
                           per step: initial state_dict, the batches, the five losses, every parameter after step 5.

    python tests/golden/make_golden.py --only headline     # just that one
    python tests/golden/make_golden.py --only meam64
    python tests/golden/make_golden.py --only widths       # gacn_w48/w96 and meam_48to48 / meam_96to96
    python tests/golden/make_golden.py --only trajectory   # msgat72_traj_n32
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

from models.attention import GraphAttention  # noqa: E402
from models.msgat import GACN, MEAM, msgat72  # noqa: E402
from loss import HuberLoss  # noqa: E402
import data_loader as ref_dl  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def synthetic_adjacency(n, n_edges, seed):
    """N nodes, E distinct undirected non-self edges, then D^-1/2 (A+I) D^-1/2."""
    rng = np.random.default_rng(seed)
    a = np.eye(n, dtype=np.float64)
    got = 0
    while got < n_edges:
        s, d = rng.integers(0, n, size=2)
        if s == d or a[s, d] != 0:
            continue
        a[s, d] = a[d, s] = 1.0
        got += 1
    r = 1.0 / np.sqrt(a.sum(1))
    return (r[:, None] * a * r[None, :]).astype(np.float32)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def gatt_case(tag, B, C, N, T, n_edges, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, N, T)).astype(np.float32)
    adj = synthetic_adjacency(N, n_edges, seed + 1)
    Wg = (rng.standard_normal((T, T)) * (2.0 / (T + T)) ** 0.5).astype(np.float32)
    alpha = rng.uniform(-C ** -0.5, C ** -0.5, size=C).astype(np.float32)
    dy = rng.standard_normal((B, C, N, T)).astype(np.float32)

    m = GraphAttention(C, T)
    with torch.no_grad():
        m.Wg.copy_(t(Wg))
        m.alpha.copy_(t(alpha))
    xt = t(x).requires_grad_(True)
    y = m(xt, t(adj))
    y.backward(t(dy))
    save(f"gatt_{tag}.npz", x=x, adj=adj, Wg=Wg, alpha=alpha, dy=dy,
         y=y, dx=xt.grad, dWg=m.Wg.grad, dalpha=m.alpha.grad)

    # GACN on the same inputs
    O = 24
    W = (rng.standard_normal((O, C)) * (2.0 / (O + C)) ** 0.5).astype(np.float32)
    dz = rng.standard_normal((B, O, N, T)).astype(np.float32)
    g = GACN(C, O, T)
    with torch.no_grad():
        g.gatt.Wg.copy_(t(Wg))
        g.gatt.alpha.copy_(t(alpha))
        g.W.copy_(t(W))
    xt = t(x).requires_grad_(True)
    z = g(xt, t(adj))
    z.backward(t(dz))
    # x/adj/Wg/alpha are shared with gatt_{tag}.npz and not stored twice
    save(f"gacn_{tag}.npz", W=W, dz=dz, z=z, dx=xt.grad, dWg=g.gatt.Wg.grad,
         dalpha=g.gatt.alpha.grad, dW=g.W.grad)


def headline_case(seed):
    """GACN(72 -> 24) on one sample of the PEMSD7-sized graph (the second-MEAM width of msgat72, msgat.py:220-229)."""
    B, C, O, N, T, E = 1, 72, 24, 883, 12, 866
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, N, T))
    x = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-5)     # LayerNorm output, as msgat.py:122 feeds it
    xq = np.clip(np.rint(x * 32), -127, 127).astype(np.int8)                           # multiples of 1/32: exact in fp32
    dzq = np.clip(np.rint(rng.standard_normal((B, O, N, T)) * 32), -127, 127).astype(np.int8)
    x32, dz32 = xq.astype(np.float32) / 32, dzq.astype(np.float32) / 32
    adj = synthetic_adjacency(N, E, seed + 1)
    Wg = (rng.standard_normal((T, T)) * (2.0 / (T + T)) ** 0.5).astype(np.float32)
    alpha = rng.uniform(-C ** -0.5, C ** -0.5, size=C).astype(np.float32)
    W = (rng.standard_normal((O, C)) * (2.0 / (O + C)) ** 0.5).astype(np.float32)
    g = GACN(C, O, T)
    with torch.no_grad():
        g.gatt.Wg.copy_(t(Wg))
        g.gatt.alpha.copy_(t(alpha))
        g.W.copy_(t(W))
    xt = t(x32).requires_grad_(True)
    z = g(xt, t(adj))
    z.backward(t(dz32))
    rows, cols = np.nonzero(adj)
    out = dict(x_q32=xq, dz_q32=dzq, adj_rows=rows.astype(np.int32), adj_cols=cols.astype(np.int32), adj_vals=adj[rows, cols],
               n_nodes=np.int32(N), Wg=Wg, alpha=alpha, W=W, z=z.detach().numpy(), dx=xt.grad.numpy(),
               dWg=g.gatt.Wg.grad.numpy(), dalpha=g.gatt.alpha.grad.numpy(), dW=g.W.grad.numpy())
    path = os.path.join(HERE, "gacn_headline_n883.npz")
    np.savez_compressed(path, **out)
    print(f"gacn_headline_n883.npz: {os.path.getsize(path) / 1024:.0f} KiB")


def gacn_width_case(tag, C, O, N, B, seed):
    """GACN(C -> O) on its own inputs (LayerNorm output, as msgat.py:122 feeds it), fp16-exact x and dz."""
    T = 12
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, N, T))
    x = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-5)
    x16, dz16 = x.astype(np.float16), rng.standard_normal((B, O, N, T)).astype(np.float16)
    adj = synthetic_adjacency(N, N + 6, seed + 1)
    Wg = (rng.standard_normal((T, T)) * (2.0 / (T + T)) ** 0.5).astype(np.float32)
    alpha = rng.uniform(-C ** -0.5, C ** -0.5, size=C).astype(np.float32)
    W = (rng.standard_normal((O, C)) * (2.0 / (O + C)) ** 0.5).astype(np.float32)
    g = GACN(C, O, T)
    with torch.no_grad():
        g.gatt.Wg.copy_(t(Wg))
        g.gatt.alpha.copy_(t(alpha))
        g.W.copy_(t(W))
    xt = t(x16.astype(np.float32)).requires_grad_(True)
    z = g(xt, t(adj))
    z.backward(t(dz16.astype(np.float32)))
    save(f"gacn_{tag}.npz", x=x16, adj=adj, Wg=Wg, alpha=alpha, W=W, dz=dz16, z=z, dx=xt.grad, dWg=g.gatt.Wg.grad,
         dalpha=g.gatt.alpha.grad, dW=g.W.grad)


def meam_case(tag, cin, cout, N, B, seed, half_inputs=False, dilations=(1, 2)):
    torch.manual_seed(seed)
    T = 12
    m = MEAM(cin, cout, n_nodes=N, n_timesteps=T, dilations=list(dilations))
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim >= 2:
                torch.nn.init.xavier_normal_(p)
            else:
                torch.nn.init.uniform_(p, -p.size(0) ** -0.5, p.size(0) ** -0.5)
    adj = synthetic_adjacency(N, N, seed + 1)
    x = torch.randn(B, cin, N, T)
    dout = torch.randn(B, cout, N, T)
    if half_inputs:   # exactly representable in fp16: stored at half the size
        x, dout = x.half().float(), dout.half().float()
    x.requires_grad_(True)
    out = m(x, t(adj))
    out.backward(dout)
    arrays = {f"p.{k}": v for k, v in m.state_dict().items()}
    arrays.update({f"g.{k}": p.grad for k, p in m.named_parameters()})
    if half_inputs:
        save(f"meam_{tag}.npz", x=x.detach().half(), adj=adj, dout=dout.half(), out=out, dx=x.grad, **arrays)
    else:
        save(f"meam_{tag}.npz", x=x, adj=adj, dout=dout, out=out, dx=x.grad, **arrays)


def msgat_case(seed):
    torch.manual_seed(seed)
    N, B, R, C, T = 32, 2, 3, 3, 12
    adj = synthetic_adjacency(N, N, seed + 1)
    net = msgat72(n_components=R, in_channels=C, in_timesteps=T, out_timesteps=T, use_te=True, adj=t(adj))
    X = torch.randn(B, R, C, N, T)
    H = torch.randint(0, 24, (B,))
    D = torch.randint(0, 7, (B,))
    Y = torch.randn(B, N, T) * 60.0  # large enough that both Huber branches are taken at delta=50
    pred = net(X, H, D)
    loss = HuberLoss(50.0)(pred, Y)
    loss.backward()
    arrays = {f"p.{k}": v for k, v in net.state_dict().items()}
    arrays.update({f"g.{k}": p.grad for k, p in net.named_parameters() if p.grad is not None})
    save("msgat72_n32.npz", X=X, H=H, D=D, Y=Y, pred=pred, loss=loss, **arrays)


def trajectory_case(seed, steps=5):
    """The reference model stepped `steps` times the way its training loop does (engine.py:56-63 with the optimizer of
    engine.py:106 and the loss of loss.py:51-52), on the CPU in fp32: no autocast, and a GradScaler that is disabled off
    CUDA, i.e. zero_grad / backward / Adam.step.  One batch per step.  Stored: the initial state_dict, the batches, the
    loss of every step and EVERY parameter after the last step."""
    torch.manual_seed(seed)
    N, B, R, C, T = 32, 2, 3, 3, 12
    adj = synthetic_adjacency(N, N, seed + 1)
    net = msgat72(n_components=R, in_channels=C, in_timesteps=T, out_timesteps=T, use_te=True, adj=t(adj))
    net.train()
    arrays = {f"p.{k}": v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
    loss_fn = HuberLoss(50.0)
    X = torch.randn(steps, B, R, C, N, T).half().float()
    H = torch.randint(0, 24, (steps, B))
    D = torch.randint(0, 7, (steps, B))
    Y = (torch.randn(steps, B, N, T) * 60.0).half().float()
    losses = []
    for k in range(steps):
        pred = net(X[k], H[k], D[k])
        loss = loss_fn(pred, Y[k])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    arrays.update({f"f.{k}": v for k, v in net.named_parameters()})
    save("msgat72_traj_n32.npz", X=X.half(), H=H, D=D, Y=Y.half(), losses=np.asarray(losses, np.float64), **arrays)


def cfg1_case(seed):
    """BASELINE.json configs[0]: PEMSD4-like (307 nodes, 3 features, T=12, B=4), the reference's default five
    components (main.py:14, `-i 1,2,3,24,168`), full msgat72 forward on the CPU.  The time-embedding tables are
    31 x 18 420 floats each component set; only the rows H and D select are stored (the others cannot
    influence the forward), which keeps the fixture under 1 MB."""
    torch.manual_seed(seed)
    N, B, R, C, T = 307, 4, 5, 3, 12
    adj = synthetic_adjacency(N, 340, seed + 1)
    net = msgat72(n_components=R, in_channels=C, in_timesteps=T, out_timesteps=T, use_te=True, adj=t(adj))
    X = torch.randn(B, R, C, N, T).half().float()   # exactly representable in fp16: stored at half the size
    H = torch.randint(0, 24, (B,))
    D = torch.randint(0, 7, (B,))
    with torch.no_grad():
        pred = net(X, H, D)
    arrays = {}
    for k, v in net.state_dict().items():
        if k == "te.h_ebd.weight":
            arrays["te_h_rows"] = v[H]
        elif k == "te.d_ebd.weight":
            arrays["te_d_rows"] = v[D]
        elif k == "adj":
            continue
        else:
            arrays[f"p.{k}"] = v
    rows, cols = np.nonzero(adj)   # the [307,307] adjacency has 987 non-zeros: stored as coordinates
    save("msgat72_cfg1_pemsd4.npz", X=X.half(), H=H, D=D, adj_rows=rows.astype(np.int16), adj_cols=cols.astype(np.int16),
         adj_vals=adj[rows, cols], pred=pred, **arrays)


def adjacency_case():
    """Runs the reference's private csv -> adjacency routine on a temp edge list."""
    n = 12
    edges = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (7, 8), (8, 9), (9, 10), (10, 7), (2, 7), (11, 0), (1, 0)]
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
        f.write("from,to,cost\n")
        for s, d in edges:
            f.write(f"{s},{d},1.0\n")
        path = f.name

    class Stub:
        pass

    stub = Stub()
    stub.num_nodes, stub.adj_file = n, path
    adj = ref_dl.DataLoaderForMSGAT._DataLoaderForMSGAT__load_adj(stub)
    os.unlink(path)
    save("adj_n12.npz", n=np.int64(n), edges=np.asarray(edges, dtype=np.int64), adj=adj)


def slices_case(seed):
    """normalize + TimeSeriesSlice on a small synthetic series (data_loader.py:92-120)."""
    rng = np.random.default_rng(seed)
    tau, q, hours = 12, 12, [1, 2, 24]
    total, N, C = 24 * 12 * 3, 5, 2
    raw = (rng.standard_normal((total, N, C)) * 10 + 50).astype(np.float32)  # npz layout [T_total,N,C]
    data = torch.from_numpy(raw).float().transpose(0, -1)  # data_loader.py:71 -> [C,N,T_total]
    in_timesteps = tau * max(hours)
    length = data.size(-1) - in_timesteps - q + 1
    split1 = int(0.6 * length)
    norm = ref_dl.normalize(data, split=in_timesteps + split1)
    interval = [in_timesteps, in_timesteps + split1]
    ds = ref_dl.TimeSeriesSlice(norm, data[0], interval, hours, q, tau)
    idx = [0, 1, 7, len(ds) - 1]
    items = [ds[i] for i in idx]
    save("slices_small.npz", raw=raw, hours=np.asarray(hours), tau=np.int64(tau), q=np.int64(q),
         norm=norm, interval=np.asarray(interval), length=np.int64(len(ds)), idx=np.asarray(idx),
         x=torch.stack([it[0] for it in items]), h=torch.stack([it[1] for it in items]),
         d=torch.stack([it[2] for it in items]), y=torch.stack([it[3] for it in items]))


def loader_case(seed):
    """The reference's file-based loader end to end (data_loader.py:29-89): a tiny csv edge list (`from,to,cost`
    with a header), an npz series (key `data`, [T_total, N, C]) and a `data/meta.yaml` entry are written to a
    temp directory, `DataLoaderForMSGAT` runs there (it opens the relative path `data/meta.yaml`), and the
    adjacency plus the first batch of every split are stored next to the INPUTS (edges, series, meta values) so
    the test can re-create the files.  The training split is read in dataset order (its loader shuffles)."""
    rng = np.random.default_rng(seed)
    n, c, tau, q, hours, bs = 7, 2, 12, 12, [1, 2, 24], 4
    total = 24 * 12 * 2 + 40
    series = (rng.standard_normal((total, n, c)) * 12 + 60).astype(np.float32)
    edges = [(0, 1), (1, 2), (2, 0), (3, 4), (5, 6), (6, 3), (4, 0)]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data"))
        with open(os.path.join(tmp, "data", "tiny.csv"), "w") as f:
            f.write("from,to,cost\n")
            for s, d in edges:
                f.write(f"{s},{d},{rng.uniform(0.5, 3.0):.3f}\n")
        np.savez(os.path.join(tmp, "data", "tiny.npz"), data=series)
        with open(os.path.join(tmp, "data", "meta.yaml"), "w") as f:
            f.write("tiny:\n    adj-file: data/tiny.csv\n    data-file: data/tiny.npz\n"
                    f"    num-nodes: {n}\n    num-channels: {c}\n    timesteps-per-hour: {tau}\n")
        os.chdir(tmp)
        try:
            dl = ref_dl.DataLoaderForMSGAT("tiny", hours, q, bs, 0)
            arrays = dict(edges=np.asarray(edges, dtype=np.int64), series=series, n=np.int64(n), c=np.int64(c),
                          tau=np.int64(tau), q=np.int64(q), hours=np.asarray(hours), batch_size=np.int64(bs), adj=dl.adj,
                          lengths=np.asarray([len(dl.training.dataset), len(dl.validation.dataset), len(dl.evaluation.dataset)]),
                          n_batches=np.asarray([len(dl.training), len(dl.validation), len(dl.evaluation)]))
            items = [dl.training.dataset[i] for i in range(bs)]
            for k, name in enumerate("xhdy"):
                arrays[f"train_{name}"] = torch.stack([it[k] for it in items])
            for split, loader in (("val", dl.validation), ("eval", dl.evaluation)):
                first = next(iter(loader))
                last = None
                for last in loader:
                    pass
                for k, name in enumerate("xhdy"):
                    arrays[f"{split}_{name}"] = first[k]
                    arrays[f"{split}_last_{name}"] = last[k]
        finally:
            os.chdir(cwd)
    save("loader_tiny.npz", **arrays)


if __name__ == "__main__":
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "headline":
        headline_case(1100)
        sys.exit(0)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "trajectory":
        trajectory_case(1700)
        sys.exit(0)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "meam64":
        meam_case("72to72_n64", 72, 72, 64, 2, 1200, half_inputs=True)
        sys.exit(0)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "widths":
        gacn_width_case("w48_n64", 48, 16, 64, 2, 1300)
        gacn_width_case("w96_n64", 96, 32, 64, 2, 1400)
        meam_case("48to48_n64", 48, 48, 64, 2, 1500, half_inputs=True, dilations=(2, 4))
        meam_case("96to96_n64", 96, 96, 64, 2, 1600, half_inputs=True, dilations=(4, 4))
        sys.exit(0)
    headline_case(1100)
    gatt_case("b2c3n16", 2, 3, 16, 12, 20, 100)
    gatt_case("b2c1n64", 2, 1, 64, 12, 70, 200)
    gatt_case("b2c72n64", 2, 72, 64, 12, 70, 300)
    gatt_case("b2c3n307", 2, 3, 307, 12, 340, 400)
    meam_case("3to72_n32", 3, 72, 32, 2, 500)
    meam_case("72to72_n32", 72, 72, 32, 2, 600)
    msgat_case(700)
    adjacency_case()
    slices_case(800)
    cfg1_case(900)
    loader_case(1000)
    meam_case("72to72_n64", 72, 72, 64, 2, 1200, half_inputs=True)
    gacn_width_case("w48_n64", 48, 16, 64, 2, 1300)
    gacn_width_case("w96_n64", 96, 32, 64, 2, 1400)
    meam_case("48to48_n64", 48, 48, 64, 2, 1500, half_inputs=True, dilations=(2, 4))
    meam_case("96to96_n64", 96, 96, 64, 2, 1600, half_inputs=True, dilations=(4, 4))
    trajectory_case(1700)
