"""CPU checks of the callers around the hot path: checkpoint layout, dense branches, data slicing,
loss / metrics, and the engine loop (with a CPU stand-in model -- the real model needs the GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from conftest import load_golden, rel_err


def _ref_state(name):
    g = load_golden(name)
    return g, {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}


def test_msgat72_state_dict_matches_reference_checkpoint_layout():
    from ms_gat_amd import model
    g, ref = _ref_state("msgat72_n32.npz")
    net = model.msgat72(n_components=3, in_channels=3, in_timesteps=12, out_timesteps=12, use_te=True,
                        adj=torch.from_numpy(g["p.adj"]))
    ours = net.state_dict()
    assert list(ours.keys()) == list(ref.keys())            # same names in the same order
    for k in ref:
        assert tuple(ours[k].shape) == tuple(ref[k].shape), k
    assert not net.adj.requires_grad and "adj" in ours      # frozen parameter, still in the checkpoint
    missing = net.load_state_dict(ref)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == sum(
        v.numel() for k, v in ref.items() if k != "adj")


@pytest.mark.parametrize("name,widths", [("ms-gat48", 48), ("ms-gat", 72), ("ms-gat96", 96)])
def test_factories_and_init_scheme(name, widths):
    from ms_gat_amd import model
    torch.manual_seed(0)
    net = model.build_msgat(name, n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True,
                            adj=torch.eye(20))
    assert net.tpcs[0].tgacns[1].gacn.W.shape == (widths // 3, widths)
    assert len(net.tpcs[0].tgacns) == 2
    for n, p in net.named_parameters():                      # msgat.py:206-217
        if p.dim() == 1:
            assert p.abs().max() <= p.size(0) ** -0.5 + 1e-6, n
    w = net.te.h_ebd.weight
    assert abs(float(w.std()) - (2.0 / (w.size(0) + w.size(1))) ** 0.5) < 0.1 * float(w.std())


def test_use_te_false_uses_the_static_gate():
    from ms_gat_amd import model
    net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=False,
                        adj=torch.eye(8))
    assert net.te is None and tuple(net.W.shape) == (2, 8, 12)
    assert "W" in net.state_dict() and not any(k.startswith("te.") for k in net.state_dict())


def test_branch_oracle_with_our_state_dict_matches_reference_meam():
    """The dense restatement of a whole MEAM (oracle/dense_torch.py: TACN, CACN, GACN, residual tail) fed
    with OUR module's parameters after loading the reference's state_dict reproduces the reference block:
    pins the oracle the GPU branch kernels are checked against, and the parameter layout."""
    from ms_gat_amd import model
    from oracle import dense_torch
    for tag, cin, n in (("3to72_n32", 3, 32), ("72to72_n32", 72, 32), ("72to72_n64", 72, 64)):
        g, ref = _ref_state(f"meam_{tag}.npz")
        m = model.MEAM(cin, 72, n_nodes=n, n_timesteps=12, dilations=[1, 2])
        m.load_state_dict(ref)
        x, adj = torch.from_numpy(g["x"]).float(), torch.from_numpy(g["adj"])
        out = dense_torch.meam_dense(x, adj, dict(m.state_dict()), [1, 2])
        assert rel_err(out.detach(), g["out"]) < 1e-5, tag


def test_layernorm_t_keeps_reference_keys_and_refuses_cpu():
    from ms_gat_amd import _lib, model
    ln = model.LayerNormT(12)
    assert set(ln.state_dict()) == {"weight", "bias"} and tuple(ln.weight.shape) == (12,)
    assert ln.eps == 1e-5
    with pytest.raises(_lib.MsgatError):
        ln(torch.zeros(2, 3, 4, 12))


def test_periodic_windows_and_zscore_match_reference_slices():
    from ms_gat_amd import data
    g = load_golden("slices_small.npz")
    raw = torch.from_numpy(g["raw"]).float().transpose(0, -1)
    hours, tau, q = g["hours"].tolist(), int(g["tau"]), int(g["q"])
    intervals, train_end = data.split_intervals(raw.size(-1), hours, q, tau)
    assert list(intervals[0]) == g["interval"].tolist()
    normed = data.zscore(raw, split=train_end)
    assert torch.allclose(normed, torch.from_numpy(g["norm"]), atol=1e-6)
    ds = data.PeriodicWindows(normed, raw[0], intervals[0], hours, q, tau)
    assert len(ds) == int(g["length"])
    for k, i in enumerate(g["idx"].tolist()):
        x, h, d, y = ds[i]
        assert torch.equal(x, torch.from_numpy(g["x"][k]))
        assert int(h) == int(g["h"][k]) and int(d) == int(g["d"][k])
        assert torch.equal(y, torch.from_numpy(g["y"][k]))
    # the three splits tile the forecast origins 60/20/20 without overlap
    assert intervals[0][1] == intervals[1][0] and intervals[1][1] == intervals[2][0]


def test_synthetic_pems_batches_have_the_model_input_shapes():
    from ms_gat_amd import data
    ds = data.SyntheticPEMS(n_nodes=20, n_edges=25, n_channels=3, in_hours=[1, 2, 24], batch_size=4, days=4)
    x, h, d, y = next(iter(ds.training))
    assert tuple(x.shape) == (4, 3, 3, 20, 12) and tuple(y.shape) == (4, 20, 12)
    assert h.dtype == torch.int64 and int(h.max()) < 24 and int(d.max()) < 7
    assert torch.allclose(ds.adj, ds.adj.t())


def test_huber_and_metrics_match_reference_definitions():
    from ms_gat_amd import engine
    m = load_golden("msgat72_n32.npz")
    pred, y = torch.from_numpy(m["pred"]), torch.from_numpy(m["Y"])
    assert abs(float(engine.huber_loss(pred, y, 50.0)) - float(m["loss"])) < 1e-5 * abs(float(m["loss"]))
    met = engine.Metrics()
    met.update(pred, y)
    met.update(pred * 0.5, y)
    n = 2 * y.numel()
    ae = (pred - y).abs().sum() + (0.5 * pred - y).abs().sum()
    mask = y > 0
    ape = 100 * (((pred - y)[mask] / y[mask]).abs().sum() + ((0.5 * pred - y)[mask] / y[mask]).abs().sum())
    se = ((pred - y) ** 2).sum() + ((0.5 * pred - y) ** 2).sum()
    assert abs(met.MAE - float(ae) / n) < 1e-4 * met.MAE
    assert abs(met.MAPE - float(ape) / n) < 1e-4 * met.MAPE       # divides by ALL entries, as metrics.py:31
    assert abs(met.RMSE - (float(se) / n) ** 0.5) < 1e-4 * met.RMSE


class _TinyModel(nn.Module):
    """Stand-in with the MSGAT call signature model(X, H, D) -> [B,N,T]."""

    def __init__(self, R, C, T):
        super().__init__()
        self.mix = nn.Conv2d(R * C, 1, 1)
        self.h = nn.Embedding(24, 1)

    def forward(self, X, H, D):
        B, R, C, N, T = X.shape
        return self.mix(X.reshape(B, R * C, N, T)).squeeze(1) + self.h(H).view(B, 1, 1)


def test_trainer_loop_checkpoint_and_evaluator_roundtrip(tmp_path):
    from ms_gat_amd import data, engine
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=6, n_edges=6, n_channels=2, in_hours=[1, 2], batch_size=16, days=3)
    model = _TinyModel(2, 2, 12)
    tr = engine.Trainer(model, loss_delta=50.0, out_dir=str(tmp_path))
    tr.max_epochs, tr.min_epochs = 3, 1
    tr.fit((ds.training, ds.validation))
    assert tr.epoch == 4 and tr.best["epoch"] >= 2
    log = (tmp_path / "run.log").read_text().strip().splitlines()
    assert len(log) == 6 and "[Train   ]" in log[0] and "epoch=1,loss=" in log[0] and "RMSE=" in log[1]
    ck = torch.load(tr.best["ckpt"], weights_only=False)
    assert set(ck) == {"best", "epoch", "model", "optimizer", "scheduler", "grad_scaler"}   # engine.py:136-143

    model2 = _TinyModel(2, 2, 12)
    ev = engine.Evaluator(model2, 50.0, str(tmp_path / "eval"), tr.best["ckpt"])
    for a, b in zip(model2.state_dict().values(), ck["model"].values()):
        assert torch.equal(a, b)
    loss = ev.eval(ds.evaluation)
    assert np.isfinite(loss) and "[Evaluate]" in (tmp_path / "eval" / "run.log").read_text()

    tr2 = engine.Trainer(_TinyModel(2, 2, 12), 50.0, str(tmp_path / "resume"))
    tr2.load(tr.best["ckpt"])
    assert tr2.epoch == ck["epoch"] + 1 and tr2.best["loss"] == ck["best"]["loss"]
    # DataParallel-style prefixes written by the reference (main.py:54) are accepted
    assert list(engine.strip_data_parallel_prefix({"module.a": 1, "module.b": 2})) == ["a", "b"]


def test_components_are_stackable_only_when_their_architectures_agree():
    from ms_gat_amd import model, stacked
    net = model.msgat72(n_components=3, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=torch.eye(6))
    assert net.stack_components and stacked.can_stack(net)
    odd = model.MSGAT([{"channels": [1, 48, 48], "dilations": [[1, 2], [2, 4]]},
                       {"channels": [1, 72, 72], "dilations": [[1, 2], [2, 4]]}], in_timesteps=12, out_timesteps=12,
                      use_te=True, adj=torch.eye(6))
    assert not stacked.can_stack(odd)
    p = torch.arange(6.0).view(2, 3)
    assert torch.equal(stacked._per_group(p, 2), torch.tensor([[0., 1, 2], [0, 1, 2], [3, 4, 5], [3, 4, 5]]))


def test_checkpoints_interchange_with_the_reference_loader(tmp_path):
    """What /root/reference/src/engine.py:148-157 does with a checkpoint: GradScaler.load_state_dict on the
    `grad_scaler` entry (an empty dict raises there), Adam.load_state_dict on `optimizer`, StepLR on `scheduler`;
    and the other direction: a checkpoint written through nn.DataParallel (`module.` keys, main.py:54,60) loads here."""
    from torch import optim
    from ms_gat_amd import data, engine
    torch.manual_seed(1)
    ds = data.SyntheticPEMS(n_nodes=6, n_edges=6, n_channels=2, in_hours=[1, 2], batch_size=16, days=2)
    model = _TinyModel(2, 2, 12)
    tr = engine.Trainer(model, 50.0, str(tmp_path))
    tr.run_epoch(ds.training, epoch=1, mode="train")
    ckpt = tmp_path / "a.pkl"
    tr.save(ckpt)
    states = torch.load(ckpt, weights_only=False)

    # the reference's load sequence, object for object
    ref_model = _TinyModel(2, 2, 12)
    ref_opt = optim.Adam(ref_model.parameters(), lr=1e-3, weight_decay=5e-4)
    ref_sched = optim.lr_scheduler.StepLR(ref_opt, step_size=30, gamma=0.1)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    fresh = torch.amp.GradScaler("cuda", enabled=True).state_dict() if torch.cuda.is_available() else None
    if scaler.is_enabled():                                  # on a CUDA-less host torch disables the scaler with a warning
        scaler.load_state_dict(states["grad_scaler"])
        assert states["grad_scaler"] == fresh
    assert set(states["grad_scaler"]) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert states["grad_scaler"] == engine.default_grad_scaler_state() and states["grad_scaler"]["scale"] == 65536.0
    ref_model.load_state_dict(states["model"])
    ref_opt.load_state_dict(states["optimizer"])
    ref_sched.load_state_dict(states["scheduler"])
    assert isinstance(states["optimizer"]["param_groups"][0]["lr"], float)
    st = ref_opt.state[next(iter(ref_model.parameters()))]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) > 0

    # a DataParallel-written checkpoint: every model key prefixed with `module.`
    states["model"] = {"module." + k: v for k, v in states["model"].items()}
    torch.save(states, tmp_path / "dp.pkl")
    tr2 = engine.Trainer(_TinyModel(2, 2, 12), 50.0, str(tmp_path / "resume"))
    tr2.load(tmp_path / "dp.pkl")
    m3 = _TinyModel(2, 2, 12)
    engine.Evaluator(m3, 50.0, str(tmp_path / "eval"), tmp_path / "dp.pkl")
    for a, b, c in zip(model.state_dict().values(), tr2.model.state_dict().values(), m3.state_dict().values()):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_file_based_loader_matches_the_reference_loader(tmp_path):
    """`MSGATData` on the on-disk formats of the reference -- csv `from,to,cost` with a header
    (data_loader.py:60-63), npz key `data` [T_total,N,C] (:71), a `data/meta.yaml` entry (:37-43) -- against what
    `DataLoaderForMSGAT` itself produced from the same files (tests/golden/make_golden.py loader_case): adjacency,
    split sizes, and the first / last batch of every split, bit for bit."""
    from ms_gat_amd import data
    g = load_golden("loader_tiny.npz")
    (tmp_path / "data").mkdir()
    with open(tmp_path / "data" / "tiny.csv", "w") as f:
        f.write("from,to,cost\n")
        for s, d in g["edges"].tolist():
            f.write(f"{s},{d},1.250\n")                    # the cost column is ignored (data_loader.py:61)
    np.savez(tmp_path / "data" / "tiny.npz", data=g["series"])
    meta = tmp_path / "data" / "meta.yaml"
    meta.write_text(f"tiny:\n    adj-file: {tmp_path}/data/tiny.csv\n    data-file: {tmp_path}/data/tiny.npz\n"
                    f"    num-nodes: {int(g['n'])}\n    num-channels: {int(g['c'])}\n    timesteps-per-hour: {int(g['tau'])}\n")
    ds = data.MSGATData("tiny", g["hours"].tolist(), int(g["q"]), int(g["batch_size"]), num_workers=0, meta_file=str(meta))
    assert (ds.num_nodes, ds.num_channels, ds.timesteps_per_hour) == (int(g["n"]), int(g["c"]), int(g["tau"]))
    assert torch.allclose(ds.adj, torch.from_numpy(g["adj"]), atol=1e-7)        # D^-1/2 (A+I) D^-1/2 (:59-66)
    loaders = (ds.training, ds.validation, ds.evaluation)
    assert [len(l.dataset) for l in loaders] == g["lengths"].tolist()
    assert [len(l) for l in loaders] == g["n_batches"].tolist()
    items = [ds.training.dataset[i] for i in range(int(g["batch_size"]))]      # the training loader shuffles: dataset order
    for k, name in enumerate("xhdy"):
        assert torch.equal(torch.stack([it[k] for it in items]), torch.from_numpy(g[f"train_{name}"])), name
    for split, loader in (("val", ds.validation), ("eval", ds.evaluation)):
        batches = list(loader)
        for k, name in enumerate("xhdy"):
            assert torch.equal(batches[0][k], torch.from_numpy(g[f"{split}_{name}"])), (split, name)
            assert torch.equal(batches[-1][k], torch.from_numpy(g[f"{split}_last_{name}"])), (split, name)
    x, h, d, y = next(iter(ds.training))                                       # and the training loader does deliver batches
    assert tuple(x.shape) == (4, 3, 2, 7, 12) and tuple(y.shape) == (4, 7, 12) and h.dtype == torch.int64
