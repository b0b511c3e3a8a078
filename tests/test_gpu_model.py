"""GPU: the reference's MEAM / msgat72 golden vectors with the HIP graph branch swapped in
(SURVEY.md section 8c, G3 and G4), and one engine step on the device."""
import numpy as np
import pytest
import torch

from conftest import assert_parity, load_golden, record_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _state(g):
    return {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}


@pytest.mark.parametrize("tag,cin,n,cout,dilations", [
    ("3to72_n32", 3, 32, 72, [1, 2]), ("72to72_n32", 72, 32, 72, [1, 2]), ("72to72_n64", 72, 64, 72, [1, 2]),
    ("48to48_n64", 48, 64, 48, [2, 4]),     # msgat48's second block: <5,4,64,3,2> (66 x 49), <3,4,128,3,1> (48 x 49)
    ("96to96_n64", 96, 64, 96, [4, 4]),     # msgat96's second block: <9,4,64,3,2> (130 x 97), <6,4,64,3,2> (96 x 97)
])
def test_meam_block_matches_reference_forward_and_backward(tag, cin, n, cout, dilations):
    """The reference's own MEAM outputs and gradients (tests/golden/make_golden.py).  N = 64 gives 768 positions per
    channel slab: the one-pass convolution backward (`k_chanpair_glds<5|7,5,64,3,2>`) and the segmented mixing forms are
    only selected from 512 positions up, so those cases are what pins them to the reference (msgat.py:121-131) -- at the
    second-block widths of all three models of the registry (msgat.py:220-229)."""
    from ms_gat_amd import model
    g = load_golden(f"meam_{tag}.npz")
    m = model.MEAM(cin, cout, n_nodes=n, n_timesteps=12, dilations=dilations)
    m.load_state_dict(_state(g))
    m.to(_dev())
    x = torch.from_numpy(g["x"]).float().to(_dev()).requires_grad_(True)
    out = m(x, torch.from_numpy(g["adj"]).to(_dev()))
    out.backward(torch.from_numpy(g["dout"]).float().to(_dev()))
    what = f"meam_{tag}"
    assert_parity(out.detach().cpu(), g["out"], what, "out")
    assert_parity(x.grad.cpu(), g["dx"], what, "dx")
    for name, p in m.named_parameters():
        assert_parity(p.grad.cpu(), g[f"g.{name}"], what, name)


def test_msgat72_forward_loss_and_all_gradients_match_reference():
    from ms_gat_amd import engine, model
    g = load_golden("msgat72_n32.npz")
    net = model.msgat72(n_components=3, in_channels=3, in_timesteps=12, out_timesteps=12, use_te=True,
                        adj=torch.from_numpy(g["p.adj"]))
    net.load_state_dict(_state(g))
    net.to(_dev())
    X, H, D, Y = (torch.from_numpy(g[k]).to(_dev()) for k in ("X", "H", "D", "Y"))
    pred = net(X, H, D)
    loss = engine.HuberLoss(50.0)(pred, Y)
    loss.backward()
    assert_parity(pred.detach().cpu(), g["pred"], "msgat72_n32", "pred")
    assert abs(float(loss) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    checked = 0
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        assert_parity(p.grad.cpu(), g[f"g.{name}"], "msgat72_n32", name)
        checked += 1
    assert checked == sum(1 for k in g if k.startswith("g."))
    # the graph branch really ran in the HIP library: its parameters received gradients
    assert net.tpcs[0].tgacns[1].gacn.gatt.Wg.grad.abs().sum() > 0


@pytest.mark.parametrize("hip_graph", [False, True])
def test_five_optimizer_steps_track_the_reference_training_loop(tmp_path, hip_graph):
    """engine.Trainer / FlatAdam against the REFERENCE model stepped five times by the reference loop's own sequence
    (engine.py:56-63 with the optimizer of engine.py:106 and loss.py:51-52; tests/golden/make_golden.py::trajectory_case):
    every step's loss to 1e-4 (achieved: 1e-8 .. 8e-8) and every parameter after step 5 to 2e-4 of the tensor's largest
    entry.  A step moves an entry by ~lr = 1e-3, five of them by up to 5e-3, so an entry stepped the WRONG way even once
    shows as 2e-3 / 0.3 ~ 7e-3 of a 0.3-sized weight: 35 x the bar.  Below that the comparison measures Adam, not the
    kernels: it divides by sqrt(v) + 1e-8, which turns fp32 rounding noise in a near-zero gradient entry into a visible
    fraction of lr.  All tensors but three agree to < 4e-6; the worst entries (a residual 1x1 convolution, a head weight)
    to 1e-5 .. 5e-5 = 1 % of one step, and which entry it is moves with the last bit of the kernels' sums
    (profiles/r06/parity_rel_err.tsv)."""
    from ms_gat_amd import engine, model
    g = load_golden("msgat72_traj_n32.npz")
    net = model.msgat72(n_components=3, in_channels=3, in_timesteps=12, out_timesteps=12, use_te=True,
                        adj=torch.from_numpy(g["p.adj"]))
    net.load_state_dict(_state(g))
    net.to(_dev())
    tr = engine.Trainer(net, 50.0, str(tmp_path), hip_graph=hip_graph)
    X, Y = torch.from_numpy(g["X"]).float(), torch.from_numpy(g["Y"]).float()
    H, D = torch.from_numpy(g["H"]), torch.from_numpy(g["D"])
    what = "msgat72_traj_n32 " + ("replayed" if hip_graph else "eager")
    for k in range(X.shape[0]):
        loss = tr.run_epoch([[X[k], H[k], D[k], Y[k]]], gpu_id=0, epoch=k + 1, mode="train")
        want = float(g["losses"][k])
        record_err(what, f"loss[{k}]", abs(loss - want) / abs(want), 1e-4)
        assert abs(loss - want) < 1e-4 * abs(want), (k, loss, want)
    checked = 0
    for name, p in net.named_parameters():
        want = g[f"f.{name}"]
        e = rel_err(p.detach().cpu(), want)
        record_err(what, name, e, 2e-4)
        assert e < 2e-4, f"{name}: {e:.3e} after five steps"
        checked += 1
    assert checked == sum(1 for k in g if k.startswith("f."))


def test_trainer_runs_an_epoch_on_the_device(tmp_path):
    from ms_gat_amd import data, engine, model
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=40, n_edges=50, n_channels=1, in_hours=[1, 2], batch_size=8, days=2)
    net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj)
    net.to(_dev())
    tr = engine.Trainer(net, 50.0, str(tmp_path))
    first = [b for _, b in zip(range(6), ds.training)]
    l0 = tr.run_epoch(first, gpu_id=0, epoch=1, mode="train")
    l1 = tr.run_epoch(first, gpu_id=0, epoch=2, mode="train")
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0
    lv = tr.run_epoch(first[:2], gpu_id=0, epoch=2, mode="validate")
    assert np.isfinite(lv)


def test_hip_graph_training_matches_eager_training(tmp_path):
    """Trainer(hip_graph=True) captures forward + loss + backward + Adam in one HIP graph per batch shape
    and replays it; losses and parameters must track the eager run (same kernels, same order; only
    Adam's capturable arithmetic differs in the last bits)."""
    import copy
    from ms_gat_amd import data, engine, model
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=40, n_edges=50, n_channels=1, in_hours=[1, 2], batch_size=8, days=2)
    net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj)
    net.to(_dev())
    twin = copy.deepcopy(net)
    batches = [b for _, b in zip(range(5), ds.training)]
    batches.append([t[:3] for t in batches[0]])          # a second batch shape: its own graph
    eager = engine.Trainer(net, 50.0, str(tmp_path / "eager"), hip_graph=False)
    graphed = engine.Trainer(twin, 50.0, str(tmp_path / "graph"), hip_graph=True)
    for epoch in (1, 2, 3):
        le = eager.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        lg = graphed.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        assert abs(le - lg) < 1e-4 * abs(le), (epoch, le, lg)
    assert len(graphed._graphs) == 2
    # Adam's first steps move every weight by ~lr * sign(grad): entries whose gradient is rounding noise
    # may step the other way in another run, so parameters are compared to a few lr, losses tightly
    for (name, p), q in zip(net.named_parameters(), twin.parameters()):
        assert rel_err(q.detach().cpu(), p.detach().cpu()) < 2e-2, name
    ve = eager.run_epoch(batches[:2], gpu_id=0, epoch=3, mode="validate")
    vg = graphed.run_epoch(batches[:2], gpu_id=0, epoch=3, mode="validate")
    assert abs(ve - vg) < 1e-4 * abs(ve)
    # the learning-rate schedule reaches the captured optimizer through device memory
    for _ in range(30):
        graphed.scheduler.step()
    graphed._sync_lr()
    assert abs(float(graphed.optimizer._dev_lr) - 1e-4) < 1e-9
    assert isinstance(graphed.optimizer.param_groups[0]["lr"], float)
    assert set(graphed.optimizer._dev_steps.tolist()) == {18.0}
    assert set(graphed.optimizer._host_steps) == set(eager.optimizer._host_steps) == {18}


def test_auto_hip_graph_mode_captures_recurring_shapes_only(tmp_path):
    """The default `Trainer(hip_graph="auto")`: a batch shape is launched eagerly until it has recurred three times, then
    captured and replayed; the ragged last batch of an epoch (seen once per epoch) stays eager; validation gets its own
    graph (grad mode is part of the key); `load()` of a checkpoint keeps the graphs valid.  Losses track an eager twin."""
    import copy
    from ms_gat_amd import data, engine, model
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=40, n_edges=50, n_channels=1, in_hours=[1, 2], batch_size=8, days=2)
    net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj)
    net.to(_dev())
    twin = copy.deepcopy(net)
    batches = [b for _, b in zip(range(5), ds.training)]
    batches.append([t[:3] for t in batches[0]])          # the ragged last batch
    eager = engine.Trainer(net, 50.0, str(tmp_path / "eager"), hip_graph=False)
    auto = engine.Trainer(twin, 50.0, str(tmp_path / "auto"))
    assert auto.hip_graph == "auto" and auto.graph_after == 3
    for epoch in (1, 2, 3):
        le = eager.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        la = auto.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        assert abs(le - la) < 1e-4 * abs(le), (epoch, le, la)
        assert len(auto._graphs) == 1                     # the full batch from its 4th occurrence on; never the ragged one
    assert set(auto.optimizer._host_steps) == set(eager.optimizer._host_steps) == {18}
    for _ in range(4):
        ve = eager.run_epoch(batches[:2], gpu_id=0, epoch=3, mode="validate")
        va = auto.run_epoch(batches[:2], gpu_id=0, epoch=3, mode="validate")
        assert abs(ve - va) < 1e-4 * abs(ve)
    assert len(auto._graphs) == 2
    for (name, p), q in zip(net.named_parameters(), twin.parameters()):
        assert rel_err(q.detach().cpu(), p.detach().cpu()) < 2e-2, name
    auto.save(str(tmp_path / "ck.pkl"))
    auto.load(str(tmp_path / "ck.pkl"))                   # same buffers: the captured steps stay valid
    assert len(auto._graphs) == 2
    le = eager.run_epoch(batches, gpu_id=0, epoch=4, mode="train")
    la = auto.run_epoch(batches, gpu_id=0, epoch=4, mode="train")
    assert abs(le - la) < 2e-4 * abs(le)
    ev = engine.Evaluator(twin, 50.0, str(tmp_path / "ev"), str(tmp_path / "ck.pkl"))
    assert ev.hip_graph == "auto"
    first = ev.eval(batches[:1], gpu_id=0)
    for _ in range(4):
        assert abs(ev.eval(batches[:1], gpu_id=0) - first) < 1e-5 * abs(first)
    assert len(ev._graphs) == 1


@pytest.mark.parametrize("factory,cin,R,N", [("msgat48", 1, 2, 23), ("msgat96", 3, 1, 23), ("msgat72", 3, 2, 23),
                                             ("msgat48", 3, 2, 64), ("msgat96", 1, 2, 64), ("msgat72", 1, 2, 64)])
def test_whole_model_matches_the_dense_op_sequence(factory, cin, R, N):
    """Every width / dilation recipe of the reference's factories (msgat.py:220-229; msgat96 stacks four
    convolutions in its first TACN) through the library, against the same parameters evaluated with the
    reference's dense op sequence (oracle/dense_torch.py) in float64: prediction and every gradient.  N = 64 gives
    768 positions per channel slab: every width then runs its LDS-DMA one-pass backward forms (mfma.hip
    MSGAT_GLDS_FORMS; below 512 positions the register-staged kernels run)."""
    from ms_gat_amd import model
    from oracle import dense_torch
    torch.manual_seed(7)
    T, B = 12, 3
    gen = torch.Generator().manual_seed(11)
    import ms_gat_amd
    adj = ms_gat_amd.synthetic_adjacency(N, N + 7, seed=5)
    net = getattr(model, factory)(n_components=R, in_channels=cin, in_timesteps=T, out_timesteps=T, use_te=True,
                                  adj=adj).to(_dev())
    X = torch.randn(B, R, cin, N, T, generator=gen).to(_dev())
    H = torch.randint(0, 24, (B,), generator=gen).to(_dev())
    D = torch.randint(0, 7, (B,), generator=gen).to(_dev())
    dout = torch.randn(B, N, T, generator=gen).to(_dev())
    pred = net(X, H, D)
    params = [p for p in net.parameters() if p.requires_grad]
    grads = torch.autograd.grad(pred, params, dout)

    # float64 restatement with the same parameters
    P = {k: v.detach().double() for k, v in net.state_dict().items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if k != "adj"}
    gate = (leaves["te.h_ebd.weight"][H] + leaves["te.d_ebd.weight"][D]).view(B, R, N, T)
    out = 0
    for r in range(R):
        x = X[:, r].double()
        tpc = net.tpcs[r]
        for l, meam in enumerate(tpc.tgacns):
            sub = {k[len(f"tpcs.{r}.tgacns.{l}."):]: v for k, v in leaves.items() if k.startswith(f"tpcs.{r}.tgacns.{l}.")}
            x = dense_torch.meam_dense(x, P["adj"], sub, meam.dilations)
        x = torch.nn.functional.layer_norm(x, [T], leaves[f"tpcs.{r}.ln.weight"], leaves[f"tpcs.{r}.ln.bias"], 1e-5)
        y = torch.nn.functional.conv2d(x.transpose(1, 3), leaves[f"tpcs.{r}.fc.weight"], leaves[f"tpcs.{r}.fc.bias"])
        out = out + y[..., 0].transpose(1, 2) * gate[:, r]
    assert rel_err(pred.double(), out) < TOL
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    g64 = torch.autograd.grad(out, [leaves[n] for n in names], dout.double(), allow_unused=True)
    for n, a, b in zip(names, grads, g64):
        if b is None:
            continue
        assert rel_err(a.double(), b) < TOL, n


def test_cfg1_pemsd4_five_components_forward_matches_the_reference():
    """BASELINE.json configs[0] on the device: msgat72, 307 nodes, 3 features, B=4, the reference's default five
    components, against the prediction the reference computed on the CPU (tests/golden/make_golden.py: cfg1_case)."""
    from ms_gat_amd import model
    from test_oracle_golden import _cfg1_state
    g = load_golden("msgat72_cfg1_pemsd4.npz")
    adj = torch.zeros(307, 307)
    adj[torch.from_numpy(g["adj_rows"].astype(np.int64)), torch.from_numpy(g["adj_cols"].astype(np.int64))] = \
        torch.from_numpy(g["adj_vals"])
    net = model.msgat72(n_components=5, in_channels=3, in_timesteps=12, out_timesteps=12, use_te=True, adj=adj)
    state = _cfg1_state(g)
    state["adj"] = adj
    net.load_state_dict(state)
    net.to(_dev())
    with torch.no_grad():
        pred = net(torch.from_numpy(g["X"]).float().to(_dev()), torch.from_numpy(g["H"]).to(_dev()),
                   torch.from_numpy(g["D"]).to(_dev()))
    assert rel_err(pred.cpu(), g["pred"]) < TOL


@pytest.mark.parametrize("factory,cin,R,use_te,T", [("msgat72", 3, 3, True, 12), ("msgat96", 1, 2, False, 12),
                                                    ("msgat48", 3, 4, True, 12), ("msgat48", 1, 2, True, 16)])
def test_stacked_components_equal_the_component_loop(factory, cin, R, use_te, T):
    """stacked.forward (all R components in each launch) against the module-by-module loop of the same model:
    prediction and every parameter gradient.  T = 16: the rank-10 temporal attention needs 2*T*10 = 320 lanes, beyond
    the fused kernel's 256 -- both schedules fall back to batched torch ops for that one matrix (round-2 advisor
    finding: the stacked schedule raised instead)."""
    import ms_gat_amd
    from ms_gat_amd import model
    torch.manual_seed(3)
    N, B = 29, 4
    adj = ms_gat_amd.synthetic_adjacency(N, 40, seed=9)
    net = getattr(model, factory)(n_components=R, in_channels=cin, in_timesteps=T, out_timesteps=12, use_te=use_te,
                                  adj=adj).to(_dev())
    gen = torch.Generator().manual_seed(5)
    X = torch.randn(B, R, cin, N, T, generator=gen).to(_dev())
    H = torch.randint(0, 24, (B,), generator=gen).to(_dev())
    D = torch.randint(0, 7, (B,), generator=gen).to(_dev())
    dout = torch.randn(B, N, 12, generator=gen).to(_dev())
    params = [p for p in net.parameters() if p.requires_grad]
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    results = []
    for stack in (True, False):
        net.stack_components = stack
        pred = net(X, H, D)
        results.append((pred, torch.autograd.grad(pred, params, dout, allow_unused=True)))
    (p1, g1), (p2, g2) = results
    assert rel_err(p1, p2) < 1e-5
    for n, a, b in zip(names, g1, g2):
        assert (a is None) == (b is None), n
        if a is not None:
            assert rel_err(a, b) < 2e-5, n


def test_training_with_the_parameter_bank_tracks_the_component_loop(tmp_path):
    """The stacked schedule keeps every component's parameters as rows of shared storage (stacked.ParamBank); a
    few optimizer steps must move them exactly as they move when the components are evaluated one by one."""
    import copy
    from ms_gat_amd import data, engine, model
    torch.manual_seed(0)
    ds = data.SyntheticPEMS(n_nodes=30, n_edges=40, n_channels=1, in_hours=[1, 2, 3], batch_size=6, days=2)
    net = model.msgat48(n_components=3, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj)
    net.to(_dev())
    twin = copy.deepcopy(net)
    twin.stack_components = False
    batches = [b for _, b in zip(range(4), ds.training)]
    a = engine.Trainer(net, 50.0, str(tmp_path / "stacked"))
    b = engine.Trainer(twin, 50.0, str(tmp_path / "loop"))
    for epoch in (1, 2):
        la = a.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        lb = b.run_epoch(batches, gpu_id=0, epoch=epoch, mode="train")
        assert abs(la - lb) < 1e-4 * abs(lb), (epoch, la, lb)
    for (name, p), q in zip(net.named_parameters(), twin.parameters()):
        assert rel_err(p.detach().cpu(), q.detach().cpu()) < 2e-2, name
    # the bank survives a round trip through state_dict (values are copied in place, aliasing intact)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(state)
    X, H, D, _ = [t.to(_dev()) for t in batches[0]]
    with torch.no_grad():
        p1 = net(X, H, D)
        net.stack_components = False
        p2 = net(X, H, D)
    assert rel_err(p1, p2) < 1e-5


def _dense_restatement(net, X, H, D, Y, dtype, masks=None):
    """The whole model as the reference's dense op sequence (oracle/dense_torch.py) in `dtype`, with `net`'s parameters:
    prediction, Huber(50) loss and the gradient of every trainable parameter.  `masks[(r, l)]`: the active units of MEAM
    l of component r as another run chose them (applied instead of the ReLU's own decision)."""
    from ms_gat_amd import engine
    from oracle import dense_torch
    R, B = len(net.tpcs), X.shape[0]
    N, T = X.shape[-2], X.shape[-1]
    P = {k: v.detach().to(dtype) for k, v in net.state_dict().items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if k != "adj"}
    gate = (leaves["te.h_ebd.weight"][H] + leaves["te.d_ebd.weight"][D]).view(B, R, N, T)
    out = 0
    for r in range(R):
        x = X[:, r].to(dtype)
        for l, meam in enumerate(net.tpcs[r].tgacns):
            sub = {k[len(f"tpcs.{r}.tgacns.{l}."):]: v for k, v in leaves.items() if k.startswith(f"tpcs.{r}.tgacns.{l}.")}
            x = dense_torch.meam_dense(x, P["adj"], sub, meam.dilations,
                                       relu_mask=None if masks is None else masks[(r, l)])
        x = torch.nn.functional.layer_norm(x, [T], leaves[f"tpcs.{r}.ln.weight"], leaves[f"tpcs.{r}.ln.bias"], 1e-5)
        y = torch.nn.functional.conv2d(x.transpose(1, 3), leaves[f"tpcs.{r}.fc.weight"], leaves[f"tpcs.{r}.fc.bias"])
        out = out + y[..., 0].transpose(1, 2) * gate[:, r]
    loss = engine.huber_loss(out, Y.to(dtype), 50.0)
    names = [k for k, p in net.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return out.detach(), float(loss), dict(zip(names, grads))


def test_cfg4_per_rank_workload_full_training_step(tmp_path):
    """BASELINE.json configs[3], the per-GPU workload of the 1/2/4/8 scaling curve: msgat72 with the reference's default
    R = 5 components on the PEMSD7-like graph (N = 883), B = 32, one whole training step.  No 8-GPU node is needed to
    pin it:
      * the stacked launch sequence against the reference's loop over components (msgat.py:204) through the same library;
      * both against the reference's dense op sequence on the same GPU, in float64 AND in float32.  At this size the
        gradients of a ReLU network are not smooth functions of rounding: a handful of the 2.4e8 pre-activations sit
        within an ulp of zero, and any two fp32 implementations mask them differently.  The reference's own fp32
        arithmetic is up to 2e-2 (max-norm, per tensor) away from float64 here.  So the free-mask gradient figures (the
        library's and the reference fp32 op sequence's distance from float64) are RECORDED in the parity log, and the
        assertion is the forced-mask run: float64 with the library's own ReLU masks, every gradient tensor at the fixed
        1e-4 bar; the prediction and the loss -- smooth -- keep the 1e-4 bar in both runs;
      * one step through engine.Trainer with the 7.8 MB flat gradient buffer Adam and the all-reduce share."""
    import copy
    import bench
    from ms_gat_amd import engine
    dev = _dev()
    ts = bench.TrainStep(bench.CFG4, dev)
    net = ts.net
    assert len(net.tpcs) == 5 and ts.n_params == 1957960                      # 1 957 960 trainable parameters
    X, H, D, Y = ts.batch
    assert tuple(X.shape) == (32, 5, 1, 883, 12)

    def fwd_bwd(model):
        model.zero_grad(set_to_none=True)
        pred = model(X, H, D)
        loss = engine.HuberLoss(50.0)(pred, Y)
        loss.backward()
        return pred.detach(), float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    loop = copy.deepcopy(net)
    loop.stack_components = False                                              # msgat.py:204: one component at a time
    p_s, l_s, g_s = fwd_bwd(net)
    masks = {}                                                                 # which units each MEAM's ReLU left active
    hooks = [m.register_forward_hook(lambda mod, inp, out, key=(r, l): masks.__setitem__(key, out.detach() > 0))
             for r, tpc in enumerate(loop.tpcs) for l, m in enumerate(tpc.tgacns)]
    p_l, l_l, g_l = fwd_bwd(loop)
    for h in hooks:
        h.remove()
    assert len(masks) == 10
    what = "cfg4 R=5 N=883 B=32: stacked vs component loop"
    e = rel_err(p_s, p_l)
    record_err(what, "pred", e, 1e-5)
    assert e < 1e-5 and abs(l_s - l_l) < 1e-5 * abs(l_l) and set(g_s) == set(g_l)
    gscale = max(float(g.abs().max()) for g in g_l.values())                   # tensors > 100x smaller than the largest gradient
    errs = []                                                                  # (e.g. the lone alpha of a 1-channel attention:
    for k in g_l:                                                              # 3e-7) are judged on that scale
        scale = max(float(g_l[k].abs().max()), 1e-2 * gscale)
        errs.append(float((g_s[k] - g_l[k]).abs().max()) / scale)
    # measured: every tensor agrees to ~1e-6 except those downstream of ONE borderline ReLU -- e.g. a single one of the
    # 72 channels of tpcs.3.tgacns.1.res.bias is off by 3.5e-4, its neighbours by < 1.4e-6 (a mask flip, see docstring)
    record_err(what, "gradients, worst tensor", max(errs), 1e-3)
    record_err(what, "gradients, median tensor", float(np.median(errs)), 1e-5)
    assert max(errs) < 1e-3 and float(np.median(errs)) < 1e-5

    p64, l64, g64 = _dense_restatement(net, X, H, D, Y, torch.float64)
    p32, l32, g32 = _dense_restatement(net, X, H, D, Y, torch.float32)
    what = "cfg4 R=5 N=883 B=32 vs dense float64 ops"
    e_lib, e_ref = rel_err(p_s.double(), p64), rel_err(p32.double(), p64)
    record_err(what, "pred (library)", e_lib, TOL)
    record_err(what, "pred (dense fp32 eager ops)", e_ref, TOL)
    assert e_lib < TOL and abs(l_s - l64) < TOL * abs(l64)
    assert set(g64) == set(g_s)
    lib = {k: rel_err(g_s[k].double(), g64[k]) for k in g64}
    ref = {k: rel_err(g32[k].double(), g64[k]) for k in g64}
    # Free ReLU masks: RECORDED, not asserted (round-4 review: "no worse than the reference's fp32" had a margin of 1.0x and
    # could never fail usefully).  A handful of the 2.4e8 pre-activations sit within an ulp of zero and any two fp32
    # implementations mask them differently; one flipped unit moves most gradient tensors of ITS component by ~1e-4, so
    # the median over all tensors jumps when a third component gets a flip (3.6e-6, 8.1e-6 and 9.7e-5 for three builds of
    # rounds 4-5 whose forced-mask errors below were the same 1.2e-5).  The assertions on the gradients are the
    # forced-mask comparisons below, at the fixed 1e-4 bar, for the component loop AND the stacked schedule.
    record_err(what, "gradients, worst tensor, free masks (library; recorded only)", max(lib.values()), float("inf"))
    record_err(what, "gradients, worst tensor, free masks (dense fp32 eager ops; recorded only)", max(ref.values()), float("inf"))
    record_err(what, "gradients, median tensor, free masks (library; recorded only)", float(np.median(list(lib.values()))), float("inf"))
    record_err(what, "gradients, median tensor, free masks (dense fp32 eager ops; recorded only)", float(np.median(list(ref.values()))),
               float("inf"))

    # The mask-flip explanation as a test: the float64 op sequence again, but with every ReLU applying the set of active
    # units the LIBRARY chose (the component-loop run above).  With the borderline pre-activations out of the comparison
    # every gradient tensor must meet the 1e-4 bar of the hot path (tensors > 100x below the largest gradient are judged
    # on that scale, as above).
    del p32, g32
    p64m, l64m, g64m = _dense_restatement(loop, X, H, D, Y, torch.float64, masks)
    what = "cfg4 R=5 N=883 B=32 vs dense float64 ops with the library's ReLU masks"
    gs64 = max(float(g.abs().max()) for g in g64m.values())
    fixed = {k: float((g_l[k].double() - g64m[k]).abs().max()) / max(float(g64m[k].abs().max()), 1e-2 * gs64) for k in g64m}
    worst = max(fixed, key=fixed.get)
    record_err(what, "pred", rel_err(p_l.double(), p64m), TOL)
    record_err(what, f"gradients, worst tensor ({worst})", fixed[worst], TOL)
    record_err(what, "gradients, median tensor", float(np.median(list(fixed.values()))), TOL)
    assert rel_err(p_l.double(), p64m) < TOL and abs(l_l - l64m) < TOL * abs(l64m)
    assert fixed[worst] < TOL, (worst, fixed[worst])
    # the stacked schedule against the same float64 run: it shares all but a few borderline units with the loop (the
    # worst tensor above), so its median tensor must meet the bar too -- and, unlike the free-mask median, stays put
    fixed_s = {k: float((g_s[k].double() - g64m[k]).abs().max()) / max(float(g64m[k].abs().max()), 1e-2 * gs64) for k in g64m}
    record_err(what, "gradients, median tensor (stacked schedule)", float(np.median(list(fixed_s.values()))), TOL)
    record_err(what, "gradients, worst tensor (stacked schedule; recorded only)", max(fixed_s.values()), float("inf"))
    assert float(np.median(list(fixed_s.values()))) < TOL

    before = [p.detach().clone() for p in net.parameters() if p.requires_grad]
    loss = ts.run(1)
    assert np.isfinite(loss) and ts.allreduce_bytes == 4 * (1957960 + 1)
    assert isinstance(ts.trainer.optimizer, engine.FlatAdam) and set(ts.trainer.optimizer._host_steps) == {1}
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, (p for p in net.parameters() if p.requires_grad)))
    assert moved == len(before)
