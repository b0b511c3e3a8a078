"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the oracle.

Bar (BASELINE.json north_star): 1e-4 relative in fp32, measured as max|a-b| / max|b|
per tensor (`rel_err` in conftest.py).  Every test here needs a real MI355X.
"""
import numpy as np
import pytest
import torch

from conftest import elementwise_violations, load_golden, load_headline_golden, record_err, rel_err
from oracle import dense_torch, gat_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def run_ours(x, adj, Wg, alpha, W, dz, sell=None):
    """One fwd+bwd through ms_gat_amd.ops.gacn with R stacked relations.
    x [G,C,N,T]; Wg [R,T,T]; alpha [R,C]; W [R,Co,C] or None; dz like the output.
    sell: None = the library's own choice of edge layout; "always" / "never" force it (SparseGraph)."""
    import ms_gat_amd
    xt = _cuda(x).requires_grad_(True)
    Wgt = _cuda(Wg).requires_grad_(True)
    at = _cuda(alpha).requires_grad_(True)
    Wt = None if W is None else _cuda(W).requires_grad_(True)
    graph = _cuda(adj) if sell is None else ms_gat_amd.SparseGraph(torch.from_numpy(np.ascontiguousarray(adj)), sell=sell)
    z = ms_gat_amd.gacn(xt, at, Wgt, Wt, graph)
    z.backward(_cuda(dz))
    torch.cuda.synchronize()
    out = dict(z=z.detach().cpu().numpy(), dx=xt.grad.cpu().numpy(), dWg=Wgt.grad.cpu().numpy(),
               dalpha=at.grad.cpu().numpy())
    if Wt is not None:
        out["dW"] = Wt.grad.cpu().numpy()
    return out


def oracle_f64(x, adj, Wg, alpha, W, dz):
    """numpy float64 oracle, relation by relation."""
    R = Wg.shape[0]
    Bg = x.shape[0] // R
    f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    zs, dxs, dWgs, das, dWs = [], [], [], [], []
    for r in range(R):
        sl = slice(r * Bg, (r + 1) * Bg)
        if W is None:
            zs.append(gat_oracle.gatt_forward(f(x[sl]), f(adj), f(Wg[r]), f(alpha[r])))
            dx, dWg, da = gat_oracle.gatt_backward(f(x[sl]), f(adj), f(Wg[r]), f(alpha[r]), f(dz[sl]))
        else:
            zs.append(gat_oracle.gacn_forward(f(x[sl]), f(adj), f(Wg[r]), f(alpha[r]), f(W[r])))
            dx, dWg, da, dW = gat_oracle.gacn_backward(f(x[sl]), f(adj), f(Wg[r]), f(alpha[r]), f(W[r]), f(dz[sl]))
            dWs.append(dW)
        dxs.append(dx), dWgs.append(dWg), das.append(da)
    out = dict(z=np.concatenate(zs), dx=np.concatenate(dxs), dWg=np.stack(dWgs), dalpha=np.stack(das))
    if W is not None:
        out["dW"] = np.stack(dWs)
    return out


def assert_close(got, want, tol=TOL, what="", floor=0.0):
    """Max-norm bar `tol` AND, per entry, |a-b| <= 1e-4 |b| + 1e-5 max|b| (an entry far below the tensor's maximum
    may not be wrong by more than ~its own 1e-4 plus a tenth of the bar); the violating fraction at the tighter
    floor 1e-6 max|b| goes on record.  `floor` is an absolute scale for tensors whose exact value is 0 (inputs are
    O(1)); they have no per-entry scale and are held to the max-norm bar only."""
    for k in want:
        e = rel_err(got[k], want[k])
        viol = None
        if floor > 0.0:
            e = min(e, float(np.abs(np.asarray(got[k], dtype=np.float64) - want[k]).max()) / floor)
        elif tol <= TOL:
            viol = (elementwise_violations(got[k], want[k], 1e-4, 1e-6), elementwise_violations(got[k], want[k], 1e-4, 1e-5))
        record_err(what, k, e, tol, viol)
        assert e < tol, f"{what} {k}: rel err {e:.3e} >= {tol}"
        assert viol is None or viol[1] == 0.0, f"{what} {k}: {viol[1]:.2e} of the entries outside 1e-4|b| + 1e-5 max|b|"


def random_problem(R, Bg, C, Co, N, T, n_edges, seed, x_scale=1.0):
    import ms_gat_amd
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((R * Bg, C, N, T)) * x_scale).astype(np.float32)
    adj = ms_gat_amd.synthetic_adjacency(N, n_edges, seed + 1).numpy()
    Wg = (rng.standard_normal((R, T, T)) * (1.0 / T) ** 0.5).astype(np.float32)
    alpha = rng.uniform(-C ** -0.5, C ** -0.5, size=(R, C)).astype(np.float32)
    W = None if Co == 0 else (rng.standard_normal((R, Co, C)) * (2.0 / (Co + C)) ** 0.5).astype(np.float32)
    dz = rng.standard_normal((R * Bg, Co if Co else C, N, T)).astype(np.float32)
    return x, adj, Wg, alpha, W, dz


# ---------------------------------------------------------------------------------------
# golden vectors produced by the reference (tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------
def test_graph_attention_matches_reference_golden(gatt_case):
    tag, g, _ = gatt_case
    got = run_ours(g["x"], g["adj"], g["Wg"][None], g["alpha"][None], None, g["dy"])
    want = dict(z=g["y"], dx=g["dx"], dWg=g["dWg"][None], dalpha=g["dalpha"][None])
    assert_close(got, want, what=f"GraphAttention[{tag}]")


def test_gacn_matches_reference_golden(gatt_case):
    tag, g, c = gatt_case
    got = run_ours(g["x"], g["adj"], g["Wg"][None], g["alpha"][None], c["W"][None], c["dz"])
    want = dict(z=c["z"], dx=c["dx"], dWg=c["dWg"][None], dalpha=c["dalpha"][None], dW=c["dW"][None])
    assert_close(got, want, what=f"GACN[{tag}]")


def test_gacn_matches_the_reference_at_the_headline_size():
    """The graph size bench.py reports on (PEMSD7-like N = 883, C = 72 -> 24: PROJ_FIRST, slab-in-LDS aggregate, fused
    du / SDDMM pass, dense passes with 7 row blocks) against the REFERENCE's own forward and autograd, not the oracle."""
    g = load_headline_golden()
    got = run_ours(g["x"], g["adj"], g["Wg"][None], g["alpha"][None], g["W"][None], g["dz"])
    want = dict(z=g["z"], dx=g["dx"], dWg=g["dWg"][None], dalpha=g["dalpha"][None], dW=g["dW"][None])
    assert_close(got, want, what="GACN[reference golden, N=883 C=72->24]")


@pytest.mark.parametrize("tag,form", [("w48_n64", "<2,3,128,3,1>"), ("w96_n64", "<3,6,64,3,2>")])
def test_gacn_matches_the_reference_at_the_other_registry_widths(tag, form):
    """GACN(48 -> 16) / GACN(96 -> 32) at N = 64 (768 positions per slab, so the LDS-DMA one-pass backward forms of
    msgat48 / msgat96 are the ones selected) against the REFERENCE's own forward and autograd (msgat.py:220-229)."""
    g = load_golden(f"gacn_{tag}.npz")
    x, dz = g["x"].astype(np.float32), g["dz"].astype(np.float32)
    got = run_ours(x, g["adj"], g["Wg"][None], g["alpha"][None], g["W"][None], dz)
    want = dict(z=g["z"], dx=g["dx"], dWg=g["dWg"][None], dalpha=g["dalpha"][None], dW=g["dW"][None])
    assert_close(got, want, what=f"GACN[reference golden {tag}, one-pass backward form {form}]")


def test_modules_are_drop_in_for_the_reference_golden():
    """nn.Module boundary: same ctor args, parameter names and forward signature."""
    import ms_gat_amd
    g, c = load_golden("gatt_b2c3n16.npz"), load_golden("gacn_b2c3n16.npz")
    m = ms_gat_amd.GACN(3, 24, 12).to(_dev())
    missing = m.load_state_dict({"gatt.Wg": _cuda(g["Wg"]), "gatt.alpha": _cuda(g["alpha"]), "W": _cuda(c["W"])})
    assert not missing.missing_keys and not missing.unexpected_keys
    x = _cuda(g["x"]).requires_grad_(True)
    z = m(x, _cuda(g["adj"]))
    z.backward(_cuda(c["dz"]))
    assert rel_err(z.detach().cpu(), c["z"]) < TOL
    assert rel_err(x.grad.cpu(), c["dx"]) < TOL
    assert rel_err(m.W.grad.cpu(), c["dW"]) < TOL
    assert rel_err(m.gatt.Wg.grad.cpu(), c["dWg"]) < TOL
    assert rel_err(m.gatt.alpha.grad.cpu(), c["dalpha"]) < TOL

    a = ms_gat_amd.GraphAttention(3, 12).to(_dev())
    a.load_state_dict({"Wg": _cuda(g["Wg"]), "alpha": _cuda(g["alpha"])})
    with torch.no_grad():
        y = a(_cuda(g["x"]), _cuda(g["adj"]))
    assert rel_err(y.cpu(), g["y"]) < TOL


def test_one_relation_without_the_leading_axis_equals_the_stacked_call_bit_for_bit():
    """ops.gacn takes the reference's own parameter shapes (alpha [C], Wg [T,T], W [Co,C]: attention.py:29-30,
    msgat.py:23) for one relation -- what the module classes pass, so that no view nodes sit between parameter and op --
    and computes exactly what the [1, ...] form computes; the gradients come back in the parameters' shapes."""
    from ms_gat_amd import ops
    import ms_gat_amd
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    adj = ms_gat_amd.synthetic_adjacency(41, 50, 2).to(dev)
    for C, Co in ((3, 24), (72, 24), (5, 0)):
        x = torch.randn(2, C, 41, 12, generator=g).to(dev)
        alpha, Wg = (torch.randn(C, generator=g) * 0.3).to(dev), (torch.randn(12, 12, generator=g) * 0.3).to(dev)
        W = (torch.randn(Co, C, generator=g) * 0.2).to(dev) if Co else None
        dz = torch.randn(2, Co or C, 41, 12, generator=g).to(dev)
        outs = []
        for lead in (False, True):
            ps = [t.clone().requires_grad_(True) for t in (alpha, Wg) + ((W,) if Co else ())]
            xs = x.clone().requires_grad_(True)
            args = [p.unsqueeze(0) if lead else p for p in ps]
            z = ops.gacn(xs, args[0], args[1], args[2] if Co else None, adj)
            z.backward(dz)
            assert all(p.grad.shape == p.shape for p in ps)
            outs.append([z.detach(), xs.grad] + [p.grad for p in ps])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    with pytest.raises(ValueError, match="Wg must be"):
        ops.gacn(x, alpha, Wg.unsqueeze(0), None, adj)


# ---------------------------------------------------------------------------------------
# oracle comparisons on seeded inputs: modes, stacked relations, odd sizes
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,Bg,C,Co,N,T,E", [
    (1, 2, 3, 0, 50, 12, 60),      # PLAIN
    (3, 2, 1, 24, 70, 12, 80),     # AGG_FIRST, stacked relations, C=1 (PEMSD7 first MEAM)
    (3, 2, 72, 24, 65, 12, 80),    # PROJ_FIRST, stacked (second MEAM of msgat72)
    (2, 1, 48, 16, 33, 12, 40),    # msgat48 widths
    (1, 2, 96, 32, 40, 12, 50),    # msgat96 widths
    (1, 1, 5, 3, 17, 12, 20),      # odd channel counts, PROJ_FIRST with Co < 4
    (1, 1, 7, 40, 19, 12, 25),     # AGG_FIRST with Co that needs the generic 8-wide tile
    (1, 2, 6, 6, 64, 4, 70),       # T = 4, N a multiple of the row tile
    (1, 1, 9, 4, 130, 8, 140),     # T = 8
    (2, 1, 4, 8, 257, 16, 300),    # T = 16, N just over a tile boundary
    (2, 2, 1, 24, 300, 4, 350),    # one input channel: q = alpha x is computed inside the score kernel (QRows<T,1>), T = 4
    (1, 2, 1, 0, 883, 8, 866),     # ... PLAIN, T = 8, and a size that takes the helper-wave form k_scores7<8,*,1>
    (1, 2, 1, 24, 257, 16, 300),   # ... T = 16
    (1, 2, 3, 24, 150, 12, 4000),  # AGG_FIRST tail with more than 1024 edges per row block: the uncached (global) operand path
    (2, 1, 1, 8, 150, 12, 4000),   # ... one channel
    (1, 1, 2, 0, 1, 12, 0),        # a single node: only the self loop
    (1, 1, 200, 180, 20, 12, 25),  # matrix too large for the MFMA kernel's LDS: VALU projection, blocked dW
    (1, 2, 33, 40, 21, 12, 25),    # AGG_FIRST with > 32 channels on both sides
])
def test_against_numpy_oracle(R, Bg, C, Co, N, T, E):
    prob = random_problem(R, Bg, C, Co, N, T, E, seed=R * 1000 + C * 10 + N)
    # a single node has softmax == 1 exactly, so dWg is exactly 0: compare on the O(1) input scale
    assert_close(run_ours(*prob), oracle_f64(*prob), what=f"R{R} Bg{Bg} C{C} Co{Co} N{N} T{T}",
                 floor=1.0 if N == 1 else 0.0)


def _ref32_cpu(x, adj, Wg, alpha, W, dz):
    """The reference's own fp32 op sequence (R = 1) on the CPU, as a yardstick."""
    t = lambda a: torch.from_numpy(a).requires_grad_(True)  # noqa: E731
    xt, Wgt, at, Wt = t(x), t(Wg[0]), t(alpha[0]), t(W[0])
    z = dense_torch.gacn_dense(xt, torch.from_numpy(adj), Wgt, at, Wt)
    z.backward(torch.from_numpy(dz))
    return dict(z=z.detach().numpy(), dx=xt.grad.numpy(), dWg=Wgt.grad.numpy()[None],
                dalpha=at.grad.numpy()[None], dW=Wt.grad.numpy()[None])


def test_large_scores_need_the_running_max():
    """Scores up to ~1e2: exp() overflows fp32 without max subtraction, rows are partly
    saturated.  Yardstick: the reference's own fp32 op sequence -- we may not be worse than
    3x its distance from the float64 oracle (or the usual 1e-4, whichever is larger)."""
    prob = random_problem(1, 2, 3, 24, 90, 12, 100, seed=5, x_scale=6.0)
    got, want, ref32 = run_ours(*prob), oracle_f64(*prob), _ref32_cpu(*prob)
    assert all(np.isfinite(v).all() for v in got.values())
    for k in want:
        bar = max(TOL, 3.0 * rel_err(ref32[k], want[k]))
        assert rel_err(got[k], want[k]) < bar, (k, rel_err(got[k], want[k]), bar)


def test_fully_saturated_softmax_stays_finite_and_accurate():
    """Scores ~1e3: every softmax row is one-hot in fp32 and the true dWg / dalpha / most of dq
    are ~1e-12 (differences of O(1e3) terms that cancel).  Outputs and the O(1) gradients
    (z, dx, dW) must meet the usual bar; the cancelling ones are held to 1e-5 of the
    magnitude of the terms that cancel (the float64 oracle supplies that magnitude)."""
    prob = random_problem(1, 2, 3, 24, 90, 12, 100, seed=5, x_scale=25.0)
    x, adj, Wg, alpha, W, dz = prob
    got, want = run_ours(*prob), oracle_f64(*prob)
    assert all(np.isfinite(v).all() for v in got.values())
    for k in ("z", "dx", "dW"):
        assert rel_err(got[k], want[k]) < TOL, (k, rel_err(got[k], want[k]))
    f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    _, c = gat_oracle.gatt_forward(f(x), f(adj), f(Wg[0]), f(alpha[0]), return_cache=True)
    dy = np.einsum("oc,bont->bcnt", f(W[0]), f(dz))
    g = np.abs(c["P"] * f(adj) * np.einsum("bcnt,bcmt->bnm", dy, f(x)))
    dS_mag = g + g.sum(-1, keepdims=True) * c["P"]            # |g| + |delta| P: the two sides that cancel
    dkW_mag = dS_mag @ np.abs(c["q"])
    dq_mag = dS_mag.transpose(0, 2, 1) @ np.abs(c["kW"]) + dkW_mag @ np.abs(f(Wg[0])).T
    scale = dict(dWg=np.einsum("bnt,bns->ts", np.abs(c["q"]), dkW_mag).max(),
                 dalpha=np.einsum("bnt,bcnt->c", dq_mag, np.abs(f(x))).max())
    for k in ("dWg", "dalpha"):
        err = np.abs(f(got[k]) - want[k]).max()
        assert err < 1e-5 * scale[k], (k, err, scale[k])


def test_arbitrary_adjacency_with_empty_rows_and_asymmetry():
    """The boundary takes any [N,N] tensor, not just sym-normalised graphs."""
    rng = np.random.default_rng(11)
    N = 45
    adj = (rng.random((N, N)) < 0.08).astype(np.float32) * rng.standard_normal((N, N)).astype(np.float32)
    adj[3, :] = 0.0   # node with no in-row edges
    adj[:, 7] = 0.0   # node nobody aggregates from
    x, _, Wg, alpha, W, dz = random_problem(1, 2, 4, 24, N, 12, 10, seed=12)
    assert_close(run_ours(x, adj, Wg, alpha, W, dz), oracle_f64(x, adj, Wg, alpha, W, dz), what="arbitrary adj")
    x, _, Wg, alpha, W, dz = random_problem(1, 2, 30, 8, N, 12, 10, seed=13)
    assert_close(run_ours(x, adj, Wg, alpha, W, dz), oracle_f64(x, adj, Wg, alpha, W, dz), what="arbitrary adj P")


def test_all_zero_adjacency_gives_zero_output_and_zero_grads():
    import ms_gat_amd
    x, _, Wg, alpha, W, dz = random_problem(1, 1, 3, 24, 20, 12, 5, seed=14)
    adj = np.zeros((20, 20), dtype=np.float32)
    got = run_ours(x, adj, Wg, alpha, W, dz)
    for k, v in got.items():
        assert np.all(v == 0), k
    assert ms_gat_amd.SparseGraph(torch.from_numpy(adj)).nnz == 0


def test_dense_adjacency_rows_sum_to_one():
    """With adj == 1 everywhere the masked softmax is the full softmax: aggregating a
    feature that is constant over nodes returns it unchanged (row sums are exactly 1)."""
    import ms_gat_amd
    N, C, T = 40, 3, 12
    rng = np.random.default_rng(3)
    base = rng.standard_normal((1, C, 1, T)).astype(np.float32)
    x = np.ascontiguousarray(np.repeat(base, N, axis=2))
    adj = np.ones((N, N), dtype=np.float32)
    Wg = rng.standard_normal((1, T, T)).astype(np.float32)
    alpha = rng.standard_normal((1, C)).astype(np.float32)
    with torch.no_grad():
        y = ms_gat_amd.graph_attention(_cuda(x), _cuda(alpha), _cuda(Wg), _cuda(adj)).cpu().numpy()
    assert rel_err(y, x) < 1e-5


def test_stage_outputs_lse_and_edge_coefficients():
    """Checks the saved intermediates the C ABI documents: q, kW, lse (log2 units), pq, E."""
    import ctypes as C
    import ms_gat_amd
    from ms_gat_amd import _lib
    prob = random_problem(2, 2, 3, 0, 77, 12, 90, seed=21)
    x, adj, Wg, alpha = prob[:4]
    G, Cc, N, T = x.shape
    g = ms_gat_amd.SparseGraph(torch.from_numpy(adj))
    gs, _keep = g.on(_dev())
    shape = _lib.Shape(2, 2, Cc, 0, N, T)
    xt, at, Wgt = _cuda(x), _cuda(alpha), _cuda(Wg)
    q = torch.empty(G, N, T, device=_dev())
    kW, pq = torch.empty_like(q), torch.empty_like(q)
    lse = torch.empty(G, N, device=_dev())
    E = torch.empty(G, g.nnz, device=_dev())
    Ec = torch.empty(G, g.nnz, device=_dev())
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.msgat_stage_project(C.byref(shape), xt.data_ptr(), at.data_ptr(), None, q.data_ptr(), None, s), "project")
    ndense = int(L.msgat_dense_scratch_bytes(C.byref(shape)))
    dense = torch.empty(max(ndense, 1), device=_dev(), dtype=torch.uint8)
    _lib.check(L.msgat_stage_scores(C.byref(shape), C.byref(gs), q.data_ptr(), Wgt.data_ptr(), kW.data_ptr(),
                                    lse.data_ptr(), pq.data_ptr(), E.data_ptr(), Ec.data_ptr(),
                                    dense.data_ptr() if ndense else None, s), "scores")
    torch.cuda.synchronize()
    assert torch.equal(Ec.cpu(), E.cpu()[:, g.cperm[: g.nnz].long()])      # the same coefficients in CSC order, bit for bit
    rows, cols = g.erow[: g.nnz].long().numpy(), g.col[: g.nnz].long().numpy()
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        x64 = x[sl].astype(np.float64)
        _, cache = gat_oracle.gatt_forward(x64, adj.astype(np.float64), Wg[r].astype(np.float64),
                                           alpha[r].astype(np.float64), return_cache=True)
        assert rel_err(q[sl].cpu(), cache["q"]) < 1e-5
        assert rel_err(kW[sl].cpu(), cache["kW"]) < 1e-5
        want_lse = gat_oracle.gatt_lse(x64, Wg[r].astype(np.float64), alpha[r].astype(np.float64))
        assert np.abs(lse[sl].cpu().numpy() * np.log(2.0) - want_lse).max() < 1e-4  # stored in log2 units
        assert rel_err(pq[sl].cpu(), cache["P"] @ cache["q"]) < TOL
        assert rel_err(E[sl].cpu(), cache["E"][:, rows, cols]) < TOL


# ---------------------------------------------------------------------------------------
# BASELINE.json sizes: dense oracle on the same GPU + size-independent properties
# ---------------------------------------------------------------------------------------
def _dense_oracle_gpu(x, adj, Wg, alpha, W, dz):
    """oracle/dense_torch.py (the reference's op sequence) run with autograd on the GPU."""
    R = Wg.shape[0]
    Bg = x.shape[0] // R
    outs = dict(z=[], dx=[], dWg=[], dalpha=[], dW=[])
    adj_t = _cuda(adj)
    for r in range(R):
        xt = _cuda(x[r * Bg:(r + 1) * Bg]).requires_grad_(True)
        Wgt, at, Wt = (_cuda(v[r]).requires_grad_(True) for v in (Wg, alpha, W))
        z = dense_torch.gacn_dense(xt, adj_t, Wgt, at, Wt)
        z.backward(_cuda(dz[r * Bg:(r + 1) * Bg]))
        for k, v in (("z", z.detach()), ("dx", xt.grad), ("dWg", Wgt.grad), ("dalpha", at.grad), ("dW", Wt.grad)):
            outs[k].append(v.cpu().numpy())
    return dict(z=np.concatenate(outs["z"]), dx=np.concatenate(outs["dx"]), dWg=np.stack(outs["dWg"]),
                dalpha=np.stack(outs["dalpha"]), dW=np.stack(outs["dW"]))


@pytest.mark.parametrize("C,Co,Bg", [(48, 16, 8), (96, 32, 8)])
def test_other_registry_widths_at_the_headline_graph_size(C, Co, Bg):
    """msgat48 and msgat96 (the other two models of the reference's registry, main.py:17, msgat.py:220-229) at N = 883:
    their second-depth GACN takes the one-pass LDS-DMA backward forms <2,3,128,3,1> resp. <3,6,64,3,2> (mfma.hip)."""
    prob = random_problem(3, Bg, C, Co, 883, 12, 866, seed=170 + C)
    got = run_ours(*prob)
    want = _dense_oracle_gpu(*prob)
    assert_close(got, want, what=f"PEMSD7 widths {C}->{Co}")


@pytest.mark.parametrize("C,Bg", [(1, 32), (72, 32)])
def test_pemsd7_full_size_against_dense_eager(C, Bg):
    """configs[2]: PEMSD7-like N=883, B=32, R=3, C in {1, 72} -> Co=24."""
    prob = random_problem(3, Bg, C, 24, 883, 12, 866, seed=70 + C)
    got = run_ours(*prob)
    want = _dense_oracle_gpu(*prob)
    assert_close(got, want, what=f"PEMSD7 C={C}")


def test_pemsd4_b64_full_size_against_dense_eager():
    """configs[1]: PEMSD4-like N=307, B=64, single relation."""
    for C in (3, 72):
        prob = random_problem(1, 64, C, 24, 307, 12, 340, seed=40 + C)
        assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what=f"PEMSD4 C={C}")


def test_full_size_linearity_in_W_and_in_cotangent():
    """Size-independent properties at N=883, B=32, C=72: z is linear in W; every gradient
    is linear in dz."""
    import ms_gat_amd
    x, adj, Wg, alpha, W, dz = random_problem(1, 32, 72, 24, 883, 12, 866, seed=99)
    rng = np.random.default_rng(100)
    W2 = rng.standard_normal(W.shape).astype(np.float32) * 0.1
    g = ms_gat_amd.SparseGraph(torch.from_numpy(adj))
    xt, at, Wgt = _cuda(x), _cuda(alpha), _cuda(Wg)
    with torch.no_grad():
        z1 = ms_gat_amd.gacn(xt, at, Wgt, _cuda(W), g)
        z2 = ms_gat_amd.gacn(xt, at, Wgt, _cuda(W2), g)
        z12 = ms_gat_amd.gacn(xt, at, Wgt, _cuda(W + W2), g)
    assert rel_err((z1 + z2).cpu(), z12.cpu()) < 1e-5
    a = run_ours(x, adj, Wg, alpha, W, dz)
    b = run_ours(x, adj, Wg, alpha, W, 2.0 * dz)
    for k in ("dx", "dWg", "dalpha", "dW"):
        assert rel_err(2.0 * a[k], b[k]) < 1e-5, k


def test_stress_graph_uses_the_large_n_path():
    """N large enough that an [N,T] slab does not fit LDS: one 4-timestep column of the slab per
    pass (N*16 B <= LDS, the BASELINE stress graph N = 8192 at degree 16), gather-from-L2 beyond."""
    prob = random_problem(1, 1, 12, 4, 8192, 12, 65536, seed=65)
    assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what="N=8192 deg16 PROJ_FIRST")
    prob = random_problem(1, 1, 6, 4, 10400, 12, 20000, seed=66)
    assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what="N=10400 PROJ_FIRST (gather from L2)")
    prob = random_problem(1, 1, 3, 24, 4000, 12, 8000, seed=61)
    got = run_ours(*prob)
    want = _dense_oracle_gpu(*prob)
    assert_close(got, want, what="N=4000 AGG_FIRST")
    prob = random_problem(1, 1, 40, 8, 3500, 12, 7000, seed=62)
    assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what="N=3500 PROJ_FIRST")


@pytest.mark.parametrize("R,Bg,C,Co,N,E,seed", [
    (1, 2, 5, 0, 50, 60, 70),        # PLAIN, one partial slice
    (2, 2, 3, 24, 130, 400, 71),     # AGG_FIRST: backward aggregate with the alpha (x) dq epilogue, SDDMM on C channels
    (1, 3, 40, 8, 200, 2500, 72),    # PROJ_FIRST, degrees ~26: several trips of 4 columns per slice
    (3, 1, 72, 24, 64, 600, 73),     # exactly one slice, dense-ish rows, R relations
    (1, 1, 12, 4, 333, 300, 74),     # sparse: most rows have 1-2 edges
])
def test_sell_edge_layout_agrees_with_the_csr_kernels_and_the_oracle(R, Bg, C, Co, N, E, seed):
    """The large-graph kernels (k_agg_sell, k_sddmm_sell: one column of a slab in LDS, edges in the degree-sorted
    sliced-ELLPACK layout) forced onto small graphs, against the fp64 oracle and against the CSR kernels."""
    prob = random_problem(R, Bg, C, Co, N, 12, E, seed)
    got = run_ours(*prob, sell="always")
    assert_close(got, oracle_f64(*prob), what=f"SELL N={N} C={C}->{Co}")
    assert_close(got, run_ours(*prob, sell="never"), tol=5e-6, what="SELL vs CSR kernels")


def test_sell_with_other_timestep_counts():
    for T, seed in ((4, 80), (8, 81), (16, 82)):
        prob = random_problem(1, 2, 6, 3, 150, T, 500, seed)
        assert_close(run_ours(*prob, sell="always"), oracle_f64(*prob), what=f"SELL T={T}")


def _stress_setup(R, Bg, seed):
    """BASELINE.json configs[4]: N = 8192 sensors, average degree 16 (+ self loops), T = 12, msgat72's second-MEAM
    widths C = 72 -> 24.  Everything stays on the device (x alone is 7.2 GB at R = 4, B = 64)."""
    import ms_gat_amd
    dev = _dev()
    N, T, C, Co = 8192, 12, 72, 24
    graph = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(N, 65536, 0))
    assert graph.has_sell and graph.nnz == N + 2 * 65536
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.nn.functional.layer_norm(torch.randn(R * Bg, C, N, T, device=dev, generator=g), (T,))
    Wg = torch.randn(R, T, T, device=dev, generator=g) * (1.0 / T) ** 0.5
    alpha = (torch.rand(R, C, device=dev, generator=g) * 2 - 1) * C ** -0.5
    W = torch.randn(R, Co, C, device=dev, generator=g) * (2.0 / (Co + C)) ** 0.5
    return graph, x, Wg, alpha, W, g


def _terr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_stress_config_full_size_properties():
    """configs[4] at its full size (R = 4, B = 64: G = 256 groups) through size-independent properties:
    determinism, z linear in W, every gradient linear in the cotangent, and the closed form of Wg = 0 (uniform
    softmax: E = adj / N, so z = W (adj x) / N) against a sparse fp64 product."""
    import ms_gat_amd
    R, Bg, N = 4, 64, 8192
    graph, x, Wg, alpha, W, gen = _stress_setup(R, Bg, 123)
    dev = x.device
    with torch.no_grad():
        z1 = ms_gat_amd.gacn(x, alpha, Wg, W, graph)
        assert torch.equal(z1, ms_gat_amd.gacn(x, alpha, Wg, W, graph)), "no atomics anywhere: bitwise reproducible"
        W2 = torch.randn(W.shape, device=dev, generator=gen) * 0.1
        z2 = ms_gat_amd.gacn(x, alpha, Wg, W2, graph)
        z12 = ms_gat_amd.gacn(x, alpha, Wg, W + W2, graph)
        e = _terr(z1 + z2, z12)
        record_err("stress full size: z linear in W", "z", e, 1e-5)
        assert e < 1e-5
        del z2, z12

        # Wg = 0: scores are 0, softmax is uniform, lse = log2 N exactly
        z0 = ms_gat_amd.gacn(x, alpha, torch.zeros_like(Wg), W, graph)
        A = torch.sparse_coo_tensor(torch.stack([graph.erow.long(), graph.col.long()]), graph.val.double(), (N, N)).to(dev)
        for grp in (0, 77, 130, 255):
            r = grp // Bg
            xg = x[grp].double().permute(1, 0, 2).reshape(N, -1)                   # [N, C*T]
            y = (torch.sparse.mm(A, xg) / N).reshape(N, x.shape[1], -1)             # [N, C, T]
            want = torch.einsum("oc,nct->ont", W[r].double(), y)
            e = _terr(z0[grp].double(), want)
            record_err("stress full size: Wg = 0 closed form", f"z[group {grp}]", e, TOL)
            assert e < TOL
        del z0, z1

    def grads(dz):
        xt = x.detach().requires_grad_(True)
        ps = [p.detach().clone().requires_grad_(True) for p in (alpha, Wg, W)]
        ms_gat_amd.gacn(xt, ps[0], ps[1], ps[2], graph).backward(dz)
        return [xt.grad] + [p.grad for p in ps]

    dz1 = torch.randn(R * Bg, 24, N, 12, device=dev, generator=gen)
    dz2 = torch.randn(R * Bg, 24, N, 12, device=dev, generator=gen)
    g1, g2 = grads(dz1), grads(dz2)
    g12 = grads(dz1 + dz2)
    for name, a, b, c in zip(("dx", "dalpha", "dWg", "dW"), g1, g2, g12):
        e = _terr(a + b, c)
        record_err("stress full size: gradients linear in dz", name, e, 2e-5)
        assert e < 2e-5, name


def test_stress_config_widths_against_the_dense_oracle():
    """configs[4]'s graph and widths (C = 72 -> 24, R = 4) against the dense [B,N,N] op sequence on the same GPU at
    8 samples per relation (G = 32 groups; every dense [8,8192,8192] tensor is 2.1 GB, autograd keeps about eight of
    them -- round 4 ran 2 per relation).  The full B = 64 is covered by the property test above."""
    prob = random_problem(4, 8, 72, 24, 8192, 12, 65536, seed=167)
    assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what="stress widths N=8192 R=4 C=72->24 vs dense eager")


def test_mid_size_graph_uses_a_large_lds_slab():
    """64 KB < N*T*4 <= 159 KB: one slab per block with the raised dynamic-LDS limit."""
    prob = random_problem(1, 2, 12, 4, 2000, 12, 4000, seed=63)
    assert_close(run_ours(*prob), _dense_oracle_gpu(*prob), what="N=2000 PROJ_FIRST")
    prob = random_problem(1, 2, 3, 0, 2000, 12, 4000, seed=64)
    x, adj, Wg, alpha, _, dz = prob
    want = oracle_f64(x, adj, Wg, alpha, None, dz)
    assert_close(run_ours(*prob), want, what="N=2000 PLAIN")


def test_results_are_bitwise_reproducible():
    """No float atomics anywhere: two runs give identical bits."""
    # the second problem has rows of 883 x 12 positions: partial last tiles in the LDS-DMA ring of the fused
    # dW / dalpha / dx pass, whose counted waits a race would show up in as run-to-run differences
    for prob in (random_problem(2, 4, 72, 24, 307, 12, 340, seed=77), random_problem(3, 8, 72, 24, 883, 12, 866, seed=78)):
        a = run_ours(*prob)
        for _ in range(3):
            b = run_ours(*prob)
            for k in a:
                assert np.array_equal(a[k], b[k]), k


def test_cpu_tensors_are_refused_not_silently_computed():
    import ms_gat_amd
    from ms_gat_amd._lib import MsgatError
    m = ms_gat_amd.GraphAttention(3, 12)
    with pytest.raises(MsgatError):
        m(torch.randn(1, 3, 8, 12), torch.eye(8))


def test_randomised_shape_sweep():
    """Seeded random shapes across every template the library instantiates: T in {4,8,12,16}, 1..1100 nodes,
    1..130 channels, all three execution modes (Co = 0, C <= Co, C > Co), 1..3 relations, edge counts from none
    to 8 per node (rows with more than 8 edges take the tail loop of the gather)."""
    rng = np.random.default_rng(123)
    for it in range(30):
        T = int(rng.choice([4, 8, 12, 16]))
        N = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 64, 100, 257, 500, 1100]))
        C = int(rng.choice([1, 2, 3, 7, 24, 40, 72, 130]))
        Co = int(rng.choice([0, 1, 4, 8, 24, 30, 72]))
        R, Bg = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 3]))
        E = int(min(N * (N - 1) // 2, rng.choice([0, 1, N // 2, N, 3 * N, 8 * N])))
        prob = random_problem(R, Bg, C, Co, N, T, E, seed=1000 + it)
        # (a single node has softmax == 1 and dWg == 0 exactly: held to the bar on the O(1) input scale, as in
        # test_against_numpy_oracle -- the fp16 payload product returns pq = q to 1e-7, not to the bit)
        assert_close(run_ours(*prob), oracle_f64(*prob), what=f"T{T} N{N} C{C} Co{Co} R{R} Bg{Bg} E{E}",
                     floor=1.0 if N == 1 else 0.0)


@pytest.mark.parametrize("R,Bg,Cu,N,T,E", [
    (3, 8, 24, 883, 12, 866),     # headline shape: three 1024-slot trips per slab, runs of 9 slabs that cross groups
    (2, 8, 32, 1024, 12, 4000),   # N T / 4 = 3072: exactly the three buffers' size; rows with more than 8 edges
    (2, 16, 16, 307, 12, 340),    # one trip per slab
    (1, 16, 32, 500, 8, 600),     # T = 8 (two float4 per row), two trips
    (4, 8, 16, 700, 16, 900),     # T = 16
    (2, 16, 16, 2000, 4, 9000),   # T = 4, dense-ish rows (mean degree 10)
])
def test_aggregate_ring_against_float64(R, Bg, Cu, N, T, E):
    """msgat_stage_aggregate on shapes that take the persistent LDS-DMA ring (k_agg_ring: G Cu >= 512 slabs of at most
    3072 float4), v[g,c,n,:] = sum_e E[g,e] u[g,c,col_e,:] against float64; repeated runs must agree bit for bit (a
    miscounted wait of the ring would not)."""
    import ctypes as C_
    import ms_gat_amd
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev = _dev()
    G = R * Bg
    graph = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(N, E, 3))
    gs, _keep = graph.on(dev)
    nnz = graph.nnz
    g = torch.Generator().manual_seed(23)
    u = torch.randn(G, Cu, N, T, generator=g).to(dev)
    Ee = torch.rand(G, nnz, generator=g).to(dev)
    shape = _lib.Shape(R, Bg, Cu, Cu, N, T)
    nscr = int(L.msgat_edge_scratch_floats(C_.byref(shape), C_.byref(gs)))
    escr = torch.empty(max(nscr, 1), device=dev)
    outs = []
    for _ in range(3):
        v = torch.full((G, Cu, N, T), float("nan"), device=dev)
        st = L.msgat_stage_aggregate(C_.byref(shape), C_.byref(gs), Cu, u.data_ptr(), Ee.data_ptr(), v.data_ptr(),
                                     escr.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "msgat_stage_aggregate")
        outs.append(v)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    rowptr = graph.rowptr.long()
    col = graph.col[:nnz].long().to(dev)
    rows = torch.repeat_interleave(torch.arange(N), rowptr[1:] - rowptr[:-1]).to(dev)
    want = torch.zeros(G, Cu, N, T, device=dev, dtype=torch.float64)
    want.index_add_(2, rows, u.double()[:, :, col, :] * Ee.double()[:, None, :, None])
    err = rel_err(outs[0], want)
    record_err(f"aggregate ring R={R} Bg={Bg} Cu={Cu} N={N} T={T}", "v", err, 1e-5)
    assert err < 1e-5


@pytest.mark.parametrize("T,N,Bg,x_scale", [(4, 883, 8, 1.0), (8, 900, 8, 1.0), (16, 883, 8, 1.0), (12, 883, 8, 6.0),
                                            (12, 1009, 6, 1.0)])
def test_dense_passes_with_a_helper_wave_at_every_timestep_count(T, N, Bg, x_scale):
    """Shapes whose dense passes take the 7 owner waves + 1 helper wave form (k_scores7 / k_bwd_dense_col7: the 128-row
    grid would load the CUs unevenly, the 112-row grid does not) for every supported T, with large scores (the helper's
    partial maximum differs from the owners': the fold re-bases both) and with a column count whose split leaves the
    helper a partial last tile; against the dense reference ops on the GPU."""
    prob = random_problem(3, Bg, 6, 24, N, T, N, seed=90 + T, x_scale=x_scale)
    got, want = run_ours(*prob), _dense_oracle_gpu(*prob)
    if x_scale == 1.0:
        assert_close(got, want, what=f"helper-wave dense passes T={T} N={N}")
    else:   # partly saturated rows: the yardstick of test_large_scores_need_the_running_max
        f64, ref32 = oracle_f64(*prob), _ref32_cpu(*prob)
        for k in f64:
            bar = max(TOL, 3.0 * rel_err(ref32[k], f64[k]))
            assert rel_err(got[k], f64[k]) < bar, (k, rel_err(got[k], f64[k]), bar)


@pytest.mark.parametrize("T,C", [(4, 5), (8, 5), (16, 5), (12, 3), (12, 1)])
def test_eight_wave_dense_passes_on_the_fp16_payload_at_every_timestep_count(T, C):
    """The 8-wave forms of the dense passes choose their payload arithmetic by the grid (dense.hip, F16P): fp32 when the blocks
    leave CUs empty -- every other small case of this file -- and the fp16 two-term product from 257 blocks up.  96 groups of
    300 nodes are 288 blocks: the fp16 form at the timestep counts the T = 12 cases do not reach, and (C = 1, 3) in the score
    kernels that finish the layer for their own rows."""
    prob = random_problem(2, 48, C, 24, 300, T, 300, seed=700 + T + C)
    got, want = run_ours(*prob), _dense_oracle_gpu(*prob)
    assert_close(got, want, what=f"8-wave dense passes, fp16 payload, T={T} C={C}")


@pytest.mark.parametrize("N,Bg", [(100, 2), (300, 48), (883, 4), (1600, 2)])
@pytest.mark.parametrize("cq,cg", [(1e-6, 1e-8), (1e3, 1e5), (3e7, 1e-3)])
def test_payload_magnitudes_far_from_one(N, Bg, cq, cg):
    """The payload product of the dense passes runs on fp16 operands behind a power-of-two scale that follows the data
    (csrc/halfsplit.hpp).  Signals of magnitude `cq` with Wg scaled by 1 / cq^2 (the scores, and with them the attention, stay
    what they are) and cotangents of magnitude `cg`: every output and gradient against float64 at the usual bar -- fp16 on its own
    would overflow at 7e4 and lose everything below 6e-8.  N = 300 in 96 groups: the 8-wave forms (288 blocks: more than the
    CUs); 883: the 7 + 1 wave forms; 1600: the split-operand forms with their per-group image scale; N = 100 in 4 groups: a grid
    that leaves CUs empty keeps the fp32 payload product (dense.hip, F16P = false) and has to pass the same bar."""
    x, adj, Wg, alpha, W, dz = random_problem(2, Bg, 5, 24, N, 12, N, seed=4000 + N)
    prob = ((x * cq).astype(np.float32), adj, (Wg / (cq * cq)).astype(np.float32), alpha, W, (dz * cg).astype(np.float32))
    got, want = run_ours(*prob), oracle_f64(*prob)
    assert all(np.isfinite(v).all() for v in got.values())
    assert_close(got, want, what=f"payload magnitudes N{N} q~{cq:g} dz~{cg:g}")


def test_double_backward_is_refused_at_the_library_node():
    """Every backward of ms_gat_amd.ops launches kernels on raw pointers and is marked `once_differentiable`:
    `create_graph=True` yields gradients whose own backward fails with PyTorch's message, not silent constants."""
    import ms_gat_amd
    dev = torch.device("cuda:0")
    m = ms_gat_amd.GACN(3, 24, 12).to(dev)
    for p in m.parameters():          # (the module allocates its parameters; the model's reset_parameters fills them)
        torch.nn.init.normal_(p, std=0.3)
    adj = ms_gat_amd.synthetic_adjacency(20, 25, seed=1).to(dev)
    x = torch.randn(2, 3, 20, 12, device=dev, requires_grad=True)
    (gx,) = torch.autograd.grad(m(x, adj).square().sum(), x, create_graph=True)
    assert gx.requires_grad
    with pytest.raises(RuntimeError, match="once_differentiable"):
        gx.sum().backward()


def test_two_host_threads_on_two_streams_share_one_graph():
    """The header promises re-entrancy (no global mutable state, the stream is an argument): two Python threads, each
    on its own stream, run forward + backward through the SAME graph object at once -- what nn.DataParallel's replica
    threads (main.py:53-54) would do.  Results must equal the same calls issued one after the other, bit for bit."""
    import threading
    import ms_gat_amd
    dev = torch.device("cuda:0")
    N = 300
    adj = ms_gat_amd.synthetic_adjacency(N, 340, seed=3).to(dev)
    mods = [ms_gat_amd.GACN(72, 24, 12), ms_gat_amd.GACN(3, 24, 12)]
    g = torch.Generator().manual_seed(7)
    for mod in mods:                  # (the module allocates its parameters; the model's reset_parameters fills them)
        for p in mod.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else p.numel() ** -0.5))
        mod.to(dev)
    xs = [torch.randn(6, 72, N, 12, generator=g).to(dev), torch.randn(6, 3, N, 12, generator=g).to(dev)]
    dzs = [torch.randn(6, 24, N, 12, generator=g).to(dev) for _ in range(2)]

    def run(k, stream, out):
        with torch.cuda.stream(stream):
            for rep in range(4):
                x = xs[k].clone().requires_grad_(True)
                for p in mods[k].parameters():
                    p.grad = None
                z = mods[k](x, adj)
                z.backward(dzs[k])
                out[k] = [z.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mods[k].parameters()]
            stream.synchronize()

    serial = {}
    for k in range(2):
        run(k, torch.cuda.current_stream(), serial)
    torch.cuda.synchronize()
    together, errs = {}, []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def guarded(k):
        try:
            run(k, streams[k], together)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    threads = [threading.Thread(target=guarded, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(2):
        for a, b in zip(serial[k], together[k]):
            assert torch.isfinite(a).all() and a.abs().max() > 0
            assert torch.equal(a, b)


def test_forced_split_arithmetic_child_run():
    """From N = 1536 nodes the two dense passes multiply on the bf16 / fp16 matrix core with split operands
    (csrc/dense_bf16.hip); the stress cases above run that way by themselves.  MSGAT_DENSE_SPLIT=1 (read once per process)
    forces it for EVERY T = 12 shape: a child process repeats the reference goldens, the float64 oracle sweep, the
    saturation cases, the stage outputs, the reproducibility and the two-thread test on those kernels -- ragged last
    tiles, a single node, empty rows, one-hot rows whose dense and sparse halves must still cancel."""
    import os
    import subprocess
    import sys
    keep = ("golden or headline or drop_in or numpy_oracle or large_scores or saturated or empty_rows or all_zero or rows_sum "
            "or stage_outputs or bitwise or randomised or two_host_threads or pemsd4_b64")
    env = dict(os.environ, MSGAT_DENSE_SPLIT="1")
    env.pop("MSGAT_PARITY_LOG", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k", keep],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout
