"""Pins the CPU oracle (oracle/) to the golden vectors produced by the reference.

CPU only.  Tolerances: the fp32 dense torch restatement must match the reference
(also fp32 torch) to 2e-6 relative; the numpy oracle run in float64 must match the
reference's fp32 results to 2e-5 relative (the reference's own rounding).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_headline_golden, rel_err
from oracle import dense_torch, gat_oracle


def test_numpy_forward_backward_matches_reference(gatt_case):
    _, g, c = gatt_case
    f64 = lambda k, d=g: d[k].astype(np.float64)  # noqa: E731
    y = gat_oracle.gatt_forward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"))
    assert rel_err(y, g["y"]) < 2e-5
    dx, dWg, dalpha = gat_oracle.gatt_backward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("dy"))
    assert rel_err(dx, g["dx"]) < 2e-5
    assert rel_err(dWg, g["dWg"]) < 2e-5
    assert rel_err(dalpha, g["dalpha"]) < 2e-5

    z = gat_oracle.gacn_forward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W", c))
    assert rel_err(z, c["z"]) < 2e-5
    dx, dWg, dalpha, dW = gat_oracle.gacn_backward(
        f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W", c), f64("dz", c))
    assert rel_err(dx, c["dx"]) < 2e-5
    assert rel_err(dWg, c["dWg"]) < 2e-5
    assert rel_err(dalpha, c["dalpha"]) < 2e-5
    assert rel_err(dW, c["dW"]) < 2e-5


def test_numpy_oracle_in_float32_is_within_bar(gatt_case):
    """The oracle run in the reference's own arithmetic type stays inside the 1e-4 bar."""
    _, g, c = gatt_case
    y = gat_oracle.gatt_forward(g["x"], g["adj"], g["Wg"], g["alpha"])
    assert y.dtype == np.float32
    assert rel_err(y, g["y"]) < 1e-5
    dx, dWg, dalpha, dW = gat_oracle.gacn_backward(g["x"], g["adj"], g["Wg"], g["alpha"], c["W"], c["dz"])
    for got, want in ((dx, c["dx"]), (dWg, c["dWg"]), (dalpha, c["dalpha"]), (dW, c["dW"])):
        assert rel_err(got, want) < 1e-4


def test_dense_torch_matches_reference(gatt_case):
    _, g, c = gatt_case
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    x = t(g["x"]).requires_grad_(True)
    Wg = t(g["Wg"]).requires_grad_(True)
    alpha = t(g["alpha"]).requires_grad_(True)
    y = dense_torch.graph_attention_dense(x, t(g["adj"]), Wg, alpha)
    y.backward(t(g["dy"]))
    assert rel_err(y.detach(), g["y"]) < 2e-6
    assert rel_err(x.grad, g["dx"]) < 2e-6
    assert rel_err(Wg.grad, g["dWg"]) < 5e-6
    assert rel_err(alpha.grad, g["dalpha"]) < 5e-6

    x = t(g["x"]).requires_grad_(True)
    W = t(c["W"]).requires_grad_(True)
    z = dense_torch.gacn_dense(x, t(g["adj"]), t(g["Wg"]), t(g["alpha"]), W)
    z.backward(t(c["dz"]))
    assert rel_err(z.detach(), c["z"]) < 2e-6
    assert rel_err(x.grad, c["dx"]) < 2e-6
    assert rel_err(W.grad, c["dW"]) < 5e-6


def test_oracles_match_the_reference_at_the_headline_size():
    """N = 883 (PEMSD7-like), C = 72 -> 24, one sample: both oracles against what the reference itself computed there."""
    g = load_headline_golden()
    f64 = lambda k: g[k].astype(np.float64)  # noqa: E731
    z = gat_oracle.gacn_forward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W"))
    assert rel_err(z, g["z"]) < 2e-5
    dx, dWg, dalpha, dW = gat_oracle.gacn_backward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W"), f64("dz"))
    for name, got in (("dx", dx), ("dWg", dWg), ("dalpha", dalpha), ("dW", dW)):
        assert rel_err(got, g[name]) < 2e-5, name
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    x = t(g["x"]).requires_grad_(True)
    W = t(g["W"]).requires_grad_(True)
    zt = dense_torch.gacn_dense(x, t(g["adj"]), t(g["Wg"]), t(g["alpha"]), W)
    zt.backward(t(g["dz"]))
    assert rel_err(zt.detach(), g["z"]) < 2e-6
    assert rel_err(x.grad, g["dx"]) < 2e-6
    assert rel_err(W.grad, g["dW"]) < 5e-6


@pytest.mark.parametrize("tag", ["w48_n64", "w96_n64"])
def test_oracles_match_the_reference_at_the_other_registry_widths(tag):
    """GACN(48 -> 16) / GACN(96 -> 32) (msgat48 / msgat96, msgat.py:220-229) at N = 64: both oracles against the
    reference's own forward and autograd, so the GPU tests of those widths stand on a pinned checker."""
    g = load_golden(f"gacn_{tag}.npz")
    f64 = lambda k: g[k].astype(np.float64)  # noqa: E731
    z = gat_oracle.gacn_forward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W"))
    assert rel_err(z, g["z"]) < 2e-5
    dx, dWg, dalpha, dW = gat_oracle.gacn_backward(f64("x"), f64("adj"), f64("Wg"), f64("alpha"), f64("W"), f64("dz"))
    for name, got in (("dx", dx), ("dWg", dWg), ("dalpha", dalpha), ("dW", dW)):
        assert rel_err(got, g[name]) < 2e-5, name
    t = lambda a: torch.from_numpy(a).float()  # noqa: E731
    x = t(g["x"]).requires_grad_(True)
    W = t(g["W"]).requires_grad_(True)
    zt = dense_torch.gacn_dense(x, t(g["adj"]), t(g["Wg"]), t(g["alpha"]), W)
    zt.backward(t(g["dz"]))
    assert rel_err(zt.detach(), g["z"]) < 2e-6
    assert rel_err(x.grad, g["dx"]) < 2e-6
    assert rel_err(W.grad, g["dW"]) < 5e-6


@pytest.mark.parametrize("tag,dilations", [("48to48_n64", [2, 4]), ("96to96_n64", [4, 4]), ("72to72_n64", [1, 2])])
def test_dense_meam_restatement_matches_the_reference_block(tag, dilations):
    """oracle/dense_torch.meam_dense (the eager baseline of bench.py and the checker of the model tests) against the
    reference's MEAM output and input gradient at the second-block widths of all three registry models."""
    g = load_golden(f"meam_{tag}.npz")
    state = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    for v in state.values():
        v.requires_grad_(True)
    x = torch.from_numpy(g["x"]).float().requires_grad_(True)
    out = dense_torch.meam_dense(x, torch.from_numpy(g["adj"]), state, dilations)
    out.backward(torch.from_numpy(g["dout"]).float())
    assert rel_err(out.detach(), g["out"]) < 5e-6
    assert rel_err(x.grad, g["dx"]) < 5e-6
    for k, v in state.items():
        assert rel_err(v.grad, g[f"g.{k}"]) < 2e-5, k


def test_lse_matches_dense_softmax_denominator():
    g = load_golden("gatt_b2c3n16.npz")
    x = g["x"].astype(np.float64)
    lse = gat_oracle.gatt_lse(x, g["Wg"].astype(np.float64), g["alpha"].astype(np.float64))
    q = np.einsum("bcnt,c->bnt", x, g["alpha"].astype(np.float64))
    S = (q @ g["Wg"].astype(np.float64)) @ q.transpose(0, 2, 1)
    assert np.allclose(np.exp(S - lse[..., None]).sum(-1), 1.0, atol=1e-12)


def test_masked_rows_do_not_sum_to_one():
    """the mask is applied AFTER the full-row softmax."""
    g = load_golden("gatt_b2c3n16.npz")
    _, cache = gat_oracle.gatt_forward(g["x"], g["adj"], g["Wg"], g["alpha"], return_cache=True)
    assert np.allclose(cache["P"].sum(-1), 1.0, atol=1e-5)
    assert cache["E"].sum(-1).max() < 0.9


def test_adjacency_matches_reference_loader():
    a = load_golden("adj_n12.npz")
    adj = gat_oracle.sym_norm_adjacency(int(a["n"]), a["edges"])
    assert np.allclose(adj, a["adj"], atol=1e-7)
    assert np.allclose(adj, adj.T)


def test_huber_matches_reference():
    m = load_golden("msgat72_n32.npz")
    got = gat_oracle.huber_loss(m["pred"].astype(np.float64), m["Y"].astype(np.float64), 50.0)
    assert abs(got - float(m["loss"])) < 1e-4 * abs(float(m["loss"]))
    got_t = dense_torch.huber(torch.from_numpy(m["pred"]), torch.from_numpy(m["Y"]), 50.0)
    assert abs(float(got_t) - float(m["loss"])) < 1e-5 * abs(float(m["loss"]))


def _cfg1_state(g, n_components=5, n_nodes=307, T=12):
    """state_dict of msgat72 from the cfg1 fixture: the time-embedding tables are rebuilt with the stored rows at
    the indices H and D select (the other rows cannot influence the forward)."""
    import torch
    state = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    width = n_components * n_nodes * T
    h, d = torch.zeros(24, width), torch.zeros(7, width)
    h[torch.from_numpy(g["H"])] = torch.from_numpy(g["te_h_rows"])
    d[torch.from_numpy(g["D"])] = torch.from_numpy(g["te_d_rows"])
    state["te.h_ebd.weight"], state["te.d_ebd.weight"] = h, d
    return state


def test_cfg1_pemsd4_full_forward_of_the_oracle_matches_the_reference():
    """BASELINE.json configs[0]: PEMSD4-like (307 nodes, 3 features, T=12, B=4), five components, msgat72 forward on
    the CPU -- the dense restatement of every block (oracle/dense_torch.py) against the reference's prediction."""
    g = load_golden("msgat72_cfg1_pemsd4.npz")
    state = _cfg1_state(g)
    X, H, D = torch.from_numpy(g["X"]).float(), torch.from_numpy(g["H"]), torch.from_numpy(g["D"])
    adj = torch.zeros(307, 307)
    adj[torch.from_numpy(g["adj_rows"].astype(np.int64)), torch.from_numpy(g["adj_cols"].astype(np.int64))] = \
        torch.from_numpy(g["adj_vals"])
    B, R, T = X.shape[0], X.shape[1], X.shape[-1]
    gate = (state["te.h_ebd.weight"][H] + state["te.d_ebd.weight"][D]).view(B, R, 307, T)
    out = 0
    with torch.no_grad():
        for r in range(R):
            x = X[:, r]
            for l in range(2):
                pre = f"tpcs.{r}.tgacns.{l}."
                sub = {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}
                x = dense_torch.meam_dense(x, adj, sub, [1, 2] if l == 0 else [2, 4])
            x = torch.nn.functional.layer_norm(x, [T], state[f"tpcs.{r}.ln.weight"], state[f"tpcs.{r}.ln.bias"], 1e-5)
            y = torch.nn.functional.conv2d(x.transpose(1, 3), state[f"tpcs.{r}.fc.weight"], state[f"tpcs.{r}.fc.bias"])
            out = out + y[..., 0].transpose(1, 2) * gate[:, r]
    assert rel_err(out, g["pred"]) < 1e-5
