"""GPU: LayerNorm over the timestep axis (the producer of every GACN
input; reference nn.LayerNorm([T]) at src/models/msgat.py:114/:122 and :152/:158) against a float64
evaluation of the same formula, with torch's own fp32 kernel as the error yardstick.

Tolerance: 1e-4 relative (the path's stated bar); in practice both fp32 implementations sit at ~1e-7.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref64(x, w, b, dy, eps):
    x64 = x.double().requires_grad_(True)
    w64 = None if w is None else w.double().requires_grad_(True)
    b64 = None if b is None else b.double().requires_grad_(True)
    y = F.layer_norm(x64, [x.shape[-1]], w64, b64, eps)
    y.backward(dy.double())
    return y.detach(), x64.grad, (None if w64 is None else w64.grad), (None if b64 is None else b64.grad)


@pytest.mark.parametrize("shape", [(1, 1, 1, 12), (2, 3, 5, 12), (4, 24, 883, 12), (3, 7, 129, 4), (2, 5, 300, 8),
                                   (1, 9, 1025, 16), (32, 72, 883, 12)])
@pytest.mark.parametrize("affine", [True, False])
def test_layernorm_matches_float64_reference(shape, affine):
    from ms_gat_amd import ops
    if not affine and shape[0] == 32:
        pytest.skip("full size once")
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(sum(shape))
    T = shape[-1]
    x = (torch.randn(shape, generator=gen) * 3 + 1.5).to(dev)
    dy = torch.randn(shape, generator=gen).to(dev)
    w = (1 + 0.3 * torch.randn(T, generator=gen)).to(dev) if affine else None
    b = (0.2 * torch.randn(T, generator=gen)).to(dev) if affine else None
    xs = x.clone().requires_grad_(True)
    ws = None if w is None else w.clone().requires_grad_(True)
    bs = None if b is None else b.clone().requires_grad_(True)
    y = ops.layer_norm_t(xs, ws, bs, 1e-5)
    y.backward(dy)
    y64, dx64, dw64, db64 = _ref64(x, w, b, dy, 1e-5)
    assert rel_err(y.detach().double(), y64) < TOL
    assert rel_err(xs.grad.double(), dx64) < TOL
    if affine:
        assert rel_err(ws.grad.double(), dw64) < TOL
        assert rel_err(bs.grad.double(), db64) < TOL
    # yardstick: no worse than 4x torch's own fp32 kernel (plus a floor for exact-zero errors)
    xt = x.clone().requires_grad_(True)
    yt = F.layer_norm(xt, [T], w, b, 1e-5)
    yt.backward(dy)
    assert rel_err(y.detach().double(), y64) <= 4 * rel_err(yt.detach().double(), y64) + 1e-6
    assert rel_err(xs.grad.double(), dx64) <= 4 * rel_err(xt.grad.double(), dx64) + 1e-6


def test_constant_rows_and_large_offsets():
    """var = 0 rows give exactly `bias`; a large common offset must not destroy the variance
    (two-pass statistics, not E[x^2] - E[x]^2)."""
    from ms_gat_amd import ops
    dev = _dev()
    x = torch.full((3, 4, 12), 7.25, device=dev)
    w, b = torch.linspace(0.5, 1.5, 12, device=dev), torch.linspace(-1, 1, 12, device=dev)
    assert torch.equal(ops.layer_norm_t(x, w, b), b.expand_as(x))
    gen = torch.Generator().manual_seed(3)
    x = (1e4 + torch.randn(5, 6, 77, 12, generator=gen)).to(dev)
    y64 = F.layer_norm(x.double(), [12], w.double(), b.double(), 1e-5)
    assert rel_err(ops.layer_norm_t(x, w, b).double(), y64) < TOL


def test_non_contiguous_input_empty_input_and_unsupported_T():
    from ms_gat_amd import _lib, ops
    dev = _dev()
    x = torch.randn(4, 12, 6, 5, device=dev).transpose(1, 3)          # [4,5,6,12] view, strided
    assert rel_err(ops.layer_norm_t(x), F.layer_norm(x, [12])) < 1e-6
    assert ops.layer_norm_t(torch.empty(0, 3, 12, device=dev)).shape == (0, 3, 12)
    with pytest.raises(_lib.MsgatError):
        ops.layer_norm_t(torch.randn(4, 10, device=dev))


def test_gradients_are_bitwise_reproducible():
    from ms_gat_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(8, 24, 883, 12, generator=gen).to(dev)
    dy = torch.randn(8, 24, 883, 12, generator=gen).to(dev)
    w, b = torch.rand(12, device=dev) + 0.5, torch.rand(12, device=dev)
    outs = []
    for _ in range(2):
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ops.layer_norm_t(xs, ws, bs).backward(dy)
        outs.append((xs.grad, ws.grad, bs.grad))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


def test_module_is_drop_in_for_nn_layernorm():
    from ms_gat_amd import model
    dev = _dev()
    ref = torch.nn.LayerNorm([12]).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    mine = model.LayerNormT(12).to(dev)
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(2, 72, 100, 12, device=dev)
    assert rel_err(mine(x), ref(x)) < 1e-6


@pytest.mark.parametrize("B,R,C,N,T,To,relu", [(6, 3, 72, 883, 12, 12, True), (2, 1, 5, 30, 12, 7, False), (4, 2, 9, 257, 8, 8, True),
                                               (2, 2, 17, 64, 16, 12, True), (3, 1, 8, 1, 4, 4, False), (2, 1, 24, 300, 12, 3, True)])
def test_ln_head_is_the_two_ops_with_one_backward_pass(B, R, C, N, T, To, relu):
    """ops.ln_head = head(layer_norm_t(x)) (msgat.py:158-160) whose backward builds the head's input gradient inside the
    LayerNorm-backward pass (msgat_layernorm_head_backward): against the float64 op sequence and against the two separate
    ops (same formulas in the same order: dx must agree to rounding of the partial sums only)."""
    from ms_gat_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(B + 3 * C + N)
    x = torch.randn(B, C, N, T, generator=gen).to(dev)
    if relu:
        x = torch.relu(x + 0.3)                          # a ReLU output, as the blocks hand it over
    shp = (To, T, 1, C) if R == 1 else (R, To, T, 1, C)
    W = (torch.randn(*shp, generator=gen) * (T * C) ** -0.5).to(dev)
    hb = (torch.randn(*((To,) if R == 1 else (R, To)), generator=gen) * 0.1).to(dev)
    lw = (1 + 0.3 * torch.randn(*((T,) if R == 1 else (R, T)), generator=gen)).to(dev)
    lb = (0.2 * torch.randn(*((T,) if R == 1 else (R, T)), generator=gen)).to(dev)
    dout = torch.randn(B, N, To, generator=gen).to(dev)

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, W, hb)]
        xs, lws, lbs, Ws, hbs = leaves
        out = (ops.ln_head(xs, lws, lbs, 1e-5, Ws, hbs, relu_input=relu) if fused
               else ops.head(ops.layer_norm_t(xs, lws, lbs, 1e-5, relu_input=relu), Ws, hbs))
        out.backward(dout)
        return [out.detach()] + [t.grad for t in leaves]

    got, two = run(True), run(False)
    names = ("out", "dx", "dln_weight", "dln_bias", "dW", "dbias")
    for name, a, b in zip(names, got, two):           # the head kernels normalise in their own lane order: rounding only
        assert rel_err(a, b) < 2e-6, name
    # float64: the reference's op sequence, relation by relation
    Bg = B // R
    x64 = x.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (lw, lb, W, hb)]
    outs = []
    for r in range(R):
        sl = slice(r * Bg, (r + 1) * Bg)
        pr = [p if R == 1 else p[r] for p in p64]
        xn = F.layer_norm(x64[sl], [T], pr[0], pr[1], 1e-5)
        outs.append(F.conv2d(xn.transpose(1, 3), pr[2], pr[3])[..., 0].transpose(1, 2))
    ref = torch.cat(outs)
    ref.backward(dout.double())
    dx64 = x64.grad if not relu else x64.grad * (x > 0)       # the mask of the ReLU that produced x (relu_input)
    for name, a, b in zip(names, got, [ref.detach(), dx64] + [p.grad for p in p64]):
        assert rel_err(a.double(), b) < TOL, name


@pytest.mark.parametrize("B,R,C,N,T,relu", [(6, 3, 72, 883, 12, True), (2, 1, 5, 30, 12, False), (4, 2, 9, 257, 8, True), (2, 2, 3, 1, 16, False)])
def test_layer_norm_pool_tee_adds_the_pooling_gradient_inside_the_layernorm_backward(B, R, C, N, T, relu):
    """ops.layer_norm_pool_tee = layer_norm_t_tee + node_pool_tee as one node (msgat.py:122-125, attention.py:89): the
    pooling's rank-one gradient joins dy inside the LayerNorm-backward kernel.  Against the two separate ops (same
    formulas; only the place of one addition differs) and against float64."""
    from ms_gat_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cpu").manual_seed(B + C + N)
    x = torch.randn(B, C, N, T, generator=gen).to(dev)
    if relu:
        x = torch.relu(x + 0.3)
    lw = (1 + 0.3 * torch.randn(*((T,) if R == 1 else (R, T)), generator=gen)).to(dev)
    lb = (0.2 * torch.randn(*((T,) if R == 1 else (R, T)), generator=gen)).to(dev)
    pw = torch.randn(*((N,) if R == 1 else (R, N)), generator=gen).to(dev)
    d_y, d_x2, d_p = (torch.randn(*s, generator=gen).to(dev) for s in ((B, C, N, T), (B, C, N, T), (B, C, T)))

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (x, lw, lb, pw)]
        xs, lws, lbs, pws = leaves
        if fused:
            y, x2, p = ops.layer_norm_pool_tee(xs, lws, lbs, 1e-5, relu, pws)
        else:
            y, x2 = ops.layer_norm_t_tee(xs, lws, lbs, 1e-5, relu)
            p, y = ops.node_pool_tee(y, pws)
        torch.autograd.backward([y, x2, p], [d_y, d_x2, d_p])
        return [y.detach(), p.detach()] + [t.grad for t in leaves]

    got, two = run(True), run(False)
    names = ("y", "pooled", "dx", "dln_weight", "dln_bias", "dpool_w")
    for name, a, b in zip(names, got, two):
        assert rel_err(a, b) < 2e-6, name
    Bg = B // R
    x64 = x.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (lw, lb, pw)]
    ys, ps = [], []
    for r in range(R):
        sl = slice(r * Bg, (r + 1) * Bg)
        pr = [p if R == 1 else p[r] for p in p64]
        yr = F.layer_norm(x64[sl], [T], pr[0], pr[1], 1e-5)
        ys.append(yr)
        ps.append(torch.einsum("bcnt,n->bct", yr, pr[2]))
    y64, pp64 = torch.cat(ys), torch.cat(ps)
    torch.autograd.backward([y64, x64 * 1.0, pp64], [d_y.double(), d_x2.double(), d_p.double()])
    dx64 = x64.grad if not relu else x64.grad * (x > 0)
    for name, a, b in zip(names, got, [y64.detach(), pp64.detach(), dx64] + [p.grad for p in p64]):
        assert rel_err(a.double(), b) < TOL, name
