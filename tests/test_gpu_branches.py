"""GPU: the temporal / channel branch kernels against the dense restatement
of the reference ops (oracle/dense_torch.py, evaluated in float64), forward and every gradient.

Reference: TemporalAttention attention.py:58-66, ChannelAttention :88-94, TACN msgat.py:57-80,
CACN :83-100, MEAM tail :130-131.  Tolerance 1e-4 relative (the path's bar)."""
import pytest
import torch

from conftest import record_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(_dev())


def _grads(out, inputs, dout):
    return torch.autograd.grad(out, inputs, dout, allow_unused=True)


def _check(ours, ref64, names):
    for name, a, b in zip(names, ours, ref64):
        assert rel_err(a.double(), b) < TOL, name


def _leaf(*ts):
    return [t.clone().requires_grad_(True) for t in ts]


def _leaf64(*ts):
    return [t.double().clone().requires_grad_(True) for t in ts]


@pytest.mark.parametrize("G,R,Ci,Co,N,T,bias,add,relu", [
    (4, 1, 72, 24, 57, 12, True, False, False),     # 1x1 convolution with bias
    (6, 6, 72, 24, 33, 12, True, False, False),     # per-sample matrix (CACN)
    (3, 1, 72, 72, 41, 12, True, True, True),       # MEAM tail
    (2, 1, 3, 72, 19, 12, True, True, True),        # first MEAM: 3 input channels
    (2, 2, 5, 4, 23, 8, False, False, True),        # few outputs: VALU kernel
    (32, 1, 72, 48, 883, 12, False, False, False),  # PEMSD7 size: TACN's channel mixing
])
def test_mix_matches_dense_ops(G, R, Ci, Co, N, T, bias, add, relu):
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(G * 100 + Ci)
    x, M = _rand(gen, G, Ci, N, T), _rand(gen, R, Co, Ci, scale=Ci ** -0.5)
    b = _rand(gen, Co) if bias else None
    a = _rand(gen, G, Co, N, T) if add else None
    dout = _rand(gen, G, Co, N, T)
    ins = [t for t in (x, M, b, a) if t is not None]
    mine = _leaf(*ins)
    it = iter(mine)
    out = ops.mix(next(it), next(it), next(it) if bias else None, next(it) if add else None, relu)
    g_mine = _grads(out, mine, dout)

    ref = _leaf64(*ins)
    it = iter(ref)
    x64, M64 = next(it), next(it)
    o64 = torch.einsum("roc,rgcnt->rgont", M64, x64.view(R, G // R, Ci, N, T)).reshape(G, Co, N, T)
    if bias:
        o64 = o64 + next(it).view(1, Co, 1, 1)
    if add:
        o64 = o64 + next(it)
    if relu:
        o64 = torch.relu(o64)
    g_ref = _grads(o64, ref, dout.double())
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, g_ref, ["dx", "dM", "dbias", "dadd"])


@pytest.mark.parametrize("B,C,N,T", [(2, 3, 17, 12), (4, 72, 307, 12), (3, 5, 64, 8), (32, 72, 883, 12)])
def test_temporal_attention_module(B, C, N, T):
    from ms_gat_amd import model
    from oracle import dense_torch
    gen = torch.Generator().manual_seed(B + C)
    m = model.TemporalAttention(C, N).to(_dev())
    with torch.no_grad():
        m.Wt1.copy_(_rand(gen, 10, N, scale=N ** -0.5))
        m.Wt2.copy_(_rand(gen, 10, N, scale=N ** -0.5))
        m.alpha.copy_(_rand(gen, C, scale=C ** -0.5))
    x = _rand(gen, B, C, N, T).requires_grad_(True)
    dout = _rand(gen, B, C, N, T)
    out = m(x)
    g_mine = _grads(out, [x, m.Wt1, m.Wt2, m.alpha], dout)
    r = _leaf64(x.detach(), m.Wt1.detach(), m.Wt2.detach(), m.alpha.detach())
    o64 = dense_torch.temporal_attention_dense(*r)
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, _grads(o64, r, dout.double()), ["dx", "dWt1", "dWt2", "dalpha"])


@pytest.mark.parametrize("B,C,N,T", [(2, 3, 17, 12), (4, 72, 307, 12), (3, 20, 64, 16), (32, 72, 883, 12)])
def test_channel_attention_module(B, C, N, T):
    from ms_gat_amd import model
    from oracle import dense_torch
    gen = torch.Generator().manual_seed(B + C + 1)
    m = model.ChannelAttention(N, T).to(_dev())
    with torch.no_grad():
        m.Wc.copy_(_rand(gen, T, T, scale=0.05))
        m.alpha.copy_(_rand(gen, N, scale=N ** -0.5))
    x = _rand(gen, B, C, N, T).requires_grad_(True)
    dout = _rand(gen, B, C, N, T)
    out = m(x)
    g_mine = _grads(out, [x, m.Wc, m.alpha], dout)
    r = _leaf64(x.detach(), m.Wc.detach(), m.alpha.detach())
    o64 = dense_torch.channel_attention_dense(*r)
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, _grads(o64, r, dout.double()), ["dx", "dWc", "dalpha"])


@pytest.mark.parametrize("B,Ci,Co,N,T,dil", [
    (2, 3, 24, 21, 12, [1, 2]), (2, 72, 24, 50, 12, [2, 4]), (2, 8, 16, 30, 12, [1, 1, 2, 2]), (2, 6, 8, 20, 12, [4, 4]),
    (2, 5, 8, 13, 8, [3]), (2, 4, 8, 11, 4, [4, 1]), (2, 6, 8, 9, 12, []), (32, 72, 24, 883, 12, [2, 4]),
])
def test_tacn_module(B, Ci, Co, N, T, dil):
    from ms_gat_amd import model
    from oracle import dense_torch
    gen = torch.Generator().manual_seed(B + Ci + len(dil))
    m = model.TACN(Ci, Co, N, dil).to(_dev())
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(_rand(gen, *p.shape, scale=(p.shape[-1] if p.dim() > 1 else p.shape[0]) ** -0.5))
    x = _rand(gen, B, Ci, N, T).requires_grad_(True)
    out = m(x)
    dout = _rand(gen, *out.shape)
    params = list(m.parameters())
    g_mine = _grads(out, [x] + params, dout)
    ta = m.seq[0]
    r = _leaf64(x.detach(), *[p.detach() for p in params])
    byname = dict(zip([n for n, _ in m.named_parameters()], r[1:]))
    convs = [(byname[f"seq.{2 * i + 1}.weight"], byname[f"seq.{2 * i + 1}.bias"], d) for i, d in enumerate(dil)]
    o64 = dense_torch.tacn_dense(r[0], byname["seq.0.Wt1"], byname["seq.0.Wt2"], byname["seq.0.alpha"], convs)
    assert out.shape == o64.shape
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, _grads(o64, r, dout.double()), ["dx"] + [n for n, _ in m.named_parameters()])


@pytest.mark.parametrize("R,Bg,Ci,Co,N,T,d", [
    (3, 2, 24, 24, 883, 12, 2), (1, 3, 16, 16, 307, 12, 4), (2, 1, 32, 32, 64, 12, 1), (1, 2, 24, 16, 13, 8, 2),
    (1, 2, 16, 24, 50, 16, 4), (2, 2, 24, 24, 5, 4, 1), (1, 2, 16, 16, 20, 4, 4), (1, 1, 24, 24, 1, 12, 3),
    (1, 2, 40, 72, 33, 12, 2),
])
def test_causal_conv_is_the_dilated_convolution_and_its_autograd_in_one_pass_each(R, Bg, Ci, Co, N, T, d):
    """ops.causal_conv = Conv2d(Ci, Co, [1,2], padding=[0,d], dilation=[1,d]) + Chomp(d) (msgat.py:69-74) for R parameter
    sets: forward and input gradient are ONE pass each (k_project_mfma<.., TAPS>: the shifted tap is an unaligned load of
    the same row of T); against torch's own convolution in float64, and the gradient also when it arrives as a channel
    slice of a wider tensor.  The two-pass form (channel mixing + time mixing) must agree to rounding."""
    from ms_gat_amd import ops
    dev, G = _dev(), R * Bg
    gen = torch.Generator().manual_seed(Ci + 3 * Co + d)
    assert ops.causal_conv_fused(Ci, Co)
    h = _rand(gen, G, Ci, N, T).requires_grad_(True)
    w = _rand(gen, R, Co, Ci, 1, 2, scale=(2 * Ci) ** -0.5)          # nn.Conv2d weight layout per relation
    bias = _rand(gen, R, Co, scale=0.3)
    taps = torch.cat([w[..., 0, 0], w[..., 0, 1]], dim=1).contiguous().requires_grad_(True)   # [R, 2Co, Ci] = [W0; W1]
    bias_l = bias.clone().requires_grad_(True)
    wide = _rand(gen, G, Co + 5, N, T)
    dout = wide[:, 2:2 + Co]                                          # a channel slice: read in place
    out = ops.causal_conv(h, taps, bias_l, d)
    out.backward(dout)
    # float64 reference: the reference's own module sequence, relation by relation
    h64 = h.detach().double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    refs = []
    for r in range(R):
        y = torch.nn.functional.conv2d(h64[r * Bg:(r + 1) * Bg], w64[r], b64[r], padding=(0, d), dilation=(1, d))
        refs.append(y[..., : y.size(-1) - d])
    ref = torch.cat(refs)
    ref.backward(dout.double())
    what = f"causal_conv R={R} {Ci}->{Co} N={N} T={T} d={d}"
    for key, got, want in (("out", out.detach(), ref.detach()), ("dh", h.grad, h64.grad), ("dbias", bias_l.grad, b64.grad),
                           ("dtaps", taps.grad, torch.cat([w64.grad[..., 0, 0], w64.grad[..., 0, 1]], dim=1))):
        err = rel_err(got, want)
        record_err(what, key, err, 1e-5)
        assert err < 1e-5, key
    # the two passes it replaces
    (mixed,) = ops.mix_multi([h.detach()], taps.detach())
    two = ops.time_mix(mixed, ops.causal_shift_taps(T, d, dev), bias)
    assert rel_err(out.detach(), two) < 2e-6
    for _ in range(2):   # repeats are bit-identical
        assert torch.equal(ops.causal_conv(h.detach(), taps.detach(), bias, d), out.detach())


@pytest.mark.parametrize("B,Ci,Co,N,T", [(2, 3, 24, 21, 12), (3, 72, 24, 50, 12), (32, 72, 24, 883, 12)])
def test_cacn_module(B, Ci, Co, N, T):
    from ms_gat_amd import model
    from oracle import dense_torch
    gen = torch.Generator().manual_seed(B + Ci + 7)
    m = model.CACN(Ci, Co, N, T).to(_dev())
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(_rand(gen, *p.shape, scale=0.1))
    x = _rand(gen, B, Ci, N, T).requires_grad_(True)
    out = m(x)
    dout = _rand(gen, *out.shape)
    params = list(m.parameters())
    g_mine = _grads(out, [x] + params, dout)
    r = _leaf64(x.detach(), *[p.detach() for p in params])
    byname = dict(zip([n for n, _ in m.named_parameters()], r[1:]))
    o64 = dense_torch.cacn_dense(r[0], byname["seq.0.Wc"], byname["seq.0.alpha"], byname["seq.1.weight"],
                                 byname["seq.1.bias"])
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, _grads(o64, r, dout.double()), ["dx"] + [n for n, _ in m.named_parameters()])


def test_branch_ops_refuse_cpu_tensors():
    from ms_gat_amd import _lib, ops
    x = torch.zeros(2, 3, 5, 12)
    for call in (lambda: ops.mix(x, torch.zeros(1, 4, 3)), lambda: ops.time_mix(x, torch.zeros(1, 1, 12, 12)),
                 lambda: ops.node_pool(x, torch.zeros(5)), lambda: ops.channel_pool(x, torch.zeros(3))):
        with pytest.raises(_lib.MsgatError):
            call()


@pytest.mark.parametrize("B,C,N,T,To", [(2, 72, 33, 12, 12), (3, 5, 300, 12, 6), (2, 9, 17, 8, 16), (32, 72, 883, 12, 12)])
def test_prediction_head_matches_the_transposed_convolution(B, C, N, T, To):
    """TPC.fc (msgat.py:153,:159-160): Conv2d(T, T_out, [1, C]) on x.transpose(1, 3), squeezed, transposed."""
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(B + C + To)
    x, W, b = _rand(gen, B, C, N, T), _rand(gen, To, T, 1, C, scale=(T * C) ** -0.5), _rand(gen, To)
    dout = _rand(gen, B, N, To)
    mine = _leaf(x, W, b)
    out = ops.head(*mine)
    g_mine = _grads(out, mine, dout)
    ref = _leaf64(x, W, b)
    o64 = torch.nn.functional.conv2d(ref[0].transpose(1, 3), ref[1], ref[2])[..., 0].transpose(1, 2)
    assert out.shape == o64.shape
    assert rel_err(out.double(), o64) < TOL
    _check(g_mine, _grads(o64, ref, dout.double()), ["dx", "dW", "dbias"])


def test_mix_multi_reads_channel_slices_in_place_and_writes_several_outputs():
    """Segment lists (include/msgat_hip.h: msgat_mix_segments): inputs that are channel slices of wider tensors,
    several inputs, several outputs, add operands given as separate tensors -- against cat + matmul in float64."""
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(21)
    G, N, T = 4, 37, 12
    wide = _rand(gen, G, 20, N, T)
    a, b = wide[:, 2:9], wide[:, 11:20]                   # 7 and 9 channels, strided in the group axis
    c = _rand(gen, G, 5, N, T)
    M = _rand(gen, 2, 30, 21, scale=0.2)                   # R = 2 relations, 21 -> 30 channels
    bias = _rand(gen, 30)
    adds = [_rand(gen, G, 8, N, T), _rand(gen, G, 22, N, T)]
    leaves = _leaf(wide, c, M, bias, *adds)
    w_, c_, M_, b_, *ad_ = leaves
    outs = ops.mix_multi([w_[:, 2:9], w_[:, 11:20], c_], M_, b_, adds=ad_, out_channels=[4, 16, 10])
    douts = [_rand(gen, *o.shape) for o in outs]
    g_mine = torch.autograd.grad(outs, leaves, douts)

    ref = _leaf64(wide, c, M, bias, *adds)
    w64, c64, M64, b64, *ad64 = ref
    x = torch.cat([w64[:, 2:9], w64[:, 11:20], c64], dim=1)
    o = torch.einsum("roc,rgcnt->rgont", M64, x.view(2, 2, 21, N, T)).reshape(G, 30, N, T) + b64.view(1, 30, 1, 1)
    o = o + torch.cat(ad64, dim=1)
    o_split = torch.split(o, [4, 16, 10], dim=1)
    for mine, want in zip(outs, o_split):
        assert rel_err(mine.double(), want) < TOL
    g_ref = torch.autograd.grad(o_split, ref, [d.double() for d in douts])
    _check(g_mine, g_ref, ["dwide", "dc", "dM", "dbias", "dadd0", "dadd1"])


def test_mix_multi_six_segments_and_limits():
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(22)
    G, N, T = 2, 11, 12
    ins = [_rand(gen, G, 3, N, T) for _ in range(6)]
    M = _rand(gen, 1, 8, 18, scale=0.3)
    (out,) = ops.mix_multi(ins, M)
    want = torch.einsum("oc,gcnt->gont", M[0].double(), torch.cat(ins, 1).double())
    assert rel_err(out.double(), want) < TOL
    with pytest.raises(ValueError):
        ops.mix_multi(ins + [ins[0]], torch.cat([M, M[:, :, :3]], dim=2))
    # few output channels: the VALU kernel's segmented form, two outputs, with its backward
    M4 = _rand(gen, 1, 4, 18, scale=0.3).requires_grad_(True)
    o1, o2 = ops.mix_multi(ins, M4, out_channels=[1, 3])
    want = torch.einsum("oc,gcnt->gont", M4[0].detach().double(), torch.cat(ins, 1).double())
    assert rel_err(torch.cat([o1, o2], 1).double(), want) < TOL
    d1, d2 = _rand(gen, *o1.shape), _rand(gen, *o2.shape)
    (dM,) = torch.autograd.grad([o1, o2], [M4], [d1, d2])
    want_dM = torch.einsum("gont,gcnt->oc", torch.cat([d1, d2], 1).double(), torch.cat(ins, 1).double())
    assert rel_err(dM[0].double(), want_dM) < TOL


def test_attention_core_equals_the_graph_attention_on_projected_features():
    """attention_core(u, q) on u = W x, q = alpha . x is GACN (msgat.py:25-28) with the projection applied first."""
    import ms_gat_amd
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(23)
    B, C, Co, N, T = 3, 40, 8, 61, 12
    adj = ms_gat_amd.synthetic_adjacency(N, 80, seed=3)
    x, W, alpha, Wg = _rand(gen, B, C, N, T), _rand(gen, Co, C, scale=0.2), _rand(gen, C, scale=0.2), _rand(gen, T, T, scale=0.3)
    dz = _rand(gen, B, Co, N, T)
    xa, Wa, aa, ga = _leaf(x, W, alpha, Wg)
    u = torch.einsum("oc,bcnt->bont", Wa, xa)
    q = torch.einsum("c,bcnt->bnt", aa, xa)
    z = ops.attention_core(u, q, ga.unsqueeze(0), adj.to(_dev()))
    g_mine = torch.autograd.grad(z, [xa, Wa, aa, ga], dz)
    xb, Wb, ab, gb = _leaf(x, W, alpha, Wg)
    z2 = ops.gacn(xb, ab.unsqueeze(0), gb.unsqueeze(0), Wb.unsqueeze(0), adj.to(_dev()))
    g_ref = torch.autograd.grad(z2, [xb, Wb, ab, gb], dz)
    assert rel_err(z, z2) < TOL
    _check(g_mine, [g.double() for g in g_ref], ["dx", "dW", "dalpha", "dWg"])


class _GradArrivesAsSlice(torch.autograd.Function):
    """Identity whose gradient reaches the producer as the channel slice [:, a:b] of a wider tensor -- what the residual
    tail of a MEAM block hands to its three branches (msgat.py:130, one dout for cat(cacn, tacn, gacn))."""

    @staticmethod
    def forward(ctx, z):
        return z.clone()

    @staticmethod
    def backward(ctx, g):
        G, Ck, N, T = g.shape
        wide = torch.full((G, 3 * Ck + 5, N, T), float("nan"), device=g.device)   # neighbours must never be read
        wide[:, Ck + 2:2 * Ck + 2] = g
        return wide[:, Ck + 2:2 * Ck + 2]


@pytest.mark.parametrize("B,C,Co,N,edges", [(3, 1, 24, 61, 80), (2, 3, 24, 50, 60), (2, 6, 24, 40, 50),
                                           (3, 72, 24, 61, 80), (2, 72, 24, 883, 866), (2, 40, 8, 2000, 2500)])
def test_gradient_channel_slices_are_read_in_place(B, C, Co, N, edges):
    """GACN and the attention core give bit-identical gradients whether dz arrives contiguous or as a channel slice of a
    wider tensor (read in place where the library accepts it -- AGG_FIRST, fused PROJ_FIRST -- copied otherwise: N = 2000)."""
    import ms_gat_amd
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(29)
    T = 12
    adj = ms_gat_amd.synthetic_adjacency(N, edges, seed=4).to(_dev())
    x, W, alpha, Wg = _rand(gen, B, C, N, T), _rand(gen, Co, C, scale=0.2), _rand(gen, C, scale=0.2), _rand(gen, T, T, scale=0.3)
    dz = _rand(gen, B, Co, N, T)

    def grads(sliced):
        xa, Wa, aa, ga = _leaf(x, W, alpha, Wg)
        z = ops.gacn(xa, aa.unsqueeze(0), ga.unsqueeze(0), Wa.unsqueeze(0), adj)
        z = _GradArrivesAsSlice.apply(z) if sliced else z
        return torch.autograd.grad(z, [xa, Wa, aa, ga], dz)

    for a, b in zip(grads(True), grads(False)):
        assert torch.equal(a, b)
    if C > Co:
        def core(sliced):
            xa, Wa, aa, ga = _leaf(x, W, alpha, Wg)
            u = torch.einsum("oc,bcnt->bont", Wa, xa)
            q = torch.einsum("c,bcnt->bnt", aa, xa)
            z = ops.attention_core(u, q, ga.unsqueeze(0), adj)
            z = _GradArrivesAsSlice.apply(z) if sliced else z
            return torch.autograd.grad(z, [xa, Wa, aa, ga], dz)
        for a, b in zip(core(True), core(False)):
            assert torch.equal(a, b)


def test_parameter_sets_per_relation_in_one_launch():
    """R parameter sets evaluated in one launch (the relation count of include/msgat_hip.h) equal R separate calls:
    LayerNorm, node pooling, channel pooling, time mixing bias and the prediction head, forward and gradients."""
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(31)
    R, B, C, N, T, To, Co = 3, 2, 5, 19, 12, 12, 4
    x = _rand(gen, R * B, C, N, T)
    lw, lb = _rand(gen, R, T) + 1.0, _rand(gen, R, T)
    nw, ca = _rand(gen, R, N), _rand(gen, R, C)
    hw, hb = _rand(gen, R, To, T, 1, C, scale=0.2), _rand(gen, R, To)
    A, tb = _rand(gen, R * B, 1, T, T, scale=0.3), _rand(gen, R, C)

    def run(stacked):
        leaves = _leaf(x, lw, lb, nw, ca, hw, hb, A, tb)
        x_, lw_, lb_, nw_, ca_, hw_, hb_, A_, tb_ = leaves
        if stacked:
            y = ops.layer_norm_t(x_, lw_, lb_)
            outs = [y, ops.node_pool(y, nw_), ops.channel_pool(y, ca_), ops.head(y, hw_, hb_), ops.time_mix(y, A_, tb_)]
        else:
            parts = [[], [], [], [], []]
            for r in range(R):
                sl = slice(r * B, (r + 1) * B)
                y = ops.layer_norm_t(x_[sl], lw_[r], lb_[r])
                for k, v in enumerate((y, ops.node_pool(y, nw_[r]), ops.channel_pool(y, ca_[r]), ops.head(y, hw_[r], hb_[r]),
                                       ops.time_mix(y, A_[sl], tb_[r]))):
                    parts[k].append(v)
            outs = [torch.cat(p, dim=0) for p in parts]
        gen2 = torch.Generator().manual_seed(77)
        douts = [_rand(gen2, *o.shape) for o in outs]
        return outs, torch.autograd.grad(outs, leaves, douts)

    (o1, g1), (o2, g2) = run(True), run(False)
    for a, b in zip(o1, o2):
        assert rel_err(a, b) < 1e-5
    for name, a, b in zip(["dx", "dlnw", "dlnb", "dnw", "dca", "dhw", "dhb", "dA", "dtb"], g1, g2):
        assert rel_err(a, b) < 2e-5, name


@pytest.mark.parametrize("R,Bg,C,cb,T", [(1, 3, 7, 4, 12), (3, 2, 72, 24, 12), (2, 2, 1, 24, 12), (1, 2, 96, 32, 8), (1, 1, 5, 3, 16)])
def test_channel_attention_mix_is_one_launch_each_way(R, Bg, C, cb, T):
    """conv @ softmax((p Wc) p^T) -- attention.py:90-92 folded with CACN's convolution weight (msgat.py:93-94) -- and
    every gradient, against the same chain of dense ops in float64."""
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(C * 7 + T)
    p, Wc, conv, dM = (_rand(gen, R * Bg, C, T, scale=0.7), _rand(gen, R, T, T, scale=0.4), _rand(gen, R, cb, C, scale=0.5),
                       _rand(gen, R * Bg, cb, C))
    a = _leaf(p, Wc, conv)
    ours = _grads(ops.channel_attention_mix(*a), a, dM)
    p64, W64, c64 = _leaf64(p, Wc, conv)
    pv = p64.view(R, Bg, C, T)
    att = torch.softmax(pv @ W64.unsqueeze(1) @ pv.transpose(2, 3), dim=-1)
    ref = (c64.unsqueeze(1) @ att).reshape(R * Bg, cb, C)
    _check(ours, _grads(ref, (p64, W64, c64), dM.double()), ["Mc", "dpooled", "dWc", "dconv"])


@pytest.mark.parametrize("R,Bg,N,K,T,dil", [(1, 3, 23, 10, 12, 1), (3, 2, 883, 10, 12, 2), (2, 1, 64, 10, 8, 4), (1, 2, 40, 10, 12, 0),
                                            (1, 1, 17, 10, 12, 12)])
def test_temporal_attention_taps_are_one_launch_each_way(R, Bg, N, K, T, dil):
    """(att shifted down by the dilation, att) with att = softmax((q^T Wt1^T)(q^T Wt2^T)^T) -- attention.py:60-64 as the
    taps of TACN's first convolution (msgat.py:66-74) -- and every gradient, against dense float64 ops."""
    from ms_gat_amd import ops
    gen = torch.Generator().manual_seed(N + T + dil)
    q, W1, W2, dtaps = (_rand(gen, R * Bg, N, T, scale=0.5), _rand(gen, R, K, N, scale=N ** -0.5), _rand(gen, R, K, N, scale=N ** -0.5),
                        _rand(gen, R * Bg, 2, T, T))
    a = _leaf(q, W1, W2)
    ours = _grads(ops.temporal_attention_taps(*a, dil), a, dtaps)
    q64, A64, B64 = _leaf64(q, W1, W2)
    per_t = q64.view(R, Bg, N, T).transpose(2, 3)
    att = torch.softmax((per_t @ A64.transpose(1, 2).unsqueeze(1)) @ (per_t @ B64.transpose(1, 2).unsqueeze(1)).transpose(2, 3), dim=-1)
    att = att.reshape(R * Bg, T, T)
    shifted = torch.zeros_like(att) if dil >= T else torch.nn.functional.pad(att[:, : T - dil], (0, 0, dil, 0))
    ref = torch.stack([shifted, att], dim=1)
    _check(ours, _grads(ref, (q64, A64, B64), dtaps.double()), ["taps", "dpooled", "dWt1", "dWt2"])
    assert ops.causal_shift_taps(T, 2, _dev()) is ops.causal_shift_taps(T, 2, _dev())
    eye = torch.eye(T)
    assert torch.equal(ops.causal_shift_taps(T, 2, _dev()).cpu()[0, 0], torch.nn.functional.pad(eye[: T - 2], (0, 0, 2, 0)))


# ---- channel-pair contraction (the weight gradients of the channel mixings) through the C ABI ------------------------
# (Ca segments, Cb, ones, N): every staging form of launch_chanpair_mfma -- LDS-DMA rings of 128- and 64-position tiles
# ([32 x 80], [80 x 80], [112 x 80] channel blocks), the register-staged kernel (narrow B, short rows) -- with rows whose
# length is no multiple of the tile (N T = 10596 = 82 x 128 + 100 = 165 x 64 + 36; 13 x 12 = 156) and channel counts
# that leave rows of the last 16-row fragment and whole row groups empty.
@pytest.mark.parametrize("segs,Cb,ones,N,Bg", [
    ((24, 1), 72, 0, 883, 2),             # dW | dalpha of the GACN projection: k_chanpair_glds<2,5,128,3>
    ((24, 48, 24, 1, 1), 72, 1, 883, 2),  # merged channel mixing of a MEAM block (98 x 73): <7,5,64,3>
    ((72,), 72, 1, 883, 1),               # residual convolution (72 x 73): <5,5,64,3>
    ((17,), 65, 0, 307, 3),               # lower edges of the [32 x 80] form
    ((81,), 79, 1, 64, 2),                # P = 768: six tiles per group; 81 rows = five full fragments + 1
    ((24, 1), 72, 0, 13, 2),              # P = 156 < 512: register-staged kernel
    ((48,), 24, 1, 883, 2),               # narrow B: register-staged kernel
    ((16, 1), 48, 0, 883, 2),             # msgat48: <2,3,128,3>
    ((16, 32, 16, 1, 1), 48, 1, 883, 2),  # msgat48 merged mixing (66 x 49): <5,4,64,3>
    ((48,), 48, 1, 307, 2),               # msgat48 residual convolution (48 x 49): <3,4,128,3>
    ((32, 1), 96, 0, 883, 2),             # msgat96: <3,6,64,3>
    ((32, 64, 32, 1, 1), 96, 1, 307, 2),  # msgat96 merged mixing (130 x 97): <9,4,64,3>, two z-blocks of 49 + 48 columns
    ((96,), 96, 1, 883, 1),               # msgat96 residual convolution (96 x 97): <6,4,64,3>, two z-blocks over B
    ((24, 1), 96, 0, 13, 2),              # P = 156, 96 columns: register-staged [32 x 80] blocks, two z-blocks over B
    ((40,), 90, 1, 20, 3),                # P = 240: register-staged [48 x 64] blocks
])
def test_contract_segments_matches_float64_and_repeats_bit_for_bit(segs, Cb, ones, N, Bg):
    import ctypes as C
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev, R, T = _dev(), 2, 12
    G, Ca = R * Bg, sum(segs)
    g = torch.Generator().manual_seed(11)
    # every segment is a channel slice of a wider tensor (group stride > channels), as the model passes them
    wide = [torch.randn(G, c + 3, N, T, generator=g).to(dev) for c in segs]
    B = torch.randn(G, Cb, N, T, generator=g).to(dev)
    arr = (_lib.Seg * len(segs))()
    for i, (t, c) in enumerate(zip(wide, segs)):
        arr[i] = _lib.Seg(t[:, 1:].data_ptr(), c, c + 3)
    nfl = int(L.msgat_contract_segments_partial_floats(R, Ca, Cb + ones))
    part = torch.empty(max(nfl, 1), device=dev)
    outs = []
    for rep in range(4):   # a race between the LDS-DMA ring and its readers would differ from run to run
        part.fill_(float("nan"))
        dst = torch.full((R, Ca, Cb + ones), float("nan"), device=dev)
        st = L.msgat_contract_segments(R, Bg, N, T, arr, len(segs), B.data_ptr(), Cb, ones, part.data_ptr(),
                                       dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "msgat_contract_segments")
        outs.append(dst)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    A64 = torch.cat([t[:, 1:1 + c] for t, c in zip(wide, segs)], dim=1).double().view(R, Bg, Ca, N * T)
    B64 = B.double().view(R, Bg, Cb, N * T)
    if ones:
        B64 = torch.cat([B64, torch.ones(R, Bg, 1, N * T, device=dev, dtype=torch.float64)], dim=2)
    want = torch.einsum("rgap,rgcp->rac", A64, B64)
    err = rel_err(outs[0], want)
    record_err(f"contract_segments {segs}x{Cb}+{ones} N={N}", "dst", err, 1e-5)
    assert err < 1e-5


@pytest.mark.parametrize("segs,Cb,ones,N,Bg", [
    ((24, 48, 24, 1, 1), 72, 1, 883, 2),  # merged channel mixing of a MEAM block: SPLIT form <7,5>
    ((72,), 72, 1, 883, 1),               # residual convolution: SPLIT form <5,5>
    ((49,), 65, 0, 307, 3),               # lower edges of the <5,5> form, no bias column
    ((81,), 79, 1, 64, 2),                # lower edge of <7,5>; P = 768
    ((24,), 72, 0, 883, 2),               # <2,5,128,3,1> (every wave contracts and mixes), plain matrix
    ((72,), 72, 1, 13, 2),                # P = 156: the two passes
    ((16, 32, 16, 1, 1), 48, 1, 883, 2),  # msgat48 merged mixing (66 x 49): <5,4,64,3,2>
    ((48,), 48, 1, 883, 1),               # msgat48 residual convolution (48 x 49): <3,4,128,3,1>
    ((33,), 40, 0, 307, 2),               # lower edge of <3,4>, B of 40 channels
    ((32, 64, 32, 1, 1), 96, 1, 883, 1),  # msgat96 merged mixing (130 x 97): <9,4,64,3,2>, z-blocks over B with all of A
    ((96,), 96, 1, 883, 1),               # msgat96 residual convolution (96 x 97): <6,4,64,3,2>, two z-blocks
    ((96,), 96, 1, 64, 3),                # the same at P = 768
    ((33,), 96, 0, 307, 2),               # <3,6,64,3,2>, no bias column
    ((16, 32, 16, 1, 1), 48, 0, 883, 2),  # msgat48 merged mixing WITHOUT a bias column (the stacked schedule): a block one
    ((48,), 48, 0, 307, 2),               #   tile wider than its 48 columns, <5,4,64,3,2> / <3,4,128,3,1>
    ((32, 64, 32, 1, 1), 96, 0, 883, 1),  # msgat96 likewise: <9,4,64,3,2>, two z-blocks of 48 columns in 64-wide blocks
    ((96,), 96, 0, 64, 2),                #   <6,4,64,3,2>
    ((72,), 1, 1, 883, 2),                # a convolution with ONE input channel (the first block of a PEMSD7 component) and a
    ((48,), 3, 1, 307, 2),                #   bias, three channels (PEMSD4), no bias column, a short row: k_aggfirst_bwd<C, ONES>
    ((24,), 1, 0, 64, 3),
    ((72,), 3, 0, 13, 2),
    ((40,), 2, 1, 50, 1),
])
def test_contract_mix_segments_gives_matrix_bias_and_input_gradients_in_one_pass(segs, Cb, ones, N, Bg):
    """msgat_contract_mix_segments = the backward of y = M x (+ bias): dM (| dbias) AND dx from one pass over dy and x,
    against float64; four repeats must agree bit for bit (the dx stores count on vmcnt next to the LDS-DMA loads: a
    miscounted wait would read a tile that has not landed, differently from run to run)."""
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev, R, T = _dev(), 2, 12
    G, Ca = R * Bg, sum(segs)
    g = torch.Generator().manual_seed(13)
    wide = [torch.randn(G, c + 2, N, T, generator=g).to(dev) for c in segs]
    B = torch.randn(G, Cb, N, T, generator=g).to(dev)
    M = (torch.randn(R, Ca, Cb, generator=g) * 0.2).to(dev)
    arr = (_lib.Seg * len(segs))()
    for i, (t, c) in enumerate(zip(wide, segs)):
        arr[i] = _lib.Seg(t[:, 2:].data_ptr(), c, c + 2)
    part = torch.empty(max(int(L.msgat_contract_mix_partial_floats(R, Bg, N, T, Ca, Cb + ones)), 1), device=dev)
    outs = []
    for rep in range(4):
        part.fill_(float("nan"))
        dst = torch.full((R, Ca, Cb + ones), float("nan"), device=dev)
        dx = torch.full((G, Cb, N, T), float("nan"), device=dev)
        st = L.msgat_contract_mix_segments(R, Bg, N, T, arr, len(segs), B.data_ptr(), Cb, ones, M.data_ptr(),
                                           part.data_ptr(), dst.data_ptr(), dx.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "msgat_contract_mix_segments")
        outs.append((dst, dx))
    torch.cuda.synchronize()
    for dst, dx in outs[1:]:
        assert torch.equal(dst, outs[0][0]) and torch.equal(dx, outs[0][1])
    A64 = torch.cat([t[:, 2:2 + c] for t, c in zip(wide, segs)], dim=1).double().view(R, Bg, Ca, N * T)
    B64 = B.double().view(R, Bg, Cb, N * T)
    want_dx = torch.einsum("rac,rgap->rgcp", M.double(), A64).reshape(G, Cb, N, T)
    if ones:
        B64 = torch.cat([B64, torch.ones(R, Bg, 1, N * T, device=dev, dtype=torch.float64)], dim=2)
    want = torch.einsum("rgap,rgcp->rac", A64, B64)
    what = f"contract_mix_segments {segs}x{Cb}+{ones} N={N}"
    for key, got, ref in (("dM", outs[0][0], want), ("dx", outs[0][1], want_dx)):
        err = rel_err(got, ref)
        record_err(what, key, err, 1e-5)
        assert err < 1e-5, key


@pytest.mark.gpu
@pytest.mark.parametrize("R,segs,Cb,N", [(96, (24, 48, 24, 1, 1), 72, 883), (40, (72,), 72, 307), (160, (24,), 72, 64), (7, (16, 32, 16, 1, 1), 48, 883)])
def test_contract_mix_with_one_matrix_per_group(R, segs, Cb, N):
    """Every group its own relation (a per-sample matrix: CACN's rows in the merged channel mixing, R = B x components):
    the block count per relation then comes from the fill of the CU rounds (chanpair_mfma_blocks), several blocks per
    group and several rounds per launch.  dM and dx against float64; two runs agree bit for bit."""
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev, T, Bg = _dev(), 12, 1
    G, Ca = R, sum(segs)
    g = torch.Generator().manual_seed(R + N)
    tens = [torch.randn(G, c, N, T, generator=g).to(dev) for c in segs]
    B = torch.randn(G, Cb, N, T, generator=g).to(dev)
    M = (torch.randn(R, Ca, Cb, generator=g) * 0.2).to(dev)
    arr = (_lib.Seg * len(segs))(*[_lib.Seg(t.data_ptr(), c, 0) for t, c in zip(tens, segs)])
    part = torch.empty(max(int(L.msgat_contract_mix_partial_floats(R, Bg, N, T, Ca, Cb)), 1), device=dev)
    outs = []
    for rep in range(2):
        part.fill_(float("nan"))
        dst = torch.full((R, Ca, Cb), float("nan"), device=dev)
        dx = torch.full((G, Cb, N, T), float("nan"), device=dev)
        _lib.check(L.msgat_contract_mix_segments(R, Bg, N, T, arr, len(segs), B.data_ptr(), Cb, 0, M.data_ptr(), part.data_ptr(),
                                                 dst.data_ptr(), dx.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "msgat_contract_mix_segments")
        outs.append((dst, dx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    A64 = torch.cat(tens, dim=1).double().view(R, Ca, N * T)
    want = torch.einsum("rap,rcp->rac", A64, B.double().view(R, Cb, N * T))
    want_dx = torch.einsum("rac,rap->rcp", M.double(), A64).reshape(G, Cb, N, T)
    for key, got, ref in (("dM", outs[0][0], want), ("dx", outs[0][1], want_dx)):
        err = rel_err(got, ref)
        record_err(f"contract_mix per-group R={R} {segs}x{Cb} N={N}", key, err, 1e-5)
        assert err < 1e-5, key


@pytest.mark.gpu
@pytest.mark.parametrize("segs,Cb,N,Bg,mix", [((72,), 72, 883, 1, True), ((24, 48, 24, 1, 1), 72, 883, 2, False), ((72,), 3, 307, 2, True),
                                             ((24,), 1, 50, 3, True), ((72,), 72, 13, 2, True), ((33,), 40, 307, 2, False)])
def test_with_ones_2_delivers_the_bias_column_apart(segs, Cb, N, Bg, mix):
    """with_ones = 2 (msgat_contract_segments, msgat_contract_mix_segments): the sums of with_ones = 1, bit for bit, as the
    plain [R,Ca,Cb] matrix followed by the ones column [R,Ca] -- every form of the reduction behind the two entry points
    (fused one-pass, two passes, the few-input-channel kernel)."""
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev, R, T = _dev(), 2, 12
    G, Ca = R * Bg, sum(segs)
    g = torch.Generator().manual_seed(17)
    tens = [torch.randn(G, c, N, T, generator=g).to(dev) for c in segs]
    B = torch.randn(G, Cb, N, T, generator=g).to(dev)
    M = (torch.randn(R, Ca, Cb, generator=g) * 0.2).to(dev)
    arr = (_lib.Seg * len(segs))(*[_lib.Seg(t.data_ptr(), c, 0) for t, c in zip(tens, segs)])
    nfl = (L.msgat_contract_mix_partial_floats(R, Bg, N, T, Ca, Cb + 1) if mix else L.msgat_contract_segments_partial_floats(R, Ca, Cb + 1))
    part = torch.empty(max(int(nfl), 1), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    got = {}
    for ones in (1, 2):
        dst = torch.full((R * Ca * (Cb + 1),), float("nan"), device=dev)
        dx = torch.full((G, Cb, N, T), float("nan"), device=dev)
        if mix:
            _lib.check(L.msgat_contract_mix_segments(R, Bg, N, T, arr, len(segs), B.data_ptr(), Cb, ones, M.data_ptr(), part.data_ptr(),
                                                     dst.data_ptr(), dx.data_ptr(), st), "msgat_contract_mix_segments")
        else:
            _lib.check(L.msgat_contract_segments(R, Bg, N, T, arr, len(segs), B.data_ptr(), Cb, ones, part.data_ptr(), dst.data_ptr(),
                                                 st), "msgat_contract_segments")
        got[ones] = (dst, dx)
    inside = got[1][0].view(R, Ca, Cb + 1)
    assert torch.equal(got[2][0][: R * Ca * Cb].view(R, Ca, Cb), inside[:, :, :Cb])
    assert torch.equal(got[2][0][R * Ca * Cb:].view(R, Ca), inside[:, :, Cb])
    if mix:
        assert torch.equal(got[1][1], got[2][1])
    want = torch.cat(tens, dim=1).double().view(R, Bg, Ca, N * T).sum(dim=(1, 3))
    assert rel_err(got[2][0][R * Ca * Cb:].view(R, Ca), want) < 1e-5


@pytest.mark.parametrize("R,Bg,C,Co,N", [(3, 2, 72, 24, 883), (2, 2, 72, 24, 13), (1, 3, 48, 16, 307), (2, 1, 80, 31, 64),
                                         (3, 2, 48, 16, 883), (3, 2, 96, 32, 883), (1, 2, 64, 20, 64), (2, 1, 90, 40, 307)])
def test_stage_project_backward_matches_float64(R, Bg, C, Co, N):
    """msgat_stage_project_backward: dW = du x^T, dalpha = dq . x, dx = W^T du + alpha (x) dq -- the fused one-pass
    form (C in 65..80, Co + 1 in 17..32, rows of >= 512 positions) and the two-pass form behind the same entry point."""
    import ctypes as C_
    from ms_gat_amd import _lib
    L = _lib.lib()
    dev, T = _dev(), 12
    G = R * Bg
    g = torch.Generator().manual_seed(17)
    du, dq = torch.randn(G, Co, N, T, generator=g).to(dev), torch.randn(G, N, T, generator=g).to(dev)
    x = torch.randn(G, C, N, T, generator=g).to(dev)
    W, alpha = (torch.randn(R, Co, C, generator=g) * 0.2).to(dev), (torch.randn(R, C, generator=g) * 0.2).to(dev)
    shape = _lib.Shape(R, Bg, C, Co, N, T)
    part = torch.empty(int(L.msgat_contract_partial_floats(C_.byref(shape), Co + 1, C)), device=dev)
    res = []
    for rep in range(3):
        dW, da, dx = (torch.full(s, float("nan"), device=dev) for s in ((R, Co, C), (R, C), (G, C, N, T)))
        st = L.msgat_stage_project_backward(C_.byref(shape), du.data_ptr(), dq.data_ptr(), x.data_ptr(), W.data_ptr(),
                                            alpha.data_ptr(), part.data_ptr(), dW.data_ptr(), da.data_ptr(), dx.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "msgat_stage_project_backward")
        res.append((dW, da, dx))
    torch.cuda.synchronize()
    for r_ in res[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r_, res[0]))
    du64, dq64, x64 = (t.double().view(R, Bg, -1, N * T) for t in (du, dq.unsqueeze(1), x))
    want = (torch.einsum("rgop,rgcp->roc", du64, x64), torch.einsum("rgop,rgcp->rc", dq64, x64),
            (torch.einsum("roc,rgop->rgcp", W.double(), du64)
             + torch.einsum("rc,rgop->rgcp", alpha.double(), dq64)).reshape(G, C, N, T))
    for key, got, ref in zip(("dW", "dalpha", "dx"), res[0], want):
        err = rel_err(got, ref)
        record_err(f"stage_project_backward R={R} C={C} Co={Co} N={N}", key, err, 1e-5)
        assert err < 1e-5, key
