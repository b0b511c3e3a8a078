"""All components of an MS-GAT model in one launch sequence.

The reference evaluates its components (one `TPC` per input period, src/models/msgat.py:191-199) in a Python
loop (msgat.py:204): R independent networks of identical architecture over the same adjacency, each on
B samples.  On an MI355X a single component at B = 32 does not fill the machine -- its dense score kernel
launches 224 blocks for 768 block slots -- and every one of its ~300 kernel launches is paid R times.  Here
the components ride on the leading axis instead: activations are [R*B, C, N, T], relation-major, and every
parameter gets a leading [R] axis -- the R modules' tensors stored back to back (`ParamBank`), so autograd hands
each module its own gradient and `state_dict` keeps the reference's per-component keys.  The library's entry
points take the relation count (include/msgat_hip.h), exactly like the graph attention has from the start
(`StackedGACN`).

`forward(model, X, H, D)` computes what `MSGAT.forward`'s loop computes, block for block:
LayerNorm -> {channel, temporal, graph} branches -> residual tail (msgat.py:117-131), twice or more per
component, then LayerNorm -> prediction head (msgat.py:158-160), gate and sum over components (msgat.py:203-204).
"""
from __future__ import annotations

import torch

from . import ops


class _StackedView(torch.autograd.Function):
    """The bank's [R, ...] storage as a differentiable function of the R parameters that are views of it: no kernel
    forward (the values are already there), no kernel backward (each parameter's gradient is row r of the incoming
    gradient, handed over as a view)."""

    @staticmethod
    def forward(ctx, storage, *params):
        ctx.R = len(params)
        return storage.view(storage.shape)

    @staticmethod
    def backward(ctx, grad):
        return (None, *grad.unbind(0))


class _StackedViewAll(torch.autograd.Function):
    """`_StackedView` for ALL parameters of the components in one autograd node: forward(n, storage_1..n, then the R
    parameters of each) -> the n storages as differentiable tensors.  One Function call per forward pass instead of one per
    parameter access (40 per training step: ~0.7 ms of host time forward + backward at R = 3)."""

    @staticmethod
    def forward(ctx, n, *args):
        ctx.n = n
        ctx.R = (len(args) - n) // n
        return tuple(s.view(s.shape) for s in args[:n])

    @staticmethod
    def backward(ctx, *grads):
        out = [None] * (1 + ctx.n)
        for g in grads:
            out.extend(g.unbind(0) if g is not None else [None] * ctx.R)
        return tuple(out)


class ParamBank:
    """Per-component parameters of identical shape stored back to back, so that "the same parameter of all R
    components" is one [R, ...] tensor without a stacking kernel.

    Each component keeps its own `nn.Parameter` objects (same names, `state_dict`, optimizer state); their `.data`
    are re-pointed to rows of the bank's storage.  `model.to(...)` or anything else that re-allocates parameter
    storage breaks the aliasing; `get` notices (pointer check) and rebuilds the row block, which costs kernels
    once.  Stacking by `torch.stack` cost ~46 concatenations per step forward and ~140 slice copies backward."""

    def __init__(self, tpcs):
        self.tpcs = list(tpcs)
        self.params = [dict(t.named_parameters()) for t in self.tpcs]
        self.storage = {}
        self.rows = {}          # name -> (parameter objects, their expected addresses = the rows of the storage)

    def get(self, name: str) -> torch.Tensor:
        hit = self.rows.get(name)
        if hit is not None:
            ps, ptrs = hit
            # the aliasing check by address only: `st[r].data_ptr()` built R views per access, 40 accesses per step --
            # 0.6 ms of host time per training step at R = 3 (tools/host_overhead_train.py)
            ok = all(p.data_ptr() == a for p, a in zip(ps, ptrs))
            st = self.storage[name]
        else:
            ps = [d[name] for d in self.params]
            st, ok = None, False
        if not ok:
            if ps[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("parameter bank must be built before a HIP-graph capture (run one eager forward)")
            ps = [d[name] for d in self.params]
            with torch.no_grad():
                st = torch.stack([p.detach() for p in ps]).contiguous()
                for r, p in enumerate(ps):
                    p.data = st[r]
            self.storage[name] = st
            self.rows[name] = (ps, [p.data_ptr() for p in ps])
        if any(p.requires_grad for p in ps):
            return _StackedView.apply(st, *ps)
        return st


def _all_views(bank: ParamBank):
    """name -> stacked [R, ...] tensor for every parameter of the components, through ONE autograd node."""
    names = list(bank.params[0])
    for name in names:              # builds / repairs the storage rows (no autograd work: the result is discarded)
        hit = bank.rows.get(name)
        if hit is None or not all(p.data_ptr() == a for p, a in zip(*hit)):
            with torch.no_grad():
                bank.get(name)
    storages = [bank.storage[name] for name in names]
    params = [p for name in names for p in bank.rows[name][0]]
    if not any(p.requires_grad for p in params):
        return dict(zip(names, storages))
    return dict(zip(names, _StackedViewAll.apply(len(names), *storages, *params)))


def _bank(model) -> ParamBank:
    bank = model.__dict__.get("_param_bank")
    if bank is None or len(bank.tpcs) != len(model.tpcs) or any(a is not b for a, b in zip(bank.tpcs, model.tpcs)):
        bank = model.__dict__["_param_bank"] = ParamBank(model.tpcs)
    return bank


def _per_group(p: torch.Tensor, B: int) -> torch.Tensor:
    """[R, ...] -> [R*B, ...]: each relation's tensor repeated for its B samples."""
    return p.unsqueeze(1).expand(p.shape[0], B, *p.shape[1:]).reshape(p.shape[0] * B, *p.shape[1:])


def can_stack(model) -> bool:
    """Stacking needs R components of identical architecture (the reference builds them that way, msgat.py:220-229)."""
    first = model.tpcs[0]
    return all(t.channels == first.channels and t.dilations == first.dilations and
               t.in_timesteps == first.in_timesteps and t.out_timesteps == first.out_timesteps for t in model.tpcs)


def _taps(P, prefix: str, layer: int) -> torch.Tensor:
    """[R, 2*Co, Ci]: both taps of convolution `layer` of the TACN stacks (model.TACN.stacked_taps), one copy kernel."""
    w = P(f"{prefix}tacn.seq.{1 + 2 * layer}.weight")                                   # [R,Co,Ci,1,2]
    R, Co, Ci = w.shape[:3]
    # one copy kernel each way (ops.relayout); indexing the size-1 axis with `[..., 0, :]` would cost a zero-fill + copy
    # in backward, a plain permute one strided copy per COMPONENT when the gradient reaches the parameters
    return ops.relayout(w.reshape(R, Co, Ci, 2), (0, 3, 1, 2)).reshape(R, 2 * Co, Ci)


def _first_taps(P, prefix: str, pooled: torch.Tensor, T: int, dilation: int) -> torch.Tensor:
    """[G,2,T,T] taps of TACN's first convolution from the alpha-weighted channel sums [G,N,T] (attention.py:60-64,
    msgat.py:66-74): one launch (`ops.temporal_attention_taps`)."""
    Wt1, Wt2 = P(prefix + "tacn.seq.0.Wt1"), P(prefix + "tacn.seq.0.Wt2")               # [R,K,N]
    if 2 * T * Wt1.shape[1] > 256:
        # beyond the fused kernel's lane budget (T = 16 with the rank-10 projections): the same batched torch ops as
        # model.TACN.first_taps, over all R relations at once
        R = Wt1.shape[0]
        per_t = pooled.transpose(1, 2).reshape(R, -1, T, pooled.shape[1])             # [R,B,T,N]
        left = per_t @ Wt1.transpose(1, 2).unsqueeze(1)                                # [R,B,T,K]
        right = per_t @ Wt2.transpose(1, 2).unsqueeze(1)
        att = torch.softmax(left @ right.transpose(-1, -2), dim=-1).reshape(-1, T, T)  # [G,T,T]
        shifted = (torch.zeros_like(att) if dilation >= T else
                   torch.nn.functional.pad(att[:, : T - dilation], (0, 0, dilation, 0)))
        return torch.stack([shifted, att], dim=1)
    return ops.temporal_attention_taps(pooled, Wt1, Wt2, dilation)


def _tacn_finish(P, prefix: str, dilations, mixed: torch.Tensor, taps0: torch.Tensor) -> torch.Tensor:
    """TACN from the channel-mixed input of its first convolution (see model.TACN.finish), R stacks at once."""
    T = taps0.size(-1)
    h = None
    for i, d in enumerate(dilations):
        bias = P(f"{prefix}tacn.seq.{1 + 2 * i}.bias")                                 # [R,Co]
        if i == 0:
            taps = taps0                                                               # [G,2,T,T]
        else:
            w = _taps(P, prefix, i)                                                    # [R,2Co,Ci]
            if ops.causal_conv_fused(w.shape[2], w.shape[1] // 2):
                # constant shift taps: channel mixing, the [G,2Co,N,T] intermediate and the time mixing are one pass
                h = ops.causal_conv(h, w, bias, d)
                continue
            taps = ops.causal_shift_taps(T, d, mixed.device)                           # [1,2,T,T], cached constant
            (mixed,) = ops.mix_multi([h], w)
        h = ops.time_mix(mixed, taps, bias)
    return h


def _meam(P, prefix: str, m0, x: torch.Tensor, adjacency, R: int, B: int, relu_input: bool) -> torch.Tensor:
    """One MEAM level of all R components.  Its output is a ReLU output whose backward mask the NEXT LayerNorm applies
    (`relu_input` there; here for the level below): the caller chains levels and must end with such a LayerNorm."""
    C, cb = m0.in_channels, m0.out_channels // 3
    T = x.shape[-1]
    # (normed, x, pooled_c): the residual tail below reads x again and CACN pools the normalised input over the nodes
    # (attention.py:89); both gradients join the LayerNorm's inside ONE backward kernel
    normed, x, pooled_c = ops.layer_norm_pool_tee(x, P(prefix + "ln.weight"), P(prefix + "ln.bias"), m0.ln.eps, relu_input,
                                                  P(prefix + "cacn.seq.0.alpha"))

    # CACN's per-sample channel matrix conv @ softmax(p Wc p^T) (attention.py:90-92, msgat.py:93-94): one launch
    conv_w = P(prefix + "cacn.seq.1.weight").flatten(2)                                # [R,cb,C,1,1] -> [R,cb,C]: a view
    conv_b = P(prefix + "cacn.seq.1.bias")                                             # [R,cb]
    Mc = ops.channel_attention_mix(pooled_c, P(prefix + "cacn.seq.0.Wc"), conv_w)
    Wg, alpha_g, W_g = P(prefix + "gacn.gatt.Wg"), P(prefix + "gacn.gatt.alpha"), P(prefix + "gacn.W")
    alpha_t = P(prefix + "tacn.seq.0.alpha")
    d0 = m0.dilations[0] if m0.dilations else 0

    if C <= cb or not m0.dilations:
        (cacn,) = ops.mix_multi([normed], Mc, _per_group(conv_b, B))
        pooled_t = ops.channel_pool(normed, alpha_t)
        if m0.dilations:
            (mixed,) = ops.mix_multi([normed], _taps(P, prefix, 0))
            tacn = _tacn_finish(P, prefix, m0.dilations, mixed, _first_taps(P, prefix, pooled_t, T, d0))
        else:   # bare temporal attention: K = 1 with the attention matrix itself (tap 1 of a dilation-0 pair)
            tacn = ops.time_mix(normed, _first_taps(P, prefix, pooled_t, T, 0)[:, 1:2])
        gacn = ops.gacn(normed, alpha_g, Wg, W_g, adjacency)
    else:
        # every channel mixing of the normalised input as row blocks of one per-sample matrix (model.MEAM._merged_branches):
        # the group's own CACN matrix, then its relation's taps / projection / pooling vectors.  CACN's bias joins the
        # residual bias below (both are per-channel constants added before the same ReLU).
        shared = torch.cat([_taps(P, prefix, 0), W_g, alpha_g.unsqueeze(1), alpha_t.unsqueeze(1)], dim=1)   # [R,3cb+2,C]
        rows = ops.assemble_rows(Mc, shared)
        cacn, mixed, u, q, pooled_t = ops.mix_multi([normed], rows, None, out_channels=[cb, 2 * cb, cb, 1, 1])
        # pooled_t, q: [G,1,N,T] -> [G,N,T] as views (a `[:, 0]` select costs a zero-fill + copy in backward)
        tacn = _tacn_finish(P, prefix, m0.dilations, mixed, _first_taps(P, prefix, pooled_t.flatten(1, 2), T, d0))
        gacn = ops.attention_core(u, q.flatten(1, 2), Wg, adjacency)
        tail_bias = ops.bias_join(P(prefix + "res.bias"), conv_b)
        return ops.mix_multi([x], P(prefix + "res.weight").flatten(2), tail_bias, adds=[cacn, tacn, gacn], relu=True,
                             relu_grad_premasked=True)[0]
    return ops.mix_multi([x], P(prefix + "res.weight").flatten(2), P(prefix + "res.bias"),
                         adds=[cacn, tacn, gacn], relu=True, relu_grad_premasked=True)[0]


def forward(model, X: torch.Tensor, H: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """X [B,R,C,N,T], H [B], D [B] -> [B,N,T_out]: `MSGAT.forward` with all R components in each kernel."""
    tpcs = list(model.tpcs)
    R, B = len(tpcs), X.shape[0]
    P = _all_views(_bank(model)).__getitem__
    x = X.transpose(0, 1).reshape(R * B, *X.shape[2:])                                 # relation-major groups
    for level, m0 in enumerate(tpcs[0].tgacns):
        x = _meam(P, f"tgacns.{level}.", m0, x, model.adj, R, B, relu_input=level > 0)
    pred = ops.ln_head(x, P("ln.weight"), P("ln.bias"), tpcs[0].ln.eps, P("fc.weight"), P("fc.bias"),
                       relu_input=len(tpcs[0].tgacns) > 0)                             # [R*B,N,T_out]
    pred = pred.view(R, B, *pred.shape[1:])
    # sum_r pred_r * gate_r (msgat.py:203-205), the gate built from the embedding tables inside the kernel
    if model.te is not None:
        return ops.gate_sum(pred, H, D, model.te.h_ebd.weight, model.te.d_ebd.weight)
    return ops.gate_sum(pred, None, None, model.W)
