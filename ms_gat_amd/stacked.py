"""All components of an MS-GAT model in one launch sequence.

The reference evaluates its components (one `TPC` per input period, src/models/msgat.py:191-199) in a Python
loop (msgat.py:204): R independent networks of identical architecture over the same adjacency, each on
B samples.  On an MI355X a single component at B = 32 does not fill the machine -- its dense score kernel
launches 224 blocks for 768 block slots -- and every one of its ~300 kernel launches is paid R times.  Here
the components ride on the leading axis instead: activations are [R*B, C, N, T], relation-major, and every
parameter gets a leading [R] axis (`torch.stack` of the R modules' tensors, so autograd hands each module its
own gradient and `state_dict` keeps the reference's per-component keys).  The library's entry points take the
relation count (include/msgat_hip.h), exactly like the graph attention has from the start (`StackedGACN`).

`forward(model, X, H, D)` computes what `MSGAT.forward`'s loop computes, block for block:
LayerNorm -> {channel, temporal, graph} branches -> residual tail (msgat.py:117-131), twice or more per
component, then LayerNorm -> prediction head (msgat.py:158-160), gate and sum over components (msgat.py:203-204).
"""
from __future__ import annotations

from typing import Callable, List

import torch

from . import ops


def _stack(mods: List, get: Callable) -> torch.Tensor:
    return torch.stack([get(m) for m in mods])


def _per_group(p: torch.Tensor, B: int) -> torch.Tensor:
    """[R, ...] -> [R*B, ...]: each relation's tensor repeated for its B samples."""
    return p.unsqueeze(1).expand(p.shape[0], B, *p.shape[1:]).reshape(p.shape[0] * B, *p.shape[1:])


def _shift_down(att: torch.Tensor, d: int) -> torch.Tensor:
    T = att.size(-2)
    if d >= T:
        return torch.zeros_like(att)
    return torch.nn.functional.pad(att[..., : T - d, :], (0, 0, d, 0))


def can_stack(model) -> bool:
    """Stacking needs R components of identical architecture (the reference builds them that way, msgat.py:220-229)."""
    first = model.tpcs[0]
    return all(t.channels == first.channels and t.dilations == first.dilations and
               t.in_timesteps == first.in_timesteps and t.out_timesteps == first.out_timesteps for t in model.tpcs)


def _channel_attention(cas, pooled: torch.Tensor, R: int, B: int) -> torch.Tensor:
    """pooled [G,C,T] -> att [R,B,C,C]  (attention.py:90-92)."""
    p = pooled.view(R, B, *pooled.shape[1:])
    return torch.softmax(p @ _stack(cas, lambda m: m.Wc).unsqueeze(1) @ p.transpose(2, 3), dim=-1)


def _temporal_attention(tas, pooled: torch.Tensor, R: int, B: int) -> torch.Tensor:
    """pooled [G,N,T] (alpha-weighted channel sum) -> att [G,T,T]  (attention.py:60-64)."""
    per_t = pooled.view(R, B, *pooled.shape[1:]).transpose(2, 3)                      # [R,B,T,N]
    left = per_t @ _stack(tas, lambda m: m.Wt1).transpose(1, 2).unsqueeze(1)          # [R,B,T,10]
    right = per_t @ _stack(tas, lambda m: m.Wt2).transpose(1, 2).unsqueeze(1)
    att = torch.softmax(left @ right.transpose(2, 3), dim=-1)
    return att.reshape(R * B, att.shape[-2], att.shape[-1])


def _tacn_finish(tacns, mixed: torch.Tensor, att: torch.Tensor) -> torch.Tensor:
    """TACN from the channel-mixed input of its first convolution (see model.TACN.finish), R stacks at once."""
    T = att.size(-1)
    h = None
    for i, d in enumerate(tacns[0].dilations):
        bias = _stack(tacns, lambda m: m.seq[1 + 2 * i].bias)                          # [R,Co]
        if i == 0:
            taps = torch.stack([_shift_down(att, d), att], dim=1)                      # [G,2,T,T]
        else:
            eye = torch.eye(T, device=att.device, dtype=att.dtype)
            taps = torch.stack([_shift_down(eye, d), eye], dim=0).unsqueeze(0)         # [1,2,T,T]
            (mixed,) = ops.mix_multi([h], _stack(tacns, lambda m: m.stacked_taps(i)))
        h = ops.time_mix(mixed, taps, bias)
    return h


def _meam(meams, x: torch.Tensor, adjacency, R: int, B: int) -> torch.Tensor:
    m0 = meams[0]
    C, cb = m0.in_channels, m0.out_channels // 3
    cas = [m.cacn.seq[0] for m in meams]
    tas = [m.tacn.seq[0] for m in meams]
    tacns = [m.tacn for m in meams]
    normed = ops.layer_norm_t(x, _stack(meams, lambda m: m.ln.weight), _stack(meams, lambda m: m.ln.bias), m0.ln.eps)

    att_c = _channel_attention(cas, ops.node_pool(normed, _stack(cas, lambda m: m.alpha)), R, B)
    conv_w = _stack(meams, lambda m: m.cacn.seq[1].weight[:, :, 0, 0])                 # [R,cb,C]
    conv_b = _stack(meams, lambda m: m.cacn.seq[1].bias)                               # [R,cb]
    Mc = (conv_w.unsqueeze(1) @ att_c).reshape(R * B, cb, C)                           # per-sample matrices
    Wg = _stack(meams, lambda m: m.gacn.gatt.Wg)
    alpha_g = _stack(meams, lambda m: m.gacn.gatt.alpha)
    alpha_t = _stack(tas, lambda m: m.alpha)
    W_g = _stack(meams, lambda m: m.gacn.W)

    if C <= cb or not m0.dilations:
        (cacn,) = ops.mix_multi([normed], Mc, _per_group(conv_b, B))
        att_t = _temporal_attention(tas, ops.channel_pool(normed, alpha_t), R, B)
        if m0.dilations:
            (mixed,) = ops.mix_multi([normed], _stack(tacns, lambda m: m.stacked_taps(0)))
            tacn = _tacn_finish(tacns, mixed, att_t)
        else:
            tacn = ops.time_mix(normed, att_t.unsqueeze(1))
        gacn = ops.gacn(normed, alpha_g, Wg, W_g, adjacency)
    else:
        # every channel mixing of the normalised input as row blocks of one per-sample matrix (model.MEAM._merged_branches)
        rows = torch.cat([Mc, _per_group(_stack(tacns, lambda m: m.stacked_taps(0)), B), _per_group(W_g, B),
                          _per_group(alpha_g.unsqueeze(1), B), _per_group(alpha_t.unsqueeze(1), B)], dim=1)
        bias = _per_group(torch.cat([conv_b, conv_b.new_zeros(R, 3 * cb + 2)], dim=1), B)
        cacn, mixed, u, q, pooled_t = ops.mix_multi([normed], rows, bias, out_channels=[cb, 2 * cb, cb, 1, 1])
        tacn = _tacn_finish(tacns, mixed, _temporal_attention(tas, pooled_t[:, 0], R, B))
        gacn = ops.attention_core(u, q[:, 0], Wg, adjacency)
    return ops.mix_multi([x], _stack(meams, lambda m: m.res.weight[:, :, 0, 0]), _stack(meams, lambda m: m.res.bias),
                         adds=[cacn, tacn, gacn], relu=True)[0]


def forward(model, X: torch.Tensor, H: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """X [B,R,C,N,T], H [B], D [B] -> [B,N,T_out]: `MSGAT.forward` with all R components in each kernel."""
    tpcs = list(model.tpcs)
    R, B = len(tpcs), X.shape[0]
    x = X.transpose(0, 1).reshape(R * B, *X.shape[2:])                                 # relation-major groups
    for level in range(len(tpcs[0].tgacns)):
        x = _meam([t.tgacns[level] for t in tpcs], x, model.adj, R, B)
    xn = ops.layer_norm_t(x, _stack(tpcs, lambda t: t.ln.weight), _stack(tpcs, lambda t: t.ln.bias), tpcs[0].ln.eps)
    pred = ops.head(xn, _stack(tpcs, lambda t: t.fc.weight), _stack(tpcs, lambda t: t.fc.bias))     # [R*B,N,T_out]
    pred = pred.view(R, B, *pred.shape[1:])
    gate = model.te(H, D).transpose(0, 1) if model.te is not None else model.W.unsqueeze(1)          # [R,B|1,N,T_out]
    return (pred * gate).sum(dim=0)
