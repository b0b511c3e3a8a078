"""The callers of the hot path: MS-GAT's blocks around `GACN`, with the reference's
`state_dict` layout so checkpoints interchange.

Reference: /root/reference/src/models/msgat.py (GACN :17, TACN :57, CACN :83, MEAM :103,
TPC :137, MSGAT :166, factories :220-229), attention.py (TemporalAttention :42,
ChannelAttention :72), embeddings.py (TimeEmbedding :12).  Only the graph branch runs in
the HIP library at first; SURVEY.md section 8 rows f-1 and f-2 then pulled in every pass over the
[B,C,N,T] activations of a MEAM block: the LayerNorm over T (`LayerNormT`), the temporal and channel
branches (`ops.channel_pool / node_pool / mix / time_mix`) and the residual tail.  What stays in
PyTorch is tiny: the [B,T,T] / [B,C,C] attention matrices and their softmax.

Parameter names and shapes are the reference's (`tpcs.{r}.tgacns.{l}.gacn.gatt.Wg`, ...):
`tests/test_model_cpu.py` checks every key and shape against a reference checkpoint.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
from torch import nn

from . import ops, stacked
from .attention import GACN


class LayerNormT(nn.LayerNorm):
    """`nn.LayerNorm([n_timesteps])` (msgat.py:114, :152) with the reference's `weight` / `bias` keys,
    evaluated by the library's one-pass kernel (`ops.layer_norm_t`): PyTorch's LayerNorm kernels spend
    a thread group per 12-element row and were 46% of the training step (profiles/r01/full_model_*)."""

    def __init__(self, n_timesteps: int, eps: float = 1e-5):
        super().__init__([n_timesteps], eps=eps)

    def forward(self, signals: torch.Tensor) -> torch.Tensor:
        return ops.layer_norm_t(signals, self.weight, self.bias, self.eps)


class TemporalAttention(nn.Module):
    """[T,T] attention from rank-10 node projections (attention.py:58-66).  signals [B,C,N,T].

    The two passes over the activation run in the library: the channel-weighted sum (`ops.channel_pool`,
    the hot path's q kernel) and the product with the [T,T] matrix (`ops.time_mix`); the rank-10
    projections and the softmax over [B,T,T] are tiny and stay in PyTorch."""

    rank = 10

    def __init__(self, n_channels: int, n_nodes: int):
        super().__init__()
        self.n_channels, self.n_nodes = n_channels, n_nodes
        self.Wt1 = nn.Parameter(torch.empty(self.rank, n_nodes))
        self.Wt2 = nn.Parameter(torch.empty(self.rank, n_nodes))
        self.alpha = nn.Parameter(torch.empty(n_channels))

    def attention(self, signals: torch.Tensor, pooled: torch.Tensor = None) -> torch.Tensor:
        """-> att [B,T,T], rows = output step.  `pooled` = sum_c alpha_c signals_c [B,N,T] when the caller has
        it already (MEAM computes it in its merged channel-mixing pass)."""
        if pooled is None:
            pooled = ops.channel_pool(signals, self.alpha)
        per_t = pooled.transpose(1, 2)                                   # [B,T,N]
        left = per_t @ self.Wt1.t()                                      # [B,T,10]
        right = per_t @ self.Wt2.t()                                     # [B,T,10]
        return torch.softmax(left @ right.transpose(1, 2), dim=-1)

    def forward(self, signals: torch.Tensor) -> torch.Tensor:
        # out[..,t] = sum_i att[t,i] x[..,i]
        return ops.time_mix(signals, self.attention(signals).unsqueeze(1))

    def extra_repr(self) -> str:
        return f"n_channels={self.n_channels}, n_nodes={self.n_nodes}"


class ChannelAttention(nn.Module):
    """[C,C] attention from node-weighted signals (attention.py:88-94).  signals [B,C,N,T]."""

    def __init__(self, n_nodes: int, n_timesteps: int):
        super().__init__()
        self.n_nodes, self.n_timesteps = n_nodes, n_timesteps
        self.Wc = nn.Parameter(torch.empty(n_timesteps, n_timesteps))
        self.alpha = nn.Parameter(torch.empty(n_nodes))

    def attention(self, signals: torch.Tensor) -> torch.Tensor:
        """-> att [B,C,C]."""
        pooled = ops.node_pool(signals, self.alpha)                      # [B,C,T]: node-weighted sum
        return torch.softmax(pooled @ self.Wc @ pooled.transpose(1, 2), dim=-1)

    def forward(self, signals: torch.Tensor) -> torch.Tensor:
        return ops.mix(signals, self.attention(signals))                 # per-sample [C,C] channel matrix

    def extra_repr(self) -> str:
        return f"n_nodes={self.n_nodes}, n_timesteps={self.n_timesteps}"


class TrimRight(nn.Module):
    """Drops the last `n` steps of the time axis: makes a padded dilated conv causal (msgat.py:34-54)."""

    def __init__(self, n: int):
        super().__init__()
        self.n = n

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x[..., : x.size(-1) - self.n]

    def extra_repr(self) -> str:
        return f"n={self.n}"


def _shift_down(att: torch.Tensor, d: int) -> torch.Tensor:
    """rows t >= d of the result are rows t - d of `att` ([..,T,T]); rows < d are zero."""
    T = att.size(-2)
    if d >= T:
        return torch.zeros_like(att)
    return torch.nn.functional.pad(att[..., : T - d, :], (0, 0, d, 0))


class TACN(nn.Module):
    """Temporal attention, then a stack of causal dilated [1,2] convolutions (msgat.py:57-80).

    `seq` keeps the reference's layout (seq.0 attention, seq.1/3/.. Conv2d weights, seq.2/4/.. trims) so
    checkpoints interchange; the forward does not run the Conv2d modules.  A padded, right-trimmed
    [1,2] convolution with dilation d is out[t] = W_0 in[t-d] + W_1 in[t]; with in = attention(x) that is
    sum_k (W_k x) applied along time with A_1 = att, A_0 = att shifted down by d.  So each layer is one
    channel-mixing pass (C -> 2 Co) and one `ops.time_mix` pass; the attention product is never stored."""

    def __init__(self, in_channels: int, out_channels: int, n_nodes: int, dilations: Sequence[int]):
        super().__init__()
        self.in_channels, self.out_channels, self.n_nodes, self.dilations = (
            in_channels, out_channels, n_nodes, list(dilations))
        layers: List[nn.Module] = [TemporalAttention(in_channels, n_nodes)]   # seq.0
        width = in_channels
        for d in self.dilations:                                              # seq.1, seq.3, ... hold weights
            layers.append(nn.Conv2d(width, out_channels, kernel_size=(1, 2), padding=(0, d), dilation=(1, d)))
            layers.append(TrimRight(d))
            width = out_channels
        self.seq = nn.Sequential(*layers)

    def stacked_taps(self, layer: int) -> torch.Tensor:
        """[2*Co, Ci]: the two taps of convolution `layer` stacked on the output axis (tap 0 acts on in[t-d])."""
        w = self.seq[1 + 2 * layer].weight
        return w.flatten(1, 2).permute(2, 0, 1).reshape(2 * w.shape[0], w.shape[1])   # [tap 0 rows; tap 1 rows], one copy

    def forward(self, signals: torch.Tensor) -> torch.Tensor:
        if not self.dilations:
            return self.seq[0](signals)
        mixed = ops.mix(signals, self.stacked_taps(0).unsqueeze(0))
        return self.finish(mixed, ops.channel_pool(signals, self.seq[0].alpha))

    def first_taps(self, pooled: torch.Tensor) -> torch.Tensor:
        """[B,2,T,T] taps of the first convolution -- (temporal attention shifted down by its dilation, temporal
        attention) -- from the alpha-weighted channel sums `pooled` [B,N,T]: one launch in the library."""
        ta = self.seq[0]
        if 2 * pooled.shape[-1] * ta.rank > 256:          # beyond the fused kernel's lane budget (T = 16): eager ops
            att = ta.attention(None, pooled=pooled)
            return torch.stack([_shift_down(att, self.dilations[0]), att], dim=1)
        return ops.temporal_attention_taps(pooled, ta.Wt1, ta.Wt2, self.dilations[0])

    def finish(self, mixed: torch.Tensor, pooled: torch.Tensor) -> torch.Tensor:
        """The stack from the channel-mixed input of its first convolution (`mixed` = stacked_taps(0) applied to
        the signals, [B,2Co,N,T]) and the pooled signals of the temporal attention (`pooled` [B,N,T])."""
        T = mixed.size(-1)
        h = None
        for i, d in enumerate(self.dilations):
            conv = self.seq[1 + 2 * i]
            if i == 0:
                taps = self.first_taps(pooled)                                                # [B,2,T,T]
            elif ops.causal_conv_fused(h.shape[1], conv.out_channels):
                # constant shift taps: channel mixing, the [B,2Co,N,T] intermediate and the time mixing are one pass
                h = ops.causal_conv(h, self.stacked_taps(i).unsqueeze(0), conv.bias, d)
                continue
            else:
                taps = ops.causal_shift_taps(T, d, mixed.device)                              # [1,2,T,T], cached constant
                mixed = ops.mix(h, self.stacked_taps(i).unsqueeze(0))
            h = ops.time_mix(mixed, taps, conv.bias)
        return h


class CACN(nn.Module):
    """Channel attention, then a 1x1 convolution (msgat.py:83-100): one pass with the per-sample
    matrix `conv.weight @ att_b`."""

    def __init__(self, in_channels: int, out_channels: int, n_nodes: int, n_timesteps: int):
        super().__init__()
        self.in_channels, self.out_channels, self.n_nodes, self.n_timesteps = (
            in_channels, out_channels, n_nodes, n_timesteps)
        self.seq = nn.Sequential(ChannelAttention(n_nodes, n_timesteps), nn.Conv2d(in_channels, out_channels, 1))

    def channel_matrix(self, signals: torch.Tensor, pooled: torch.Tensor = None) -> torch.Tensor:
        """[B,Co,C] = conv.weight @ channel attention: node pooling in one pass (or `pooled` [B,C,T], when the caller has
        it already: MEAM takes it out of its LayerNorm pass), the rest in one launch."""
        ca, conv = self.seq[0], self.seq[1]
        if pooled is None:
            pooled = ops.node_pool(signals, ca.alpha)
        return ops.channel_attention_mix(pooled, ca.Wc, conv.weight.flatten(1))

    def forward(self, signals: torch.Tensor, pooled: torch.Tensor = None) -> torch.Tensor:
        return ops.mix(signals, self.channel_matrix(signals, pooled), self.seq[1].bias)


class MEAM(nn.Module):
    """LayerNorm -> {channel, temporal, graph} branches -> concat + 1x1 residual -> ReLU (msgat.py:103-134).

    The graph branch is the hot path: `self.gacn(normed, adjacency)` (msgat.py:127)."""

    def __init__(self, in_channels: int, out_channels: int, n_nodes: int, n_timesteps: int,
                 dilations: Sequence[int]):
        if out_channels % 3:
            raise ValueError("out_channels must be divisible by the 3 branches")
        super().__init__()
        self.in_channels, self.out_channels, self.n_nodes, self.n_timesteps, self.dilations = (
            in_channels, out_channels, n_nodes, n_timesteps, list(dilations))
        branch = out_channels // 3
        self.ln = LayerNormT(n_timesteps)
        self.res = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.cacn = CACN(in_channels, branch, n_nodes=n_nodes, n_timesteps=n_timesteps)
        self.tacn = TACN(in_channels, branch, n_nodes=n_nodes, dilations=dilations)
        self.gacn = GACN(in_channels, branch, n_timesteps=n_timesteps)

    def forward(self, signals: torch.Tensor, adjacency, relu_input: bool = False, premasked: bool = False) -> torch.Tensor:
        """`relu_input` / `premasked` are for a caller that chains blocks itself (TPC): `relu_input` says `signals` is a
        ReLU output whose backward mask this block's LayerNorm applies; `premasked` says the consumer of THIS block's
        output will do the same for it, so the block skips its own mask pass.  Defaults: plain autograd semantics."""
        # (normed, signals): the residual convolution below reads the block input again (msgat.py:130); routed
        # through the LayerNorm op, its gradient is added inside the LayerNorm-backward kernel
        # ... and CACN pools the normalised input over the nodes (attention.py:89): the pooled signal comes out of the
        # LayerNorm pass, its gradients go back inside the LayerNorm-backward kernel (ops.layer_norm_pool_tee)
        normed, signals, pooled_c = ops.layer_norm_pool_tee(signals, self.ln.weight, self.ln.bias, self.ln.eps, relu_input,
                                                            self.cacn.seq[0].alpha)
        res_w = self.res.weight.flatten(1).unsqueeze(0)
        if self.in_channels <= self.out_channels // 3 or not self.dilations:
            # few input channels (the first block of a component): the graph branch aggregates before it
            # projects, nothing to merge -- branch by branch
            branches = [self.cacn(normed, pooled_c), self.tacn(normed), self.gacn(normed, adjacency)]
        else:
            branches = self._merged_branches(normed, adjacency, pooled_c)
        # relu(cat(branches) + res(signals)): the 1x1 residual convolution reads the three branch tensors as
        # its add operand (no concatenation) and applies the ReLU in its store epilogue
        return ops.mix_multi([signals], res_w, self.res.bias, adds=branches, relu=True, relu_grad_premasked=premasked)[0]

    def _merged_branches(self, normed: torch.Tensor, adjacency, pooled_c: torch.Tensor = None):
        """All channel mixings of the normalised input in ONE pass: CACN's per-sample
        matrix `conv @ att_b` (msgat.py:93-94), the two taps of TACN's first convolution (msgat.py:66-74), GACN's
        projection W (msgat.py:27, applied before the aggregation as C > out/3) and the two alpha-weighted channel
        poolings (attention.py:33 and :59) are rows of one [B, 4*Cb + 2, C] matrix.  The backward is then ONE
        transposed pass producing the gradient at the LayerNorm output, instead of five passes plus four adds."""
        B, C = normed.shape[:2]
        cb = self.out_channels // 3
        ca, ta, gatt = self.cacn.seq[0], self.tacn.seq[0], self.gacn.gatt
        conv_c = self.cacn.seq[1]
        rows = torch.cat([
            self.cacn.channel_matrix(normed, pooled_c),                          # [B,cb,C]   from the pooled signal
            self.tacn.stacked_taps(0).unsqueeze(0).expand(B, -1, -1),            # [B,2cb,C]
            self.gacn.W.unsqueeze(0).expand(B, -1, -1),                          # [B,cb,C]
            gatt.alpha.view(1, 1, C).expand(B, -1, -1),                          # [B,1,C]    q of the graph attention
            ta.alpha.view(1, 1, C).expand(B, -1, -1)], dim=1)                    # [B,1,C]    pooled signal of the temporal attention
        bias = torch.cat([conv_c.bias, conv_c.bias.new_zeros(3 * cb + 2)])
        cacn, mixed, u, q, pooled_t = ops.mix_multi([normed], rows, bias, out_channels=[cb, 2 * cb, cb, 1, 1])
        tacn = self.tacn.finish(mixed, pooled_t.flatten(1, 2))
        gacn = ops.attention_core(u, q.flatten(1, 2), gatt.Wg.unsqueeze(0), adjacency)
        return [cacn, tacn, gacn]


class TPC(nn.Module):
    """One component: a stack of MEAMs, LayerNorm, and a [1,C] convolution mapping T_in -> T_out (msgat.py:137-163)."""

    def __init__(self, channels: Sequence[int], n_nodes: int, in_timesteps: int, out_timesteps: int,
                 dilations: Sequence[Sequence[int]]):
        super().__init__()
        self.channels, self.n_nodes, self.in_timesteps, self.out_timesteps, self.dilations = (
            list(channels), n_nodes, in_timesteps, out_timesteps, [list(d) for d in dilations])
        self.tgacns = nn.ModuleList(
            MEAM(channels[i], channels[i + 1], n_nodes=n_nodes, n_timesteps=in_timesteps, dilations=d)
            for i, d in enumerate(dilations))
        self.ln = LayerNormT(in_timesteps)
        self.fc = nn.Conv2d(in_timesteps, out_timesteps, kernel_size=(1, channels[-1]))

    def forward(self, signals: torch.Tensor, adjacency) -> torch.Tensor:
        # every block's output is a ReLU output consumed by exactly one LayerNorm (the next block's, or self.ln): that
        # LayerNorm's backward kernel applies the ReLU mask, the blocks skip their own mask pass
        # (only among this package's own modules: a swapped-in block or LayerNorm gets plain autograd semantics)
        own = isinstance(self.ln, LayerNormT) and all(isinstance(b, MEAM) for b in self.tgacns)
        for i, block in enumerate(self.tgacns):
            signals = block(signals, adjacency, relu_input=i > 0, premasked=True) if own else block(signals, adjacency)
        if own:
            # LayerNorm, then fc over the transposed activation, squeezed and transposed back (msgat.py:158-160); backward
            # is one pass for the head's input gradient and the LayerNorm together
            return ops.ln_head(signals, self.ln.weight, self.ln.bias, self.ln.eps, self.fc.weight, self.fc.bias,
                               relu_input=len(self.tgacns) > 0)                    # [B,N,T_out]
        normed = self.ln(signals)
        return ops.head(normed, self.fc.weight, self.fc.bias)                      # [B,N,T_out]


class TimeEmbedding(nn.Module):
    """Hour-of-day + day-of-week gate [B,R,N,T_out] (embeddings.py:12-39)."""

    def __init__(self, n_components: int, n_nodes: int, n_timesteps: int):
        super().__init__()
        self.n_components, self.n_nodes, self.n_timesteps = n_components, n_nodes, n_timesteps
        width = n_components * n_nodes * n_timesteps
        self.d_ebd = nn.Embedding(7, width)
        self.h_ebd = nn.Embedding(24, width)

    def forward(self, H: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
        gate = self.h_ebd(H) + self.d_ebd(D)
        return gate.view(-1, self.n_components, self.n_nodes, self.n_timesteps)


class MSGAT(nn.Module):
    """X [B,R,C,N,T], H [B], D [B] -> [B,N,T_out]  (msgat.py:166-217).

    `use_te=False` uses the learned static gate `W [R,N,T_out]` that the reference declares
    (msgat.py:189) but cannot reach (its forward reads `self.te` unconditionally, msgat.py:203).
    """

    def __init__(self, components: Sequence[Dict], in_timesteps: int, out_timesteps: int, use_te: bool,
                 adj: torch.Tensor):
        super().__init__()
        n_nodes = len(adj)
        if use_te:
            self.te = TimeEmbedding(len(components), n_nodes, out_timesteps)
        else:
            self.te = None
            self.W = nn.Parameter(torch.empty(len(components), n_nodes, out_timesteps))
        self.adj = nn.Parameter(adj.detach().clone().float(), requires_grad=False)
        self.tpcs = nn.ModuleList(
            TPC(channels=c["channels"], n_nodes=n_nodes, in_timesteps=in_timesteps, out_timesteps=out_timesteps,
                dilations=c["dilations"]) for c in components)
        self.stack_components = True   # False: evaluate the components one by one, as the reference does
        self.reset_parameters()

    def forward(self, X: torch.Tensor, H: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
        if self.stack_components and stacked.can_stack(self):
            # all components in each kernel launch (stacked.py) instead of the reference's loop (msgat.py:204)
            return stacked.forward(self, X, H, D)
        gates = self.te(H, D).unbind(1) if self.te is not None else self.W.unbind(0)
        out = None
        for tpc, x, gate in zip(self.tpcs, X.unbind(1), gates):
            term = tpc(x, self.adj) * gate
            out = term if out is None else out + term
        return out

    def reset_parameters(self) -> None:
        """xavier_normal_ for >= 2-D, U(-size0^-1/2, +size0^-1/2) for 1-D, every trainable tensor (msgat.py:206-217)."""
        with torch.no_grad():
            for p in self.parameters():
                if not p.requires_grad:
                    continue
                if p.dim() >= 2:
                    nn.init.xavier_normal_(p)
                else:
                    bound = p.size(0) ** -0.5
                    p.uniform_(-bound, bound)


_WIDTHS = {  # msgat.py:220-229
    "ms-gat48": dict(hidden=48, dilations=[[1, 2], [2, 4]]),
    "ms-gat72": dict(hidden=72, dilations=[[1, 2], [2, 4]]),
    "ms-gat96": dict(hidden=96, dilations=[[1, 1, 2, 2], [4, 4]]),
}
_WIDTHS["ms-gat"] = _WIDTHS["ms-gat72"]  # main.py:17


def build_msgat(name: str, n_components: int, in_channels: int, **kwargs) -> MSGAT:
    cfg = _WIDTHS[name]
    comp = {"channels": [in_channels, cfg["hidden"], cfg["hidden"]], "dilations": cfg["dilations"]}
    return MSGAT([comp] * n_components, **kwargs)


def msgat48(n_components: int, in_channels: int, **kwargs) -> MSGAT:
    return build_msgat("ms-gat48", n_components, in_channels, **kwargs)


def msgat72(n_components: int, in_channels: int, **kwargs) -> MSGAT:
    return build_msgat("ms-gat72", n_components, in_channels, **kwargs)


def msgat96(n_components: int, in_channels: int, **kwargs) -> MSGAT:
    return build_msgat("ms-gat96", n_components, in_channels, **kwargs)


models = {name: (lambda n=name: (lambda **kw: build_msgat(n, **kw)))() for name in _WIDTHS}
