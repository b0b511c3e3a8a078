"""Drop-in modules for the reference's graph-attention operators.

Same constructor arguments, parameter names/shapes and `forward(signals, adjacency)`
signature as /root/reference/src/models/attention.py:12-39 (`GraphAttention`) and
src/models/msgat.py:17-31 (`GACN`), so `state_dict`s interchange key for key
(`Wg`, `alpha`, `gatt.Wg`, `gatt.alpha`, `W`).  The arithmetic runs in the HIP library.

`StackedGACN` is the MI355X-first form: the R components of MS-GAT (msgat.py:191-199)
evaluate the same operator with R parameter sets over one adjacency, so their R*B
attention problems go to the GPU as one launch sequence instead of R Python iterations.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class GraphAttention(nn.Module):
    """signals [B,C,N,T], adjacency [N,N] -> [B,C,N,T]   (attention.py:21-23)."""

    def __init__(self, n_channels: int, n_timesteps: int):
        super().__init__()
        self.n_channels, self.n_timesteps = n_channels, n_timesteps
        # uninitialised like the reference's (attention.py:29-30); MSGAT.reset_parameters fills them
        self.Wg = nn.Parameter(torch.empty(n_timesteps, n_timesteps))
        self.alpha = nn.Parameter(torch.empty(n_channels))

    def forward(self, signals: torch.Tensor, adjacency: torch.Tensor) -> torch.Tensor:
        return ops.gacn(signals, self.alpha, self.Wg, None, adjacency)

    def extra_repr(self) -> str:
        return f"n_channels={self.n_channels}, n_timesteps={self.n_timesteps}"


class GACN(nn.Module):
    """Graph attention followed by the per-node channel projection W (msgat.py:25-28).

    The two commute, so the library projects first when in_channels > out_channels
    (72 -> 24 in the second MEAM: a third of the gather traffic) and aggregates first
    otherwise (1 or 3 -> 24 in the first MEAM).
    """

    def __init__(self, in_channels: int, out_channels: int, n_timesteps: int):
        super().__init__()
        self.in_channels, self.out_channels, self.n_timesteps = in_channels, out_channels, n_timesteps
        self.gatt = GraphAttention(n_channels=in_channels, n_timesteps=n_timesteps)
        self.W = nn.Parameter(torch.empty(out_channels, in_channels))

    def forward(self, signals: torch.Tensor, adjacency: torch.Tensor) -> torch.Tensor:
        return ops.gacn(signals, self.gatt.alpha, self.gatt.Wg, self.W, adjacency)

    def extra_repr(self) -> str:
        return (f"in_channels={self.in_channels}, out_channels={self.out_channels}, "
                f"n_timesteps={self.n_timesteps}")


class StackedGACN(nn.Module):
    """R independent GACNs over one adjacency, evaluated together.

    signals [R,B,C,N,T] -> [R,B,Co,N,T].  `from_modules` stacks the parameters of
    R ordinary `GACN`s (whose keys are the reference's checkpoint names).
    """

    def __init__(self, n_relations: int, in_channels: int, out_channels: int, n_timesteps: int):
        super().__init__()
        self.n_relations, self.in_channels, self.out_channels, self.n_timesteps = (
            n_relations, in_channels, out_channels, n_timesteps)
        self.Wg = nn.Parameter(torch.empty(n_relations, n_timesteps, n_timesteps))
        self.alpha = nn.Parameter(torch.empty(n_relations, in_channels))
        self.W = nn.Parameter(torch.empty(n_relations, out_channels, in_channels))

    @classmethod
    def from_modules(cls, gacns):
        gacns = list(gacns)
        g0 = gacns[0]
        m = cls(len(gacns), g0.in_channels, g0.out_channels, g0.n_timesteps)
        with torch.no_grad():
            m.Wg.copy_(torch.stack([g.gatt.Wg for g in gacns]))
            m.alpha.copy_(torch.stack([g.gatt.alpha for g in gacns]))
            m.W.copy_(torch.stack([g.W for g in gacns]))
        return m.to(g0.W.device)

    def forward(self, signals: torch.Tensor, adjacency) -> torch.Tensor:
        R, B = signals.shape[:2]
        z = ops.gacn(signals.reshape(R * B, *signals.shape[2:]), self.alpha, self.Wg, self.W, adjacency)
        return z.view(R, B, *z.shape[1:])

    def extra_repr(self) -> str:
        return (f"n_relations={self.n_relations}, in_channels={self.in_channels}, "
                f"out_channels={self.out_channels}, n_timesteps={self.n_timesteps}")
