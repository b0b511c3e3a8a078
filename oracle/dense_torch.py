"""Dense PyTorch restatement of the reference hot path (TEST / BASELINE ONLY).

The op sequence the reference runs for GraphAttention + GACN
(/root/reference/src/models/attention.py:33-36 and src/models/msgat.py:27-28),
written as plain functions over explicit parameter tensors.  Every intermediate
is a dense [B,N,N] tensor exactly like the reference's eager path, so this is
what `bench.py` times as

  * `cpu_baseline`  (kind "port"): on the GPU box's host cores, and
  * `eager_rocm`    : the same ops on PyTorch-ROCm eager (the ">= 10x" target
    of BASELINE.json is quoted against this).

It is pinned against the imported reference by `tests/test_oracle_golden.py`
(golden vectors in `tests/golden/`).  The product package never imports it.
"""
from __future__ import annotations

import torch


def graph_attention_dense(x: torch.Tensor, adj: torch.Tensor, Wg: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """x [B,C,N,T] -> [B,C,N,T];  attention.py:33 (q=k), :34 (softmax over all N), :36 (mask, aggregate)."""
    q = torch.einsum("bcnt,c->bnt", x, alpha)
    score = torch.matmul(torch.matmul(q, Wg), q.transpose(1, 2))
    coeff = torch.softmax(score, dim=-1) * adj
    return torch.einsum("bnm,bcmt->bcnt", coeff, x)


def gacn_dense(x, adj, Wg, alpha, W):
    """-> [B,C_out,N,T];  msgat.py:26 (attention), :27 (per-node channel projection), :28 (transpose back)."""
    y = graph_attention_dense(x, adj, Wg, alpha)
    z = torch.matmul(y.transpose(1, -1), W.t())
    return z.transpose(1, -1)


def huber(output: torch.Tensor, target: torch.Tensor, delta: float) -> torch.Tensor:
    """loss.py:51-52."""
    err = (output - target).abs()
    return torch.where(err <= delta, 0.5 * err * err, delta * err - 0.5 * delta * delta).mean()
