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


# ---- the dense branches of MEAM, same rules: test / baseline only ----------

def temporal_attention_dense(x, Wt1, Wt2, alpha):
    """attention.py:58-66.  x [B,C,N,T] -> [B,C,N,T]: [T,T] attention from rank-10 node projections."""
    mixed = torch.tensordot(x, alpha, dims=([1], [0]))                  # [B,N,T]
    per_t = mixed.transpose(1, 2)                                       # [B,T,N]
    att = torch.softmax((per_t @ Wt1.t()) @ (per_t @ Wt2.t()).transpose(1, 2), dim=-1)
    return x @ att.transpose(1, 2).unsqueeze(1)


def channel_attention_dense(x, Wc, alpha):
    """attention.py:88-94.  [C,C] attention from node-weighted signals."""
    B, C, N, T = x.shape
    pooled = torch.tensordot(x, alpha, dims=([2], [0]))                 # [B,C,T]
    att = torch.softmax(pooled @ Wc @ pooled.transpose(1, 2), dim=-1)
    return (att @ x.reshape(B, C, N * T)).view(B, C, N, T)


def tacn_dense(x, Wt1, Wt2, alpha, convs):
    """msgat.py:57-80.  convs = [(weight [Co,Ci,1,2], bias [Co], dilation), ...]: padded dilated conv, right-trimmed."""
    h = temporal_attention_dense(x, Wt1, Wt2, alpha)
    for weight, bias, d in convs:
        h = torch.nn.functional.conv2d(h, weight, bias, padding=(0, d), dilation=(1, d))
        h = h[..., : h.size(-1) - d]
    return h


def cacn_dense(x, Wc, alpha, weight, bias):
    """msgat.py:83-100: channel attention, then a 1x1 convolution."""
    return torch.nn.functional.conv2d(channel_attention_dense(x, Wc, alpha), weight, bias)


def meam_dense(x, adj, p, dilations, eps=1e-5, relu_mask=None):
    """msgat.py:117-131 with the parameters of one MEAM in `p` (reference state_dict keys).
    `relu_mask` (bool, output-shaped): apply THIS mask instead of the ReLU's own -- a test pins the set of active units
    to the one another implementation chose, so that pre-activations within rounding of zero do not decide a comparison."""
    T = x.shape[-1]
    normed = torch.nn.functional.layer_norm(x, [T], p["ln.weight"], p["ln.bias"], eps)
    convs = [(p[f"tacn.seq.{2 * i + 1}.weight"], p[f"tacn.seq.{2 * i + 1}.bias"], d) for i, d in enumerate(dilations)]
    branches = torch.cat([
        cacn_dense(normed, p["cacn.seq.0.Wc"], p["cacn.seq.0.alpha"], p["cacn.seq.1.weight"], p["cacn.seq.1.bias"]),
        tacn_dense(normed, p["tacn.seq.0.Wt1"], p["tacn.seq.0.Wt2"], p["tacn.seq.0.alpha"], convs),
        gacn_dense(normed, adj, p["gacn.gatt.Wg"], p["gacn.gatt.alpha"], p["gacn.W"])], dim=1)
    pre = branches + torch.nn.functional.conv2d(x, p["res.weight"], p["res.bias"])
    return torch.relu(pre) if relu_mask is None else pre * relu_mask.to(pre.dtype)
