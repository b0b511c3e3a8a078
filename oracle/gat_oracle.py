"""CPU oracle for the MS-GAT graph-attention hot path (TEST INFRASTRUCTURE ONLY).

This file is a dense numpy restatement of the reference algorithm.  It exists to
check the HIP kernels; it is never imported by the product package
(`ms_gat_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference
modules from `/root/reference/src` (allowed in the build container, SURVEY.md
section 8c) and stores input/output vectors under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against them.

Reference lines restated here (paths under /root/reference/):
  src/models/attention.py:33   q = k = sum_c alpha[c] * x[b,c,n,t]
  src/models/attention.py:34   att = softmax_over_all_N( (k @ Wg) @ q^T )
  src/models/attention.py:36   y[b,c,n,t] = sum_m (att*adj)[b,n,m] x[b,c,m,t]
  src/models/msgat.py:27-28    z[b,o,n,t] = sum_c W[o,c] y[b,c,n,t]
  src/data_loader.py:59-66     adj = D^-1/2 (A + I) D^-1/2
  src/loss.py:51-52            mean Huber loss

The softmax runs over the FULL row of N columns and the adjacency mask is
applied AFTER it (attention.py:34 then :36), so rows of att*adj do not sum to 1.

All functions take and return numpy arrays.  dtype follows the inputs: pass
float64 for a high-precision oracle, float32 to mimic the reference's
arithmetic type.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------
# adjacency (data_loader.py:59-66)
# --------------------------------------------------------------------------
def sym_norm_adjacency(n_nodes: int, edges: np.ndarray, dtype=np.float32) -> np.ndarray:
    """D^-1/2 (A + I) D^-1/2 for an undirected edge list [[src, dst], ...].

    Follows data_loader.py:59 (A starts as I), :60-63 (A[s,d] = A[d,s] = 1) and
    :65-66 (row sums -> rsqrt -> two-sided scaling).  A repeated or self edge
    stays 1, exactly as the assignment in the reference does.
    """
    a = np.eye(n_nodes, dtype=np.float64)
    for s, d in np.asarray(edges, dtype=np.int64).reshape(-1, 2):
        a[s, d] = 1.0
        a[d, s] = 1.0
    d_rsqrt = 1.0 / np.sqrt(a.sum(axis=1))
    return (d_rsqrt[:, None] * a * d_rsqrt[None, :]).astype(dtype)


# --------------------------------------------------------------------------
# GraphAttention forward (attention.py:33-36)
# --------------------------------------------------------------------------
def gatt_forward(x, adj, Wg, alpha, return_cache: bool = False):
    """x [B,C,N,T], adj [N,N], Wg [T,T], alpha [C] -> y [B,C,N,T]."""
    q = np.einsum("bcnt,c->bnt", x, alpha)            # attention.py:33
    kW = q @ Wg                                         # attention.py:34 (k @ Wg)
    S = kW @ q.transpose(0, 2, 1)                       # [B,N,N]
    S = S - S.max(axis=-1, keepdims=True)               # stable softmax, all N columns
    P = np.exp(S)
    P = P / P.sum(axis=-1, keepdims=True)
    E = P * adj                                         # attention.py:36 (mask AFTER softmax)
    y = np.einsum("bnm,bcmt->bcnt", E, x)               # attention.py:36
    if return_cache:
        return y, dict(q=q, kW=kW, P=P, E=E)
    return y


def gatt_lse(x, Wg, alpha):
    """Row log-sum-exp of the score matrix, lse[b,n] = log sum_m exp(S[b,n,m])."""
    q = np.einsum("bcnt,c->bnt", x, alpha)
    S = (q @ Wg) @ q.transpose(0, 2, 1)
    mx = S.max(axis=-1)
    return mx + np.log(np.exp(S - mx[..., None]).sum(axis=-1))


# --------------------------------------------------------------------------
# GraphAttention backward (autograd of attention.py:33-36, derived by hand)
# --------------------------------------------------------------------------
def gatt_backward(x, adj, Wg, alpha, dy):
    """Gradients of sum(y * dy) w.r.t. x, Wg, alpha (adj has no gradient,
    msgat.py:190 registers it with requires_grad=False)."""
    _, c = gatt_forward(x, adj, Wg, alpha, return_cache=True)
    q, kW, P, E = c["q"], c["kW"], c["P"], c["E"]
    # y = E x  ->  dE[b,n,m] = sum_{c,t} dy[b,c,n,t] x[b,c,m,t];  dx1 = E^T dy
    dE = np.einsum("bcnt,bcmt->bnm", dy, x)
    dx = np.einsum("bnm,bcnt->bcmt", E, dy)
    # E = P * adj ; softmax backward over the full row
    g = P * adj * dE
    delta = g.sum(axis=-1, keepdims=True)
    dS = g - delta * P
    # S = kW q^T
    dkW = dS @ q
    dq = dS.transpose(0, 2, 1) @ kW
    # kW = q Wg
    dWg = np.einsum("bnt,bns->ts", q, dkW)
    dq = dq + dkW @ Wg.T
    # q = sum_c alpha_c x_c
    dalpha = np.einsum("bnt,bcnt->c", dq, x)
    dx = dx + alpha[None, :, None, None] * dq[:, None, :, :]
    return dx, dWg, dalpha


# --------------------------------------------------------------------------
# GACN = GraphAttention + channel projection (msgat.py:25-28)
# --------------------------------------------------------------------------
def gacn_forward(x, adj, Wg, alpha, W):
    """-> z [B,C_out,N,T] with z[b,o,n,t] = sum_c W[o,c] y[b,c,n,t] (msgat.py:27-28)."""
    y = gatt_forward(x, adj, Wg, alpha)
    return np.einsum("oc,bcnt->bont", W, y)


def gacn_backward(x, adj, Wg, alpha, W, dz):
    y = gatt_forward(x, adj, Wg, alpha)
    dW = np.einsum("bont,bcnt->oc", dz, y)
    dy = np.einsum("oc,bont->bcnt", W, dz)
    dx, dWg, dalpha = gatt_backward(x, adj, Wg, alpha, dy)
    return dx, dWg, dalpha, dW


# --------------------------------------------------------------------------
# Huber loss (loss.py:51-52), used by the engine-level fixtures
# --------------------------------------------------------------------------
def huber_loss(output, target, delta=1.0):
    err = np.abs(output - target)
    quad = 0.5 * (output - target) ** 2
    lin = delta * err - 0.5 * delta ** 2
    return np.where(err <= delta, quad, lin).mean()
