"""Autograd binding of the HIP hot path.

`gacn(x, alpha, Wg, W, adjacency)` is the functional form of the reference's
`GACN.forward` (/root/reference/src/models/msgat.py:25-28) -- and of
`GraphAttention.forward` (src/models/attention.py:32-36) when `W is None` -- for
R stacked relations at once.  PyTorch only owns memory and streams here: every
arithmetic step runs in libmsgat_hip.so through the C ABI of include/msgat_hip.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .graph import SparseGraph, graph_of


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _require_device_tensor(name: str, t: torch.Tensor, device=None):
    if not t.is_cuda:
        raise _lib.MsgatError(
            f"{name} is on {t.device}: ms_gat_amd runs on MI355X (PyTorch-ROCm 'cuda' device) only; "
            "there is no CPU path -- the CPU oracle lives under oracle/ and is test infrastructure.")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference arithmetic type), got {t.dtype}")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")


def _stream_handle(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _GACNFunction(torch.autograd.Function):
    """x[G,C,N,T], alpha[R,C], Wg[R,T,T], W[R,Co,C] or None -> z[G,Co|C,N,T];  G = R*Bg."""

    @staticmethod
    def forward(ctx, x, alpha, Wg, W, graph: SparseGraph):
        L = _lib.lib()
        dev = x.device
        G, Cin, N, T = x.shape
        R = alpha.shape[0]
        if G % R != 0:
            raise ValueError(f"{G} groups cannot be split over {R} relations")
        Co = 0 if W is None else W.shape[1]
        shape = _lib.Shape(R, G // R, Cin, Co, N, T)
        mode = L.msgat_gacn_mode(Cin, Co)
        need_bwd = any(ctx.needs_input_grad)

        x = x.contiguous()
        alpha, Wg = alpha.contiguous(), Wg.contiguous()
        W = None if W is None else W.contiguous()
        gstruct, _keep = graph.on(dev)
        nnz = max(graph.nnz, 1)

        def new(*s):
            return torch.empty(s, device=dev, dtype=torch.float32)

        z = new(G, Co if Co else Cin, N, T)
        q, kW, lse, E = new(G, N, T), new(G, N, T), new(G, N), new(G, nnz)
        pq = new(G, N, T) if need_bwd else None
        if mode == _lib.MODE_PROJ_FIRST:
            u = new(G, Co, N, T)
        elif mode == _lib.MODE_AGG_FIRST and need_bwd:
            u = new(G, Cin, N, T)
        else:
            u = None
        io = _lib.Fwd(_ptr(x), _ptr(alpha), _ptr(Wg), _ptr(W), _ptr(z), _ptr(q), _ptr(kW), _ptr(lse), _ptr(pq),
                      _ptr(E), _ptr(u), int(need_bwd))
        st = L.msgat_gacn_forward(C.byref(shape), C.byref(gstruct), C.byref(io), _stream_handle(dev))
        _lib.check(st, "msgat_gacn_forward")

        if need_bwd:
            ctx.graph, ctx.dims, ctx.has_W = graph, (R, G // R, Cin, Co, N, T), W is not None
            saved = [x, alpha, Wg, q, kW, lse, pq, E]
            if W is not None:
                saved.append(W)
            if u is not None:
                saved.append(u)
            ctx.has_u = u is not None
            ctx.save_for_backward(*saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        L = _lib.lib()
        saved = list(ctx.saved_tensors)
        x, alpha, Wg, q, kW, lse, pq, E = saved[:8]
        rest = saved[8:]
        W = rest.pop(0) if ctx.has_W else None
        u = rest.pop(0) if ctx.has_u else None
        dev = x.device
        R, Bg, Cin, Co, N, T = ctx.dims
        shape = _lib.Shape(R, Bg, Cin, Co, N, T)
        gstruct, _keep = ctx.graph.on(dev)
        dz = dz.contiguous()

        dx = torch.empty_like(x)
        dalpha = torch.empty_like(alpha)
        dWg = torch.empty_like(Wg)
        dW = None if W is None else torch.empty_like(W)
        nbytes = L.msgat_bwd_workspace_bytes(C.byref(shape), ctx.graph.nnz)
        ws = torch.empty(max(int(nbytes), 256), device=dev, dtype=torch.uint8)
        io = _lib.Bwd(_ptr(x), _ptr(alpha), _ptr(Wg), _ptr(W), _ptr(q), _ptr(kW), _ptr(lse), _ptr(pq), _ptr(E),
                      _ptr(u), _ptr(dz), _ptr(dx), _ptr(dalpha), _ptr(dWg), _ptr(dW), _ptr(ws), ws.numel())
        st = L.msgat_gacn_backward(C.byref(shape), C.byref(gstruct), C.byref(io), _stream_handle(dev))
        _lib.check(st, "msgat_gacn_backward")
        return dx, dalpha, dWg, dW, None


def gacn(x: torch.Tensor, alpha: torch.Tensor, Wg: torch.Tensor, W: Optional[torch.Tensor],
         adjacency) -> torch.Tensor:
    """Graph attention (+ channel projection when `W` is given) over R stacked relations.

    x [R*Bg, C, N, T] (relation-major), alpha [R,C], Wg [R,T,T], W [R,Co,C] or None,
    adjacency: dense [N,N] tensor or a prebuilt `SparseGraph`.  Returns [R*Bg, Co|C, N, T].
    """
    if x.dim() != 4:
        raise ValueError(f"signals must be [batch, channels, nodes, timesteps], got {tuple(x.shape)}")
    _require_device_tensor("signals", x)
    for name, t in (("alpha", alpha), ("Wg", Wg)) + ((("W", W),) if W is not None else ()):
        _require_device_tensor(name, t, x.device)
    G, Cin, N, T = x.shape
    if alpha.dim() != 2 or alpha.shape[1] != Cin:
        raise ValueError(f"alpha must be [R,{Cin}], got {tuple(alpha.shape)}")
    R = alpha.shape[0]
    if tuple(Wg.shape) != (R, T, T):
        raise ValueError(f"Wg must be [{R},{T},{T}], got {tuple(Wg.shape)}")
    if W is not None and (W.dim() != 3 or W.shape[0] != R or W.shape[2] != Cin):
        raise ValueError(f"W must be [{R},Co,{Cin}], got {tuple(W.shape)}")
    graph = adjacency if isinstance(adjacency, SparseGraph) else graph_of(adjacency)
    if graph.n_nodes != N:
        raise ValueError(f"adjacency has {graph.n_nodes} nodes, signals have {N}")
    return _GACNFunction.apply(x, alpha, Wg, W, graph)


def graph_attention(x, alpha, Wg, adjacency):
    """`GraphAttention.forward` (attention.py:32-36) for R stacked relations."""
    return gacn(x, alpha, Wg, None, adjacency)


class _LayerNormTFunction(torch.autograd.Function):
    """x[..., T], weight[T] | None, bias[T] | None -> LayerNorm over the last axis."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float):
        L = _lib.lib()
        x = x.contiguous()
        T = x.shape[-1]
        rows = x.numel() // T if T else 0
        y = torch.empty_like(x)
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        st = L.msgat_layernorm_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, T, eps, _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_forward")
        ctx.eps, ctx.has_w, ctx.has_b = eps, weight is not None, bias is not None
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(*([x] + ([w] if w is not None else [])))
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, w = saved[0], (saved[1] if ctx.has_w else None)
        T = x.shape[-1]
        rows = x.numel() // T
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(T, device=x.device, dtype=torch.float32) if ctx.has_w else None
        db = torch.empty(T, device=x.device, dtype=torch.float32) if ctx.has_b else None
        part = torch.empty(max(int(L.msgat_layernorm_partial_floats(rows, T)), 1), device=x.device,
                           dtype=torch.float32)
        st = L.msgat_layernorm_backward(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), _ptr(part),
                                        rows, T, ctx.eps, _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_backward")
        return dx, dw, db, None


def layer_norm_t(x: torch.Tensor, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                 eps: float = 1e-5) -> torch.Tensor:
    """`F.layer_norm(x, [T], weight, bias, eps)` over the timestep axis, the op the reference applies
    to every GACN input (msgat.py:122, :158), as one HBM-speed pass in libmsgat_hip.so."""
    _require_device_tensor("signals", x)
    T = x.shape[-1]
    for name, t in (("weight", weight), ("bias", bias)):
        if t is not None:
            _require_device_tensor(name, t, x.device)
            if tuple(t.shape) != (T,):
                raise ValueError(f"{name} must be [{T}], got {tuple(t.shape)}")
    if x.numel() == 0:
        return torch.empty_like(x)
    return _LayerNormTFunction.apply(x, weight, bias, float(eps))
