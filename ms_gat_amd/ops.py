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
    if t.dtype != torch.float32 and not (t.dtype in (torch.float16, torch.bfloat16) and torch.is_autocast_enabled("cuda")):
        # inside an autocast region (the reference's engine.py:54) half-precision outputs of neighbouring matmuls are
        # accepted and cast back to float32 at the op boundary (see _guard below)
        raise TypeError(f"{name} must be float32 (the reference arithmetic type), got {t.dtype}")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle(device) -> int:
    """The current HIP stream of `device` as an integer handle.  (`torch.cuda.current_stream(...).cuda_stream` builds a
    Stream object per call, ~5 us -- a tenth of the host time of a PEMSD4-sized forward.)"""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class _GacnPlan:
    """Everything of a GACN call that depends only on its dimensions, its graph and whether backward state is kept:
    the shape / graph structures handed to the library, the layout of the one buffer that carries what a forward saves
    (q, kW, lse, pq, E, E in CSC order, u, the SELL scratch) and the backward workspace size.  Built once per
    (graph, device, dims, need_bwd): the library queries and the offset arithmetic were ~10 us of host time per call."""
    __slots__ = ("shape", "gstruct", "keep", "mode", "sizes", "offs", "total", "bwd_bytes", "z_channels", "nscratch", "ndense")

    def __init__(self, graph, dev, R, Bg, Cin, Co, N, T, need_bwd):
        L = _lib.lib()
        G = R * Bg
        self.shape = _lib.Shape(R, Bg, Cin, Co, N, T)
        self.gstruct, self.keep = graph.on(dev)
        self.mode = L.msgat_gacn_mode(Cin, Co)
        nnz = max(graph.nnz, 1)
        if self.mode == _lib.MODE_PROJ_FIRST:
            n_u = G * Co * N * T
        elif self.mode == _lib.MODE_AGG_FIRST and need_bwd:
            n_u = G * Cin * N * T
        else:
            n_u = 0
        nscratch = int(L.msgat_edge_scratch_floats(C.byref(self.shape), C.byref(self.gstruct)))  # E in the SELL layout's order
        # (the SELL scratch is the forward's own workspace: allocated per call and dropped at its end, not carved from the
        # saved buffer -- it is at least the size of E and would stay resident until backward for every GACN layer)
        self.nscratch = nscratch
        # operand images of the score pass on large graphs (msgat_dense_scratch_bytes; 0 at PEMS sizes): per call as well
        self.ndense = int(L.msgat_dense_scratch_bytes(C.byref(self.shape)))
        self.sizes = (G * N * T, G * N * T, G * N, G * N * T if need_bwd else 0, G * nnz, G * nnz if need_bwd else 0, n_u)
        offs, total = [], 0
        for n in self.sizes:                 # 256-byte aligned pieces
            offs.append(total if n else -1)
            total += (n + 63) & ~63
        self.offs, self.total = tuple(offs), max(total, 64)
        self.bwd_bytes = None                # asked for by the first backward
        self.z_channels = Co if Co else Cin


def _gacn_plan(graph, dev, R, Bg, Cin, Co, N, T, need_bwd) -> _GacnPlan:
    # the plans live ON the graph object and die with it (a module-level table keyed by id(graph) kept every graph a
    # process had ever used, and its device tensors, alive)
    plans = graph.__dict__.get("_gacn_plans")
    if plans is None:
        plans = graph.__dict__["_gacn_plans"] = {}
    key = (dev.index, R, Bg, Cin, Co, N, T, need_bwd)
    plan = plans.get(key)
    if plan is None:
        if len(plans) > 256:
            plans.clear()
        plan = plans[key] = _GacnPlan(graph, dev, R, Bg, Cin, Co, N, T, need_bwd)
    return plan


class _GACNFunction(torch.autograd.Function):
    """x[G,C,N,T], alpha[R,C], Wg[R,T,T], W[R,Co,C] or None -> z[G,Co|C,N,T];  G = R*Bg."""

    @staticmethod
    def forward(ctx, x, alpha, Wg, W, graph: SparseGraph, recording: bool = True):
        L = _lib.lib()
        dev = x.device
        G, Cin, N, T = x.shape
        R = alpha.shape[0] if alpha.dim() == 2 else 1     # alpha [C], Wg [T,T], W [Co,C]: ONE relation, no leading axis
        if G % R != 0:
            raise ValueError(f"{G} groups cannot be split over {R} relations")
        Co = 0 if W is None else W.shape[-2]
        # `needs_input_grad` is True under torch.no_grad() as well (it mirrors requires_grad); whether a graph is being
        # recorded is known only to the caller (inside forward grad mode is always off).  Inference then skips everything
        # backward alone needs: pq (4 of the 7 matrix-core instructions per score tile), E in CSC order, the saved y.
        need_bwd = bool(recording) and any(ctx.needs_input_grad)
        plan = _gacn_plan(graph, dev, R, G // R, Cin, Co, N, T, need_bwd)

        if not x.is_contiguous():
            x = x.contiguous()
        alpha, Wg = alpha.contiguous(), Wg.contiguous()
        W = None if W is None else W.contiguous()

        # Everything backward needs besides the inputs (q, kW, lse, pq, E, E in CSC order, the projected / aggregated
        # features u) and the SELL scratch lives in ONE allocation, addressed by offset: seven `torch.empty` calls
        # per forward were a quarter of its host time, which is what a PEMSD4-sized step is bound by.
        z = torch.empty((G, plan.z_channels, N, T), device=dev, dtype=torch.float32)
        buf = torch.empty(plan.total, device=dev, dtype=torch.float32)
        base = buf.data_ptr()
        q, kW, lse, pq, E, Ec, u = (None if o < 0 else base + 4 * o for o in plan.offs)
        scratch_t = torch.empty(plan.nscratch, device=dev, dtype=torch.float32) if plan.nscratch else None
        scratch = _ptr(scratch_t)
        dense_t = torch.empty(plan.ndense, device=dev, dtype=torch.uint8) if plan.ndense else None
        io = _lib.Fwd(_ptr(x), _ptr(alpha), _ptr(Wg), _ptr(W), _ptr(z), q, kW, lse, pq, E, u, int(need_bwd), scratch, Ec,
                      _ptr(dense_t))
        st = L.msgat_gacn_forward(C.byref(plan.shape), C.byref(plan.gstruct), C.byref(io), _stream_handle(dev))
        _lib.check(st, "msgat_gacn_forward")

        if need_bwd:
            ctx.plan, ctx.has_W = plan, W is not None
            if W is not None:
                ctx.save_for_backward(x, alpha, Wg, buf, W)
            else:
                ctx.save_for_backward(x, alpha, Wg, buf)
        return z

    @staticmethod
    def backward(ctx, dz):
        # All four gradients are computed whatever `ctx.needs_input_grad` says: they are by-products of shared passes, not
        # separate work.  PROJ_FIRST: dW, dalpha AND dx leave ONE launch that reads du, dq and x once
        # (msgat_stage_project_backward); AGG_FIRST: dW rides in the pass that forms dy, dalpha in the transposed aggregate
        # that forms dx; dWg is a [T,T] partial of the row pass that every other gradient needs.  The only input that
        # is ever frozen in the reference's models is none of these (adj, msgat.py:190, gets no gradient here at all), and
        # x always requires one (it is a LayerNorm output with learnable weights, msgat.py:122).
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, alpha, Wg, buf = saved[:4]
        W = saved[4] if ctx.has_W else None
        base = buf.data_ptr()
        plan = ctx.plan
        q, kW, lse, pq, E, Ec, u = (None if o < 0 else base + 4 * o for o in plan.offs)
        dev = x.device
        shape, gstruct = plan.shape, plan.gstruct
        # a gradient that arrives as a channel slice dout[:, a:b] of a wider tensor is read in place where the library
        # can (one 98 MB copy less per GACN at PEMSD7 size), copied otherwise
        dz, dz_gs = _sliced_grad(dz, lambda: L.msgat_bwd_accepts_strided_dz(C.byref(shape), C.byref(gstruct)))

        dx = torch.empty_like(x)
        dalpha = torch.empty_like(alpha)
        dWg = torch.empty_like(Wg)
        dW = None if W is None else torch.empty_like(W)
        if plan.bwd_bytes is None:
            plan.bwd_bytes = max(int(L.msgat_bwd_workspace_bytes(C.byref(shape), C.byref(gstruct))), 256)
        ws = torch.empty(plan.bwd_bytes, device=dev, dtype=torch.uint8)
        io = _lib.Bwd(_ptr(x), _ptr(alpha), _ptr(Wg), _ptr(W), q, kW, lse, pq, E, u, _ptr(dz), _ptr(dx), _ptr(dalpha),
                      _ptr(dWg), _ptr(dW), _ptr(ws), ws.numel(), dz_gs, Ec)
        st = L.msgat_gacn_backward(C.byref(shape), C.byref(gstruct), C.byref(io), _stream_handle(dev))
        _lib.check(st, "msgat_gacn_backward")
        return dx, dalpha, dWg, dW, None, None


def gacn(x: torch.Tensor, alpha: torch.Tensor, Wg: torch.Tensor, W: Optional[torch.Tensor],
         adjacency) -> torch.Tensor:
    """Graph attention (+ channel projection when `W` is given) over R stacked relations.

    x [R*Bg, C, N, T] (relation-major), alpha [R,C], Wg [R,T,T], W [R,Co,C] or None,
    adjacency: dense [N,N] tensor or a prebuilt `SparseGraph`.  Returns [R*Bg, Co|C, N, T].
    One relation may come without the leading axis -- alpha [C], Wg [T,T], W [Co,C], the reference's own parameter
    shapes (attention.py:29-30, msgat.py:23): the module classes call it that way, so that no view nodes sit between
    the parameters and the op (three `unsqueeze` forward and three more nodes backward were a sixth of a call's host time).
    """
    if x.dim() != 4:
        raise ValueError(f"signals must be [batch, channels, nodes, timesteps], got {tuple(x.shape)}")
    _require_device_tensor("signals", x)
    for name, t in (("alpha", alpha), ("Wg", Wg)) + ((("W", W),) if W is not None else ()):
        _require_device_tensor(name, t, x.device)
    G, Cin, N, T = x.shape
    if alpha.dim() == 1:                                  # one relation, the reference's parameter shapes
        if alpha.shape[0] != Cin:
            raise ValueError(f"alpha must be [{Cin}], got {tuple(alpha.shape)}")
        if tuple(Wg.shape) != (T, T):
            raise ValueError(f"Wg must be [{T},{T}], got {tuple(Wg.shape)}")
        if W is not None and (W.dim() != 2 or W.shape[1] != Cin):
            raise ValueError(f"W must be [Co,{Cin}], got {tuple(W.shape)}")
    else:
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
    return _GACNFunction.apply(x, alpha, Wg, W, graph, torch.is_grad_enabled())


def graph_attention(x, alpha, Wg, adjacency):
    """`GraphAttention.forward` (attention.py:32-36) for R stacked relations."""
    return gacn(x, alpha, Wg, None, adjacency)


class _LayerNormTFunction(torch.autograd.Function):
    """x[..., T], weight[T] | None, bias[T] | None -> LayerNorm over the last axis."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu_input: bool = False):
        L = _lib.lib()
        x = x.contiguous()
        T = x.shape[-1]
        rows = x.numel() // T if T else 0
        ctx.relu_input = bool(relu_input)
        y = torch.empty_like(x)
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        R = 1 if w is None else w.numel() // T          # weight [T] or [R,T]: R parameter sets, relation-major rows
        st = L.msgat_layernorm_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, T, eps, R, _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_forward")
        ctx.eps, ctx.has_w, ctx.has_b, ctx.R = eps, weight is not None, bias is not None, R
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
        R = ctx.R
        dw = torch.empty_like(w) if ctx.has_w else None
        db = torch.empty(w.shape if ctx.has_w else (T,), device=x.device, dtype=torch.float32) if ctx.has_b else None
        part = torch.empty(max(int(L.msgat_layernorm_partial_floats(rows, T, R)), 1), device=x.device,
                           dtype=torch.float32)
        st = L.msgat_layernorm_backward(_ptr(x), _ptr(w), _ptr(dy), None, _ptr(dx), _ptr(dw), _ptr(db), _ptr(part),
                                        rows, T, ctx.eps, R, int(ctx.relu_input), _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_backward")
        return dx, dw, db, None, None


class _LayerNormTeeFunction(torch.autograd.Function):
    """x -> (LayerNorm(x), x): the second output is x itself, for the consumer that reads the un-normalised input
    beside the LayerNorm (MEAM's residual convolution, msgat.py:122 and :130).  Routing that use through here lets the
    backward add its gradient inside the LayerNorm-backward kernel instead of in a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu_input: bool = False):
        L = _lib.lib()
        x = x.contiguous()
        T = x.shape[-1]
        rows = x.numel() // T
        ctx.relu_input = bool(relu_input)
        y = torch.empty_like(x)
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        R = 1 if w is None else w.numel() // T
        st = L.msgat_layernorm_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, T, eps, R, _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_forward")
        ctx.eps, ctx.has_w, ctx.has_b, ctx.R = eps, weight is not None, bias is not None, R
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(*([x] + ([w] if w is not None else [])))
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dx_other):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, w = saved[0], (saved[1] if ctx.has_w else None)
        T = x.shape[-1]
        rows = x.numel() // T
        if dy is None:
            if dx_other is not None and ctx.relu_input:
                dx_other = torch.ops.aten.threshold_backward(dx_other.contiguous(), x, 0.0)
            return dx_other, None, None, None, None
        dy = dy.contiguous()
        other = None if dx_other is None else dx_other.contiguous()
        dx = torch.empty_like(x)
        R = ctx.R
        dw = torch.empty_like(w) if ctx.has_w else None
        db = torch.empty(w.shape if ctx.has_w else (T,), device=x.device, dtype=torch.float32) if ctx.has_b else None
        part = torch.empty(max(int(L.msgat_layernorm_partial_floats(rows, T, R)), 1), device=x.device, dtype=torch.float32)
        st = L.msgat_layernorm_backward(_ptr(x), _ptr(w), _ptr(dy), _ptr(other), _ptr(dx), _ptr(dw), _ptr(db), _ptr(part),
                                        rows, T, ctx.eps, R, int(ctx.relu_input), _stream_handle(x.device))
        _lib.check(st, "msgat_layernorm_backward")
        return dx, dw, db, None, None


class _LnPoolTeeFunction(torch.autograd.Function):
    """x -> (LayerNorm(x), x, node_pool(LayerNorm(x), pool_w)): MEAM's first three reads of its input (msgat.py:122-125 with
    attention.py:89 inside CACN) as one autograd node, so that backward adds the pooling's rank-one gradient
    pool_w[n] dpooled[s,t] inside the LayerNorm-backward kernel (msgat_layernorm_backward_pooled) instead of in a pass of
    its own over the activation, next to the residual path's gradient (`_LayerNormTeeFunction`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, relu_input: bool, pool_w):
        L = _lib.lib()
        x = x.contiguous()
        B, Cc, N, T = x.shape
        rows = B * Cc * N
        y = torch.empty_like(x)
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        pw = pool_w.contiguous()
        R = 1 if w is None else w.numel() // T
        stream = _stream_handle(x.device)
        pooled = _new(x, B, Cc, T)
        Rp = pw.numel() // N
        if Rp == R and N >= 64:     # the pooling's sums come out of the LayerNorm pass itself (+ a small launch adding the shares)
            part = _new(x, max(int(L.msgat_layernorm_pool_partial_floats(rows, T, R)), 1))
            _lib.check(L.msgat_layernorm_forward_pooled(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(pw), N, _ptr(pooled), _ptr(part),
                                                        rows, T, eps, R, stream), "msgat_layernorm_forward_pooled")
        else:
            _lib.check(L.msgat_layernorm_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, T, eps, R, stream), "msgat_layernorm_forward")
            _lib.check(L.msgat_node_pool(_ptr(y), _ptr(pw), _ptr(pooled), B * Cc, N, T, Rp, 0, 0, stream), "msgat_node_pool")
        ctx.eps, ctx.relu_input, ctx.has_w, ctx.has_b, ctx.R, ctx.Rp = eps, bool(relu_input), w is not None, b is not None, R, Rp
        ctx.save_for_backward(*([x, y, pw] + ([w] if w is not None else []) + ([b] if b is not None else [])))
        return y, x.view_as(x), pooled

    @staticmethod
    def backward(ctx, dy, dx_other, dpooled):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, y, pw = saved[:3]
        w = saved[3] if ctx.has_w else None
        lnb = saved[3 + int(ctx.has_w)] if ctx.has_b else None      # the LayerNorm's bias: y is rebuilt from x for dpool_w
        B, Cc, N, T = x.shape
        rows = B * Cc * N
        stream = _stream_handle(x.device)
        need = ctx.needs_input_grad
        dpw = None
        fused = dpooled is not None and ctx.Rp == ctx.R           # the pooling's gradients inside the LayerNorm-backward kernel

        def pool_weight_grad():     # its own pass over the stored y (msgat_node_pool_grad_weight)
            g = torch.empty_like(pw)
            part = _new(x, max(int(L.msgat_node_pool_partial_floats(B, Cc, N)), 1))
            _lib.check(L.msgat_node_pool_grad_weight(_ptr(y), _ptr(dpooled.contiguous()), _ptr(g), _ptr(part), B, Cc, N, T, ctx.Rp,
                                                     stream), "msgat_node_pool_grad_weight")
            return g
        if dy is None and dpooled is None:
            if dx_other is not None and ctx.relu_input:
                dx_other = torch.ops.aten.threshold_backward(dx_other.contiguous(), x, 0.0)
            return dx_other, None, None, None, None, None
        dy = torch.zeros_like(x) if dy is None else dy.contiguous()
        other = None if dx_other is None else dx_other.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w) if ctx.has_w else None
        db = torch.empty(w.shape if ctx.has_w else (T,), device=x.device, dtype=torch.float32) if ctx.has_b else None
        part = _new(x, max(int(L.msgat_layernorm_partial_floats(rows, T, ctx.R)), 1))
        if fused:
            want_pw = bool(need[5])
            dpw = torch.empty_like(pw) if want_pw else None
            rows_scratch = _new(x, rows) if want_pw else None
            st = L.msgat_layernorm_backward_pooled(_ptr(x), _ptr(w), _ptr(lnb), _ptr(dy), _ptr(other), _ptr(pw),
                                                   _ptr(dpooled.contiguous()), N, _ptr(dx), _ptr(dw), _ptr(db), _ptr(dpw),
                                                   _ptr(rows_scratch), _ptr(part), rows, T, ctx.eps, ctx.R,
                                                   int(ctx.relu_input), stream)
            _lib.check(st, "msgat_layernorm_backward_pooled")
        else:
            if dpooled is not None:      # parameter-set counts differ: the pooling's gradients in their own passes, then the LayerNorm
                if need[5]:
                    dpw = pool_weight_grad()
                dyp = torch.empty_like(x)
                _lib.check(L.msgat_node_pool_grad_signal(_ptr(pw), _ptr(dpooled.contiguous()), _ptr(dy), _ptr(dyp), B * Cc, N, T,
                                                         ctx.Rp, stream), "msgat_node_pool_grad_signal")
                dy = dyp
            st = L.msgat_layernorm_backward(_ptr(x), _ptr(w), _ptr(dy), _ptr(other), _ptr(dx), _ptr(dw), _ptr(db), _ptr(part),
                                            rows, T, ctx.eps, ctx.R, int(ctx.relu_input), stream)
            _lib.check(st, "msgat_layernorm_backward")
        return dx, dw, db, None, None, dpw


def layer_norm_pool_tee(x: torch.Tensor, weight, bias, eps: float, relu_input: bool, pool_w: torch.Tensor):
    """(layer_norm_t(x), x, node_pool(layer_norm_t(x), pool_w)): `layer_norm_t_tee` followed by `node_pool_tee`, as one
    node whose backward adds the pooling's gradient inside the LayerNorm-backward kernel.  pool_w [N] or [R,N]."""
    _require_device_tensor("signals", x)
    _require_device_tensor("weights", pool_w, x.device)
    T = x.shape[-1]
    if x.dim() != 4 or pool_w.shape[-1] != x.shape[2] or pool_w.dim() > 2 or (pool_w.dim() == 2 and x.shape[0] % pool_w.shape[0]):
        raise ValueError(f"layer_norm_pool_tee: signals {tuple(x.shape)}, pooling weights {tuple(pool_w.shape)}")
    if x.numel() == 0 or not x.requires_grad:
        normed = layer_norm_t(x, weight, bias, eps, relu_input)
        return normed, x, node_pool(normed, pool_w)
    for name, t in (("weight", weight), ("bias", bias)):
        if t is not None:
            _require_device_tensor(name, t, x.device)
            if t.shape[-1] != T or t.dim() > 2 or (t.dim() == 2 and x.shape[0] % t.shape[0]):
                raise ValueError(f"{name} must be [{T}] or [R,{T}] with R dividing the leading axis, got {tuple(t.shape)}")
    return _LnPoolTeeFunction.apply(x, weight, bias, float(eps), bool(relu_input), pool_w)


def layer_norm_t_tee(x: torch.Tensor, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                     eps: float = 1e-5, relu_input: bool = False):
    """(layer_norm_t(x), x): use the second value wherever the block reads its un-normalised input again.
    `relu_input`: see `layer_norm_t` (the mask then covers the gradients of both uses)."""
    _require_device_tensor("signals", x)
    if x.numel() == 0 or not x.requires_grad:
        return layer_norm_t(x, weight, bias, eps, relu_input), x
    T = x.shape[-1]
    for name, t in (("weight", weight), ("bias", bias)):
        if t is not None:
            _require_device_tensor(name, t, x.device)
            if t.shape[-1] != T or t.dim() > 2 or (t.dim() == 2 and x.shape[0] % t.shape[0]):
                raise ValueError(f"{name} must be [{T}] or [R,{T}] with R dividing the leading axis, got {tuple(t.shape)}")
    return _LayerNormTeeFunction.apply(x, weight, bias, float(eps), bool(relu_input))


def layer_norm_t(x: torch.Tensor, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                 eps: float = 1e-5, relu_input: bool = False) -> torch.Tensor:
    """`F.layer_norm(x, [T], weight, bias, eps)` over the timestep axis, the op the reference applies
    to every GACN input (msgat.py:122, :158), as one HBM-speed pass in libmsgat_hip.so.

    `relu_input=True`: x is the output of a ReLU whose backward mask (gradient 0 where x <= 0) this op's backward
    applies -- for a producer that was told to skip it (`mix_multi(..., relu=True, relu_grad_premasked=True)`)."""
    _require_device_tensor("signals", x)
    T = x.shape[-1]
    for name, t in (("weight", weight), ("bias", bias)):
        if t is not None:
            _require_device_tensor(name, t, x.device)
            if t.shape[-1] != T or t.dim() > 2 or (t.dim() == 2 and x.shape[0] % t.shape[0]):
                raise ValueError(f"{name} must be [{T}] or [R,{T}] with R dividing the leading axis, got {tuple(t.shape)}")
    if weight is not None and bias is not None and weight.shape != bias.shape:
        raise ValueError("weight and bias must have the same shape")
    if x.numel() == 0:
        return torch.empty_like(x)
    return _LayerNormTFunction.apply(x, weight, bias, float(eps), bool(relu_input))


# ---- the temporal / channel branches of MEAM and its residual tail -----------

def _new(like: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(shape, device=like.device, dtype=torch.float32)


_ones_cache = {}


def _channel_sums(t: torch.Tensor, R: int = 0) -> torch.Tensor:
    """[G,C,N,T] -> [C] (or [R,C] per relation when R > 0): the bias gradient of a convolution.  One streaming
    pass (node pooling with unit weights) instead of torch's strided reduction kernel (87 us vs 12 us at
    [32,24,883,12])."""
    G, Cc, N, T = t.shape
    key = (t.device, N)
    ones = _ones_cache.get(key)
    if ones is None:
        ones = _ones_cache[key] = torch.ones(N, device=t.device, dtype=torch.float32)
    pooled = _new(t, G, Cc, T)
    keep, seg = _as_segment(t)                    # a channel slice of a wider gradient tensor is read in place
    st = _lib.lib().msgat_node_pool(seg.ptr, _ptr(ones), _ptr(pooled), G * Cc, N, T, 1, Cc if seg.group_stride else 0,
                                    seg.group_stride, _stream_handle(t.device))
    _lib.check(st, "msgat_node_pool")
    if R > 0:
        return pooled.view(R, G // R, Cc, T).sum(dim=(1, 3))
    return pooled.sum(dim=(0, 2))


class _MixFunction(torch.autograd.Function):
    """x[G,Ci,N,T], M[R,Co,Ci] (R | G), bias[Co] | None, add[G,Co,N,T] | None -> relu?(M x + bias + add)."""

    @staticmethod
    def forward(ctx, x, M, bias, add, relu: bool):
        L = _lib.lib()
        x, M = x.contiguous(), M.contiguous()
        G, Ci, N, T = x.shape
        R, Co = M.shape[0], M.shape[1]
        shape = _lib.Shape(R, G // R, Ci, Co, N, T)
        out = _new(x, G, Co, N, T)
        b = None if bias is None else bias.contiguous()
        a = None if add is None else add.contiguous()
        st = L.msgat_stage_mix_epilogue(C.byref(shape), Ci, Co, _ptr(x), _ptr(M), 0, _ptr(b), 0, _ptr(a), int(relu),
                                        _ptr(out), _stream_handle(x.device))
        _lib.check(st, "msgat_stage_mix_epilogue")
        ctx.relu, ctx.has_bias, ctx.has_add = relu, bias is not None, add is not None
        ctx.save_for_backward(x, M, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, M, out = ctx.saved_tensors
        G, Ci, N, T = x.shape
        R, Co = M.shape[0], M.shape[1]
        # ReLU mask in one pass: aten's threshold_backward is dout where out > 0, else 0
        dpre = torch.ops.aten.threshold_backward(dout.contiguous(), out, 0.0) if ctx.relu else dout.contiguous()
        stream = _stream_handle(x.device)
        need = ctx.needs_input_grad
        dx = dM = dbias = None
        if need[0]:
            dx = torch.empty_like(x)
            shape = _lib.Shape(R, G // R, Co, Ci, N, T)
            st = L.msgat_stage_mix(C.byref(shape), Co, Ci, _ptr(dpre), _ptr(M), 1, None, None, _ptr(dx), stream)
            _lib.check(st, "msgat_stage_mix")
        if need[1]:
            dM = torch.empty_like(M)
            shape = _lib.Shape(R, G // R, Ci, Co, N, T)
            part = _new(x, max(int(L.msgat_contract_partial_floats(C.byref(shape), Co, Ci)), 1))
            st = L.msgat_stage_contract(C.byref(shape), Co, Ci, _ptr(dpre), None, _ptr(x), _ptr(part), _ptr(dM),
                                        Co * Ci, None, 0, stream)
            _lib.check(st, "msgat_stage_contract")
        if ctx.has_bias and need[2]:
            dbias = _channel_sums(dpre)
        return dx, dM, dbias, (dpre if ctx.has_add and need[3] else None), None


def mix(x: torch.Tensor, M: torch.Tensor, bias: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
        relu: bool = False) -> torch.Tensor:
    """out[g,o] = relu?(sum_c M[r,o,c] x[g,c] + bias[o] + add[g,o]), r = g // (G/R): a 1x1 convolution (R = 1), a
    per-sample channel matrix (R = batch), or MEAM's tail relu(cat(branches) + res(x)) (msgat.py:130-131)."""
    _require_device_tensor("signals", x)
    _require_device_tensor("matrix", M, x.device)
    if x.dim() != 4 or M.dim() != 3 or M.shape[2] != x.shape[1] or x.shape[0] % M.shape[0]:
        raise ValueError(f"mix: signals {tuple(x.shape)} and matrix {tuple(M.shape)} do not match")
    if bias is not None and tuple(bias.shape) != (M.shape[1],):
        raise ValueError(f"bias must be [{M.shape[1]}]")
    if add is not None and tuple(add.shape) != (x.shape[0], M.shape[1], x.shape[2], x.shape[3]):
        raise ValueError(f"add must be [{x.shape[0]},{M.shape[1]},{x.shape[2]},{x.shape[3]}], got {tuple(add.shape)}")
    return _MixFunction.apply(x, M, bias, add, bool(relu))


class _TimeMixFunction(torch.autograd.Function):
    """y[G,K*Co,N,T], A[G|1,K,T,T], bias[Co] | None -> out[g,o,n,t] = bias[o] + sum_k sum_i A[g,k,t,i] y[g,k*Co+o,n,i]."""

    @staticmethod
    def forward(ctx, y, A, bias):
        L = _lib.lib()
        y, A = y.contiguous(), A.contiguous()
        G, KC, N, T = y.shape
        K = A.shape[1]
        Co = KC // K
        per_group = int(A.shape[0] != 1 or G == 1)
        out = _new(y, G, Co, N, T)
        b = None if bias is None else bias.contiguous()
        Rb = 1 if b is None or b.dim() == 1 else b.shape[0]     # bias [Co] or [R,Co]
        st = L.msgat_time_mix(_ptr(y), _ptr(A), per_group, _ptr(b), _ptr(out), G, Co, K, N, T, 0, Rb, 0,
                              _stream_handle(y.device))
        _lib.check(st, "msgat_time_mix")
        ctx.dims, ctx.per_group, ctx.has_bias = (G, Co, K, N, T), per_group, bias is not None
        ctx.bias_R = 0 if b is None or b.dim() == 1 else b.shape[0]
        ctx.save_for_backward(y, A)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        y, A = ctx.saved_tensors
        G, Co, K, N, T = ctx.dims
        dout, seg = _as_segment(dout)              # a channel slice of the block's concatenated gradient is read in place
        stream = _stream_handle(y.device)
        need = ctx.needs_input_grad
        dy = dA = dbias = None
        if need[0]:
            dy = torch.empty_like(y)
            st = L.msgat_time_mix(seg.ptr, _ptr(A), ctx.per_group, None, _ptr(dy), G, Co, K, N, T, 1, 1, seg.group_stride,
                                  stream)
            _lib.check(st, "msgat_time_mix (backward)")
        if need[1]:
            dAg = _new(y, G, K, T, T)
            part = _new(y, max(int(L.msgat_time_mix_partial_floats(G, K, T)), 1))
            st = L.msgat_time_mix_grad_matrix(seg.ptr, _ptr(y), _ptr(dAg), _ptr(part), G, Co, K, N, T, seg.group_stride,
                                              stream)
            _lib.check(st, "msgat_time_mix_grad_matrix")
            dA = dAg if A.shape[0] == G else dAg.sum(dim=0, keepdim=True)
        if ctx.has_bias and need[2]:
            dbias = _channel_sums(dout, ctx.bias_R)
        return dy, dA, dbias


def time_mix(y: torch.Tensor, A: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Applies K [T,T] matrices along the time axis of K channel groups and adds them up (see
    include/msgat_hip.h: msgat_time_mix).  K = 1 with A = att is TemporalAttention's product
    (attention.py:66); K = 2 is a causal dilated [1,2] convolution after its channel mixing."""
    _require_device_tensor("signals", y)
    _require_device_tensor("matrices", A, y.device)
    if y.dim() != 4 or A.dim() != 4 or A.shape[2] != y.shape[3] or A.shape[3] != y.shape[3]:
        raise ValueError(f"time_mix: signals {tuple(y.shape)} and matrices {tuple(A.shape)} do not match")
    if A.shape[0] not in (1, y.shape[0]) or y.shape[1] % A.shape[1]:
        raise ValueError(f"time_mix: {tuple(A.shape)} matrices for signals {tuple(y.shape)}")
    return _TimeMixFunction.apply(y, A, bias)


class _CausalConvFunction(torch.autograd.Function):
    """h[G,Ci,N,T], taps[R,2Co,Ci] = [W0; W1], bias[Co] | [R,Co] | None -> bias + W0 h[t-d] + W1 h[t]: one pass each way
    (msgat_causal_conv); the weight gradient contracts [dout[t+d]; dout] with h."""

    @staticmethod
    def forward(ctx, h, taps, bias, dilation: int):
        L = _lib.lib()
        h, taps = h.contiguous(), taps.contiguous()
        G, Ci, N, T = h.shape
        R, Co = taps.shape[0], taps.shape[1] // 2
        out = _new(h, G, Co, N, T)
        b = None if bias is None else bias.contiguous()
        st = L.msgat_causal_conv(_ptr(h), _ptr(taps), _ptr(b), int(b is not None and b.dim() == 2), _ptr(out), R, G // R,
                                 Ci, Co, N, T, int(dilation), 0, 0, _stream_handle(h.device))
        _lib.check(st, "msgat_causal_conv")
        ctx.dilation, ctx.bias_R = int(dilation), (None if b is None else (b.shape[0] if b.dim() == 2 else 0))
        ctx.save_for_backward(h, taps)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        h, taps = ctx.saved_tensors
        G, Ci, N, T = h.shape
        R, Co = taps.shape[0], taps.shape[1] // 2
        keep, seg = _as_segment(dout)             # a channel slice of the block's concatenated gradient is read in place
        stream = _stream_handle(h.device)
        need = ctx.needs_input_grad
        dh = dtaps = dbias = None
        if need[0]:
            dh = torch.empty_like(h)
            st = L.msgat_causal_conv(seg.ptr, _ptr(taps), None, 0, _ptr(dh), R, G // R, Ci, Co, N, T, ctx.dilation, 1,
                                     seg.group_stride, stream)
            _lib.check(st, "msgat_causal_conv (backward)")
        want_bias = ctx.bias_R is not None and need[2]
        if need[1] or want_bias:
            # d[W0; W1] = [dout[t+d]; dout] h^T with the 2 Co gradient rows read as time-shifted views of dout inside the
            # contraction (and a virtual channel of ones: the bias gradient is its tap-1 half)
            ones = int(bool(want_bias))
            buf = _new(h, R * 2 * Co * (Ci + ones))
            part = _new(h, max(int(L.msgat_contract_segments_partial_floats(R, 2 * Co, Ci + ones)), 1))
            # with_ones = 2: the matrix [R,2Co,Ci] and, behind it, the ones column [R,2Co] (no slice copies)
            st = L.msgat_causal_conv_grad_weight(seg.ptr, seg.group_stride, _ptr(h), _ptr(part), _ptr(buf), R, G // R, Ci, Co,
                                                 N, T, ctx.dilation, 2 * ones, stream)
            _lib.check(st, "msgat_causal_conv_grad_weight")
            dM = buf[: R * 2 * Co * Ci].view(R, 2 * Co, Ci)
            if ones:
                colsum = buf[R * 2 * Co * Ci:].view(R, 2, Co)[:, 1]      # [R,Co]: sum of dout over the relation's groups and positions
                dbias = colsum if ctx.bias_R else colsum.sum(dim=0)
            dtaps = dM if need[1] else None
        return dh, dtaps, dbias, None


def causal_conv_fused(Ci: int, Co: int) -> bool:
    """Whether `causal_conv` has its one-pass kernels for these widths (else use mix_multi + time_mix)."""
    return bool(_lib.lib().msgat_causal_conv_fused(int(Ci), int(Co)))


def causal_conv(h: torch.Tensor, taps: torch.Tensor, bias: Optional[torch.Tensor], dilation: int) -> torch.Tensor:
    """A causal dilated [1,2] convolution -- `Conv2d(Ci, Co, [1,2], padding=[0,d], dilation=[1,d])` + `Chomp(d)`,
    msgat.py:69-74 -- with both taps stacked on the output axis, taps [R, 2*Co, Ci] = [W0; W1] (W0 acts on h[t-d]):
    out = bias + W0 h[t-d] + W1 h[t], R parameter sets over R*Bg groups, in one pass over h."""
    _require_device_tensor("signals", h)
    _require_device_tensor("taps", taps, h.device)
    if h.dim() != 4 or taps.dim() != 3 or taps.shape[2] != h.shape[1] or taps.shape[1] % 2 or h.shape[0] % taps.shape[0]:
        raise ValueError(f"causal_conv: signals {tuple(h.shape)} and taps {tuple(taps.shape)} do not match")
    Co = taps.shape[1] // 2
    if bias is not None and tuple(bias.shape) not in ((Co,), (taps.shape[0], Co)):
        raise ValueError(f"bias must be [{Co}] or [{taps.shape[0]},{Co}]")
    if dilation <= 0:
        raise ValueError("dilation must be positive")
    return _CausalConvFunction.apply(h, taps, bias, int(dilation))


class _NodePoolFunction(torch.autograd.Function):
    """x[B,C,N,T], w[N] -> pooled[B,C,T] = sum_n w[n] x[b,c,n,:]   (attention.py:89)."""

    @staticmethod
    def forward(ctx, x, w):
        L = _lib.lib()
        x, w = x.contiguous(), w.contiguous()
        B, Cc, N, T = x.shape
        pooled = _new(x, B, Cc, T)
        R = w.numel() // N                                   # weights [N] or [R,N]
        st = L.msgat_node_pool(_ptr(x), _ptr(w), _ptr(pooled), B * Cc, N, T, R, 0, 0, _stream_handle(x.device))
        _lib.check(st, "msgat_node_pool")
        ctx.save_for_backward(x, w)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        B, Cc, N, T = x.shape
        dp = dp.contiguous()
        stream = _stream_handle(x.device)
        R = w.numel() // N
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.msgat_node_pool_grad_signal(_ptr(w), _ptr(dp), None, _ptr(dx), B * Cc, N, T, R, stream),
                       "msgat_node_pool_grad_signal")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            part = _new(x, max(int(L.msgat_node_pool_partial_floats(B, Cc, N)), 1))
            _lib.check(L.msgat_node_pool_grad_weight(_ptr(x), _ptr(dp), _ptr(dw), _ptr(part), B, Cc, N, T, R, stream),
                       "msgat_node_pool_grad_weight")
        return dx, dw


class _NodePoolTeeFunction(torch.autograd.Function):
    """x, w -> (node_pool(x, w), x): like `_LayerNormTeeFunction`, for the activation that is pooled AND read by another
    consumer (MEAM's normalised input feeds the channel attention's pooling and the channel-mixing pass): the other
    consumer's gradient comes back through the second output and joins inside the pooling's backward kernel."""

    @staticmethod
    def forward(ctx, x, w):
        L = _lib.lib()
        x, w = x.contiguous(), w.contiguous()
        B, Cc, N, T = x.shape
        pooled = _new(x, B, Cc, T)
        R = w.numel() // N
        st = L.msgat_node_pool(_ptr(x), _ptr(w), _ptr(pooled), B * Cc, N, T, R, 0, 0, _stream_handle(x.device))
        _lib.check(st, "msgat_node_pool")
        ctx.save_for_backward(x, w)
        return pooled, x.view_as(x)

    @staticmethod
    def backward(ctx, dp, dx_other):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        B, Cc, N, T = x.shape
        stream = _stream_handle(x.device)
        R = w.numel() // N
        if dp is None:
            return dx_other, None
        dp = dp.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            other = None if dx_other is None else dx_other.contiguous()
            dx = torch.empty_like(x)
            _lib.check(L.msgat_node_pool_grad_signal(_ptr(w), _ptr(dp), _ptr(other), _ptr(dx), B * Cc, N, T, R, stream),
                       "msgat_node_pool_grad_signal")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            part = _new(x, max(int(L.msgat_node_pool_partial_floats(B, Cc, N)), 1))
            _lib.check(L.msgat_node_pool_grad_weight(_ptr(x), _ptr(dp), _ptr(dw), _ptr(part), B, Cc, N, T, R, stream),
                       "msgat_node_pool_grad_weight")
        return dx, dw


def node_pool_tee(x: torch.Tensor, w: torch.Tensor):
    """(node_pool(x, w), x): use the second value for the other consumer of x."""
    pooled_only = not x.requires_grad
    if pooled_only:
        return node_pool(x, w), x
    _require_device_tensor("signals", x)
    _require_device_tensor("weights", w, x.device)
    if x.dim() != 4 or w.shape[-1] != x.shape[2] or w.dim() > 2 or (w.dim() == 2 and x.shape[0] % w.shape[0]):
        raise ValueError(f"node_pool: signals {tuple(x.shape)}, weights {tuple(w.shape)}")
    return _NodePoolTeeFunction.apply(x, w)


def node_pool(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _require_device_tensor("signals", x)
    _require_device_tensor("weights", w, x.device)
    if x.dim() != 4 or w.shape[-1] != x.shape[2] or w.dim() > 2 or (w.dim() == 2 and x.shape[0] % w.shape[0]):
        raise ValueError(f"node_pool: signals {tuple(x.shape)}, weights {tuple(w.shape)}")
    return _NodePoolFunction.apply(x, w)


class _ChannelPoolFunction(torch.autograd.Function):
    """x[B,C,N,T], alpha[C] -> mixed[B,N,T] = sum_c alpha[c] x[b,c]   (attention.py:59; the hot path's q)."""

    @staticmethod
    def forward(ctx, x, alpha):
        L = _lib.lib()
        x, alpha = x.contiguous(), alpha.contiguous()
        B, Cc, N, T = x.shape
        R = alpha.numel() // Cc                              # alpha [C] or [R,C]
        shape = _lib.Shape(R, B // R, Cc, 0, N, T)
        q = _new(x, B, N, T)
        st = L.msgat_stage_project(C.byref(shape), _ptr(x), _ptr(alpha), None, _ptr(q), None, _stream_handle(x.device))
        _lib.check(st, "msgat_stage_project")
        ctx.save_for_backward(x, alpha)
        return q

    @staticmethod
    def backward(ctx, dq):
        L = _lib.lib()
        x, alpha = ctx.saved_tensors
        B, Cc, N, T = x.shape
        dq = dq.contiguous()
        stream = _stream_handle(x.device)
        R = alpha.numel() // Cc
        dx = dalpha = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            shape = _lib.Shape(R, B // R, 1, Cc, N, T)
            st = L.msgat_stage_mix(C.byref(shape), 1, Cc, _ptr(dq), _ptr(alpha), 0, None, None, _ptr(dx), stream)
            _lib.check(st, "msgat_stage_mix")
        if ctx.needs_input_grad[1]:
            dalpha = torch.empty_like(alpha)
            shape = _lib.Shape(R, B // R, Cc, 0, N, T)
            part = _new(x, max(int(L.msgat_contract_partial_floats(C.byref(shape), 1, Cc)), 1))
            st = L.msgat_stage_contract(C.byref(shape), 1, Cc, None, _ptr(dq), _ptr(x), _ptr(part), _ptr(dalpha), Cc,
                                        None, 0, stream)
            _lib.check(st, "msgat_stage_contract")
        return dx, dalpha


def channel_pool(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    _require_device_tensor("signals", x)
    _require_device_tensor("alpha", alpha, x.device)
    if x.dim() != 4 or alpha.shape[-1] != x.shape[1] or alpha.dim() > 2 or (alpha.dim() == 2 and x.shape[0] % alpha.shape[0]):
        raise ValueError(f"channel_pool: signals {tuple(x.shape)}, alpha {tuple(alpha.shape)}")
    return _ChannelPoolFunction.apply(x, alpha)


class _HeadFunction(torch.autograd.Function):
    """x[B,C,N,T], W[To,T,1,C], bias[To] | None -> out[B,N,To] = bias + sum_{c,t} W[o,t,0,c] x[b,c,n,t]  (msgat.py:153,:159)."""

    @staticmethod
    def forward(ctx, x, W, bias):
        L = _lib.lib()
        x, W = x.contiguous(), W.contiguous()
        B, Cc, N, T = x.shape
        To = W.shape[-4]
        R = 1 if W.dim() == 4 else W.shape[0]               # weight [To,T,1,C] or [R,To,T,1,C]
        out = _new(x, B, N, To)
        part = _new(x, max(int(L.msgat_head_forward_partial_floats(B, Cc, N, To)), 1))
        b = None if bias is None else bias.contiguous()
        st = L.msgat_head_forward(_ptr(x), _ptr(W), _ptr(b), _ptr(out), _ptr(part), B, Cc, N, T, To, R,
                                  _stream_handle(x.device))
        _lib.check(st, "msgat_head_forward")
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, W = ctx.saved_tensors
        B, Cc, N, T = x.shape
        To = W.shape[-4]
        R = 1 if W.dim() == 4 else W.shape[0]
        dout = dout.contiguous()
        stream = _stream_handle(x.device)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.msgat_head_grad_signal(_ptr(dout), _ptr(W), _ptr(dx), B, Cc, N, T, To, R, stream),
                       "msgat_head_grad_signal")
        if ctx.needs_input_grad[1]:
            dWc = _new(x, R, Cc, To, T)
            part = _new(x, max(int(L.msgat_head_grad_weight_partial_floats(Cc, T, To, R)), 1))
            _lib.check(L.msgat_head_grad_weight(_ptr(dout), _ptr(x), _ptr(dWc), _ptr(part), B, Cc, N, T, To, R, stream),
                       "msgat_head_grad_weight")
            # [R,To,T,1,C], made contiguous HERE, once for all R components: each component's parameter receives its row,
            # and a non-contiguous row would be copied per component when it is accumulated into .grad
            dW = dWc.permute(0, 2, 3, 1).contiguous().unsqueeze(3)
            if W.dim() == 4:
                dW = dW[0]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if To in (4, 8, 12, 16):   # one streaming pass over dout (node pooling with unit weights), then a tiny sum
                key = (dout.device, N)
                ones = _ones_cache.get(key)
                if ones is None:
                    ones = _ones_cache[key] = torch.ones(N, device=dout.device, dtype=torch.float32)
                pooled = _new(dout, B, To)
                _lib.check(L.msgat_node_pool(_ptr(dout), _ptr(ones), _ptr(pooled), B, N, To, 1, 0, 0, stream), "msgat_node_pool")
                db = pooled.view(R, B // R, To).sum(dim=1) if W.dim() == 5 else pooled.sum(dim=0)
            else:
                db = dout.view(R, B // R, N, To).sum(dim=(1, 2)) if W.dim() == 5 else dout.sum(dim=(0, 1))
        return dx, dW, db


class _LnHeadFunction(torch.autograd.Function):
    """x[B,C,N,T] -> head(layer_norm_t(x)): a component's last two steps (msgat.py:158-160).  Forward is one pass over x:
    the head kernel normalises the rows it loads (msgat_head_forward_ln) and writes LayerNorm(x) only when the weight
    gradient will need it.  Backward builds the head's input gradient in registers inside the LayerNorm-backward pass
    (msgat_layernorm_head_backward), so the [B,C,N,T] gradient between the two is never written."""

    @staticmethod
    def forward(ctx, x, lnw, lnb, eps: float, W, bias, relu_input: bool):
        L = _lib.lib()
        x, W = x.contiguous(), W.contiguous()
        B, Cc, N, T = x.shape
        To = W.shape[-4]
        R = 1 if W.dim() == 4 else W.shape[0]
        stream = _stream_handle(x.device)
        w = None if lnw is None else lnw.contiguous()
        lb = None if lnb is None else lnb.contiguous()
        out = _new(x, B, N, To)
        part = _new(x, max(int(L.msgat_head_forward_partial_floats(B, Cc, N, To)), 1))
        b = None if bias is None else bias.contiguous()
        # the head normalises what it loads; LayerNorm(x) is written only when the weight gradient will read it
        keep = ctx.needs_input_grad[4]
        xn = torch.empty_like(x) if keep else None
        st = L.msgat_head_forward_ln(_ptr(x), _ptr(w), _ptr(lb), eps, _ptr(W), _ptr(b), _ptr(out), _ptr(xn), _ptr(part), B, Cc, N, T,
                                     To, R, stream)
        _lib.check(st, "msgat_head_forward_ln")
        ctx.eps, ctx.relu_input, ctx.has_lnw, ctx.has_lnb, ctx.has_bias = eps, bool(relu_input), w is not None, lb is not None, bias is not None
        ctx.has_xn = keep
        ctx.save_for_backward(*([x, W] + ([xn] if keep else []) + ([w] if w is not None else [])))
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        saved = ctx.saved_tensors
        x, W = saved[:2]
        xn = saved[2] if ctx.has_xn else None
        w = saved[2 + int(ctx.has_xn)] if ctx.has_lnw else None
        B, Cc, N, T = x.shape
        To = W.shape[-4]
        R = 1 if W.dim() == 4 else W.shape[0]
        dout = dout.contiguous()
        stream = _stream_handle(x.device)
        need = ctx.needs_input_grad
        dx = dlnw = dlnb = dW = db = None
        if need[0] or need[1] or need[2]:
            dx = torch.empty_like(x)
            dlnw = torch.empty_like(w) if (ctx.has_lnw and need[1]) else None
            dlnb = torch.empty(w.shape if ctx.has_lnw else (T,), device=x.device, dtype=torch.float32) if (ctx.has_lnb and need[2]) else None
            part = _new(x, max(int(L.msgat_layernorm_head_backward_partial_floats(B, Cc, N, T)), 1))
            st = L.msgat_layernorm_head_backward(_ptr(dout), _ptr(W), _ptr(x), _ptr(w), _ptr(dx), _ptr(dlnw), _ptr(dlnb),
                                                 _ptr(part), B, Cc, N, T, To, R, ctx.eps, int(ctx.relu_input), stream)
            _lib.check(st, "msgat_layernorm_head_backward")
        if need[4]:
            dWc = _new(x, R, Cc, To, T)
            part = _new(x, max(int(L.msgat_head_grad_weight_partial_floats(Cc, T, To, R)), 1))
            _lib.check(L.msgat_head_grad_weight(_ptr(dout), _ptr(xn), _ptr(dWc), _ptr(part), B, Cc, N, T, To, R, stream),
                       "msgat_head_grad_weight")
            dW = dWc.permute(0, 2, 3, 1).contiguous().unsqueeze(3)     # [R,To,T,1,C], contiguous once (see _HeadFunction)
            if W.dim() == 4:
                dW = dW[0]
        if ctx.has_bias and need[5]:
            if To in (4, 8, 12, 16):
                key = (dout.device, N)
                ones = _ones_cache.get(key)
                if ones is None:
                    ones = _ones_cache[key] = torch.ones(N, device=dout.device, dtype=torch.float32)
                pooled = _new(dout, B, To)
                _lib.check(L.msgat_node_pool(_ptr(dout), _ptr(ones), _ptr(pooled), B, N, To, 1, 0, 0, stream), "msgat_node_pool")
                db = pooled.view(R, B // R, To).sum(dim=1) if W.dim() == 5 else pooled.sum(dim=0)
            else:
                db = dout.view(R, B // R, N, To).sum(dim=(1, 2)) if W.dim() == 5 else dout.sum(dim=(0, 1))
        return dx if need[0] else None, dlnw, dlnb, None, dW, db, None


def ln_head(x: torch.Tensor, ln_weight: Optional[torch.Tensor], ln_bias: Optional[torch.Tensor], eps: float,
            W: torch.Tensor, bias: Optional[torch.Tensor] = None, relu_input: bool = False) -> torch.Tensor:
    """`head(layer_norm_t(x, ln_weight, ln_bias, eps, relu_input), W, bias)` -- TPC's last two steps, msgat.py:158-160 --
    with ONE backward pass over x for the head's input gradient and the LayerNorm backward together."""
    _require_device_tensor("signals", x)
    _require_device_tensor("weight", W, x.device)
    T = x.shape[-1]
    if (x.dim() != 4 or W.dim() not in (4, 5) or W.shape[-3] != T or W.shape[-2] != 1 or W.shape[-1] != x.shape[1]
            or (W.dim() == 5 and x.shape[0] % W.shape[0])):
        raise ValueError(f"ln_head: signals {tuple(x.shape)} and weight {tuple(W.shape)} do not match")
    for name, t in (("ln_weight", ln_weight), ("ln_bias", ln_bias)):
        if t is not None:
            _require_device_tensor(name, t, x.device)
            if t.shape[-1] != T or t.dim() > 2 or (t.dim() == 2 and x.shape[0] % t.shape[0]):
                raise ValueError(f"{name} must be [{T}] or [R,{T}] with R dividing the leading axis, got {tuple(t.shape)}")
    if ln_weight is not None and W.dim() == 5 and ln_weight.dim() == 2 and ln_weight.shape[0] != W.shape[0]:
        raise ValueError("ln_head: the LayerNorm and the head must have the same number of parameter sets")
    if x.numel() == 0 or (ln_weight is not None and ln_weight.numel() // T != (1 if W.dim() == 4 else W.shape[0])):
        return head(layer_norm_t(x, ln_weight, ln_bias, eps, relu_input), W, bias)   # mixed parameter-set counts: the two ops
    return _LnHeadFunction.apply(x, ln_weight, ln_bias, float(eps), W, bias, bool(relu_input))


def head(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The component's prediction head: `fc(x.transpose(1, 3))[..., 0].transpose(1, 2)` of msgat.py:159-160
    with `fc = Conv2d(T, T_out, [1, C])`, as one pass over x."""
    _require_device_tensor("signals", x)
    _require_device_tensor("weight", W, x.device)
    if (x.dim() != 4 or W.dim() not in (4, 5) or W.shape[-3] != x.shape[3] or W.shape[-2] != 1 or W.shape[-1] != x.shape[1]
            or (W.dim() == 5 and x.shape[0] % W.shape[0])):
        raise ValueError(f"head: signals {tuple(x.shape)} and weight {tuple(W.shape)} do not match")
    return _HeadFunction.apply(x, W, bias)


class _GateSumFunction(torch.autograd.Function):
    """pred [R,B,N,To], H [B], D [B], h_w [nh, R*N*To], d_w [nd, R*N*To] -> sum_r pred_r * (h_w[H] + d_w[D])_r
    (msgat.py:203-205 with embeddings.py:36-39 inside); H = D = d_w = None: the static gate h_w = W [R,N,To]."""

    @staticmethod
    def forward(ctx, pred, H, D, h_w, d_w):
        L = _lib.lib()
        pred, h_w = pred.contiguous(), h_w.contiguous()
        d_w = None if d_w is None else d_w.contiguous()
        R, B = pred.shape[:2]
        E = pred[0, 0].numel()
        nh = h_w.shape[0] if H is not None else 1
        nd = d_w.shape[0] if d_w is not None else 0
        out = _new(pred, *pred.shape[1:])
        st = L.msgat_gate_sum(_ptr(pred), _ptr(H), _ptr(D), _ptr(h_w), _ptr(d_w), _ptr(out), R, B, E, nh, nd,
                              _stream_handle(pred.device))
        _lib.check(st, "msgat_gate_sum")
        ctx.dims = (R, B, E, nh, nd)
        ctx.has_idx, ctx.has_day = H is not None, d_w is not None
        ctx.save_for_backward(*([pred, h_w] + ([H, D] if H is not None else []) + ([d_w] if d_w is not None else [])))
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        saved = list(ctx.saved_tensors)
        pred, h_w = saved[:2]
        H, D = (saved[2], saved[3]) if ctx.has_idx else (None, None)
        d_w = saved[-1] if ctx.has_day else None
        R, B, E, nh, nd = ctx.dims
        need = ctx.needs_input_grad
        dout = dout.contiguous()
        dpred = torch.empty_like(pred) if need[0] else None
        dh = torch.empty_like(h_w) if need[3] else None
        dd = torch.empty_like(d_w) if (d_w is not None and need[4]) else None
        st = L.msgat_gate_sum_backward(_ptr(dout), _ptr(pred), _ptr(H), _ptr(D), _ptr(h_w), _ptr(d_w), _ptr(dpred), _ptr(dh),
                                       _ptr(dd), R, B, E, nh, nd, _stream_handle(pred.device))
        _lib.check(st, "msgat_gate_sum_backward")
        return dpred, None, None, dh, dd


def gate_sum(pred: torch.Tensor, H: Optional[torch.Tensor], D: Optional[torch.Tensor], h_weight: torch.Tensor,
             d_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`sum_r pred[r] * gate[:, r]` with `gate = (h_ebd(H) + d_ebd(D)).view(B, R, N, T_out)` -- the model's last line
    (msgat.py:203-205, embeddings.py:36-39) -- as one launch forward and one backward; the embedding tables receive
    dense gradients, as nn.Embedding's do.  `H = D = d_weight = None`: the static gate `h_weight = W [R,N,T_out]`
    (msgat.py:189).  Indices outside a table are clamped to its ends (torch's gather traps on the device)."""
    _require_device_tensor("pred", pred)
    _require_device_tensor("h_weight", h_weight, pred.device)
    if pred.dim() < 3:
        raise ValueError(f"gate_sum: pred {tuple(pred.shape)} must be [R, B, ...]")
    R, B = pred.shape[:2]
    RE = pred[:, 0].numel()
    if (H is None) != (D is None) or (H is None) != (d_weight is None):
        raise ValueError("gate_sum: H, D and d_weight come together (time embedding) or not at all (static gate)")
    if H is None:
        if h_weight.numel() != RE:
            raise ValueError(f"gate_sum: static gate {tuple(h_weight.shape)} does not match pred {tuple(pred.shape)}")
    else:
        _require_device_tensor("d_weight", d_weight, pred.device)
        if H.dtype in (torch.int32, torch.int16, torch.uint8) and D.dtype in (torch.int32, torch.int16, torch.uint8):
            H, D = H.long(), D.long()                      # nn.Embedding takes these too
        if (h_weight.dim() != 2 or d_weight.dim() != 2 or h_weight.shape[1] != RE or d_weight.shape[1] != RE
                or H.shape != (B,) or D.shape != (B,) or H.dtype != torch.int64 or D.dtype != torch.int64
                or H.device != pred.device or D.device != pred.device):
            raise ValueError(f"gate_sum: pred {tuple(pred.shape)}, H {tuple(H.shape)} {H.dtype}, D {tuple(D.shape)} {D.dtype}, "
                             f"tables {tuple(h_weight.shape)} / {tuple(d_weight.shape)} do not match")
        H, D = H.contiguous(), D.contiguous()
    return _GateSumFunction.apply(pred, H, D, h_weight, d_weight)


# ---- channel axes assembled from several tensors: one pass instead of cat / several mixes -------------------

def _sliced_grad(dz: torch.Tensor, accepts):
    """(tensor, group stride in channels) for an incoming [G,Ck,N,T] gradient: used in place when it is contiguous
    (stride 0 = "its own width") or a channel slice [:, a:b] of a contiguous wider tensor AND the library reads such
    slices in place for this shape and graph (`accepts()`); a contiguous copy otherwise."""
    if dz.is_contiguous():
        return dz, 0
    G, Ck, N, T = dz.shape
    st = dz.stride()
    if dz.numel() > 0 and st[1:] == (N * T, T, 1) and st[0] % (N * T) == 0 and st[0] // (N * T) > Ck and accepts():
        return dz, st[0] // (N * T)
    return dz.contiguous(), 0


def _as_segment(t: torch.Tensor):
    """(tensor to keep alive, Seg) for a [G,Ck,N,T] tensor: used in place when it is contiguous or a channel
    slice [:, a:b] of a contiguous wider tensor (the library addresses such slices directly), copied otherwise."""
    G, Ck, N, T = t.shape
    st = t.stride()
    if t.numel() > 0 and st[1:] == (N * T, T, 1) and st[0] % (N * T) == 0 and st[0] // (N * T) >= Ck:
        return t, _lib.Seg(t.data_ptr(), Ck, st[0] // (N * T))
    t = t.contiguous()
    return t, _lib.Seg(t.data_ptr(), Ck, 0)


def _seg_array(tensors):
    keep, segs = [], []
    for t in tensors:
        k, sg = _as_segment(t)
        keep.append(k)
        segs.append(sg)
    arr = (_lib.Seg * max(len(segs), 1))(*segs)
    return keep, arr, len(segs)


class _MixMultiFunction(torch.autograd.Function):
    """outs = split(relu?(M cat(ins) + bias + cat(adds))): see include/msgat_hip.h, msgat_mix_segments."""

    @staticmethod
    def forward(ctx, M, bias, relu, n_in, n_add, out_channels, premasked, *tensors):
        L = _lib.lib()
        ins, adds = tensors[:n_in], tensors[n_in:n_in + n_add]
        G, _, N, T = ins[0].shape
        M = M.contiguous()
        R = M.shape[0]
        outs = [_new(ins[0], G, c, N, T) for c in out_channels]
        kin, ain, _ = _seg_array(ins)
        _kad, aad, _ = _seg_array(adds)
        _, aout, _ = _seg_array(outs)
        b = None if bias is None else bias.contiguous()
        per_rel = int(b is not None and b.dim() == 2)         # bias [Co] or [R,Co] (one row per matrix)
        st = L.msgat_mix_segments(R, G // R, N, T, ain, n_in, _ptr(M), 0, _ptr(b), per_rel, aad, n_add, int(relu), aout,
                                  len(outs), _stream_handle(M.device))
        _lib.check(st, "msgat_mix_segments")
        relu_bwd = bool(relu and not premasked)   # premasked: the consumer of the output applies the ReLU's backward mask
        ctx.meta = (relu_bwd, n_in, tuple(out_channels), (0 if bias is None else (R if per_rel else -1)),
                    [t.shape[1] for t in ins], [t.shape[1] for t in adds])
        ctx.save_for_backward(M, *kin, *(outs if relu_bwd else []))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        L = _lib.lib()
        relu, n_in, out_channels, has_bias, in_channels, add_channels = ctx.meta
        saved = ctx.saved_tensors
        M, ins, outs = saved[0], saved[1:1 + n_in], saved[1 + n_in:]
        like = ins[0]
        G, _, N, T = like.shape
        R = M.shape[0]
        stream = _stream_handle(M.device)
        dpre = []
        for i, (d, c) in enumerate(zip(douts, out_channels)):
            if d is None:
                d = torch.zeros(G, c, N, T, device=like.device, dtype=torch.float32)
            if relu:  # ReLU mask in one pass: dout where out > 0, else 0
                d = torch.ops.aten.threshold_backward(d.contiguous(), outs[i], 0.0)
            dpre.append(d)
        kd, ad, nd = _seg_array(dpre)
        need = ctx.needs_input_grad
        dM = dbias = None
        d_ins = [None] * n_in
        # one input, its gradient and the matrix gradient both wanted: ONE pass over cat(dpre) and the input gives both
        both = n_in == 1 and need[0] and need[7]
        if any(need[7:7 + n_in]):   # d cat(ins) = M^T cat(dpre), each input's range to its own tensor
            d_ins = [_new(like, G, c, N, T) for c in in_channels]
        if any(need[7:7 + n_in]) and not both:
            _, ai, _ = _seg_array(d_ins)
            st = L.msgat_mix_segments(R, G // R, N, T, ad, nd, _ptr(M), 1, None, 0, None, 0, 0, ai, n_in, stream)
            _lib.check(st, "msgat_mix_segments (backward)")
        want_bias = bool(has_bias and need[1])   # has_bias: 0 none, -1 shared [Co], R per matrix [R,Co]
        if need[0]:
            Co = sum(out_channels)
            parts = []
            for i, (x, c) in enumerate(zip(ins, in_channels)):
                x = x.contiguous()
                ones = int(want_bias and i == 0)   # the bias gradient = contraction with a virtual channel of ones
                buf = _new(like, R * Co * (c + ones))
                nfl = (L.msgat_contract_mix_partial_floats(R, G // R, N, T, Co, c + ones) if both
                       else L.msgat_contract_segments_partial_floats(R, Co, c + ones))
                part = _new(like, max(int(nfl), 1))
                # with_ones = 2: the matrix [R,Co,c] and, behind it, the ones column [R,Co] -- two contiguous tensors (a
                # column slice handed down to the components' parameters would cost a copy per component in AccumulateGrad)
                if both:
                    st = L.msgat_contract_mix_segments(R, G // R, N, T, ad, nd, _ptr(x), c, 2 * ones, _ptr(M), _ptr(part),
                                                       _ptr(buf), _ptr(d_ins[0]), stream)
                    _lib.check(st, "msgat_contract_mix_segments")
                else:
                    st = L.msgat_contract_segments(R, G // R, N, T, ad, nd, _ptr(x), c, 2 * ones, _ptr(part), _ptr(buf), stream)
                    _lib.check(st, "msgat_contract_segments")
                if ones:
                    colsum = buf[R * Co * c:].view(R, Co)                   # [R,Co]: sum over the relation's groups and positions
                    dbias = colsum if has_bias > 0 else colsum.sum(dim=0)
                parts.append(buf[: R * Co * c].view(R, Co, c))
            dM = parts[0] if len(parts) == 1 else torch.cat(parts, dim=2)
        elif want_bias:
            dbias = torch.cat([_channel_sums(k.contiguous(), max(has_bias, 0)) for k in kd], dim=-1)
        d_adds = []
        if add_channels:   # channel ranges of cat(dpre), as views (the library reads channel slices in place)
            whole = dpre[0] if len(dpre) == 1 else torch.cat(dpre, dim=1)
            a = 0
            for c in add_channels:
                d_adds.append(whole[:, a:a + c])
                a += c
        return (dM, dbias, None, None, None, None, None, *d_ins, *d_adds)


def mix_multi(ins, M, bias=None, adds=(), relu=False, out_channels=None, relu_grad_premasked=False):
    """One channel-mixing pass over cat(ins) with the [R, Co, Ci] matrix M (R | batch): relu?(M cat(ins) + bias +
    cat(adds)), the output channel ranges `out_channels` written to separate tensors (default: one).  Returns a
    tuple.  Replaces torch.cat + several 1x1 convolutions by a single read of the inputs.

    `relu_grad_premasked=True` (with relu): EVERY consumer of the output promises to hand back a gradient that is
    already zero where the output is <= 0 (`layer_norm_t(..., relu_input=True)` does, inside its backward kernel), so
    this op's backward skips its own mask pass over the activation."""
    ins, adds = list(ins), list(adds)
    for t in ins + adds:
        _require_device_tensor("signals", t)
    _require_device_tensor("matrix", M, ins[0].device)
    Ci, Co = sum(t.shape[1] for t in ins), M.shape[1]
    out_channels = [Co] if out_channels is None else list(out_channels)
    if M.dim() != 3 or M.shape[2] != Ci or sum(out_channels) != Co or ins[0].shape[0] % M.shape[0]:
        raise ValueError(f"mix_multi: matrix {tuple(M.shape)} for {Ci} input / {out_channels} output channels")
    if adds and sum(t.shape[1] for t in adds) != Co:
        raise ValueError("mix_multi: the add operands must cover the output channels")
    if relu and len(out_channels) != 1:
        raise ValueError("mix_multi: relu needs a single output tensor")
    if bias is not None and tuple(bias.shape) not in ((Co,), (M.shape[0], Co)):
        raise ValueError(f"bias must be [{Co}] or [{M.shape[0]},{Co}]")
    if max(len(ins), len(adds), len(out_channels)) > 6:
        raise ValueError("mix_multi: at most 6 tensors per channel axis")
    return _MixMultiFunction.apply(M, bias, bool(relu), len(ins), len(adds), tuple(out_channels),
                                   bool(relu and relu_grad_premasked), *ins, *adds)


class _AttentionCoreFunction(torch.autograd.Function):
    """u[G,Cu,N,T], q[G,N,T], Wg[R,T,T] -> z = (softmax(q Wg q^T) . adj) u: attention.py:34-36 on features and
    pooled signals that were produced elsewhere (msgat_stage_scores + msgat_stage_aggregate; backward
    msgat_attention_backward)."""

    @staticmethod
    def forward(ctx, u, q, Wg, graph: SparseGraph, recording: bool = True):
        L = _lib.lib()
        u, q, Wg = u.contiguous(), q.contiguous(), Wg.contiguous()
        G, Cu, N, T = u.shape
        R = Wg.shape[0]
        dev = u.device
        shape = _lib.Shape(R, G // R, Cu, 0, N, T)
        gstruct, _keep = graph.on(dev)
        need_bwd = bool(recording) and any(ctx.needs_input_grad)   # see _GACNFunction.forward
        kW, lse, E = _new(u, G, N, T), _new(u, G, N), _new(u, G, max(graph.nnz, 1))
        pq = _new(u, G, N, T) if need_bwd else None
        Ec = _new(u, G, max(graph.nnz, 1)) if need_bwd else None     # E in CSC order, for backward's transposed pass
        z = torch.empty_like(u)
        stream = _stream_handle(dev)
        # operand images of the score pass (large graphs only; the size depends on the dimensions alone: asked once per shape)
        sizes = graph.__dict__.setdefault("_dense_scratch_bytes", {})
        ndense = sizes.get((R, G, N, T))
        if ndense is None:
            ndense = sizes[(R, G, N, T)] = int(L.msgat_dense_scratch_bytes(C.byref(shape)))
        dense_t = torch.empty(ndense, device=dev, dtype=torch.uint8) if ndense else None
        _lib.check(L.msgat_stage_scores(C.byref(shape), C.byref(gstruct), _ptr(q), _ptr(Wg), _ptr(kW), _ptr(lse), _ptr(pq),
                                        _ptr(E), _ptr(Ec), _ptr(dense_t), stream), "msgat_stage_scores")
        nscratch = int(L.msgat_edge_scratch_floats(C.byref(shape), C.byref(gstruct)))
        scratch = _new(u, nscratch) if nscratch else None
        _lib.check(L.msgat_stage_aggregate(C.byref(shape), C.byref(gstruct), Cu, _ptr(u), _ptr(E), _ptr(z), _ptr(scratch),
                                           stream), "msgat_stage_aggregate")
        if need_bwd:
            ctx.graph, ctx.R = graph, R
            ctx.save_for_backward(u, q, Wg, kW, lse, pq, E, Ec)
        return z

    @staticmethod
    def backward(ctx, dz):
        L = _lib.lib()
        u, q, Wg, kW, lse, pq, E, Ec = ctx.saved_tensors
        G, Cu, N, T = u.shape
        dev = u.device
        shape = _lib.Shape(ctx.R, G // ctx.R, Cu, 0, N, T)
        gstruct, _keep = ctx.graph.on(dev)
        dz, dz_gs = _sliced_grad(dz, lambda: L.msgat_attention_bwd_accepts_strided_dv(C.byref(shape), C.byref(gstruct)))
        du, dq, dWg = torch.empty_like(u), torch.empty_like(q), torch.empty_like(Wg)
        nbytes = L.msgat_attention_bwd_workspace_bytes(C.byref(shape), C.byref(gstruct))
        ws = torch.empty(max(int(nbytes), 256), device=dev, dtype=torch.uint8)
        st = L.msgat_attention_backward(C.byref(shape), C.byref(gstruct), _ptr(u), _ptr(dz), dz_gs, _ptr(q), _ptr(kW), _ptr(lse),
                                        _ptr(pq), _ptr(E), _ptr(Ec), _ptr(Wg), _ptr(du), _ptr(dq), _ptr(dWg), _ptr(ws),
                                        ws.numel(), _stream_handle(dev))
        _lib.check(st, "msgat_attention_backward")
        return du, dq, dWg, None, None


def attention_core(u: torch.Tensor, q: torch.Tensor, Wg: torch.Tensor, adjacency) -> torch.Tensor:
    """Graph attention on already projected features `u` [G,Cu,N,T] with pooled signals `q` [G,N,T]."""
    _require_device_tensor("features", u)
    _require_device_tensor("pooled signals", q, u.device)
    _require_device_tensor("Wg", Wg, u.device)
    G, Cu, N, T = u.shape
    if tuple(q.shape) != (G, N, T) or Wg.dim() != 3 or tuple(Wg.shape[1:]) != (T, T) or G % Wg.shape[0]:
        raise ValueError(f"attention_core: u {tuple(u.shape)}, q {tuple(q.shape)}, Wg {tuple(Wg.shape)}")
    graph = adjacency if isinstance(adjacency, SparseGraph) else graph_of(adjacency)
    if graph.n_nodes != N:
        raise ValueError(f"adjacency has {graph.n_nodes} nodes, signals have {N}")
    return _AttentionCoreFunction.apply(u, q, Wg, graph, torch.is_grad_enabled())


# ---- the tiny attention matrices of a MEAM block, one launch each way ----------------------------------------------

class _ChannelAttentionMixFunction(torch.autograd.Function):
    """pooled[G,C,T], Wc[R,T,T], conv[R,cb,C] -> Mc[G,cb,C] = conv_r softmax((p Wc_r) p^T): attention.py:88-94 folded
    with CACN's 1x1 convolution weight (msgat.py:93-94)."""

    @staticmethod
    def forward(ctx, pooled, Wc, conv):
        L = _lib.lib()
        pooled, Wc, conv = pooled.contiguous(), Wc.contiguous(), conv.contiguous()
        G, Cc, T = pooled.shape
        R, cb = Wc.shape[0], conv.shape[1]
        att, Mc = _new(pooled, G, Cc, Cc), _new(pooled, G, cb, Cc)
        st = L.msgat_channel_attention_forward(_ptr(pooled), _ptr(Wc), _ptr(conv), _ptr(att), _ptr(Mc), G, R, Cc, cb, T,
                                               _stream_handle(pooled.device))
        _lib.check(st, "msgat_channel_attention_forward")
        ctx.save_for_backward(pooled, Wc, conv, att)
        return Mc

    @staticmethod
    def backward(ctx, dMc):
        L = _lib.lib()
        pooled, Wc, conv, att = ctx.saved_tensors
        G, Cc, T = pooled.shape
        R, cb = Wc.shape[0], conv.shape[1]
        dMc = dMc.contiguous()
        dp, dWc, dconv = torch.empty_like(pooled), torch.empty_like(Wc), torch.empty_like(conv)
        part = _new(pooled, max(int(L.msgat_channel_attention_partial_floats(G, Cc, cb, T)), 1))
        st = L.msgat_channel_attention_backward(_ptr(dMc), _ptr(att), _ptr(pooled), _ptr(Wc), _ptr(conv), _ptr(dp),
                                                _ptr(dWc), _ptr(dconv), _ptr(part), G, R, Cc, cb, T,
                                                _stream_handle(pooled.device))
        _lib.check(st, "msgat_channel_attention_backward")
        return dp, dWc, dconv


def channel_attention_mix(pooled: torch.Tensor, Wc: torch.Tensor, conv: torch.Tensor) -> torch.Tensor:
    """The per-sample channel matrix of CACN: conv @ softmax(pooled Wc pooled^T)  ([G,cb,C]).  pooled [G,C,T] (node-weighted
    sums, `node_pool`), Wc [T,T] or [R,T,T], conv [cb,C] or [R,cb,C] with R dividing G."""
    _require_device_tensor("pooled signals", pooled)
    _require_device_tensor("Wc", Wc, pooled.device)
    _require_device_tensor("conv weight", conv, pooled.device)
    if Wc.dim() == 2:
        Wc, conv = Wc.unsqueeze(0), conv.unsqueeze(0)
    G, Cc, T = pooled.shape
    if tuple(Wc.shape[1:]) != (T, T) or conv.dim() != 3 or conv.shape[0] != Wc.shape[0] or conv.shape[2] != Cc or G % Wc.shape[0]:
        raise ValueError(f"channel_attention_mix: pooled {tuple(pooled.shape)}, Wc {tuple(Wc.shape)}, conv {tuple(conv.shape)}")
    return _ChannelAttentionMixFunction.apply(pooled, Wc, conv)


class _TemporalAttentionTapsFunction(torch.autograd.Function):
    """pooled[G,N,T], Wt1/Wt2[R,K,N] -> taps[G,2,T,T]: (att shifted down by the dilation, att) with
    att = softmax((q^T Wt1^T)(q^T Wt2^T)^T), attention.py:58-66 as the taps of TACN's first convolution (msgat.py:66-74)."""

    @staticmethod
    def forward(ctx, pooled, Wt1, Wt2, dilation: int):
        L = _lib.lib()
        pooled, Wt1, Wt2 = pooled.contiguous(), Wt1.contiguous(), Wt2.contiguous()
        G, N, T = pooled.shape
        R, K = Wt1.shape[0], Wt1.shape[1]
        lr, att, taps = _new(pooled, G, 2, T, K), _new(pooled, G, T, T), _new(pooled, G, 2, T, T)
        st = L.msgat_temporal_attention_forward(_ptr(pooled), _ptr(Wt1), _ptr(Wt2), _ptr(lr), _ptr(att), _ptr(taps), G, R, N,
                                                K, T, int(dilation), _stream_handle(pooled.device))
        _lib.check(st, "msgat_temporal_attention_forward")
        ctx.dilation = int(dilation)
        ctx.save_for_backward(pooled, Wt1, Wt2, lr, att)
        return taps

    @staticmethod
    def backward(ctx, dtaps):
        L = _lib.lib()
        pooled, Wt1, Wt2, lr, att = ctx.saved_tensors
        G, N, T = pooled.shape
        R, K = Wt1.shape[0], Wt1.shape[1]
        dtaps = dtaps.contiguous()
        dp, dW1, dW2 = torch.empty_like(pooled), torch.empty_like(Wt1), torch.empty_like(Wt2)
        part = _new(pooled, max(int(L.msgat_temporal_attention_partial_floats(G, K, N)), 1))
        st = L.msgat_temporal_attention_backward(_ptr(dtaps), _ptr(att), _ptr(lr), _ptr(pooled), _ptr(Wt1), _ptr(Wt2), _ptr(dp),
                                                 _ptr(dW1), _ptr(dW2), _ptr(part), G, R, N, K, T, ctx.dilation,
                                                 _stream_handle(pooled.device))
        _lib.check(st, "msgat_temporal_attention_backward")
        return dp, dW1, dW2, None


def temporal_attention_taps(pooled: torch.Tensor, Wt1: torch.Tensor, Wt2: torch.Tensor, dilation: int) -> torch.Tensor:
    """[G,2,T,T] taps of TACN's first causal convolution: (temporal attention shifted down by `dilation`, temporal
    attention).  pooled [G,N,T] (alpha-weighted channel sums, `channel_pool`), Wt1 / Wt2 [K,N] or [R,K,N]."""
    _require_device_tensor("pooled signals", pooled)
    _require_device_tensor("Wt1", Wt1, pooled.device)
    _require_device_tensor("Wt2", Wt2, pooled.device)
    if Wt1.dim() == 2:
        Wt1, Wt2 = Wt1.unsqueeze(0), Wt2.unsqueeze(0)
    G, N, T = pooled.shape
    if Wt1.shape != Wt2.shape or Wt1.dim() != 3 or Wt1.shape[2] != N or G % Wt1.shape[0] or 2 * T * Wt1.shape[1] > 256:
        raise ValueError(f"temporal_attention_taps: pooled {tuple(pooled.shape)}, Wt1 {tuple(Wt1.shape)}, Wt2 {tuple(Wt2.shape)}")
    return _TemporalAttentionTapsFunction.apply(pooled, Wt1, Wt2, int(dilation))


class _BiasJoinFunction(torch.autograd.Function):
    """wide [R,Co] + narrow [R,cb] added into its first cb columns: CACN's convolution bias joining the residual bias in
    front of MEAM's ReLU (both per-channel constants, msgat.py:94,123-131).  pad + add cost three launches forward and a
    slice copy backward; here a copy and an in-place add, and backward hands out views of the incoming gradient."""

    @staticmethod
    def forward(ctx, wide, narrow):
        out = wide.clone()
        out[:, : narrow.shape[1]].add_(narrow)
        ctx.cb = narrow.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g[:, : ctx.cb]


def bias_join(wide: torch.Tensor, narrow: torch.Tensor) -> torch.Tensor:
    """`wide + pad(narrow)`: [R,Co] plus [R,cb <= Co] in the leading columns."""
    if wide.dim() != 2 or narrow.dim() != 2 or wide.shape[0] != narrow.shape[0] or narrow.shape[1] > wide.shape[1]:
        raise ValueError(f"bias_join: {tuple(wide.shape)} and {tuple(narrow.shape)}")
    return _BiasJoinFunction.apply(wide, narrow)


class _AssembleRowsFunction(torch.autograd.Function):
    """per_group [G,A,C], per_relation [R,Kr,C] -> rows [G,A+Kr,C]: each group's own rows followed by its relation's.
    Two copies forward; backward = a view of the first block and ONE sum over each relation's groups for the second
    (torch.cat of expanded tensors costs a copy per operand forward and a reduction per operand backward)."""

    @staticmethod
    def forward(ctx, per_group, per_relation):
        G, A, Cc = per_group.shape
        R, Kr, _ = per_relation.shape
        rows = torch.empty(G, A + Kr, Cc, device=per_group.device, dtype=torch.float32)
        rows[:, :A].copy_(per_group)
        rows.view(R, G // R, A + Kr, Cc)[:, :, A:].copy_(per_relation.unsqueeze(1))
        ctx.dims = (G, A, Cc, R, Kr)
        return rows

    @staticmethod
    def backward(ctx, drows):
        G, A, Cc, R, Kr = ctx.dims
        d_rel = drows.view(R, G // R, A + Kr, Cc)[:, :, A:].sum(dim=1) if ctx.needs_input_grad[1] else None
        return (drows[:, :A] if ctx.needs_input_grad[0] else None), d_rel


def assemble_rows(per_group: torch.Tensor, per_relation: torch.Tensor) -> torch.Tensor:
    """[G,A,C] rows of every group followed by the [Kr,C] rows of its relation (per_relation [R,Kr,C], R | G)."""
    if per_group.dim() != 3 or per_relation.dim() != 3 or per_group.shape[2] != per_relation.shape[2] or \
            per_group.shape[0] % per_relation.shape[0]:
        raise ValueError(f"assemble_rows: {tuple(per_group.shape)} and {tuple(per_relation.shape)}")
    return _AssembleRowsFunction.apply(per_group, per_relation)


_shift_taps_cache = {}


def causal_shift_taps(T: int, dilation: int, device) -> torch.Tensor:
    """[1,2,T,T] constant taps of a plain causal dilated [1,2] convolution (identity shifted down by `dilation`, identity):
    built once per (T, dilation, device)."""
    key = (T, int(dilation), str(device))
    taps = _shift_taps_cache.get(key)
    if taps is None:
        eye = torch.eye(T, device=device, dtype=torch.float32)
        shifted = torch.zeros_like(eye)
        if dilation < T:
            shifted[dilation:] = eye[: T - dilation]
        taps = _shift_taps_cache[key] = torch.stack([shifted, eye], dim=0).unsqueeze(0).contiguous()
    return taps


class _Relayout(torch.autograd.Function):
    """t.permute(perm).contiguous() whose backward is ALSO one contiguous copy (autograd's own backward of a permute is a
    strided view: handed down to R per-component parameters it is copied R times by AccumulateGrad)."""

    @staticmethod
    def forward(ctx, t, perm):
        ctx.perm = tuple(perm)
        return t.permute(*perm).contiguous()

    @staticmethod
    def backward(ctx, g):
        inv = [0] * len(ctx.perm)
        for i, p in enumerate(ctx.perm):
            inv[p] = i
        return g.permute(*inv).contiguous(), None


def relayout(t: torch.Tensor, perm) -> torch.Tensor:
    return _Relayout.apply(t, tuple(perm))


# ---- step tail: fused Huber loss + metric sums ---------------------------------------

class _HuberMetricsFunction(torch.autograd.Function):
    """pred, truth (same shape) -> mean Huber loss (0-dim); adds |e|, 100|e/y| (y > mask) and e^2 sums plus the loss
    itself to the running fp64 totals `sums` [4] on the device (engine.py:56,66-70; loss.py:51-52; metrics.py:20-35)."""

    @staticmethod
    def forward(ctx, pred, truth, delta: float, mask_value: float, sums, loss_weight: float = 1.0):
        L = _lib.lib()
        pred_c, truth_c = pred.contiguous(), truth.contiguous()
        n = pred_c.numel()
        part = torch.empty(max(int(L.msgat_huber_partial_doubles(n)), 1), device=pred.device, dtype=torch.float64)
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        st = L.msgat_huber_metrics(_ptr(pred_c), _ptr(truth_c), n, float(delta), float(mask_value), _ptr(part), _ptr(loss),
                                   _ptr(sums), float(loss_weight), _stream_handle(pred.device))
        _lib.check(st, "msgat_huber_metrics")
        ctx.delta = float(delta)
        ctx.save_for_backward(pred_c, truth_c)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        pred, truth = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        st = _lib.lib().msgat_huber_grad(_ptr(pred), _ptr(truth), _ptr(dloss.contiguous()), pred.numel(), ctx.delta,
                                         _ptr(dpred), _stream_handle(pred.device))
        _lib.check(st, "msgat_huber_grad")
        return dpred, None, None, None, None, None


def huber_metrics(pred: torch.Tensor, truth: torch.Tensor, delta: float, mask_value: float = 0.0,
                  sums: Optional[torch.Tensor] = None, loss_weight: float = 1.0) -> torch.Tensor:
    """Mean Huber loss of `pred` against `truth` in one pass that also feeds the epoch's metric totals.
    `loss_weight` scales what is added to the running loss total (a rank's share n_r / n_b of a global batch)."""
    _require_device_tensor("prediction", pred)
    _require_device_tensor("target", truth, pred.device)
    if pred.shape != truth.shape or pred.numel() == 0:
        raise ValueError(f"prediction {tuple(pred.shape)} and target {tuple(truth.shape)} must match and be non-empty")
    if sums is not None and (sums.dtype != torch.float64 or sums.numel() != 4 or sums.device != pred.device):
        raise ValueError("sums must be a float64 [4] tensor on the prediction's device")
    return _HuberMetricsFunction.apply(pred, truth, delta, mask_value, sums, float(loss_weight))


# ---- boundary hygiene for every autograd.Function above -----------------------------------------------------------
# (a) the library's launches act on the CURRENT HIP device (hipFuncSetAttribute for > 64 KB LDS, occupancy queries),
#     while the reference's API lets the caller name any device (`run_epoch(..., gpu_id=k)`, engine.py:40,50):
#     make the tensors' device current for the duration of the call;
# (b) the reference runs the forward under CUDA AMP (engine.py:54): inside an autocast region these ops receive
#     half-precision outputs of neighbouring matmuls -- cast them back to fp32 (the parity type) and switch
#     autocast off inside, as torch.amp.custom_fwd(cast_inputs=torch.float32) does.
def _guard(fn, is_forward: bool):
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, *args):
        dev = next((a.device for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)

    # Outside an autocast region -- every call of this package's own engine, and the autograd thread unless the caller
    # ran backward inside one -- torch.amp.custom_fwd is a no-op and custom_bwd an `autocast(enabled=False)` context
    # around a region that has autocast off already: ~5 us of host time per backward call for nothing (a sixth of a
    # GACN module call's host time, tools/host_overhead.py --dropin).  Both keep their exact semantics inside a region.
    if is_forward:
        in_region = torch.amp.custom_fwd(wrapped, device_type="cuda", cast_inputs=torch.float32)

        @functools.wraps(fn)
        def forward(ctx, *args):
            if torch.is_autocast_enabled("cuda"):
                return in_region(ctx, *args)
            return wrapped(ctx, *args)
        return forward

    @functools.wraps(fn)
    def backward(ctx, *args):
        if torch.is_autocast_enabled("cuda"):
            with torch.autocast(device_type="cuda", enabled=False):
                return wrapped(ctx, *args)
        return wrapped(ctx, *args)
    return backward


# (c) every backward here launches library kernels on raw pointers: it cannot be differentiated a second time.
#     `once_differentiable` says so to autograd -- `backward(create_graph=True)` / a grad-of-grad then fails AT this node
#     with PyTorch's own message ("trying to differentiate twice a function that was marked with @once_differentiable")
#     instead of silently returning gradients that are constants of the outer graph.
from torch.autograd.function import once_differentiable  # noqa: E402


def _once(fn):
    """`once_differentiable`, entered only when it has something to do: in an ordinary backward grad mode is off already and
    the decorator's `no_grad` context is ~2 us of host time per call for nothing (28 calls per training step)."""
    import functools
    guarded = once_differentiable(fn)

    @functools.wraps(fn)
    def backward(ctx, *args):
        if torch.is_grad_enabled():      # backward(create_graph=True): hand the gradients over behind an error node
            return guarded(ctx, *args)
        return fn(ctx, *args)
    return backward


for _name, _cls in list(globals().items()):
    if isinstance(_cls, type) and issubclass(_cls, torch.autograd.Function) and _cls is not torch.autograd.Function:
        _cls.forward = staticmethod(_guard(_cls.forward, True))
        _cls.backward = staticmethod(_once(_guard(_cls.backward, False)))
