"""Sensor-graph adjacency: the sym-normalised dense matrix the reference uses and the
CSR/CSC form the HIP kernels walk.

Reference: /root/reference/src/data_loader.py:49-66 builds `D^-1/2 (A + I) D^-1/2` from a
csv edge list; src/models/msgat.py:190 keeps it as a frozen parameter `adj`;
src/models/attention.py:36 applies it as a dense mask.  Only its non-zeros matter to that
product, so `SparseGraph` holds them (built by the native host routine
`msgat_graph_build`, include/msgat_hip.h).
"""
from __future__ import annotations

import collections
import ctypes as C
import weakref
from typing import Iterable, Tuple

import numpy as np
import torch

from . import _lib


def sym_norm_adjacency(n_nodes: int, edges: Iterable[Tuple[int, int]]) -> torch.Tensor:
    """`D^-1/2 (A + I) D^-1/2` as a dense fp32 [N,N] tensor (data_loader.py:59-66).

    `edges` are undirected (src, dst) pairs; duplicates and self pairs collapse to a
    single unit entry, as the assignment `A[s, d] = A[d, s] = 1` does in the reference.
    """
    a = torch.eye(n_nodes, dtype=torch.float32)
    e = torch.as_tensor(np.asarray(list(edges), dtype=np.int64).reshape(-1, 2))
    if e.numel():
        a[e[:, 0], e[:, 1]] = 1.0
        a[e[:, 1], e[:, 0]] = 1.0
    d = a.sum(dim=1).rsqrt()
    return d[:, None] * a * d[None, :]


def random_edges(n_nodes: int, n_edges: int, seed: int = 0) -> np.ndarray:
    """`n_edges` distinct undirected non-self edges, uniform over node pairs."""
    max_edges = n_nodes * (n_nodes - 1) // 2
    if n_edges > max_edges:
        raise ValueError(f"{n_edges} edges requested, a simple graph on {n_nodes} nodes has {max_edges}")
    rng = np.random.default_rng(seed)
    seen, out = set(), []
    while len(out) < n_edges:
        s, d = (int(v) for v in rng.integers(0, n_nodes, size=2))
        if s == d:
            continue
        key = (s, d) if s < d else (d, s)
        if key in seen:
            continue
        seen.add(key)
        out.append((s, d))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def synthetic_adjacency(n_nodes: int, n_edges: int, seed: int = 0) -> torch.Tensor:
    """PEMS-like synthetic sensor graph (no PEMS files ship with the reference)."""
    return sym_norm_adjacency(n_nodes, random_edges(n_nodes, n_edges, seed))


# one float4 per node must fit LDS for the SELL kernels (csrc/common.hpp: kLdsMax - 1024); below _SELL_AUTO_MIN
# nodes a whole [N,T] slab fits LDS for every supported T, so the layout would never be used
_SELL_MAX_NODES = (160 * 1024 - 1024) // 16
_SELL_AUTO_MIN = 2048


class SparseGraph:
    """CSR + CSC of a dense adjacency, host arrays plus (lazily) device copies.

    `sell`: "auto" also builds the degree-sorted sliced-ELLPACK layouts (include/msgat_hip.h, msgat_sell_t) for
    graphs large enough that an [N,T] slab may not fit LDS; "always" builds them and makes the library prefer the
    SELL kernels (tests exercise them on small graphs this way); "never" omits them."""

    def __init__(self, adjacency: torch.Tensor, sell: str = "auto"):
        if adjacency.dim() > 2:
            # the reference documents `adjacency: [..., n_nodes, n_nodes]` (attention.py:22: `att * adjacency`
            # broadcasts, attention.py:36); its only caller passes ONE [N,N] matrix (msgat.py:127, :190), and so does
            # everything here: the CSR / CSC / SELL structures are built per graph, not per sample
            raise ValueError(
                f"adjacency must be one [n_nodes, n_nodes] matrix, got {tuple(adjacency.shape)}: the batched form "
                "`[..., n_nodes, n_nodes]` that the reference's attention.py:22 documents is not supported -- call the "
                "op once per adjacency (one SparseGraph each)")
        if adjacency.dim() != 2 or adjacency.size(0) != adjacency.size(1):
            raise ValueError(f"adjacency must be [N,N], got {tuple(adjacency.shape)}")
        a = adjacency.detach().to(device="cpu", dtype=torch.float32).contiguous()
        n = a.size(0)
        L = _lib.lib()
        nnz = C.c_int32(0)
        _lib.check(L.msgat_graph_count(a.data_ptr(), n, n, C.byref(nnz)), "msgat_graph_count")
        self.n_nodes, self.nnz = n, int(nnz.value)
        m = max(self.nnz, 1)
        self.rowptr = torch.zeros(n + 1, dtype=torch.int32)
        self.colptr = torch.zeros(n + 1, dtype=torch.int32)
        self.col = torch.zeros(m, dtype=torch.int32)
        self.val = torch.zeros(m, dtype=torch.float32)
        self.erow = torch.zeros(m, dtype=torch.int32)
        self.crow = torch.zeros(m, dtype=torch.int32)
        self.cperm = torch.zeros(m, dtype=torch.int32)
        self.cpos = torch.zeros(m, dtype=torch.int32)
        _lib.check(L.msgat_graph_build(a.data_ptr(), n, n, self.nnz, self.rowptr.data_ptr(), self.col.data_ptr(),
                                       self.val.data_ptr(), self.erow.data_ptr(), self.colptr.data_ptr(),
                                       self.crow.data_ptr(), self.cperm.data_ptr(), self.cpos.data_ptr()), "msgat_graph_build")
        self._dev = {}
        if sell not in ("auto", "always", "never"):
            raise ValueError(f"sell must be 'auto', 'always' or 'never', got {sell!r}")
        self.sell_prefer = sell == "always"
        self._sell = {}
        if sell != "never" and self.nnz > 0 and n <= _SELL_MAX_NODES and (self.sell_prefer or n >= _SELL_AUTO_MIN):
            self._sell["sell_rows"] = self._build_sell(self.rowptr, self.col, None, with_pos=True)
            self._sell["sell_cols"] = self._build_sell(self.colptr, self.crow, self.cperm, with_pos=False)

    _FIELDS = ("rowptr", "col", "val", "erow", "colptr", "crow", "cperm", "cpos")
    _SELL_FIELDS = ("slice_off", "lane_row", "idx", "src", "pos")

    def _build_sell(self, ptr, idx, perm, with_pos: bool):
        L = _lib.lib()
        ns, npos, pair = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(L.msgat_graph_sell_count(ptr.data_ptr(), self.n_nodes, C.byref(ns), C.byref(npos), C.byref(pair)),
                   "msgat_graph_sell_count")
        ns, npos = int(ns.value), int(npos.value)
        t = dict(slice_off=torch.zeros(ns + 1, dtype=torch.int32), lane_row=torch.zeros(64 * ns, dtype=torch.int32),
                 idx=torch.zeros(npos + _lib.SELL_SLACK, dtype=torch.int16), src=torch.zeros(max(npos, 1), dtype=torch.int32))
        if with_pos:
            t["pos"] = torch.zeros(self.nnz, dtype=torch.int32)
        _lib.check(L.msgat_graph_sell_build(ptr.data_ptr(), idx.data_ptr(), None if perm is None else perm.data_ptr(),
                                            self.n_nodes, self.nnz, ns, npos, t["slice_off"].data_ptr(),
                                            t["lane_row"].data_ptr(), t["idx"].data_ptr(), t["src"].data_ptr(),
                                            t["pos"].data_ptr() if with_pos else None), "msgat_graph_sell_build")
        t["n_slices"], t["n_pos"], t["pair_trips"] = ns, npos, int(pair.value)
        return t

    @property
    def has_sell(self) -> bool:
        return bool(self._sell)

    def _struct(self, tensors) -> _lib.Graph:
        g = _lib.Graph()
        g.n_nodes, g.nnz = self.n_nodes, self.nnz
        for name in self._FIELDS:
            setattr(g, name, tensors[name].data_ptr())
        for form, host in self._sell.items():
            j = getattr(g, form)
            j.n_slices, j.n_pos, j.prefer = host["n_slices"], host["n_pos"], int(self.sell_prefer)
            j.pair_trips = host["pair_trips"]
            for name in self._SELL_FIELDS:
                if name in host:
                    setattr(j, name, tensors[f"{form}.{name}"].data_ptr())
        return g

    def _host_tensors(self):
        t = {k: getattr(self, k) for k in self._FIELDS}
        for form, host in self._sell.items():
            for name in self._SELL_FIELDS:
                if name in host:
                    t[f"{form}.{name}"] = host[name]
        return t

    def host_struct(self) -> _lib.Graph:
        return self._struct(self._host_tensors())

    def validate(self) -> None:
        hs = self.host_struct()
        _lib.check(_lib.lib().msgat_graph_validate(C.byref(hs)), "msgat_graph_validate")

    def on(self, device: torch.device):
        """(ctypes struct of device pointers, tensors kept alive) for `device`."""
        key = str(device)
        if key not in self._dev:
            tensors = {k: v.to(device) for k, v in self._host_tensors().items()}
            self._dev[key] = (self._struct(tensors), tensors)
        return self._dev[key]

    def dense(self) -> torch.Tensor:
        a = torch.zeros(self.n_nodes, self.n_nodes)
        if self.nnz:
            a[self.erow[: self.nnz].long(), self.col[: self.nnz].long()] = self.val[: self.nnz]
        return a


# The adjacency is a frozen parameter (msgat.py:190): one CSR build per tensor version.
_CACHE: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_CACHE_MAX = 16


def graph_of(adjacency: torch.Tensor) -> SparseGraph:
    """Cached `SparseGraph` of a dense adjacency tensor.

    Keyed on (storage address, shape, device, version).  A hit on the very same tensor
    object is free; a hit through a different object (a view, or a new tensor the allocator
    placed at a recycled address) is confirmed by comparing contents before it is trusted.
    """
    key = (adjacency.data_ptr(), tuple(adjacency.shape), str(adjacency.device), adjacency._version)
    hit = _CACHE.get(key)
    if hit is not None:
        g, ref, snapshot = hit
        if ref() is adjacency:
            _CACHE.move_to_end(key)
            return g
        # another tensor object at a cached address: compare contents ONCE (a device read-back, which a HIP-graph capture
        # cannot contain) and remember the new object, so that its later calls -- the captured one included -- hit by
        # identity.  (A model built where a freed model's adjacency lived used to compare on every call, and its first
        # capture failed with "operation not permitted when stream is capturing".)
        if not (adjacency.is_cuda and torch.cuda.is_current_stream_capturing()) and torch.equal(snapshot, adjacency.detach()):
            _CACHE[key] = (g, weakref.ref(adjacency), snapshot)
            _CACHE.move_to_end(key)
            return g
    if adjacency.is_cuda and torch.cuda.is_current_stream_capturing():
        # building the CSR reads the adjacency back to the host, which a stream capture cannot contain
        raise _lib.MsgatError("the CSR of this adjacency is not cached yet: run one forward outside the HIP-graph "
                              "capture first (engine.Trainer does this in its warm-up)")
    g = SparseGraph(adjacency)
    _CACHE[key] = (g, weakref.ref(adjacency), adjacency.detach().clone())
    _CACHE.move_to_end(key)
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return g
