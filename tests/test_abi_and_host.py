"""CPU-side checks: the C-ABI library builds, loads and exports every symbol
include/msgat_hip.h declares; the native host graph builder; the loud refusal of CPU
tensors.  No device compute is called here.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import gat_oracle


@pytest.fixture(scope="session", autouse=True)
def built_library():
    from ms_gat_amd import build
    return build.build(verbose=False)


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "msgat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msgat_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ms_gat_amd import _lib
    names = _declared_functions()
    assert len(names) >= 12
    h = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/msgat_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names, "ctypes prototypes and header disagree"
    assert _lib.lib().msgat_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every ABI struct as the C compiler lays out include/msgat_hip.h vs the ctypes mirror."""
    import subprocess
    from ms_gat_amd import _lib
    structs = {"msgat_shape_t": _lib.Shape, "msgat_sell_t": _lib.Sell, "msgat_graph_t": _lib.Graph,
               "msgat_fwd_t": _lib.Fwd, "msgat_bwd_t": _lib.Bwd, "msgat_seg_t": _lib.Seg}
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for field, *_ in cls._fields_:
            lines.append(f'printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "msgat_hip.h"\nint main(void) {\n' + "\n".join(lines)
                   + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for field, *_ in cls._fields_:
            assert int(got[f"{cname}.{field}"]) == getattr(cls, field).offset, f"{cname}.{field}"
    assert f"#define MSGAT_SELL_SLACK {_lib.SELL_SLACK} " in open(os.path.join(ROOT, "include", "msgat_hip.h")).read()


def test_mode_selection_and_status_strings():
    from ms_gat_amd import _lib
    L = _lib.lib()
    assert L.msgat_gacn_mode(3, 0) == _lib.MODE_PLAIN
    assert L.msgat_gacn_mode(1, 24) == _lib.MODE_AGG_FIRST
    assert L.msgat_gacn_mode(24, 24) == _lib.MODE_AGG_FIRST
    assert L.msgat_gacn_mode(72, 24) == _lib.MODE_PROJ_FIRST
    assert L.msgat_status_string(0) == b"ok"
    assert b"workspace" in L.msgat_status_string(-4)


def test_bad_arguments_come_back_as_status_codes():
    """Argument errors must be reported before anything is enqueued (no GPU needed)."""
    from ms_gat_amd import _lib
    L = _lib.lib()
    shape = _lib.Shape(1, 2, 3, 24, 16, 12)
    assert L.msgat_gacn_forward(C.byref(shape), None, None, None) == -1          # NULL graph
    bad_t = _lib.Shape(1, 2, 3, 24, 16, 10)
    g = _lib.Graph()
    g.n_nodes = 16
    assert L.msgat_gacn_forward(C.byref(bad_t), C.byref(g), None, None) == -3    # T = 10 unsupported
    assert L.msgat_gacn_forward(C.byref(_lib.Shape(0, 2, 3, 24, 16, 12)), C.byref(g), None, None) == -2
    g.n_nodes = 15
    assert L.msgat_gacn_forward(C.byref(shape), C.byref(g), None, None) == -2    # graph/shape mismatch
    import ms_gat_amd
    hs = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(16, 12, 0)).host_struct()
    assert L.msgat_bwd_workspace_bytes(C.byref(bad_t), C.byref(hs)) == 0
    assert L.msgat_bwd_workspace_bytes(C.byref(shape), None) == 0
    assert L.msgat_bwd_workspace_bytes(C.byref(shape), C.byref(hs)) > 0
    with pytest.raises(_lib.MsgatError):
        _lib.check(-4, "x")


def _numpy_csr(adj):
    rows, cols = np.nonzero(adj)
    return rows, cols, adj[rows, cols]


@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (7, 5, 1), (64, 70, 2), (307, 340, 3), (883, 866, 4)])
def test_native_csr_csc_build_matches_numpy(n, e, seed):
    import ms_gat_amd
    adj = ms_gat_amd.synthetic_adjacency(n, e, seed)
    g = ms_gat_amd.SparseGraph(adj)
    g.validate()
    rows, cols, vals = _numpy_csr(adj.numpy())
    assert g.nnz == len(rows) == n + 2 * e
    assert np.array_equal(g.erow[: g.nnz].numpy(), rows)
    assert np.array_equal(g.col[: g.nnz].numpy(), cols)
    assert np.array_equal(g.val[: g.nnz].numpy(), vals)
    assert np.array_equal(g.rowptr.numpy(), np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]))
    # CSC: entries sorted by column then row, cperm maps back into the CSR order
    order = np.lexsort((rows, cols))
    assert np.array_equal(g.cperm[: g.nnz].numpy(), order)
    inverse = np.empty_like(order)
    inverse[order] = np.arange(len(order))
    assert np.array_equal(g.cpos[: g.nnz].numpy(), inverse)      # CSR edge -> CSC position
    assert np.array_equal(g.crow[: g.nnz].numpy(), rows[order])
    assert np.array_equal(g.colptr.numpy(), np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=n))]))
    assert torch.equal(g.dense(), adj)


def test_csr_build_on_asymmetric_weighted_matrix_with_empty_rows():
    import ms_gat_amd
    rng = np.random.default_rng(5)
    a = (rng.random((23, 23)) < 0.15) * rng.standard_normal((23, 23))
    a[4, :] = 0
    a[:, 9] = 0
    g = ms_gat_amd.SparseGraph(torch.from_numpy(a.astype(np.float32)))
    g.validate()
    assert torch.equal(g.dense(), torch.from_numpy(a.astype(np.float32)))
    assert g.rowptr[4] == g.rowptr[5] and g.colptr[9] == g.colptr[10]


def test_graph_validate_rejects_corruption():
    import ms_gat_amd
    from ms_gat_amd import _lib
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(20, 25, 0))
    g.validate()
    g.col[3] = 99
    with pytest.raises(_lib.MsgatError):
        g.validate()
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(20, 25, 0))
    g.cpos[[0, 1]] = g.cpos[[1, 0]]         # the CSR -> CSC map must invert cperm (the forward scatters E through it)
    with pytest.raises(_lib.MsgatError):
        g.validate()


# the lane groups a ds_read_b128 is served in (MI355X_MICROARCH.md, LDS table)
_B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
                list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _numpy_sell(ptr, idx, ids):
    """Restatement of msgat_graph_sell_build: rows sorted by degree (descending, stable), slices of 64, every slice
    padded to its first row's degree rounded up to a multiple of 4 and stored in lane-interleaved trips of 4 columns;
    every column filled group by group (the 16 lanes a ds_read_b128 serves together) with a maximum bipartite
    matching lane -> LDS bank quad (neighbour index mod 16) over the lanes' remaining edges."""
    n = len(ptr) - 1
    deg = np.diff(ptr)
    order = np.argsort(-deg, kind="stable")
    ns = (n + 63) // 64
    off, lane_row, sidx, ssrc = [0], [], [], []
    for s in range(ns):
        rows = order[64 * s: 64 * s + 64].tolist()
        width = (int(deg[rows[0]]) + 3) // 4 * 4
        lane_row += rows + [-1] * (64 - len(rows))
        rem = [list(range(ptr[r], ptr[r + 1])) for r in rows] + [[] for _ in range(64 - len(rows))]
        cols = [[None] * 64 for _ in range(width)]
        for k in range(width):
            for grp in _B128_GROUPS:
                active = [l for l in sorted(grp, reverse=True) if rem[l]]
                owner = {}

                def try_lane(l, seen):
                    for q in sorted({int(idx[e]) & 15 for e in rem[l]}):
                        if q in seen:
                            continue
                        seen.add(q)
                        if q not in owner or try_lane(owner[q], seen):
                            owner[q] = l
                            return True
                    return False

                for l in active:
                    try_lane(l, set())
                lane_q = {l: q for q, l in owner.items()}
                cnt = [0] * 16
                for q in owner:
                    cnt[q] = 1
                for l in active:
                    if l in lane_q:
                        i = next(i for i, e in enumerate(rem[l]) if int(idx[e]) & 15 == lane_q[l])
                    else:
                        i = min(range(len(rem[l])), key=lambda i: (cnt[int(idx[rem[l][i]]) & 15], i))
                        cnt[int(idx[rem[l][i]]) & 15] += 1
                    cols[k][l] = rem[l].pop(i)
        for trip in range(width // 4):                       # a trip = 4 columns, lane-interleaved
            for lane in range(64):
                for k in range(4 * trip, 4 * trip + 4):
                    e = cols[k][lane]
                    sidx.append(0 if e is None else int(idx[e]))
                    ssrc.append(-1 if e is None else int(ids[e]))
        off.append(len(sidx))
    return np.array(off), np.array(lane_row), np.array(sidx, dtype=np.int64), np.array(ssrc, dtype=np.int64)


def _b128_cycles(sidx_trips):
    """LDS cycles the gathers of one column pass cost in the guide's model: per instruction, per lane group, the
    largest number of DISTINCT addresses on one bank quad."""
    total = 0
    for trip in sidx_trips.reshape(-1, 64, 4):
        for k in range(4):
            for grp in _B128_GROUPS:
                quads = {}
                for lane in grp:
                    quads.setdefault(int(trip[lane, k]) & 15, set()).add(int(trip[lane, k]))
                total += max(len(v) for v in quads.values())
    return total


@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (7, 5, 1), (64, 300, 2), (65, 70, 3), (200, 1500, 4), (883, 866, 5)])
def test_native_sell_build_matches_numpy(n, e, seed):
    """The degree-sorted sliced-ELLPACK edge layout of the large-graph kernels (msgat_sell_t): pure index work,
    bit-exact."""
    import ms_gat_amd
    from ms_gat_amd import _lib
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(n, e, seed), sell="always")
    assert g.has_sell
    g.validate()
    nnz = g.nnz
    for form, ptr, idx, ids in (("sell_rows", g.rowptr, g.col, np.arange(nnz)), ("sell_cols", g.colptr, g.crow, g.cperm.numpy())):
        j = g._sell[form]
        off, lr, si, ss = _numpy_sell(ptr.numpy(), idx.numpy(), ids)
        assert j["n_slices"] == (n + 63) // 64 and j["n_pos"] == off[-1] and j["n_pos"] % 256 == 0
        trips = np.diff(off) // 256
        ns = len(trips)
        assert j["pair_trips"] == max(trips[i] + (trips[ns - 1 - i] if ns - 1 - i > i else 0) for i in range((ns + 1) // 2))
        assert np.array_equal(j["slice_off"].numpy(), off)
        assert np.array_equal(j["lane_row"].numpy(), lr)
        assert j["idx"].dtype == torch.int16 and np.array_equal(j["idx"].numpy()[: j["n_pos"]], si)
        assert not j["idx"].numpy()[j["n_pos"]:].any() and len(j["idx"]) == j["n_pos"] + _lib.SELL_SLACK
        assert np.array_equal(j["src"].numpy()[: j["n_pos"]], ss)
        assert sorted(ss[ss >= 0].tolist()) == list(range(nnz))      # every CSR edge exactly once
        if "pos" in j:
            assert np.array_equal(ss[j["pos"].numpy()], np.arange(nnz))


def test_sell_padding_stays_small_on_the_stress_graph_shape():
    """Sorting by degree keeps the slices homogeneous: Poisson-like degrees (the BASELINE stress graph's shape,
    scaled down) pad by a few percent, not by the max/mean degree ratio an unsorted sliced layout would."""
    import ms_gat_amd
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(2048, 16384, 0))
    assert g.has_sell and g.nnz == 2048 + 2 * 16384
    assert g._sell["sell_rows"]["n_pos"] < 1.12 * g.nnz
    deg = np.diff(g.rowptr.numpy())
    unsorted = sum(64 * int(deg[i:i + 64].max()) for i in range(0, 2048, 64))
    assert unsorted > 1.4 * g.nnz


def test_sell_column_order_cuts_the_lds_bank_conflicts_of_the_gathers():
    """Every row's edges are dealt to the columns so that the 16 lanes a ds_read_b128 serves together hit different
    bank quads where possible: against the CSR order the modelled LDS cycles of a column pass drop by a third or more
    on a Poisson-degree graph (11.5 -> 6.5 per instruction at the stress graph; measured on the GPU the same way)."""
    import ms_gat_amd
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(2048, 16384, 3))
    j = g._sell["sell_rows"]
    ours = _b128_cycles(j["idx"].numpy()[: j["n_pos"]].astype(np.int64))
    ptr, col = g.rowptr.numpy(), g.col.numpy()
    plain = np.zeros(j["n_pos"], dtype=np.int64)                           # the same layout with each row in CSR order
    lane_row, off = j["lane_row"].numpy(), j["slice_off"].numpy()
    for s in range(j["n_slices"]):
        for lane in range(64):
            r = lane_row[64 * s + lane]
            for k in range(ptr[r + 1] - ptr[r] if r >= 0 else 0):
                plain[off[s] + 256 * (k // 4) + 4 * lane + k % 4] = col[ptr[r] + k]
    assert ours < 0.7 * _b128_cycles(plain)


def test_sell_is_built_for_large_graphs_only_and_validate_catches_corruption():
    import ms_gat_amd
    from ms_gat_amd import _lib
    assert not ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(300, 340, 0)).has_sell       # a slab fits LDS
    assert not ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(2100, 3000, 0), sell="never").has_sell
    big = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(2100, 3000, 0))
    assert big.has_sell and not big.sell_prefer
    big.validate()
    hs = big.host_struct()
    assert hs.sell_rows.n_slices == 33 and hs.sell_cols.n_slices == 33 and hs.sell_rows.prefer == 0
    for field, at in (("idx", 5), ("src", 7), ("lane_row", 3), ("slice_off", 1)):
        g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(100, 300, 1), sell="always")
        t = g._sell["sell_cols"][field]
        t[at] = t[at] + 1
        with pytest.raises(_lib.MsgatError):
            g.validate()


def test_sym_norm_adjacency_matches_reference_and_oracle():
    import ms_gat_amd
    a = load_golden("adj_n12.npz")
    ours = ms_gat_amd.sym_norm_adjacency(int(a["n"]), a["edges"]).numpy()
    assert np.allclose(ours, a["adj"], atol=1e-7)
    e = ms_gat_amd.random_edges(50, 80, seed=3)
    assert len({tuple(sorted(p)) for p in e.tolist()}) == 80 and all(s != d for s, d in e.tolist())
    assert np.allclose(ms_gat_amd.sym_norm_adjacency(50, e).numpy(), gat_oracle.sym_norm_adjacency(50, e), atol=1e-7)


@pytest.mark.parametrize("hidden", [48, 72, 96])
def test_every_convolution_backward_of_the_registry_models_takes_a_one_pass_lds_dma_form(hidden):
    """The three models of the reference's registry (main.py:17, msgat.py:220-229: 48 / 72 / 96 hidden channels, a third
    per branch) produce three 1x1-convolution backward shapes per second-level block; at rows of >= 512 positions each
    must take a one-pass LDS-DMA form (contraction AND input gradient from one read of the operands) -- a silent
    fall-back to the register-staged kernels plus a projection pass is what `msgat_contract_form_name` makes visible."""
    from ms_gat_amd import _lib
    co = hidden // 3
    # (the merged mixing carries a bias column in the per-component schedule, model.MEAM._merged_branches, and none in the
    # stacked one, stacked._meam -- round 5 found the bias-less 48- and 96-channel shapes falling back to two passes)
    sites = {"GACN projection [Co+1 x C]": (co + 1, hidden, False),
             "merged channel mixing [4Co+2 x C+1]": (4 * co + 2, hidden, True),
             "merged channel mixing, no bias column [4Co+2 x C]": (4 * co + 2, hidden, False),
             "residual convolution [C x C+1]": (hidden, hidden, True),
             "residual convolution, no bias column [C x C]": (hidden, hidden, False)}
    for P in (512, 64 * 12, 307 * 12, 883 * 12):
        for what, (Ca, Cb, ones) in sites.items():
            name = _lib.contract_form_name(Ca, Cb, ones, P, True)
            assert name.startswith("k_chanpair_glds<") and "projection pass" not in name, (hidden, what, P, name)
            mode = int(name.split(">")[0].split(",")[-1])
            assert mode in (1, 2), name                                   # a MIX form, not the plain contraction
    # below 512 positions per row, and for channel counts outside the registry, the two-pass form is reported as such
    assert _lib.contract_form_name(25, 72, False, 156, True) == "k_chanpair_mfma<2,5,true,256> + projection pass"
    assert "projection pass" in _lib.contract_form_name(150, 120, True, 883 * 12, True)
    # plain contractions report their kernel too
    assert _lib.contract_form_name(98, 72, True, 883 * 12, False).startswith("k_chanpair_glds<7,5,64,3,0>")
    with pytest.raises(_lib.MsgatError):
        _lib.contract_form_name(0, 72, False, 100, True)


def test_batched_adjacency_is_refused_with_a_pointer_to_the_reference():
    """attention.py:22 documents `adjacency: [..., n_nodes, n_nodes]`; the sole caller passes [N,N] (msgat.py:127) and so
    does everything here -- a batched adjacency must not be mistaken for a malformed one."""
    import ms_gat_amd
    from ms_gat_amd import graph
    with pytest.raises(ValueError, match=r"attention\.py:22"):
        ms_gat_amd.SparseGraph(torch.zeros(2, 5, 5))
    with pytest.raises(ValueError, match=r"attention\.py:22"):
        graph.graph_of(torch.eye(4).expand(3, 4, 4))
    with pytest.raises(ValueError, match=r"\[N,N\]"):
        ms_gat_amd.SparseGraph(torch.zeros(5, 4))


def test_gacn_plans_live_on_their_graph_and_die_with_it():
    """The per-shape plans of ops.gacn are stored on the SparseGraph (round-4 advisor finding: a module-level table keyed
    by id(graph) kept every graph a process had used alive)."""
    import gc
    import weakref
    import ms_gat_amd
    from ms_gat_amd import ops
    assert not hasattr(ops, "_PLANS")
    g = ms_gat_amd.SparseGraph(ms_gat_amd.synthetic_adjacency(20, 20, 0))
    g.__dict__["_gacn_plans"] = {"probe": object()}      # what _gacn_plan() creates on first use
    ref = weakref.ref(g)
    del g
    gc.collect()
    assert ref() is None


def test_graph_cache_hits_same_tensor_and_detects_recycled_storage():
    import ms_gat_amd
    from ms_gat_amd import graph
    a = ms_gat_amd.synthetic_adjacency(30, 30, 0)
    g1 = graph.graph_of(a)
    assert graph.graph_of(a) is g1
    assert graph.graph_of(a.view(30, 30)) is g1          # a view of the same storage: contents equal
    a.add_(0.0)                                          # version bump -> rebuilt
    assert graph.graph_of(a) is not g1
    b = a.clone()
    g2 = graph.graph_of(b)
    b.data.copy_(ms_gat_amd.synthetic_adjacency(30, 31, 1))  # same storage & version, new contents
    assert graph.graph_of(b.data.view(30, 30)) is not g2


def test_graph_cache_compares_a_new_object_at_a_cached_address_once(monkeypatch):
    """A tensor OBJECT the cache has not seen at an address it has (a new model built where a freed model's adjacency
    lived) is confirmed by a content comparison ONCE and then hit by identity: the comparison reads the device back, so
    a per-call comparison cost a host sync per GACN call and made the first HIP-graph capture of such a model fail
    ("operation not permitted when stream is capturing", found by bench.py's full_step_cfg3 in round 4)."""
    import ms_gat_amd
    from ms_gat_amd import graph
    a = ms_gat_amd.synthetic_adjacency(20, 22, 5)
    g1 = graph.graph_of(a)
    v = a.view(20, 20)                                   # another object, same storage address and version
    calls = []
    real = torch.equal
    monkeypatch.setattr(torch, "equal", lambda x, y: calls.append(1) or real(x, y))
    assert graph.graph_of(v) is g1 and len(calls) == 1
    assert graph.graph_of(v) is g1 and len(calls) == 1   # by identity now
    assert graph.graph_of(a) is g1 and len(calls) == 2   # the first object is the stranger now: compared once, remembered
    assert graph.graph_of(a) is g1 and len(calls) == 2


def test_cpu_tensors_are_refused_not_silently_computed():
    import ms_gat_amd
    from ms_gat_amd._lib import MsgatError
    m = ms_gat_amd.GACN(3, 24, 12)
    assert sorted(k for k, _ in m.named_parameters()) == ["W", "gatt.Wg", "gatt.alpha"]
    assert tuple(m.W.shape) == (24, 3) and tuple(m.gatt.Wg.shape) == (12, 12) and tuple(m.gatt.alpha.shape) == (3,)
    with pytest.raises(MsgatError):
        m(torch.randn(1, 3, 8, 12), torch.eye(8))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ms_gat_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ms_gat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "dense_torch" not in text and "gat_oracle" not in text, f


def test_gate_sum_and_bias_join_reject_mismatched_operands():
    """Host-side argument checks of the two step-tail ops (no device work: they fail before any launch)."""
    from ms_gat_amd import ops
    pred = torch.zeros(3, 4, 5, 2)
    with pytest.raises((ValueError, RuntimeError)):
        ops.gate_sum(pred, None, None, torch.zeros(3, 5, 2))          # CPU tensors: the product has no CPU path
    wide, narrow = torch.zeros(3, 8), torch.zeros(3, 9)
    with pytest.raises(ValueError):
        ops.bias_join(wide, narrow)                                   # the narrow bias must fit the wide one
    with pytest.raises(ValueError):
        ops.bias_join(torch.zeros(2, 8), torch.zeros(3, 4))           # one row per relation in both
    out = ops.bias_join(torch.ones(2, 4), torch.full((2, 3), 2.0))    # plain torch ops: runs anywhere
    assert out.tolist() == [[3.0, 3.0, 3.0, 1.0]] * 2
