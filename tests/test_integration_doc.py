"""INTEGRATION.md section 4 ("Option B -- bind the C ABI directly") is executable documentation: the ctypes stub a
maintainer of the reference would paste into `src/models/msgat_hip.py`.  These tests extract that Python block from the
document and run it, so an ABI change that leaves the document behind fails here:

* CPU: the block imports against the built library (its `msgat_abi_version()` assert included) and its structures
  have the layouts of `ms_gat_amd/_lib.py` (which tests/test_abi_and_host.py pins to include/msgat_hip.h);
* GPU: the stub's `build_graph` + `_GACN` reproduce the reference's GACN golden vectors (tests/golden/gacn_b2c3n16.npz,
  gatt_b2c3n16.npz: /root/reference/src/models/msgat.py:25-28, attention.py:32-36) -- forward and all four gradients.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_parity, load_golden


def _stub_namespace():
    from ms_gat_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 4. Option B"):text.index("## 5.")]
    blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
    assert blocks, "INTEGRATION.md section 4 lost its python block"
    code = blocks[0]
    assert 'C.CDLL("libmsgat_hip.so")' in code
    code = code.replace('C.CDLL("libmsgat_hip.so")', f"C.CDLL({_lib.LIB_PATH!r})")   # the document names it by soname only
    ns = {}
    exec(compile(code, "INTEGRATION.md#4", "exec"), ns)
    return ns


def test_documented_stub_imports_and_its_structs_match_the_binding():
    from ms_gat_amd import _lib
    ns = _stub_namespace()
    for name in ("Sell", "Graph", "Shape", "Fwd", "Bwd"):
        doc, ours = ns[name], getattr(_lib, name)
        assert C.sizeof(doc) == C.sizeof(ours), name
        assert [f[0] for f in doc._fields_] == [f[0] for f in ours._fields_], name
        for f in doc._fields_:
            assert getattr(doc, f[0]).offset == getattr(ours, f[0]).offset, (name, f[0])
            assert getattr(doc, f[0]).size == getattr(ours, f[0]).size, (name, f[0])
    assert f"== {_lib.ABI_VERSION}" in open(os.path.join(ROOT, "INTEGRATION.md")).read().split("## 4. Option B")[1].split("## 5.")[0]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["b2c3n16", "b2c72n64"])
def test_documented_stub_reproduces_the_reference_gacn_golden(tag):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    ns = _stub_namespace()
    g, c = load_golden(f"gatt_{tag}.npz"), load_golden(f"gacn_{tag}.npz")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    graph = ns["build_graph"](to(g["adj"]))
    x = to(g["x"]).requires_grad_(True)
    alpha, Wg, W = (to(a).requires_grad_(True) for a in (g["alpha"], g["Wg"], c["W"]))
    z = ns["_GACN"].apply(x, alpha, Wg, W, graph)
    z.backward(to(c["dz"]))
    torch.cuda.synchronize()
    what = f"integration_doc_stub_{tag}"
    assert_parity(z.detach().cpu(), c["z"], what, "z")
    for key, t in (("dx", x), ("dWg", Wg), ("dalpha", alpha), ("dW", W)):
        assert_parity(t.grad.cpu(), c[key], what, key)
    # GraphAttention alone (W = None) through the same stub
    x2 = to(g["x"]).requires_grad_(True)
    y = ns["_GACN"].apply(x2, alpha.detach().requires_grad_(True), Wg.detach().requires_grad_(True), None, graph)
    y.backward(to(g["dy"]))
    assert_parity(y.detach().cpu(), g["y"], what, "y")
    assert_parity(x2.grad.cpu(), g["dx"], what, "gatt_dx")
