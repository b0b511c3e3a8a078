import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """max|a-b| / max|b| -- the parity figure quoted everywhere (bar: 1e-4 in fp32)."""
    a, b = (t.detach().cpu().numpy() if hasattr(t, "detach") else t for t in (a, b))
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / scale


# Achieved errors of the parity tests: `record_err(test, tensor, err, bar)` rows are written at session end to
# $MSGAT_PARITY_LOG (a .tsv; committed per round under profiles/) so the margin under the bar is on record.
_PARITY_ROWS = []


def record_err(what, key, err, bar):
    _PARITY_ROWS.append((str(what), str(key), float(err), float(bar)))


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("MSGAT_PARITY_LOG")
    if not path or not _PARITY_ROWS:
        return
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# achieved parity errors, max|a-b| / max|b| per tensor (tests/conftest.py rel_err)\n")
        f.write("test\ttensor\trel_err\tbar\tmargin_x\n")
        for what, key, err, bar in _PARITY_ROWS:
            f.write(f"{what}\t{key}\t{err:.3e}\t{bar:.0e}\t{bar / max(err, 1e-30):.1f}\n")


GATT_CASES = ["b2c3n16", "b2c1n64", "b2c72n64", "b2c3n307"]


@pytest.fixture(params=GATT_CASES)
def gatt_case(request):
    g = load_golden(f"gatt_{request.param}.npz")
    c = load_golden(f"gacn_{request.param}.npz")
    return request.param, g, c
