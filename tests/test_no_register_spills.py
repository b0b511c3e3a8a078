"""CPU: no kernel of libmsgat_hip.so spills vector registers.

Round 3 shipped register-staged contraction blocks ([48 x 96], [48 x 80], [32 x 96] channels) and a seven-tile two-pass
projection that the dispatch could select for msgat96's widths (reference main.py:17, msgat.py:220-229) and that spilled up
to 93 VGPRs to scratch.  Every .hip source is compiled to gfx950 assembly here (device side only, the build's flags) and
the kernel metadata is read: `vgpr_spill_count` must be 0 for EVERY kernel in the library -- whatever a factory selects
is then spill-free by construction.  Scalar spills (SGPR -> VGPR lanes, no memory traffic) are reported, not refused.
"""
import concurrent.futures as cf
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _kernel_metadata(src, out_dir):
    from ms_gat_amd import build
    asm = os.path.join(out_dir, os.path.basename(src) + ".s")
    cmd = [build._hipcc(), *build.FLAGS, "--cuda-device-only", "-S", "-o", asm, src]
    cmd = [c for c in cmd if c != "-fPIC"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = []
    for blk in open(asm).read().split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
        rows.append((name.group(1), get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count")))
    return rows


def test_no_kernel_of_the_library_spills_vector_registers(tmp_path):
    from ms_gat_amd import build
    srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES if s.endswith(".hip")]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        tables = list(ex.map(lambda s: _kernel_metadata(s, str(tmp_path)), srcs))
    kernels = [row for t in tables for row in t]
    assert len(kernels) > 150, "kernel metadata not found in the assembly"
    spilled = [(n, v, sp) for n, v, sp, _ in kernels if sp > 0]
    assert not spilled, f"kernels with VGPR spills: {spilled}"
    if os.environ.get("MSGAT_REGS_LOG"):   # tools: keep the table (profiles/rNN/kernel_registers.txt)
        with open(os.environ["MSGAT_REGS_LOG"], "w") as f:
            f.write("# hipcc -S metadata of every kernel in libmsgat_hip.so: vgpr_count vgpr_spill_count sgpr_spill_count\n")
            for n, v, sp, ss in sorted(kernels):
                f.write(f"{v:4d} {sp:3d} {ss:3d}  {n}\n")
