"""Builds libmsgat_hip.so (gfx950 only) in-tree with hipcc.

    python -m ms_gat_amd.build          # incremental
    python -m ms_gat_amd.build --force  # rebuild everything

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJDIR = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "libmsgat_hip.so")

SOURCES = ["api.hip", "project.hip", "mfma.hip", "dense.hip", "dense_bf16.hip", "scores.hip", "aggregate.hip", "reduce.hip", "layernorm.hip", "branches.hip", "smallatt.hip", "tail.hip", "graph_host.cpp"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "sell.hpp"), os.path.join(CSRC, "rowtile.hpp"), os.path.join(CSRC, "halfsplit.hpp"), os.path.join(INCLUDE, "msgat_hip.h")]
# diagnostic translation units (never part of the product library): `build(lab=True)` / `--lab` adds them and their
# extra, undeclared entry points for tools/stress_kernels.py --lab
LAB_SOURCES = [os.path.join(ROOT, "tools", "agg_sell_lab.hip"), os.path.join(ROOT, "tools", "bwd_sell_lab.hip")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         f"--offload-arch={ARCH}",
         "-I" + INCLUDE, "-I" + CSRC]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmsgat_hip.so cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(hipcc: str, src: str, obj: str, extra) -> None:
    cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
    if src.endswith(".cpp"):
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-Wall", "-I" + INCLUDE, "-x", "c++", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force: bool = False, verbose: bool = True, extra_flags=(), lab: bool = False, out: str = LIB,
          objdir: str = OBJDIR) -> str:
    """Compile every HIP source for gfx950 and link libmsgat_hip.so; returns its path.
    lab=True: a diagnostic build (give it another `out` and `objdir`) that also carries the tools/*_lab.hip units
    and is compiled with -DMSGAT_LAB (environment switches for A/B runs of kernel forms)."""
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    if lab:
        extra_flags = [*extra_flags, "-DMSGAT_LAB"]
    jobs, objs = [], []
    for name in SOURCES + (LAB_SOURCES if lab else []):
        src = name if os.path.isabs(name) else os.path.join(CSRC, name)
        obj = os.path.join(objdir, os.path.basename(name).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, *HEADERS, os.path.abspath(__file__)]):
            jobs.append((src, obj))
    if jobs:
        if verbose:
            print(f"[ms_gat_amd.build] hipcc --offload-arch={ARCH}: {', '.join(os.path.basename(s) for s, _ in jobs)}",
                  flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for f in [ex.submit(_compile, hipcc, s, o, list(extra_flags)) for s, o in jobs]:
                f.result()
    if jobs or force or _stale(out, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[ms_gat_amd.build] linked {out}", flush=True)
    return out


if __name__ == "__main__":
    if "--lab" in sys.argv:   # python -m ms_gat_amd.build --lab  ->  build/lab/libmsgat_lab.so
        os.makedirs(os.path.join(ROOT, "build", "lab"), exist_ok=True)
        build(force="--force" in sys.argv, lab=True, out=os.path.join(ROOT, "build", "lab", "libmsgat_lab.so"),
              objdir=os.path.join(ROOT, "build", "lab", "obj"))
    else:
        build(force="--force" in sys.argv)
