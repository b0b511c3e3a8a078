#!/usr/bin/env python3
"""Repeats the tests that would show a miscounted wait of the LDS-DMA rings (k_chanpair_glds, k_agg_ring) as run-to-run
differences -- each pytest run is a fresh process, so schedules differ.  On the GPU box:
    python tools/soak_tests.py [runs=6]"""
import subprocess
import sys

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rc = 0
for i in range(runs):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_branches.py", "tests/test_gpu_parity.py", "-q", "-x", "-k",
                        "contract or project_backward or ring or reproducible or headline", "-p", "no:cacheprovider"],
                       capture_output=True, text=True)
    print(i, (r.stdout.strip().splitlines() or ["?"])[-1], flush=True)
    rc |= r.returncode
sys.exit(rc)
