#!/usr/bin/env python3
"""profiles/rNN/hbm_traffic.json from the four per-kernel PMC tables tools/profile_round.sh leaves
(pmc_{hot,stress}_{FETCH,WRITE}_SIZE.txt: FETCH_SIZE and WRITE_SIZE in KB per launch, separate passes).

    python tools/pmc_tables_to_json.py gpurun_out/r03 profiles/r03/hbm_traffic.json

Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 on gfx950 for 16-B-per-lane streaming reads, WRITE_SIZE x1."""
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
G, N, T, C, Co, nnz = 96, 883, 12, 72, 24, 2615
sG, sN, sCu, snnz = 256, 8192, 24, 139264
ALG = {  # algorithmic bytes per launch, the workloads of tools/kbench.py / tools/stress_kernels.py
    "msgat::k_agg_ring<3, 3>": 2 * 4 * G * Co * N * T + 4 * G * nnz + 8 * nnz + 4 * (N + 1),
    "msgat::k_agg_lds<3, false>": 2 * 4 * G * Co * N * T + 4 * G * nnz + 8 * nnz + 4 * (N + 1),
    "msgat::k_agg_sell<3>": 2 * 4 * sG * sCu * sN * T + 4 * sG * snnz + 8 * snnz + 4 * (sN + 1),
    "msgat::k_sddmm_sellreg<3>": 2 * 4 * sG * sCu * sN * T,
    "msgat::k_project_mfma<2, true, true, false, false, false>": 4 * G * N * T * (C + Co + 1),
    "msgat::k_chanpair_glds<2, 5, 128, 3, 0>": 4 * G * N * T * (Co + 1 + C),          # dW, dalpha alone (kbench contract)
    "msgat::k_chanpair_glds<2, 5, 128, 3, 1>": 4 * G * N * T * (Co + 1 + 2 * C),      # dW, dalpha AND dx (the hot path's pass)
    "msgat::k_chanpair_glds<7, 5, 64, 3, 2>": 4 * G * N * T * (98 + 2 * C),           # 98 x 73 mixing backward, one pass
    "msgat::k_chanpair_glds<5, 5, 64, 3, 2>": 4 * G * N * T * (3 * C),                # 72 x 73 mixing backward, one pass
}


def table(path):
    d, name = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+\((\d+) launches; per launch\)", line)
        if m:
            name = m.group(1)
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
        if m and name:
            d[name] = float(m.group(2))
    return d


kernels = {}
for wl in ("hot", "stress"):
    f, w = table(os.path.join(src, f"pmc_{wl}_FETCH_SIZE.txt")), table(os.path.join(src, f"pmc_{wl}_WRITE_SIZE.txt"))
    for k in sorted(f):
        if not k.startswith("msgat::"):
            continue
        e = {"workload": "pemsd7" if wl == "hot" else "stress", "fetch_kb_raw": f[k], "write_kb_raw": w.get(k, 0.0),
             "fetch_factor": 2, "hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0.0)) * 1024),
             "hbm_bytes_per_launch_uncorrected": int((f[k] + w.get(k, 0.0)) * 1024)}
        if k in ALG:
            e["algorithmic_bytes"] = ALG[k]
            e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / ALG[k], 3)
        kernels[k] = e
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile_round.sh, part pmc): hot-path "
                     "kernels: python3 tools/kbench.py --eager --reps 5 (pemsd7: G=96, N=883, T=12, C=72, Co=24); stress "
                     "kernels: python3 tools/stress_kernels.py --reps 2 (N=8192, degree 16, R=4, B=64: G=256, Cu=24)",
           "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced streaming reads; for the 16-B pieces at a "
                         "48-B stride of the column staging the x2 figure is an upper bound, the raw one a lower bound); WRITE_SIZE x1",
           "kernels": kernels}, open(out, "w"), indent=1)
for k, e in kernels.items():
    if "algorithmic_bytes" in e:
        print(k, e["hbm_bytes_per_launch"], e["traffic_over_algorithmic"])
