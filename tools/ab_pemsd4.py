"""Same-box A/B of the PEMSD4 object (configs[1]): run in two trees (see tools/ab_step.py)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402

o = bench.small_graph_object("pemsd4", torch.device("cuda:0"))
print(os.path.basename(os.getcwd()) or "repo", "pemsd4 hot-path step replayed ms:", o.get("ms_per_step"), "gpu busy:", o.get("gpu_busy_ms_per_step"),
      "train step eager / graph:", o.get("train_step_eager_launch_ms"), o.get("train_step_hip_graph_ms"), flush=True)
