"""msgat72 training step (R = 3) with a lab build of the library: MSGAT_LAB_* knobs in the environment."""
import os, statistics, sys
sys.path.insert(0, os.getcwd())
from ms_gat_amd import _lib
_lib.LIB_PATH = os.path.abspath("build/lab/libmsgat_lab.so")
import torch
import bench
dev = torch.device("cuda:0")
sync = lambda: torch.cuda.synchronize(dev)
R = int(os.environ.get("LAB_R", "3"))
ts = bench.TrainStep(dict(bench.CFG4, R=R), dev)
for rep in range(2):
    w3, p3 = bench.time_train_step(ts, 20, 5, sync)
    print("CCPB", os.environ.get("MSGAT_LAB_CCPB", "-"), f"training step R={R} wall / median ms:", round(w3 / 20 * 1e3, 3), round(statistics.median(p3), 3), flush=True)
