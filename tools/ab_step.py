import sys, os, json
sys.path.insert(0, os.getcwd())
from ms_gat_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, bench, statistics
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
wall, per = bench.timed_steps(hp.step, 50, 10, dev, lambda: torch.cuda.synchronize(dev))
ts = bench.TrainStep(dict(bench.CFG4, R=3), dev)
w3, p3 = bench.time_train_step(ts, 20, 5, lambda: torch.cuda.synchronize(dev))
print(sys.argv[1] if len(sys.argv) > 1 else "in-tree", "hot", round(wall / 50 * 1e3, 4), round(statistics.median(per), 4), "full R=3", round(w3 / 20 * 1e3, 3), round(statistics.median(p3), 3))
