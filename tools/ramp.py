import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
w, per = bench.timed_steps(hp.step, 60, 0, dev, lambda: torch.cuda.synchronize(dev))
print(" ".join(f"{p:.3f}" for p in per))
