import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
ts = bench.TrainStep(dict(bench.CFG4, R=3), dev)
sync = lambda: torch.cuda.synchronize(dev)
for rep in range(3):
    wall, per = bench.time_train_step(ts, 20, 5, sync)
    print(f"rep {rep}: wall {wall/20*1e3:.3f} ms/step; per-step:", " ".join(f"{p:.2f}" for p in per))
