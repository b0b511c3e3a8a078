"""Same-box A/B of two builds: hot-path step and msgat72 training step (R = 3), wall and median by HIP events.

    python tools/ab_step.py                       # the tree it is started in (cwd), its own in-tree library
    (cd build/ab/r05 && python tools/ab_step.py)  # another round's tree (source + its built .so), on the SAME box

Every round keeps the previous round's tree under build/ab/<round>/ (`git archive` of the round's last commit + the library
built from it; not tracked) and commits the pair of lines as profiles/<round>/ab_*.txt.  `.gpurunignore` lists build/ab/ and
build/lab/ so that an ordinary lease stays small: take those two lines out for an A/B or lab call."""
import os
import statistics
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
out = []
for rep in range(3):
    wall, per = bench.timed_steps(hp.step, 50, 10, dev, sync)
    out.append((round(wall / 50 * 1e3, 4), round(statistics.median(per), 4)))
ts = bench.TrainStep(dict(bench.CFG4, R=3), dev)
w3, p3 = bench.time_train_step(ts, 20, 5, sync)
print(os.path.basename(os.getcwd()) or "repo", "hot-path step ms (wall, median) x3:", out, "| msgat72 training step R=3:",
      round(w3 / 20 * 1e3, 3), round(statistics.median(p3), 3), flush=True)
