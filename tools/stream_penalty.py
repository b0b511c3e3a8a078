import sys, os, statistics
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
sync = lambda: torch.cuda.synchronize(dev)
w, per = bench.timed_steps(hp.step, 50, 10, dev, sync)
print("plain", round(w / 50 * 1e3, 4), round(statistics.median(per), 4))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    t = torch.zeros(16, device=dev) + 1
sync()
w, per = bench.timed_steps(hp.step, 50, 10, dev, sync)
print("after a side stream was used", round(w / 50 * 1e3, 4), round(statistics.median(per), 4))
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
w, per = bench.timed_steps(hp.step, 50, 10, dev, sync)
print("after init_process_group(nccl)", round(w / 50 * 1e3, 4), round(statistics.median(per), 4))
x = torch.ones(4, device=dev); dist.all_reduce(x); sync()
w, per = bench.timed_steps(hp.step, 50, 10, dev, sync)
print("after one all_reduce", round(w / 50 * 1e3, 4), round(statistics.median(per), 4))
dist.destroy_process_group()
