import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from ms_gat_amd import data, engine, model
variant = sys.argv[1]
torch.manual_seed(0)
dev = torch.device("cuda:0")
ds = data.SyntheticPEMS(n_nodes=40, n_edges=50, n_channels=1, in_hours=[1, 2], batch_size=8, days=2)
net = model.msgat48(n_components=2, in_channels=1, in_timesteps=12, out_timesteps=12, use_te=True, adj=ds.adj).to(dev)
batches = [b for _, b in zip(range(6), ds.training)]
tr = engine.Trainer(net, 50.0, "/tmp/dbg_" + variant, hip_graph=False)
if variant == "a":      # one eager training epoch of 1 batch, then capture
    tr.run_epoch(batches[:1], gpu_id=0, epoch=1, mode="train")
elif variant == "b":    # eager forward only
    with torch.no_grad():
        net(*[t.to(dev) for t in batches[0][:-1]])
elif variant == "c":    # eager fwd+bwd without optimizer step / metrics
    b = [t.to(dev) for t in batches[0]]
    tr._loss(net(*b[:-1]), b[-1], None).backward()
elif variant == "d":    # eager fwd+bwd+step, no metrics
    b = [t.to(dev) for t in batches[0]]
    tr.optimizer.zero_grad(set_to_none=True)
    tr._loss(net(*b[:-1]), b[-1], None).backward()
    tr.optimizer.step()
elif variant == "e":
    pass
elif variant in ("f", "g", "h"):
    tr.hip_graph = "auto"
    tr.graph_after = {"f": 3, "g": 0, "h": 1}[variant]
    l = tr.run_epoch(batches, gpu_id=0, epoch=1, mode="train")
    print(variant, "ok", l, flush=True)
    sys.exit(0)
torch.cuda.synchronize()
print("eager part done", flush=True)
tr.hip_graph = True
l = tr.run_epoch(batches, gpu_id=0, epoch=2, mode="train")
print(variant, "ok", l, flush=True)
