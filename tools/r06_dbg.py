import os, sys
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
import ms_gat_amd
from ms_gat_amd import _lib, graph as G_
dev = torch.device("cuda:0")
N = 300
adj = ms_gat_amd.synthetic_adjacency(N, 340, seed=3).to(dev)
g = torch.Generator().manual_seed(7)
for Cc in (72, 3):
    m = ms_gat_amd.GACN(Cc, 24, 12).to(dev)
    x = torch.randn(6, Cc, N, 12, generator=g).to(dev)
    z = m(x, adj)
    print(Cc, "z nan", torch.isnan(z).sum().item(), "of", z.numel())
    # stage by stage
    L = _lib.lib()
    gr = G_.graph_of(adj)
    gs, keep = gr.on(dev)
    shape = _lib.Shape(1, 6, Cc, 24, N, 12)
    q = torch.einsum("bcnt,c->bnt", x, m.gatt.alpha.detach()).contiguous()
    kW, pq = torch.empty_like(q), torch.empty_like(q)
    lse = torch.empty(6, N, device=dev); E = torch.empty(6, gr.nnz, device=dev)
    nd = int(L.msgat_dense_scratch_bytes(C.byref(shape)))
    ds = torch.zeros(max(nd, 1), device=dev, dtype=torch.uint8)
    st = L.msgat_stage_scores(C.byref(shape), C.byref(gs), q.data_ptr(), m.gatt.Wg.detach().contiguous().data_ptr(), kW.data_ptr(), lse.data_ptr(), pq.data_ptr(), E.data_ptr(), None, ds.data_ptr() if nd else None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(" st", st, "nd", nd, "q absmax", q.abs().max().item(), "kW nan", torch.isnan(kW).sum().item(), "lse nan/inf", torch.isnan(lse).sum().item(), torch.isinf(lse).sum().item(), "pq nan", torch.isnan(pq).sum().item(), "E nan", torch.isnan(E).sum().item())
    S = torch.einsum("bnt,bmt->bnm", kW.double(), q.double())
    print(" S range", S.min().item(), S.max().item(), "lse ref diff", (torch.logsumexp(S, -1) / 0.6931471805599453 - lse.double()).abs().max().item())
    if nd:
        nb = nd - 256
        sc = ds[-(((6 * 12 + 255) // 256) * 256):].view(torch.float32)[:12]
        print(" scales", sc.tolist())
