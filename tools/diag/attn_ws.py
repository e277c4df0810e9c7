import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vilco_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(8)
B, Tq, Tk, H, hd = 2, 200, 157, 3, 32
q, k, v = torch.randn(B, Tq, H * hd, device=dev) * 3, torch.randn(B, Tk, H * hd, device=dev) * 1e-3, torch.randn(B, Tk, H * hd, device=dev) * 50
lens = torch.tensor([Tk, Tk - 9], dtype=torch.int32, device=dev)
prec = 2
o = torch.full_like(q, 7.0)
lse = torch.full((B, H, Tq), 7.0, device=dev)
nws = lib.vilco_attn_fwd_workspace(B, H, Tq, Tk, hd, prec)
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
rc = lib.vilco_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, lens.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, Tq, Tk, hd, 0.2, 0, prec, 0.0, 0, ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rc", rc, "o max", float(o.abs().max()), "lse", float(lse.abs().max()), float(lse.min()))
w16 = ws.view(torch.int16)
nz = (w16 != 0)
idx = nz.nonzero()
print("ws bytes", nws, "nonzero int16:", int(nz.sum()), "first", int(idx[0]) if len(idx) else None, "last", int(idx[-1]) if len(idx) else None)
# coarse histogram of nonzero density per 64 KB
per = 32768
for i in range(0, w16.numel(), per):
    print(i * 2, int(nz[i:i + per].sum()), end=" | ")
print()
