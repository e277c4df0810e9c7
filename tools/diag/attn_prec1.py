import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vilco_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(8)
B, Tq, Tk, H, hd = 2, 200, 157, 3, 32
q, k, v = torch.randn(B, Tq, H * hd, device=dev) * 3, torch.randn(B, Tk, H * hd, device=dev) * 1e-3, torch.randn(B, Tk, H * hd, device=dev) * 50
lens = torch.tensor([Tk, Tk - 9], dtype=torch.int32, device=dev)
ops.set_precision("split3")
o = ops.attention(q, k, v, lens, H, 0.2)
torch.cuda.synchronize()
print("split3", float(o.abs().max()))
