import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops
dev = torch.device("cuda:0")
B, T, H, hd = 2, 2304, 16, 64
C = H * hd
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
q, k, v = [torch.randn(B, T, C, device=dev) for _ in range(3)]
lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
dense = torch.randn(B, H, T, T, device=dev) * 0.1
full = torch.randn(B, H, T, 2 * T, device=dev) * 0.1
for name, bias, mode, want in [("mode0 nobias", None, 0, False), ("mode1 nobias", None, 1, False), ("mode1 dense bias", dense, 1, False),
                               ("mode1 dense bias +dbias", dense, 1, True), ("mode3 full bias", full, 3, False), ("mode3 full bias +dbias", full, 3, True)]:
    o, lse = ops._flash_fwd(q, k, v, bias, lens, H, 0.125, mode)
    do = torch.randn_like(o)
    tf = timeit(lambda: ops._flash_fwd(q, k, v, bias, lens, H, 0.125, mode))
    tb = timeit(lambda: ops._flash_bwd(q, k, v, bias, lens, o, lse, do, H, 0.125, mode, want))
    print("%-26s fwd %.3f ms  bwd %.3f ms" % (name, tf, tb))
