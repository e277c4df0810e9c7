import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

dev = torch.device("cuda:0")
for prec in (sys.argv[1:] or ["split3"]):
    ops.set_precision(prec)
    for (B, T, H, hd) in [(2, 2304, 16, 64), (2, 1152, 16, 64)]:
        C = H * hd
        q, k, v = [torch.randn(B, T, C, device=dev) for _ in range(3)]
        lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
        o, lse = ops._flash_fwd(q, k, v, None, lens, H, 0.125, 0)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops._flash_fwd(q, k, v, None, lens, H, 0.125, 0))
        tb = timeit(lambda: ops._flash_bwd(q, k, v, None, lens, o, lse, do, H, 0.125, 0, False))
        fl = 4.0 * B * T * T * C
        print("%s T=%d: fwd %.3f ms (%.0f TF)  bwd %.3f ms (%.0f TF)" % (prec, T, tf, fl / tf / 1e9, tb, 3.5 * fl / tb / 1e9))
