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
qw, qr, k, v = [torch.randn(B, T, C, device=dev, requires_grad=True) for _ in range(4)]
kr = torch.randn(2 * T, C, device=dev, requires_grad=True)
lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
def rel():
    o = ops.rel_attention(qw, qr, k, v, kr, lens, H, 0.125)
    o.backward(torch.ones_like(o))
def plain():
    o = ops.attention(qw, k, v, lens, H, 0.125)
    o.backward(torch.ones_like(o))
print("rel_attention fwd+bwd %.3f ms ; plain attention fwd+bwd %.3f ms" % (timeit(rel), timeit(plain)))
