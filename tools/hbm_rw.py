"""HBM read / write / copy rates of this MI355X with plain streaming kernels (torch fill_, copy_, a reduction) and with
vilco_axpby: what a kernel whose traffic is mostly WRITES can reach."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops, _lib
dev = torch.device("cuda:0")
n = 256 * 1024 * 1024          # 1 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
a.normal_(); b.zero_()
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
gb = n * 4 / 1e9
print("fill_ (write only)      %.2f TB/s" % (gb / t(lambda: b.fill_(1.0)) / 1e3))
print("sum (read only)         %.2f TB/s" % (gb / t(lambda: a.sum()) / 1e3))
print("copy_ (1 read 1 write)  %.2f TB/s of bytes moved" % (2 * gb / t(lambda: b.copy_(a)) / 1e3))
lib = _lib.load()
print("vilco_axpby out=2a      %.2f TB/s of bytes moved" % (2 * gb / t(lambda: lib.vilco_axpby(b.data_ptr(), a.data_ptr(), None, 2.0, 0.0, n, ops._stream())) / 1e3))
c = torch.empty(3, n // 4, device=dev)
a4 = a[: n // 4]
def one_in_three_out():
    c[0].copy_(a4); c[1].copy_(a4); c[2].copy_(a4)
print("3 copies of a quarter   %.2f TB/s of bytes moved (1.5 GB written, 0.75 GB read)" % ((6 * gb / 4) / t(one_in_three_out) / 1e3))
torch.add(a4, 1.0, out=c[0])
print("mul out-of-place        %.2f TB/s" % (2 * gb / 4 / t(lambda: torch.mul(a4, 2.0, out=c[0])) / 1e3))
