"""flash attention (hd = 64 fast path and general kernels) vs an fp64 softmax reference, ragged key lengths; timing"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops

dev = torch.device("cuda:0")
ops.set_precision("f16x2")


def ref(q, k, v, lens, H, scale):
    B, Tq, C = q.shape
    Tk = k.shape[1]
    hd = C // H
    qh, kh, vh = [t.double().view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v)]
    s = qh @ kh.transpose(-1, -2) * scale
    m = torch.arange(Tk, device=q.device)[None, :] < lens[:, None].long()
    s = s.masked_fill(~m[:, None, None, :], float('-inf'))
    p = torch.softmax(s, -1)
    o = (p @ vh).transpose(1, 2).reshape(B, Tq, C)
    return o, torch.logsumexp(s, -1)


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


torch.manual_seed(0)
for (B, Tq, Tk, H, hd, lens) in [(2, 200, 200, 2, 64, [200, 131]), (2, 130, 77, 3, 64, [77, 5]), (1, 64, 64, 1, 64, [64]),
                                 (2, 333, 333, 2, 64, [1, 333]), (2, 2304, 2304, 16, 64, [2304, 2287]), (2, 1152, 77, 16, 64, [77, 60])]:
    C = H * hd
    q, k, v = [torch.randn(B, T, C, device=dev) * s for T, s in ((Tq, 1.0), (Tk, 2.0), (Tk, 0.5))]
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    o, lse = ops._flash_fwd(q, k, v, None, lt, H, 0.125, 0)
    ow, lw = ref(q, k, v, lt, H, 0.125)
    eo = ((o.double() - ow).abs().max() / ow.abs().max()).item()
    el = (lse.double() - lw).abs().max().item()
    t = timeit(lambda: ops._flash_fwd(q, k, v, None, lt, H, 0.125, 0))
    do = torch.randn_like(o) * 3.0
    dq, dk, dv, _ = ops._flash_bwd(q, k, v, None, lt, o, lse, do, H, 0.125, 0, False)
    qd, kd, vd = [x.detach().double().requires_grad_(True) for x in (q, k, v)]
    ref(qd, kd, vd, lt, H, 0.125)[0].backward(do.double())
    eg = [((g.double() - w.grad).abs().max() / w.grad.abs().max()).item() for g, w in ((dq, qd), (dk, kd), (dv, vd))]
    tb = timeit(lambda: ops._flash_bwd(q, k, v, None, lt, o, lse, do, H, 0.125, 0, False))
    print("B%d Tq%d Tk%d H%d: o err %.2e  lse err %.2e  dq %.2e dk %.2e dv %.2e   fwd call %.3f ms  bwd call %.3f ms" %
          (B, Tq, Tk, H, eo, el, eg[0], eg[1], eg[2], t, tb))
