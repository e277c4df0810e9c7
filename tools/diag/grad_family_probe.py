"""Where does the fp32 reference's 0.1-0.2 (of the tensor maximum) distance from an exact run of itself on `branch.*.mlp.3.weight`
come from?  (DESIGN_LOG.md 4, round 4.)  dW = sum_t dY[t] X[t]^T: this script runs the oracle at config P (no dropout) in fp32 and in
fp64, catches both operands of branch.0's second MLP conv, and compares them.
Findings (build container, 8 cores, ~3 min): X agrees to 8e-7; dY differs by 0.39 of its maximum on single elements, at PAIRS of
adjacent tokens (18/19, 77/78) with equal error -- a gradient element that one run routes to token 2j and the other to token 2j+1:
the stride-2 max-pool of the next block's skip path (blocks.py:553-556) picks its argmax among near-tied neighbours differently once
the inputs differ in the last bit.  dW(x fp64, dY fp32) reproduces the whole 0.11; dW(x fp32, dY fp64) is exact to 3e-7, and the sum
itself is well conditioned (sum |terms| / |sum| ~ 3).  The routed element is a full residual-stream gradient, the tensors it lands in
carry gradients scaled by the 1e-4 AffineDropPath factors: one misrouted element is ~10 % of their maximum."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import vilco_amd.modeling as vm
from oracle import mq_oracle
cfg = bench.p_config()
KEY = 'backbone.branch.0.mlp.3.weight'
real = F.conv1d
def run(dtype):
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet()))
    p = {k: (v.detach().to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    vl = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in bench.synth_batch(2, "cpu")]
    c = {}
    def conv1d(x, w, *a, **k):
        y = real(x, w, *a, **k)
        if w is p[KEY]:
            c['x'] = x.detach().double()
            y.register_hook(lambda g: c.__setitem__('dy', g.detach().double()))
        return y
    mq_oracle.F.conv1d = conv1d
    try:
        l, _ = mq_oracle.forward_losses(p, cfg, vl); l['final_loss'].backward()
    finally:
        mq_oracle.F.conv1d = real
    c['dw'] = p[KEY].grad.detach().double()[:, :, 0]
    return c
a, b = run(torch.float32), run(torch.float64)
for k in ('x', 'dy', 'dw'):
    d = (a[k] - b[k]).abs()
    print(k, "max rel-to-max %.3e" % float(d.max() / b[k].abs().max()), "l2 %.3e" % float((a[k] - b[k]).norm() / b[k].norm()), "max|.| %.3e" % float(b[k].abs().max()))
dy32, dy64 = a['dy'], b['dy']
# where in (b, n, t) is dy most different; per-token and per-channel error profiles
d = (dy32 - dy64).abs()
print("dy err by token (top 5 t):", torch.topk(d.amax(dim=(0, 1)), 5), "valid len clip1 at this level:", (2287 + 1) // 2)
print("dy err by sample:", d.amax(dim=(1, 2)))
# recompute dW from fp64 x with fp32 dy and vice versa
print("dW(x64, dy32) vs dW64 rel-to-max %.3e" % float((torch.einsum('bnt,bkt->nk', dy32, b['x']) - b['dw']).abs().max() / b['dw'].abs().max()))
print("dW(x32, dy64) vs dW64 rel-to-max %.3e" % float((torch.einsum('bnt,bkt->nk', dy64, a['x']) - b['dw']).abs().max() / b['dw'].abs().max()))
