import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd import ops
dev = torch.device("cuda:0")
dp, dpath, xl = [float(a) for a in (sys.argv[1:4] or (0.1, 0.1, 0.1))]
cfg = bench.p_config(dp, dpath)
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(xl))).to(dev).train()
batch = bench.synth_batch(2, dev)
out = model(batch, is_training=True)
print({k: float(v) for k, v in out.items()})
out['final_loss'].backward()
bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(len(bad), "non-finite grads; first:", bad[:12])
for it in range(8):
    model.zero_grad(set_to_none=True)
    with torch.autograd.detect_anomaly(check_nan=True):
        out = model(batch, is_training=True)
        out['final_loss'].backward()
    bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(it, float(out['final_loss']), len(bad), bad[:6])
    if bad:
        break
