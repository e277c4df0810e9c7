import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
import vilco_amd.modeling as vm
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)
counts = {}
import functools
def wrap(name):
    f = getattr(model, name)
    @functools.wraps(f)
    def w(*a, **k):
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            out = f(*a, **k)
            torch.cuda.synchronize()
        ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        counts.setdefault(name, []).append((len(ev), sum(e.device_time for e in ev) if hasattr(ev[0], 'device_time') else sum(e.cuda_time for e in ev)))
        return out
    setattr(model, name, w)
for n in ("preprocessing", "label_points", "losses"):
    wrap(n)
for _ in range(3):
    model.zero_grad(set_to_none=True)
    model(batch, is_training=True)['final_loss'].backward()
for k, v in counts.items():
    print(k, "forward kernels %d, GPU time %.0f us" % v[-1])
