import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
import vilco_amd.modeling as vm
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)
def step():
    model.zero_grad(set_to_none=True)
    with record_function("FWD"):
        l = model(batch, is_training=True)
    with record_function("BWD"):
        l['final_loss'].backward()
for _ in range(3): step()
torch.cuda.synchronize()
# phase-wise kernel counts: wrap model methods
import types
names = ["preprocessing", "label_points", "losses"]
for n in names:
    f = getattr(model, n)
    def mk(f, n):
        def w(*a, **k):
            with record_function("PH_" + n):
                return f(*a, **k)
        return w
    setattr(model, n, mk(f, n))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
for e in rows[:45]:
    print("%-60s count %5d  cuda %8.1f us  cpu %8.1f us" % (e.key[:60], e.count, e.device_time_total, e.cpu_time_total))
