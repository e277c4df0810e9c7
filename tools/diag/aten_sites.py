"""Which ATen kernels does an eager P step still launch, from where?  torch.profiler with stacks: prints the ATen ops
(name, input shapes, python call site) that own the remaining non-vilco kernels of the step."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
def step():
    model.zero_grad(set_to_none=True)
    l = model(batch, is_training=True)
    l['final_loss'].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = [e for e in ka if e.key.startswith("aten::") and e.self_device_time_total > 0]
for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:45]:
    st = [x for x in (e.stack or []) if "vilco_amd" in x or "bench.py" in x]
    print("%3d x %8.1f us  %-20s %-80s %s" % (e.count, e.self_device_time_total, e.key, str(e.input_shapes)[:80], st[0][-80:] if st else "(autograd engine)"))
