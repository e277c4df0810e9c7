"""staged vs one-graph replay at config P: which gradients differ after the k-th call, and does it grow with the replays?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
import vilco_amd.modeling as vm
from vilco_amd import ops
from vilco_amd.dist import GradReducer
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
if os.environ.get('NOGC') == '1':
    import gc; gc.disable()
def run(seg, calls, reduce=True):
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    batch = bench.synth_batch(2, dev, seed=0)
    red = GradReducer(model)
    g = GraphedStep(model, None, eager_steps=2, reducer=red, segments=seg)
    snaps = {}
    for c in range(max(calls) + 1):
        g(batch)
        if c in calls:
            torch.cuda.synchronize()
            snaps[c] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    red.remove()
    return snaps
calls = tuple(int(x) for x in os.environ.get('CALLS', '2,3,6').split(','))
if os.environ.get('STREAM') == '1':
    _st = torch.cuda.Stream(); _ctx = torch.cuda.stream(_st); _ctx.__enter__()
a = run(False, calls); b = run(True, calls); b2 = run(True, calls) if os.environ.get('TWICE', '1') == '1' else b
for c in calls:
    bad = sorted(((float((a[c][k] - b[c][k]).abs().max() / (a[c][k].abs().max() + 1e-30)), k) for k in a[c] if not torch.equal(a[c][k], b[c][k])), reverse=True)
    rep = sum(int(not torch.equal(b[c][k], b2[c][k])) for k in b[c])
    big = [(e, k) for e, k in bad if e > 1e-3 and float(a[c][k].abs().max()) > 1e-10]
    print("call %d: %d differ (staged run-to-run: %d differ); > 1e-3 on non-noise tensors: %s" % (c, len(bad), rep, [(round(e, 4), k.replace('backbone.', '')) for e, k in big][:10]), flush=True)
    print("     next: %s" % [(("%.1e" % e), k.replace('backbone.', '')) for e, k in bad if (e, k) not in big][:6])
dist.destroy_process_group()
