"""per call: r_w_bias.grad of the staged asynchronous replay vs the one-graph replay -- when does it go stale, does the tensor move?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29536")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
import vilco_amd.modeling as vm
from vilco_amd.dist import GradReducer
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
def run(seg, n=16):
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    batch = bench.synth_batch(2, dev, seed=0)
    red = GradReducer(model)
    g = GraphedStep(model, None, eager_steps=2, reducer=red, segments=seg)
    p = dict(model.named_parameters())['backbone.xlnet.layer.0.rel_attn.r_w_bias']
    bi, view = red._slot.get(id(p), (None, None)) if red._slot else (None, None)
    rows = []
    for c in range(n):
        g(batch)
        torch.cuda.synchronize()
        if red._slot and bi is None:
            bi, view = red._slot[id(p)]
        rows.append((c, float(p.grad.abs().sum()), p.grad.data_ptr(), float(view.abs().sum()) if view is not None else -1, dict(g.stats), red._next))
    ent = [e for e in g._graphs.values() if 'graph' in e][0]
    print("segments=%s bucket of r_w_bias %s of %d; schedule %s" % (seg, bi, len(red.buckets), ent.get('seg_upto')))
    red.remove()
    return rows
a = run(False); b = run(True)
for x, y in zip(a, b):
    print("call %2d  one-graph |g|1 %.6e ptr %x  | staged |g|1 %.6e ptr %x view %.6e  stats %s next %s" % (x[0], x[1], x[2], y[1], y[2], y[3], y[4], y[5]))
dist.destroy_process_group()
