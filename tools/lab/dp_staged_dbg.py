"""staged vs one-graph replay: which switch makes the XLNet bias gradients differ?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
import vilco_amd.modeling as vm
from vilco_amd import ops
from vilco_amd.dist import GradReducer
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
def run(seg, **sw):
    for k, v in sw.items(): setattr(ops, k, v)
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    batch = bench.synth_batch(2, dev, seed=0)
    red = GradReducer(model)
    g = GraphedStep(model, None, eager_steps=2, reducer=red, segments=seg)
    for _ in range(5): g(batch)
    torch.cuda.synchronize()
    out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None and ('r_r_bias' in n or 'r_w_bias' in n or n.endswith('rel_attn.q') or n.endswith('rel_attn.r'))}
    red.remove()
    return out
base = run(False)
for name, sw in (("default", {}), ("fp32 dS + pack", dict(xl_ds_planes=False)), ("band GEMM scores", dict(xl_scores_kernel=False)), ("no skip fold", dict(fold_skip_grads=False)),
                 ("no forks", dict(_FORKS=set()))):
    got = run(True, **sw)
    ref = run(False, **sw) if sw else base
    print(name, {k.split('.')[-1]: "%.3g (max %.3g / %.3g)" % (float((got[k] - ref[k]).abs().max()), float(got[k].abs().max()), float(ref[k].abs().max())) for k in ref}, flush=True)
    for k, v in dict(xl_ds_planes=True, xl_scores_kernel=True, fold_skip_grads=True).items(): setattr(ops, k, v)
dist.destroy_process_group()
