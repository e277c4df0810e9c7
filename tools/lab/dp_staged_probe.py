"""The data-parallel P step on a ONE-rank RCCL group: backward replayed as one graph + exchange vs in stages with the buckets
launched between the stage graphs (vilco_amd/graph.py segments).  Same gradients?  What does staging cost / hide on one GPU?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
import vilco_amd.modeling as vm
from vilco_amd.dist import GradReducer
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
res = {}
for seg in (None, False, True):
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    for m in model.modules():                      # no dropout: the two runs must see the same arithmetic
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    batch = bench.synth_batch(2, dev, seed=0)
    red = GradReducer(model)
    g = GraphedStep(model, None, eager_steps=2, reducer=red, segments=bool(seg), enabled=seg is not None)
    for _ in range(6):
        out = g(batch)
    torch.cuda.synchronize()
    ent = ([e for e in g._graphs.values() if 'graph' in e] or [{}])[0]
    t0 = time.perf_counter()
    for _ in range(10):
        out = g(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    if seg:
        marks = ent['seg_marks']; names = [n for n, _ in model.named_parameters()]
        prev = [None] * len(names); last = {}
        for k, m in enumerate(marks):
            ch = [n for n, x, y in zip(names, prev, m) if y is not None and x != y]
            print("  stage %d touched %d gradients, e.g. %s" % (k, len(ch), ch[:3] + ch[-3:]))
            prev = m
        print("  bucket 0 holds", [n for n, p in model.named_parameters() if any(p is q for q in red.buckets[0]["params"])][:6])
    res[seg] = (ms, float(out['final_loss']), grads, len(ent.get('seg_graphs') or ()), ent.get('seg_upto'), len(red.buckets))
    print("segments=%s: %.2f ms/step, loss %.6f, stage graphs %d, buckets %d, buckets ready after each stage %s" % (seg, ms, res[seg][1], res[seg][3], res[seg][5], res[seg][4]), flush=True)
    red.remove(); del g, model, red
for x, y in ((None, False), (None, True), (False, True)):
    a, b = res[x][2], res[y][2]
    bad = sorted(((float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)), k) for k in a if not torch.equal(a[k], b[k])), reverse=True)
    print("%s vs %s: %d tensors, identical %d" % ({None: "eager", False: "one graph", True: "staged"}[x], {None: "eager", False: "one graph", True: "staged"}[y], len(a), len(a) - len(bad)))
    for e, k in bad[:8]:
        print("   %.3g  %s %s  max|a| %.3g max|b| %.3g" % (e, k, tuple(a[k].shape), float(a[k].abs().max()), float(b[k].abs().max())))
dist.destroy_process_group()
