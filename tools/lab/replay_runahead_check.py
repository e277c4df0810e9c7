"""Does a replayed step read the right inputs / return the right losses when the host runs many steps ahead?  60 steps over
rotating batches, once without any synchronisation and once with one after every step: same losses and same final gradients?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
def run(sync):
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    batches = [bench.synth_batch(2, dev, seed=s) for s in range(5)]
    g = GraphedStep(model, None, eager_steps=2)
    for i in range(4):
        g(batches[i % 5]); torch.cuda.synchronize()
    losses = []
    for i in range(60):
        out = g(batches[(i * 3 + 1) % 5])
        losses.append(out['final_loss'])
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return torch.stack(losses).cpu(), grads
la, ga = run(False); lb, gb = run(True)
print("losses equal:", bool(torch.equal(la, lb)), " max |diff|", float((la - lb).abs().max()), " distinct loss values", len(set(la.tolist())))
print("final gradients: %d tensors, identical %d" % (len(ga), sum(int(torch.equal(ga[k], gb[k])) for k in ga)))
for k in ga:
    if not torch.equal(ga[k], gb[k]):
        print("   differs:", k, tuple(ga[k].shape), "max|a| %.3g max|diff| %.3g" % (float(ga[k].abs().max()), float((ga[k] - gb[k]).abs().max())))
