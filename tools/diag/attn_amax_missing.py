"""Which operands of the attention calls of an eager P step come WITHOUT their producer's max|x| partials (ops._attn_amax_in)?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
log = collections.Counter()
on = [False]
orig = ops._attn_amax_in


def spy(q, k, v, do=None):
    if on[0]:
        for name, t in (("q", q), ("k", k), ("v", v), ("dout", do)):
            if t is not None and ops._amax_of(t)[0] is None:
                log[("bwd" if do is not None else "fwd", name, tuple(t.shape), tuple(k.shape))] += 1
    return orig(q, k, v, do)


ops._attn_amax_in = spy


def step():
    model.zero_grad(set_to_none=True)
    model(batch, is_training=True)['final_loss'].backward()


for _ in range(3):
    step()
on[0] = True
step()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print("%2d x %s %-4s shape %s (k %s)" % (v, k[0], k[1], k[2], k[3]))
