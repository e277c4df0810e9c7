"""single-part weight-gradient products (ops.dw_precision = 4) against the three-MFMA products at config P, train mode
with the reference's dropout, over several mask realisations: worst tensors (max |fast - full| / max |full|)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from vilco_amd import ops
import vilco_amd.modeling as vm
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)

def run(seed, dwp):
    ops.dw_precision = dwp
    torch.manual_seed(seed)
    ops._drop_counter[0] = 1000 * seed
    from vilco_amd.modeling import blocks
    blocks.reset_drop_pool()
    model.zero_grad(set_to_none=True)
    model.loss_normalizer = 100.0
    model(batch, is_training=True)['final_loss'].backward()
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

worst = {}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    g1 = run(seed, 4)
    g2 = run(seed, None)
    errs = sorted((((g1[k] - g2[k]).abs().max() / g2[k].abs().max().clamp_min(1e-7)).item(), k) for k in g1)
    print("seed", seed, ["%.2e %s" % e for e in errs[-6:]])
    for e, k in errs:
        worst[k] = max(worst.get(k, 0.0), e)
top = sorted(((e, k) for k, e in worst.items()), reverse=True)
print("tensors over 5e-4:", [(round(e, 5), k) for e, k in top if e > 5e-4])
