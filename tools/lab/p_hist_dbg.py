"""the full-size parity comparison after the history that makes it fail in-process: what do the masks / worst tensors look like?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
if len(sys.argv) > 1 and sys.argv[1] == "hist":
    pytest.main([os.path.join(ROOT, "tests/test_dist_gpu.py"), os.path.join(ROOT, "tests/test_episode.py"), "-q", "-m", "gpu", "-k",
                 "reducer_paths or (run_episodes_end_to_end and True)", "-p", "no:cacheprovider"])
import bench
import vilco_amd.modeling as vm
from oracle import mq_oracle
from vilco_amd import ops
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)
ops.dropout_log = []
losses = model(batch, is_training=True)
losses['final_loss'].backward()
log = list(ops.dropout_log)
ops.dropout_log = None
torch.cuda.synchronize()
print("droppath factors:", [tuple(round(float(x), 2) for x in e[1].cpu()) for e in log if e[0] == "droppath"])
got = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
p = {k: (v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())) for k, v in model.state_dict().items()}
vl = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.items()} for d in batch]
mq_oracle.DROP = mq_oracle.DropReplay(log, lambda pr, seed, shape, site: ops.dropout_mask(pr, seed, shape, dev, site).cpu())
want, _ = mq_oracle.forward_losses(p, cfg, vl)
want['final_loss'].backward()
mq_oracle.DROP = None
rows = []
for k, g in got.items():
    if p[k].grad is not None and not k.endswith(('key_norm.bias', '.key.bias')):
        w = p[k].grad
        rows.append((float((g - w).abs().max() / w.abs().max().clamp_min(1e-7)), k, float(w.abs().max()), float(g.abs().max())))
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.4f %-46s max|want| %.3e max|got| %.3e" % r)
