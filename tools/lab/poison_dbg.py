"""uninitialised-read hunt: fill the caching allocator's free memory with NaN patterns, run one P step, list tensors with NaN / Inf"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import vilco_amd.modeling as vm
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)
def poison(gb=40):
    ts = [torch.full((1 << 28,), float('nan'), device=dev) for _ in range(gb)]      # 1 GiB each
    del ts
    # half-precision NaNs in both halves of every word as well
    ts = [torch.full((1 << 29,), float('nan'), device=dev, dtype=torch.float16) for _ in range(gb)]
    del ts
poison()
losses = model(batch, is_training=True)
losses['final_loss'].backward()
torch.cuda.synchronize()
print({k: float(v) for k, v in losses.items()})
bad = [(k, int((~torch.isfinite(p.grad)).sum())) for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(len(bad), "tensors with non-finite gradients")
for k, n in bad[:30]:
    print("  ", k, n)
