"""How long does the host take to ENQUEUE one P step (no waiting for the GPU)?  If this is close to the GPU step time the
step is launch-bound on the CPU."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
import vilco_amd.modeling as vm
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
def step():
    model.zero_grad(set_to_none=True)
    l = model(batch, is_training=True)
    l['final_loss'].backward()
for _ in range(5):
    step()
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.1f ms, then waited %.1f ms for the GPU (total %.1f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
