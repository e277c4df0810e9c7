"""What the host-side part of a replayed step costs on the device timeline: GraphedStep.__call__ (prepare -> copies into
the static inputs -> replay -> loss clone) vs the bare graph replay, same box, interleaved."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
gs = GraphedStep(model, None, eager_steps=2)
for _ in range(40):
    gs(batch)
torch.cuda.synchronize()
ent = [e for e in gs._graphs.values() if 'graph' in e][0]
def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
inp = model.prepare(batch, True, gt_pad=gs.gt_pad)
def full(): gs(batch)
def bare(): ent['graph'].replay()
def noprep(): gs._replay(ent, inp)
for r in range(4):
    print("full %.3f ms | prepared once + copies + replay %.3f | bare replay %.3f" % (timed(full), timed(noprep), timed(bare)), flush=True)
t0 = time.perf_counter()
for _ in range(20): model.prepare(batch, True, gt_pad=gs.gt_pad)
print("prepare host ms", (time.perf_counter() - t0) / 20 * 1e3)
