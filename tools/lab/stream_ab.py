"""The replayed P step with the DEFAULT (null) stream as the current stream vs a created stream (round 5: on the null stream the
staged data-parallel replay went wrong under host run-ahead; does the plain step care?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd.graph import GraphedStep
dev = torch.device("cuda:0")
def run(side):
    ctx = torch.cuda.stream(torch.cuda.Stream()) if side else None
    if ctx: ctx.__enter__()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    batch = bench.synth_batch(2, dev, seed=0)
    g = GraphedStep(model, None, eager_steps=2)
    for _ in range(40): g(batch)
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(20): g(batch)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    if ctx: ctx.__exit__(None, None, None)
    return ts
for side in (False, True, False, True):
    print("created stream" if side else "default stream", ["%.3f" % t for t in run(side)], flush=True)
