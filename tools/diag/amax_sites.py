"""Which tensors of an eager P step still get an amax launch of their own (VILCO_AMAX_TRACE prints one line per operand from
pack.h: launch_amax)?  Aggregates the library's stderr lines of one step."""
import os, sys, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AMAX_CHILD"):
    sys.path.insert(0, ROOT)
    import torch, bench
    import vilco_amd.modeling as vm
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    batch = bench.synth_batch(2, dev, seed=0)
    def step():
        model.zero_grad(set_to_none=True)
        model(batch, is_training=True)['final_loss'].backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print("==STEP==", file=sys.stderr, flush=True)
    step()
    torch.cuda.synchronize()
    sys.exit(0)
env = dict(os.environ, AMAX_CHILD="1", VILCO_AMAX_TRACE="1")
out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
lines = out.split("==STEP==")[-1].splitlines()
cnt = collections.Counter(l.strip() for l in lines if l.startswith("amax_launch"))
for k, v in cnt.most_common():
    print("%3d x %s" % (v, k))
print("operands with an amax pass of their own per step:", sum(cnt.values()))
