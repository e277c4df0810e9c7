"""diagnostic: which operand packs of a config-P step still need their own amax launch (no producer-emitted partials)"""
import os, sys, traceback
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from vilco_amd import ops
import vilco_amd.modeling as vm

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
def step():
    model.zero_grad(set_to_none=True)
    model(batch, is_training=True)['final_loss'].backward()
step(); step()
cnt = Counter()
real = ops.pack
def pack(x, rows, cols, precision=None):
    tagged = ops._amax_of(x)[0] is not None
    fr = traceback.extract_stack(limit=6)
    site = " < ".join("%s:%d" % (f.name, f.lineno) for f in reversed(fr[:-1]) if 'ops.py' in f.filename or 'modeling' in f.filename)
    cnt[(tagged, int(rows), int(cols), site)] += 1
    return real(x, rows, cols, precision)
ops.pack = pack
step()
torch.cuda.synchronize()
for (tagged, r, c, site), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%-8s %3d x [%5d, %5d]  %s" % ("tagged" if tagged else "AMAX", n, r, c, site))
print("untagged packs:", sum(n for k, n in cnt.items() if not k[0]), "tagged:", sum(n for k, n in cnt.items() if k[0]))
