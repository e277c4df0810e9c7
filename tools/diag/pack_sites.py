"""Which operand packs does one eager P step launch, from where, and does the source carry its producer's amax partials?
Wraps the library's vilco_pack / vilco_pack_many entry points (ctypes attributes) and records the python call site."""
import sys, os, collections, traceback, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd import _lib, ops
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
def step():
    model.zero_grad(set_to_none=True)
    l = model(batch, is_training=True)
    l['final_loss'].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
lib = _lib.load()
rec = collections.Counter()
def site():
    st = [f for f in traceback.extract_stack() if "vilco_amd" in f.filename and not f.filename.endswith("_lib.py")]
    st = [f for f in st if f.name not in ("pack", "pack_many", "pack_tap", "weight_planes", "_cached", "<lambda>", "gemm")]
    f = st[-1] if st else None
    g = st[-2] if len(st) > 1 else None
    return "%s:%d %s < %s" % (os.path.basename(f.filename), f.lineno, f.name, g.name if g else "") if f else "?"
orig_pack, orig_many = lib.vilco_pack, lib.vilco_pack_many
def w_pack(src, rows, cols, ld, prec, planes, nbytes, stream):
    rec[(site(), rows, cols, "untagged", 1)] += 1
    return orig_pack(src, rows, cols, ld, prec, planes, nbytes, stream)
def w_many(items, n, prec, stream):
    arr = [items._obj] if hasattr(items, "_obj") else items
    for i in range(n):
        it = arr[i]
        rec[(site(), it.rows, it.cols, ("tagged" if it.amax else "untagged") + (" seq" if it.seq_len else "") + (" nb%d" % it.nbatch if it.nbatch > 1 else ""), n)] += 1
    return orig_many(items, n, prec, stream)
class Wrap:
    def __init__(self, lib): self.__dict__['_l'] = lib
    def __getattr__(self, k):
        if k == "vilco_pack": return w_pack
        if k == "vilco_pack_many": return w_many
        return getattr(self._l, k)
real_load = _lib.load
wrapped = Wrap(lib)
_lib.load = lambda: wrapped
step()
torch.cuda.synchronize()
_lib.load = real_load
tot = 0
for (s, rows, cols, tag, n), c in sorted(rec.items(), key=lambda kv: (-kv[1] * kv[0][1] * kv[0][2])):
    print("%3d x [%5d, %5d] %-14s (launch of %d)  %s" % (c, rows, cols, tag, n, s))
    tot += c
print("pack items per step:", tot)
