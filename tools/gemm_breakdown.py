"""Per-shape time breakdown of one P-config training step (GEMM launches + everything else)."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import vilco_amd, vilco_amd.modeling as vm
from vilco_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "split3"
ops.set_precision(prec)
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)

def step():
    model.zero_grad(set_to_none=True)
    l = model(batch, is_training=True)
    l['final_loss'].backward()

step(); torch.cuda.synchronize()
recs = []
real = ops.gemm
def timed(A, B, Cc, M, N, K, a_kc, b_kc, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); real(A, B, Cc, M, N, K, a_kc, b_kc, *a, **k); e1.record()
    bt = k.get("batch", (1, 1))
    recs.append((e0, e1, (M, N, K, bt[0] * bt[1], "NT" if (a_kc and b_kc) else ("NN" if a_kc else "TN"), k.get("tap", 0))))
ops.gemm = timed
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); step(); t1.record(); torch.cuda.synchronize()
ops.gemm = real
agg = collections.defaultdict(lambda: [0, 0.0])
for a, b, key in recs:
    agg[key][0] += 1; agg[key][1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print("precision", prec, "step %.1f ms, gemm %.1f ms in %d launches" % (t0.elapsed_time(t1), tot, len(recs)))
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (M, N, K, bt, form, tap), (cnt, ms) in rows[:40]:
    fl = 2.0 * M * N * K * bt * cnt
    print("%-3s tap%d M=%-6d N=%-6d K=%-6d batch=%-3d x%-3d %8.2f ms  %6.1f TF  %4.1f%%" % (form, tap, M, N, K, bt, cnt, ms, fl / ms / 1e9, 100 * ms / tot))
