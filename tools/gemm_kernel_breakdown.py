"""Per-shape MFMA-kernel-only time of one P-config training step (vilco_gemm_profile_* around every call)."""
import sys, os, collections, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import vilco_amd, vilco_amd.modeling as vm
from vilco_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
cfg = bench.p_config()
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
batch = bench.synth_batch(2, dev)

def step():
    model.zero_grad(set_to_none=True)
    l = model(batch, is_training=True)
    l['final_loss'].backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
real = ops.gemm
def timed(A, B, Cc, M, N, K, a_kc, b_kc, *a, **k):
    lib.vilco_gemm_profile_begin()
    real(A, B, Cc, M, N, K, a_kc, b_kc, *a, **k)
    ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.vilco_gemm_profile_end(ctypes.byref(ms), ctypes.byref(cnt))
    bt = k.get("batch", (1, 1))
    key = (M, N, K, bt[0] * bt[1], "NT" if (a_kc and b_kc) else ("NN" if a_kc else "TN"), k.get("tap", 0))
    agg[key][0] += 1; agg[key][1] += ms.value
ops.gemm = timed
step(); torch.cuda.synchronize()
ops.gemm = real
tot = sum(v[1] for v in agg.values())
print("kernel-only gemm %.2f ms in %d launches" % (tot, sum(v[0] for v in agg.values())))
for (M, N, K, bt, form, tap), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:36]:
    fl = 2.0 * M * N * K * bt * cnt
    tiles256 = ((M + 255) // 256) * ((N + 127) // 128) * bt
    print("%-3s tap%d M=%-6d N=%-6d K=%-6d batch=%-3d x%-3d %7.3f ms %6.1f us/call %6.1f TF  tiles256=%d" % (form, tap, M, N, K, bt, cnt, ms, ms / cnt * 1e3, fl / ms / 1e9, tiles256))
