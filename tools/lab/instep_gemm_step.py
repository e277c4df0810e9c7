"""one eager P step whose GEMM launches are listed in order (M, N, K, tile, splits, precision, orientations) -> /tmp/instep_records.json;
run under rocprofv3 --pmc by tools/lab/instep_gap.sh: the LAST len(records) GEMM dispatches of the trace are that step's, in order"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
import vilco_amd.modeling as vm
from vilco_amd import _lib, ops
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0)
def step():
    for p in model.parameters():
        p.grad = None
    model(batch, is_training=True)['final_loss'].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
lib = _lib.load()
_lib.check(lib.vilco_gemm_profile_begin())
step()
torch.cuda.synchronize()
ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
_lib.check(lib.vilco_gemm_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
n = int(cnt.value)
desc = (ctypes.c_int64 * (10 * n))()
tms = (ctypes.c_double * n)()
lib.vilco_gemm_profile_records.restype = ctypes.c_int64
lib.vilco_gemm_profile_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
got = lib.vilco_gemm_profile_records(desc, tms, n)
recs = [[int(desc[i * 10 + j]) for j in range(10)] + [float(tms[i])] for i in range(min(n, got))]
json.dump(recs, open(os.environ.get("INSTEP_RECORDS", "/tmp/instep_records.json"), "w"))
