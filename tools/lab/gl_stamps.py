"""Per-phase cycle stamps of gemm_gl_kernel (lab build: LABFLAGS=... tools/lab/build_lab.sh; block 0, waves 0 and 4)."""
import os, sys, ctypes
os.environ["VILCO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvilco_lab.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, numpy as np
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
M, N, K = [int(x) for x in (sys.argv[1:4] or (4608, 1024, 4096))]
form = sys.argv[4] if len(sys.argv) > 4 else "NT"
dev = torch.device("cuda:0")
if form == "NT":
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
elif form == "NN":
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
else:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
C = torch.empty(M, N, device=dev)
prec = 4 if form == "TN" else None
pa, pb = ops.pack(A, A.shape[0], A.shape[1]), ops.pack(B, B.shape[0], B.shape[1])
epi = set(x for x in os.environ.get("EPI", "").split(",") if x)          # EPI=bias,res,gelu,amax: epilogue features of the timed call
kw = {}
if "bias" in epi: kw["bias"] = torch.randn(N, device=dev)
if "res" in epi: kw["residual"] = torch.randn(M, N, device=dev)
if "gelu" in epi: kw.update(act=2, preact=torch.empty(M, N, device=dev))
if "amax" in epi: kw["want_amax"] = True
for _ in range(5):
    ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=prec, **kw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 8))()
lib = _lib.load()
lib.vilco_lab_read.argtypes = [ctypes.c_void_p]
print("rc", lib.vilco_lab_read(buf))
st = np.array(buf, dtype=np.uint64).reshape(2, 64, 8).astype(np.int64)
print("early: t | reads(+drain) | DMA issue | ->X | MFMA | vmcnt wait | ->Y | total")
for t in range(2, 10):
    s = st[0, t]; r = s[1] if s[1] else s[0]
    print("  %2d | %5d | %5d | %5d | %5d | %5d | %5d | %6d" % (t, r - s[0], s[2] - r, s[4] - s[2], s[5] - s[4], s[6] - s[5], s[7] - s[6], st[0, t + 1, 0] - s[0]))
print("late:  t | MFMA | vmcnt wait | ->X | reads(+drain) | DMA issue | ->Y | total")
for t in range(2, 10):
    s = st[1, t]; r = s[4] if s[4] else s[3]
    print("  %2d | %5d | %5d | %5d | %5d | %5d | %5d | %6d" % (t, s[1] - s[0], s[2] - s[1], s[3] - s[2], r - s[3], s[5] - r, s[7] - s[5], st[1, t + 1, 0] - s[0]))
x = st[:, 63, :]
for grp in range(2):
    print("group", grp, "entry->first DMAs issued %d | ->chunk 0 landed + barrier %d | loop %d | epilogue %d cycles" %
          (x[grp, 4] - x[grp, 0], x[grp, 1] - x[grp, 4], x[grp, 2] - x[grp, 1], x[grp, 3] - x[grp, 2]))
    if x[grp, 5]:
        print("   prologue: entry -> tile + descriptors %d | -> DMA piece addresses %d | -> fragment addresses + acc %d | -> K range + first issue %d" %
              (x[grp, 5] - x[grp, 0], x[grp, 6] - x[grp, 5], x[grp, 7] - x[grp, 6], x[grp, 4] - x[grp, 7]))
