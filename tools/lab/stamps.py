"""Prints the per-phase cycle stamps of the ping-pong GEMM (lab build: tools/lab/build_lab.sh)."""
import os, sys, ctypes
os.environ["VILCO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvilco_lab.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f16x2")
M, N, K = [int(x) for x in (sys.argv[2:5] or (4608, 1024, 1024))]
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(3):
    ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 8))()
lib = _lib.load()
lib.vilco_lab_read.argtypes = [ctypes.c_void_p]
print("rc", lib.vilco_lab_read(buf))
import numpy as np
st = np.array(buf, dtype=np.uint64).reshape(2, 64, 8).astype(np.int64)
base = st[0, 0, 0]
nk = min(64, (K + 31) // 32)
for grp in range(2):
    print("group", grp, "t | start | %s | %s | barrier wait | step total" % (("MEM", "MFMA") if grp == 0 else ("MFMA(t-1)", "MEM")))
    for t in range(min(nk, 10)):
        s = st[grp, t]
        nxt = st[grp, t + 1, 0] if t + 1 < nk else s[6]
        print("  %2d | %8d | %6d | %6d | %6d | %6d" % (t, s[0] - base, s[3] - s[0], s[5] - s[3], s[6] - s[5], nxt - s[0]))
x = st[:, 63, :4]
for grp in range(2):
    print("group", grp, "entry->prologue done %d | loop %d | epilogue %d cycles" % (x[grp,1]-x[grp,0], x[grp,2]-x[grp,1], x[grp,3]-x[grp,2]))
if st[0, 2, 1] != 0:
    for grp in range(2):
        print("group", grp, "fine MEM split: vmcnt-wait+ds_write | gload issue | ds_read+wait")
        for t in range(2, 8):
            s = st[grp, t]; m0 = s[0] if grp == 0 else s[3]; m3 = s[3] if grp == 0 else s[5]
            print("  %2d | %6d | %6d | %6d" % (t, s[1] - m0, s[2] - s[1], m3 - s[2]))
