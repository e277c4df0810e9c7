"""cycle stamps of softnms_reg_kernel (lab build with -DVILCO_LAB_NMS): picks 100..131 of one class of 30 000 candidates"""
import os, sys, ctypes
os.environ["VILCO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvilco_lab.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from tools.nms_bench import candidates
from vilco_amd import _lib
from vilco_amd.utils.nms import nms_1d_cpu
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
segs, scores, cls = candidates(n, 1, 1)
dets = torch.zeros(n, 3, device=dev)
for _ in range(2):
    idx = nms_1d_cpu.softnms(segs.to(dev), scores.to(dev), dets, 0.1, 0.75, 0.01, 2)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (32 * 8))()
lib.vilco_lab_nms_read.argtypes = [ctypes.c_void_p]
print("rc", lib.vilco_lab_nms_read(buf), "kept", idx.numel())
st = np.array(buf, dtype=np.uint64).reshape(32, 8).astype(np.int64)
print("pick | argmax+reduce | ->B1+fold+owner | B2 | decay | B3 | compaction | total")
for t in range(4, 20):
    s = st[t]
    nxt = st[t + 1, 0]
    print(" %3d | %6d | %6d | %5d | %6d | %5d | %6d | %6d" % (100 + t, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], (s[6] - s[5]) if s[6] else 0, nxt - s[0]))
