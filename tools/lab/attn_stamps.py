"""Per-phase cycle stamps of the hd = 64 forward attention kernel (lab build: tools/lab/build_lab.sh)."""
import os, sys, ctypes
os.environ["VILCO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvilco_lab.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
B, T, H = [int(x) for x in (sys.argv[1:4] or (2, 2304, 16))]
dev = torch.device("cuda:0")
q, k, v = [torch.randn(B, T, H * 64, device=dev) for _ in range(3)]
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(3):
    ops._flash_fwd(q, k, v, None, lens, H, 0.125, 0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 8))()
lib = _lib.load()
lib.vilco_lab_attn_read.argtypes = [ctypes.c_void_p]
print("rc", lib.vilco_lab_attn_read(buf))
st = np.array(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
print(" t | barrier1 | lstore+barrier2 | gload+S | softmax | PV | total")
for t in range(min(36, T // 64)):
    s = st[t]
    nxt = st[t + 1, 0] if t + 1 < T // 64 else s[5]
    print("%2d | %6d | %6d | %6d | %6d | %6d | %6d" % (t, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], nxt - s[0]))
