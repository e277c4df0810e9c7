"""Per-phase cycle stamps of the XLNet variant of the hd = 64 forward attention kernel (lab build)."""
import os, sys, ctypes
os.environ["VILCO_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvilco_lab.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
B, T, H = 2, 2304, 16
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dev = torch.device("cuda:0")
q, k, v = [torch.randn(B, T, H * 64, device=dev) for _ in range(3)]
bd = torch.randn(B, H, T, 2 * T, device=dev)
lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
for _ in range(3):
    ops._flash_fwd(q, k, v, bd, lens, H, 0.125, ops.MASK_XLNET_REL, (drop, 1234))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 8))()
lib = _lib.load()
lib.vilco_lab_attn_read.argtypes = [ctypes.c_void_p]
print("rc", lib.vilco_lab_attn_read(buf))
st = np.array(buf, dtype=np.uint64).reshape(64, 8).astype(np.int64)
print(" t | barrier+lstore | gload+bias+S | softmax | PV | total")
for t in range(2, 10):
    s = st[t]
    print("%2d | %6d | %6d | %6d | %6d | %6d" % (t, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[5] - s[3], st[t + 1, 0] - s[0]))
