"""Is the hd = 64 attention bound by the partial last round of workgroups?  (576 workgroups on 512 resident slots at
B = 2, H = 16, T = 2304.)  Sweep T so the workgroup count crosses multiples of 512; kernel times come from the rocprofv3 trace
of this script (tools/lab/attn_tail.sh groups them by grid size)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
B, H, hd = 2, 16, 64
for T in [int(x) for x in (sys.argv[1:] or "1024 1536 1920 2048 2176 2304 2560 3072 4096 4224".split())]:
    C = H * hd
    q, k, v = [torch.randn(B, T, C, device=dev) for _ in range(3)]
    lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
    for _ in range(4):
        o, lse = ops._flash_fwd(q, k, v, None, lens, H, 0.125, 0)
        do = torch.randn_like(o)
        ops._flash_bwd(q, k, v, None, lens, o, lse, do, H, 0.125, 0, False)
    torch.cuda.synchronize()
