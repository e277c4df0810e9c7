import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "split3")
M, N, K = [int(x) for x in (sys.argv[2:5] or (4608, 4096, 1024))]
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N)
torch.cuda.synchronize()
