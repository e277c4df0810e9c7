"""the main GEMM shapes of a P step launched back to back, operands packed once (tools/lab/instep_gap.sh)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops
ops.set_precision("f16x2")
dev = torch.device("cuda:0")
for form, M, N, K, prec in (("NT", 4608, 1024, 1024, None), ("NN", 4608, 1024, 1024, None), ("NT", 4608, 4096, 1024, None), ("NT", 4608, 1024, 4096, None),
                            ("NN", 4608, 4096, 1024, None), ("NN", 4608, 1024, 4096, None), ("TN", 1024, 1024, 4608, 4), ("TN", 1024, 4096, 4608, 4),
                            ("TN", 4096, 1024, 4608, 4), ("NT", 2304, 1024, 1024, None)):
    if form == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
    elif form == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
    C = torch.empty(M, N, device=dev)
    pa, pb = ops.pack(A, A.shape[0], A.shape[1]), ops.pack(B, B.shape[0], B.shape[1])
    for _ in range(12):
        ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=prec)
torch.cuda.synchronize()
