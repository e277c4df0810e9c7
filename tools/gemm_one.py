import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f16x2")
M, N, K = [int(x) for x in (sys.argv[2:5] or (4608, 4096, 1024))]
form = sys.argv[5] if len(sys.argv) > 5 else "NT"
PREC = int(sys.argv[6]) if len(sys.argv) > 6 else None          # 4: single-part products
dev = torch.device("cuda:0")
if form == "NT":
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
elif form == "NN":
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
else:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
C = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N)
torch.cuda.synchronize()
# timing with the operands packed once (the GEMM kernel + split-K finish only), 30 calls
pa = ops.pack(A, A.shape[0], A.shape[1])
pb = ops.pack(B, B.shape[0], B.shape[1])
for _ in range(3):
    ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=PREC)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=PREC)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 30 * 1e3
print("%d x %d x %d %s: %.1f us  %.0f TFLOP/s algorithmic" % (M, N, K, form, us, 2.0 * M * N * K / us / 1e6))
