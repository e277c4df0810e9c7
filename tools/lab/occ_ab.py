"""Kernel-only time (operands packed once) of the step's many-tile GEMM shapes under forced tile heights, for one library
build (VILCO_HIP_LIB selects it).  Used for the two-workgroups-per-CU experiment: a -DVILCO_GEMM_WPE=4 build holds the 128-row
kernel at <= 128 VGPRs, so two of its 64 KB workgroups share a CU."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
lib = _lib.load()
shapes = [("NT", 4608, 4096, 1024), ("NN", 4608, 4096, 1024), ("NT", 4608, 3072, 1024), ("NT", 4608, 2048, 1024), ("NT", 9216, 1024, 1024),
          ("NT", 9082, 1024, 3072), ("NT", 4608, 1024, 1024), ("NN", 4608, 1024, 1024), ("NT", 4608, 1024, 4096), ("NT", 2304, 4096, 1024),
          ("TN", 1024, 4096, 4608), ("TN", 1024, 1024, 4608)]
bms = [int(x) for x in os.environ.get("BMS", "0,128,192,256").split(",")]
def t_of(fn, n=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for form, M, N, K in shapes:
    if form == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
    elif form == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
    C = torch.empty(M, N, device=dev)
    pa, pb = ops.pack(A, A.shape[0], A.shape[1]), ops.pack(B, B.shape[0], B.shape[1])
    prec = 4 if form == "TN" else None
    out = []
    for bm in bms:
        _lib.check(lib.vilco_gemm_force(bm, 1 if bm else 0))
        us = t_of(lambda: ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=prec))
        out.append("BM%-3d %7.1f us %4.0f TF" % (bm, us, 2.0 * M * N * K / us / 1e6))
    _lib.check(lib.vilco_gemm_force(0, 0))
    print("%s %5d x %4d x %4d | " % (form, M, N, K) + " | ".join(out), flush=True)
