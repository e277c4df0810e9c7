"""the same product with the operands resident (one buffer set, back to back) and cold (NSETS buffer sets cycled, far more
bytes than the 256 MB Infinity Cache between two uses of a set): how much of the step's GEMM slow-down is cache state?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops
ops.set_precision("f16x2")
dev = torch.device("cuda:0")
for (M, N, K) in [(4608, 1024, 1024), (4608, 4096, 1024), (4608, 1024, 4096), (9082, 1024, 3072)]:
    per = (M * K + N * K) * 4 + M * N * 4
    for nsets in (1, max(2, int(1.5e9 // per))):
        sets = []
        for _ in range(nsets):
            A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
            sets.append((A, B, C, ops.pack(A, M, K), ops.pack(B, N, K)))
        def run(n):
            for i in range(n):
                A, B, C, pa, pb = sets[i % nsets]
                ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb)
        run(nsets + 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(30, 2 * nsets)
        e0.record(); run(n); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print("%d x %d x %d  %3d buffer set(s): %.1f us  %.0f TFLOP/s" % (M, N, K, nsets, us, 2.0 * M * N * K / us / 1e6), flush=True)
        del sets
