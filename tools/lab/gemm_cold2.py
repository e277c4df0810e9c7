"""which operand's cache state costs the cold GEMM its 16 %?  4608 x 1024 x 1024, 35 buffer sets cycled; variants:
all cold / A planes packed immediately before the product (as in the step) / one resident B / one resident C / touch B first"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops
ops.set_precision("f16x2")
ops._pack_cache = False
dev = torch.device("cuda:0")
M, N, K = 4608, 1024, 1024
nsets = 35
sets = []
for _ in range(nsets):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    sets.append([A, B, C, ops.pack(A, M, K), ops.pack(B, N, K)])
def timed(fn, n=70):
    fn(nsets + 3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(n); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def all_cold(n):
    for i in range(n):
        A, B, C, pa, pb = sets[i % nsets]
        ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb)
def pack_only(n):
    for i in range(n):
        A = sets[i % nsets][0]
        ops.pack(A, M, K)
def pack_then_gemm(n):
    for i in range(n):
        A, B, C, pa, pb = sets[i % nsets]
        pa = ops.pack(A, M, K)
        ops.gemm(A, B, C, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb)
def resident_b(n):
    B0, pb0 = sets[0][1], sets[0][4]
    for i in range(n):
        A, B, C, pa, pb = sets[i % nsets]
        ops.gemm(A, B0, C, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb0)
def resident_c(n):
    C0 = sets[0][2]
    for i in range(n):
        A, B, C, pa, pb = sets[i % nsets]
        ops.gemm(A, B, C0, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb)
def resident_a(n):
    A0, pa0 = sets[0][0], sets[0][3]
    for i in range(n):
        A, B, C, pa, pb = sets[i % nsets]
        ops.gemm(A0, B, C, M, N, K, 1, 1, K, K, N, a_planes=pa0, b_planes=pb)
t_cold = timed(all_cold)
t_pack = timed(pack_only)
print("all cold                      %.1f us" % t_cold)
print("pack A then product           %.1f us  (pack alone %.1f -> product %.1f)" % (timed(pack_then_gemm), t_pack, timed(pack_then_gemm) - t_pack))
print("resident A planes, rest cold  %.1f us" % timed(resident_a))
print("resident B planes, rest cold  %.1f us" % timed(resident_b))
print("resident C, rest cold         %.1f us" % timed(resident_c))
