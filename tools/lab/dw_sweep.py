"""Round 6: the single-part weight-gradient shapes of a P step (TN: A = dY [K tok][M Cout], B = X [K tok][N Cin], precision 4) under
forced tile heights / split counts, on the kernel this process selects (VILCO_GEMM_GL_SINGLE=1: gemm_gl_kernel<.., SINGLE>; default:
gemm_pp_kernel K2).  Operands packed once; 20 calls per point; kernel + split-K finish."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [(1024, 3072, 9082), (1024, 1024, 4608), (1024, 4096, 4608), (4096, 1024, 4608), (3072, 1024, 4608), (1024, 6912, 4608),
          (1024, 1024, 2304), (1024, 1024, 9216)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
tag = "gl-single" if os.environ.get("VILCO_GEMM_GL_SINGLE") == "1" else "pp-K2"
for M, N, K in shapes:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    pa, pb = ops.pack(A, K, M), ops.pack(B, K, N)
    def run():
        ops.gemm(A, B, C, M, N, K, 0, 0, M, N, N, a_planes=pa, b_planes=pb, precision=4)
    res = []
    for bm in (0, 128, 192, 256):
        for ks in (0, 1, 2, 3, 4, 6, 8, 12):
            if bm == 0 and ks != 0:
                continue
            _lib.check(lib.vilco_gemm_force(bm, ks))
            try:
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run()
                e1.record(); torch.cuda.synchronize()
                res.append((e0.elapsed_time(e1) / 20 * 1e3, bm, ks))
            except RuntimeError as e:
                res.append((float('inf'), bm, ks))
    _lib.check(lib.vilco_gemm_force(0, 0))
    d = [r for r in res if r[1] == 0][0][0]
    res.sort()
    fl = 2.0 * M * N * K
    print("%-9s %5d x %5d x %5d  default %6.1f us (%4.0f TF) | best: %s" % (tag, M, N, K, d, fl / d / 1e6,
          "  ".join("BM%d ks%d %.1f (%.0f TF)" % (bm, ks, t, fl / t / 1e6) for t, bm, ks in res[:5])), flush=True)
