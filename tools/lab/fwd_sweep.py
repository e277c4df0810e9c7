"""Round 6: the forward / dX shapes of a P step whose 192-row tiles leave a quarter of the chip idle (4608 x 1024 x K: 192 tiles on 256
CUs) and the sub-round shapes of the branch levels, under forced tile heights / split counts (vilco_gemm_force), precision 3, operands
packed once; 20 calls per point back to back; kernel + split-K finish."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [(4608, 1024, 1024), (4608, 1024, 4096), (4608, 1024, 3072), (4608, 1024, 2048), (9216, 1024, 1024), (2304, 1024, 1024),
          (2304, 1024, 4096), (1152, 1024, 1024), (1152, 1024, 4096), (1152, 4096, 1024), (576, 1024, 1024), (576, 1024, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "2":      # second set: the other sub-round / between-rounds shapes of the step
    shapes = [(4608, 2048, 1024), (2304, 4096, 1024), (2304, 2048, 1024), (1152, 2048, 1024), (576, 4096, 1024), (288, 4096, 1024),
              (288, 1024, 4096), (288, 1024, 1024), (144, 1024, 1024), (154, 1024, 1024), (154, 4096, 1024), (154, 1024, 4096),
              (154, 1024, 768), (9082, 1024, 1024), (4541, 1024, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "3":      # profiling run: the default plan only, one tiny shape
    shapes = [(288, 1024, 1024)]
for M, N, K in shapes:
    for tb, name in ((1, "NT"), (0, "NN")):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
        Cc = torch.empty(M, N, device=dev)
        pa = ops.pack(A, M, K)
        pb = ops.pack(B, N, K) if tb else ops.pack(B, K, N)

        def run():
            ops.gemm(A, B, Cc, M, N, K, 1, tb, K, K if tb else N, N, a_planes=pa, b_planes=pb)
        res = []
        for bm in ((0,) if (len(sys.argv) > 1 and sys.argv[1] == "3") else (0, 128, 192)):
            for ks in (0, 1, 2, 3, 4, 6, 8, 12, 16):
                if bm == 0 and ks != 0:
                    continue
                _lib.check(lib.vilco_gemm_force(bm, ks))
                try:
                    for _ in range(3):
                        run()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        run()
                    e1.record(); torch.cuda.synchronize()
                    res.append((e0.elapsed_time(e1) / 20 * 1e3, bm, ks))
                except RuntimeError:
                    res.append((float('inf'), bm, ks))
        _lib.check(lib.vilco_gemm_force(0, 0))
        d = [r for r in res if r[1] == 0][0][0]
        res.sort()
        fl = 2.0 * M * N * K
        print("%s %5d x %5d x %5d  default %6.1f us (%4.0f TF) | best: %s" % (name, M, N, K, d, fl / d / 1e6,
              "  ".join("BM%d ks%d %.1f" % (bm, ks, t) for t, bm, ks in res[:4])), flush=True)
