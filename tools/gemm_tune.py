"""Times the main P-config GEMM shapes under forced (BM, split-K) choices (vilco_gemm_force)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vilco_amd import ops, _lib

def timeit(fn, n=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

dev = torch.device("cuda:0")
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "split3")
shapes = [("NT", 4608, 1024, 1024), ("NN", 4608, 1024, 1024), ("TN", 1024, 1024, 4608), ("NT", 4608, 4096, 1024),
          ("NT", 4608, 1024, 4096), ("TN", 4096, 1024, 4608), ("NT", 2304, 1024, 1024), ("NT", 1152, 1024, 1024),
          ("NT", 576, 1024, 1024), ("NT", 154, 1024, 1024), ("NT", 4608, 1024, 3072), ("NT", 288, 1024, 3072),
          ("NT", 144, 1024, 3072), ("TN", 1024, 3072, 1152), ("TN", 1024, 1024, 154), ("TN", 1024, 3072, 4608),
          ("NN", 4608, 4096, 1024), ("NT", 4608, 2304, 2304), ("TN", 1024, 1024, 288)]
for form, M, N, K in shapes:
    if form == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
    elif form == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
    C = torch.empty(M, N, device=dev)
    res = []
    for bm in (128, 256):
        for ks in (0, 1, 2, 3, 4, 6, 8):
            _lib.check(_lib.load().vilco_gemm_force(bm, ks))
            t = timeit(lambda: ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N))
            res.append((t, bm, ks))
    _lib.check(_lib.load().vilco_gemm_force(0, 0))
    t0 = timeit(lambda: ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N))
    res.sort()
    fl = 2.0 * M * N * K
    if os.environ.get("TUNE_FULL"):
        tab = {(bm, ks): t for t, bm, ks in res}
        print("%s M=%d N=%d K=%d default %.3f | " % (form, M, N, K, t0) + " ".join(
            "%d/%d:%.3f" % (bm, ks, tab[(bm, ks)]) for bm in (128, 256) for ks in (1, 2, 3, 4, 6, 8)))
    else:
        print("%s M=%d N=%d K=%d default %.3f ms (%.0f TF) | best: %s" % (form, M, N, K, t0, fl / t0 / 1e9,
              "  ".join("BM%d ks%d %.3f (%.0f TF)" % (bm, ks, t, fl / t / 1e9) for t, bm, ks in res[:4])))
