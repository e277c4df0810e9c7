"""Round 5 lab: gemm_gl_kernel against fp64 and against gemm_pp_kernel, one process (vilco_gemm_set_gl), kernel-only times with the
operands packed once.  Usage: python tools/lab/gl_check.py [quick]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
lib = _lib.load()
torch.manual_seed(0)

def t_of(fn, n=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def mk(form, M, N, K):
    if form == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); a_kc, b_kc, lda, ldb = 1, 1, K, K
        ref = lambda: A.double() @ B.double().t()
    elif form == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 1, 0, K, N
        ref = lambda: A.double() @ B.double()
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); a_kc, b_kc, lda, ldb = 0, 0, M, N
        ref = lambda: A.double().t() @ B.double()
    return A, B, a_kc, b_kc, lda, ldb, ref

# correctness: odd sizes, tails, split-K, both precisions
cases = [("NT", 200, 130, 96), ("NT", 192, 128, 64), ("NT", 777, 300, 1000), ("NN", 500, 260, 224), ("TN", 300, 200, 500),
         ("NT", 4608, 1024, 1024), ("NN", 4608, 1024, 1024), ("TN", 1024, 1024, 4608), ("NT", 154, 1024, 1024), ("NN", 154, 1024, 4096),
         ("TN", 1024, 1024, 154), ("NT", 2304, 1024, 1056), ("TN", 1024, 3072, 9082), ("NT", 64, 64, 32), ("TN", 100, 72, 40)]
bad = 0
for form, M, N, K in cases:
    A, B, a_kc, b_kc, lda, ldb, ref = mk(form, M, N, K)
    R = ref()
    scale = R.abs().max().item()
    for prec in ((None, 4) if form == "TN" else (None,)):
        out = {}
        for gl in (1, 0):
            _lib.check(lib.vilco_gemm_set_gl(gl))
            C = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, precision=prec)
            torch.cuda.synchronize()
            out[gl] = C
        e1 = (out[1].double() - R).abs().max().item() / scale
        e0 = (out[0].double() - R).abs().max().item() / scale
        d = (out[1] - out[0]).abs().max().item() / scale
        tol = 2e-3 if prec == 4 else 2e-6
        ok = e1 <= max(tol, 2 * e0) and e1 == e1
        bad += 0 if ok else 1
        print("%s %5d x %5d x %5d prec %s: gl err %.2e  pp err %.2e  gl-pp %.2e  %s" % (form, M, N, K, prec, e1, e0, d, "ok" if ok else "FAIL"), flush=True)
_lib.check(lib.vilco_gemm_set_gl(1))
print("correctness failures:", bad)
if len(sys.argv) > 1 and sys.argv[1] == "quick": sys.exit(1 if bad else 0)

shapes = [("NT", 4608, 1024, 1024), ("NN", 4608, 1024, 1024), ("NT", 4608, 4096, 1024), ("NN", 4608, 4096, 1024), ("NT", 4608, 1024, 4096),
          ("NN", 4608, 1024, 4096), ("NT", 9082, 1024, 3072), ("NT", 4608, 3072, 1024), ("NT", 2304, 1024, 1024), ("NT", 9216, 1024, 1024),
          ("TN", 1024, 4096, 4608), ("TN", 1024, 1024, 4608), ("TN", 4096, 1024, 4608), ("TN", 1024, 3072, 9082), ("NT", 154, 1024, 1024),
          ("NT", 1152, 1024, 1024), ("NT", 8192, 8192, 8192)]
for form, M, N, K in shapes:
    A, B, a_kc, b_kc, lda, ldb, ref = mk(form, M, N, K)
    C = torch.empty(M, N, device=dev)
    pa, pb = ops.pack(A, A.shape[0], A.shape[1]), ops.pack(B, B.shape[0], B.shape[1])
    prec = 4 if form == "TN" else None
    res = []
    for rnd in range(2):
        for gl in (0, 1):
            _lib.check(lib.vilco_gemm_set_gl(gl))
            for bm in ((0,) if rnd else (0, 128, 192)):
                _lib.check(lib.vilco_gemm_force(bm, 1 if bm else 0))
                us = t_of(lambda: ops.gemm(A, B, C, M, N, K, a_kc, b_kc, lda, ldb, N, a_planes=pa, b_planes=pb, precision=prec))
                res.append("%s bm%-3d %6.1f us %4.0f TF" % ("gl" if gl else "pp", bm, us, 2.0 * M * N * K / us / 1e6))
    _lib.check(lib.vilco_gemm_force(0, 0)); _lib.check(lib.vilco_gemm_set_gl(1))
    print("%s %5d x %4d x %4d | " % (form, M, N, K) + " | ".join(res), flush=True)
sys.exit(1 if bad else 0)
