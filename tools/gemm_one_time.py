import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, ctypes
from vilco_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (M, N, K, form) in [(4608, 1024, 1024, "NT"), (4608, 4096, 1024, "NT"), (4608, 1024, 4096, "NT"), (4608, 1024, 3072, "NN")]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if form == "NT" else torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    b_kc, ldb = (1, K) if form == "NT" else (0, N)
    for _ in range(3): ops.gemm(A, B, C, M, N, K, 1, b_kc, K, ldb, N)
    lib.vilco_gemm_profile_begin()
    for _ in range(20): ops.gemm(A, B, C, M, N, K, 1, b_kc, K, ldb, N)
    ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.vilco_gemm_profile_end(ctypes.byref(ms), ctypes.byref(cnt))
    print("  %s M=%d N=%d K=%d kernel %.1f us (%.0f TF)" % (form, M, N, K, ms.value * 1e3 / cnt.value, 2.0 * M * N * K * cnt.value / ms.value / 1e9))
