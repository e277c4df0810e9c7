"""Round 6: few-row NT products (the pyramid levels at T' <= 288, the text tokens, cfg1's levels) on gemm_skinny_kernel against the
tiled kernels' plan (vilco_gemm_set_skinny 1 / 0), operands packed once, 20 dependent calls per hipGraph replay (no host in the loop), kernel + split-K finish."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vilco_amd import ops, _lib
ops.set_precision("f16x2")
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [(144, 1024, 1024), (154, 1024, 1024), (154, 1024, 768), (288, 1024, 1024), (576, 1024, 1024), (144, 4096, 1024), (288, 4096, 1024),
          (576, 4096, 1024), (144, 1024, 4096), (288, 1024, 4096), (576, 1024, 4096), (154, 4096, 1024), (154, 1024, 4096),
          (512, 512, 512), (256, 512, 512), (128, 512, 512), (64, 512, 512), (32, 512, 512), (16, 512, 512), (512, 2048, 512), (512, 512, 2048),
          (640, 1024, 1024), (1152, 1024, 1024)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); Cc = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    pa, pb = ops.pack(A, M, K), ops.pack(B, N, K)

    def run():
        ops.gemm(A, B, Cc, M, N, K, 1, 1, K, K, N, a_planes=pa, b_planes=pb, bias=bias, want_amax=True)
    t = {}
    for on in (0, 1, 0, 1):
        _lib.check(lib.vilco_gemm_set_skinny(on))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):          # (host-free: 20 dependent calls per replay)
                for _ in range(20):
                    run()
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
        t.setdefault(on, []).append(e0.elapsed_time(e1) / 200 * 1e3)
        del gr
    _lib.check(lib.vilco_gemm_set_skinny(1))
    print("NT %5d x %5d x %5d  tiled %5.1f us  few-row %5.1f us" % (M, N, K, min(t[0]), min(t[1])), flush=True)
