"""Per-kernel timings at the P config shapes (T=2304, D=1024, B=2): GEMM TFLOP/s, LN / dwconv GB/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vilco_amd import ops


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def main():
    dev = torch.device("cuda:0")
    B, T, D = 2, 2304, 1024
    print(torch.cuda.get_device_name(0))
    for prec in ("split", "bf16"):
        ops.set_precision(prec)
        for (M, N, K) in [(B * T, D, D), (B * T, 4 * D, D), (B * T, D, 4 * D), (B * T, D, 3 * 2304), (8192, 8192, 8192)]:
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev)
            y = torch.empty(M, N, device=dev)
            t = timeit(lambda: ops.gemm(x, w, y, M, N, K, 1, 1, K, K, N))
            print("gemm NT %-6s M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s" % (prec, M, N, K, t * 1e3, 2 * M * N * K / t / 1e12))
            if M <= 8192 and N <= 4096:
                t = timeit(lambda: ops.gemm(y, w, x, M, K, N, 1, 0, N, K, K))
                print("gemm NN %-6s (dX)                 : %.3f ms  %.1f TFLOP/s" % (prec, t * 1e3, 2 * M * N * K / t / 1e12))
                dw = torch.empty_like(w)
                t = timeit(lambda: ops.gemm(y, x, dw, N, K, M, 0, 0, N, K, K))
                print("gemm TN %-6s (dW)                 : %.3f ms  %.1f TFLOP/s" % (prec, t * 1e3, 2 * M * N * K / t / 1e12))
    ops.set_precision("split")
    for C in (1024, 2304):
        x = torch.randn(B, T, C, device=dev)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        t = timeit(lambda: ops.layernorm(x, g, b))
        print("layernorm fwd C=%d: %.1f us  %.2f TB/s" % (C, t * 1e6, 8 * B * T * C / t / 1e12))
        w = torch.randn(C, 1, 3, device=dev)
        lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
        for s in (1, 2):
            t = timeit(lambda: ops.dwconv3(x, w, lens, s))
            print("dwconv3 fwd C=%d s=%d: %.1f us  %.2f TB/s" % (C, s, t * 1e6, 4 * B * T * C * (1 + 1 / s) / t / 1e12))
    H = 16
    q = torch.randn(B, T, D, device=dev); k = torch.randn(B, T, D, device=dev); v = torch.randn(B, T, D, device=dev)
    lens = torch.tensor([T, T - 17], dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.attention(q, k, v, lens, H))
    print("attention fwd (materialised) T=%d: %.3f ms  %.1f TFLOP/s" % (T, t * 1e3, 4 * B * T * T * D / t / 1e12))


if __name__ == "__main__":
    main()
