"""Micro-benchmarks of the two kernels BASELINE.json's north_star sets explicit targets on, at T = C = 2304:
  (a) the fused LN -> depthwise-conv x3 -> LN x3 q/k/v pre-projection (vilco_qkv_pre_fwd), HBM-bound: GB/s of ALGORITHMIC
      bytes (read x once, write q, k, v once: 4*C*T + 3*4*C*T/stride per clip) against the 8 TB/s HBM3E peak;
  (b) the cross-attention block of a stride-2 level (MaskedMHA: q-proj, k/v-proj of the 77 text tokens, attention,
      out-proj at T' = 1152, L = 77, D = 1024): algorithmic TFLOP/s against the 2.5 PFLOP/s dense fp16 peak, and the
      MFMA-issue fraction (x3 MFMAs per product in the parity precision).
Prints one JSON object; bench.py embeds it under "targets"."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters=20, warm=5, graph=True):
    """seconds per call.  `iters` calls are captured into ONE hipGraph and the graph is replayed, so the figure is device
    time: launched call by call from Python a 30-40 us kernel is hidden behind ~40 us of host work per call (allocations,
    ctypes) and the event pair measures the host.  Falls back to eager launches if the capture fails."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = None
    if graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
        except Exception:
            g = None
            torch.cuda.synchronize()
    reps = 3 if g is not None else 1
    e0.record()
    for _ in range(reps):
        if g is not None:
            g.replay()
        else:
            for _ in range(iters):
                fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e-3


def qkv_pre_target(dev, B, T=2304, C=2304, stride=1):
    from vilco_amd import ops
    x = torch.randn(B, T, C, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    mk = lambda *s: torch.randn(*s, device=dev)
    ln1 = (1 + 0.1 * mk(1, C, 1), 0.1 * mk(1, C, 1), 1e-5)
    convs = tuple(mk(C, 1, 3) for _ in range(3))
    norms = tuple((1 + 0.1 * mk(1, C, 1), 0.1 * mk(1, C, 1)) for _ in range(3)) + (1e-5,)
    with torch.no_grad():
        dt = timeit(lambda: ops.qkv_pre(x, ln1, convs, norms, lens, stride, False))

        def unfused():
            h = ops.layernorm(x, ln1[0], ln1[1], 1e-5)
            for w, (g, b) in zip(convs, norms[:3]):
                ops.layernorm(ops.dwconv3(h, w, lens, stride), g, b, 1e-5)
        dt_u = timeit(unfused)
    alg = 4.0 * C * T * B * (1 + 3.0 / stride)
    return {"shape": [B, T, C], "stride": stride, "algorithmic_bytes": alg, "us": dt * 1e6, "GBps": alg / dt / 1e9,
            "hbm_frac": alg / dt / 8e12, "unfused_7_launches_us": dt_u * 1e6, "unfused_GBps_same_algorithmic_bytes": alg / dt_u / 1e9}


def cross_attn_target(dev, B, T=1152, L=77, D=1024, H=16):
    import vilco_amd.modeling as vm
    from vilco_amd import ops
    torch.manual_seed(0)
    mha = vm.MaskedMHA(D, H).to(dev)
    x = torch.randn(B, T, D, device=dev, requires_grad=True)
    enc = torch.randn(B, L, D, device=dev, requires_grad=True)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    elens = torch.full((B,), L, dtype=torch.int32, device=dev)
    fwd_flop = B * (2 * 2.0 * T * D * D + 2 * 2.0 * L * D * D + 4.0 * T * L * D)      # q/out proj, k/v proj, QK^T + PV

    def fwd():
        with torch.no_grad():
            mha.forward_tm(x, lens, enc, elens)

    def fwdbwd():
        mha.zero_grad(set_to_none=True)
        x.grad = enc.grad = None
        y, _ = mha.forward_tm(x, lens, enc, elens)
        y.backward(y.detach())
    tf, tfb = timeit(fwd), timeit(fwdbwd, graph=False)
    peak = 2.5e15
    return {"shape": {"B": B, "T": T, "L": L, "D": D, "H": H}, "fwd_gflop": fwd_flop / 1e9, "fwd_us": tf * 1e6,
            "fwd_tflops": fwd_flop / tf / 1e12, "fwd_frac_of_2.5PF": fwd_flop / tf / peak,
            "fwd_mfma_issue_frac": 3 * fwd_flop / tf / peak, "fwd_bwd_us": tfb * 1e6,
            "fwd_bwd_tflops": 3 * fwd_flop / tfb / 1e12, "fwd_bwd_mfma_issue_frac": 9 * fwd_flop / tfb / peak}


def main():
    dev = torch.device("cuda:0")
    import vilco_amd._lib as L
    L.load()
    out = {"qkv_pre": [qkv_pre_target(dev, 2), qkv_pre_target(dev, 8), qkv_pre_target(dev, 2, C=1024), qkv_pre_target(dev, 8, stride=2)],
           "cross_attention": [cross_attn_target(dev, 2), cross_attn_target(dev, 8), cross_attn_target(dev, 32)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
