"""Per-shape table of the MFMA GEMM launches of one config-P training step (HIP events around each launch, recorded by
vilco_gemm_profile_*): which shapes the 13 ms go to and at what rate.  python tools/gemm_shapes.py [steps]"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from vilco_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    import vilco_amd.modeling as vm
    model = vm.make_meta_arch('LocPointTransformer', **dict(bench.p_config(), xlnet_config=bench.P_XLNET)).to(dev).train()
    batch = bench.synth_batch(2, dev, seed=0)

    def step():
        model.zero_grad(set_to_none=True)
        model(batch, is_training=True)['final_loss'].backward()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    lib = _lib.load()
    _lib.check(lib.vilco_gemm_profile_begin())
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    _lib.check(lib.vilco_gemm_profile_end(ctypes.byref(ms), ctypes.byref(cnt)))
    N = int(cnt.value)
    desc = (ctypes.c_int64 * (N * 10))()
    tms = (ctypes.c_double * N)()
    lib.vilco_gemm_profile_records(desc, tms, N)
    agg = defaultdict(lambda: [0, 0.0])
    for i in range(N):
        key = tuple(desc[i * 10:(i + 1) * 10])
        agg[key][0] += 1
        agg[key][1] += tms[i]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print("GEMM kernel time per step %.2f ms over %d launches" % (tot / n, N // n))
    print("%6s %6s %6s %4s %4s %3s %2s %2s %2s %3s | %5s %8s %8s %7s %6s" %
          ("M", "N", "K", "nb", "BM", "ks", "pr", "ak", "bk", "tap", "n/st", "us/call", "ms/step", "TF(alg)", "share"))
    for k, (c, t) in rows:
        M, Nn, K, nb = k[0], k[1], k[2], k[3]
        fl = 2.0 * M * Nn * K * nb
        print("%6d %6d %6d %4d %4d %3d %2d %2d %2d %3d | %5.1f %8.1f %8.3f %7.0f %5.1f%%" %
              (k + (c / n, t / c * 1e3, t / n, fl * c / (t * 1e-3) / 1e12, 100 * t / tot)))


if __name__ == "__main__":
    main()
