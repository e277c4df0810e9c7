"""device time of ops.pack on the step's activation shapes (graph replays); VILCO_PACK_CAP = workgroup cap of the kc pack"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench_targets as bt
from vilco_amd import ops
dev = torch.device("cuda:0")
for (M, K) in [(4608, 1024), (4608, 4096), (9082, 1024), (2304, 1024), (1152, 1024), (154, 1024)]:
    x = torch.randn(M, K, device=dev)
    y = ops.layernorm(x.view(1, M, K), torch.ones(K, device=dev), torch.zeros(K, device=dev)).view(M, K)   # carries amax partials
    def f():
        ops._pack_cache = False
        ops.pack(y, M, K)
    dt = bt.timeit(f, iters=50)
    print("cap=%s  pack [%d, %d]: %.2f us  %.2f TB/s" % (os.environ.get("VILCO_PACK_CAP", "2048"), M, K, dt * 1e6, 8.0 * M * K / dt / 1e12), flush=True)
