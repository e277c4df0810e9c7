"""forward time of the fused q/k/v pre-projection (device time: graph replays) at the target shapes and at the shapes of the
P / cfg1 / W steps; run with VILCO_QKV_RING=0/1 to compare the three-pass kernel with the ring kernel"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_targets as bt
dev = torch.device("cuda:0")
shapes = [(2, 2304, 2304, 1), (8, 2304, 2304, 1), (8, 2304, 2304, 2), (2, 2304, 2304, 2),
          (2, 2304, 1024, 1), (2, 2304, 1024, 2), (2, 1152, 1024, 2), (2, 576, 1024, 2), (2, 288, 1024, 2), (2, 144, 1024, 1),
          (8, 2304, 1024, 1), (2, 256, 512, 1), (2, 128, 512, 2)]
for (B, T, C, s) in shapes:
    o = bt.qkv_pre_target(dev, B, T=T, C=C, stride=s)
    print("RING=" + os.environ.get("VILCO_QKV_RING", "-"), o["shape"], "stride", o["stride"],
          "%.1f us  %.0f GB/s  frac %.3f" % (o["us"], o["GBps"], o["hbm_frac"]), flush=True)
