"""forward time of the fused q/k/v pre-projection at the target shapes; run with VILCO_QKV_CS / VILCO_QKV_CS_TB set"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_targets as bt
dev = torch.device("cuda:0")
out = [bt.qkv_pre_target(dev, 2), bt.qkv_pre_target(dev, 8), bt.qkv_pre_target(dev, 8, stride=2), bt.qkv_pre_target(dev, 2, C=1024)]
for o in out:
    print("LDS=" + os.environ.get("VILCO_QKV_LDS", "1"), "TB=" + os.environ.get("VILCO_QKV_TB", "-"), o["shape"], "stride", o["stride"],
          "%.1f us  %.0f GB/s  frac %.3f" % (o["us"], o["GBps"], o["hbm_frac"]))
