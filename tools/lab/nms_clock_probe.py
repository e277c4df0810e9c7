"""Is the one-workgroup soft-NMS slow in CYCLES or in CLOCK?  Times vilco_softnms_1d (30 000 candidates, one class) alone and with
a large matmul kept running on a second stream (which holds the chip's clock up)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from tools.nms_bench import candidates
from vilco_amd.utils.nms import nms_1d_cpu
dev = torch.device("cuda:0")
segs, scores, cls = candidates(30000, 1, 1)
segs, scores = segs.to(dev), scores.to(dev)
def run():
    dets = torch.zeros(30000, 3, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = nms_1d_cpu.softnms(segs, scores, dets, 0.1, 0.75, 0.01, 2)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, idx.numel()
for _ in range(2): run()
print("alone: %.1f ms (%d kept)" % run())
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
with torch.cuda.stream(side):
    for _ in range(400): c = a @ b
print("with a matmul stream running: %.1f ms (%d kept)" % run())
torch.cuda.synchronize()
