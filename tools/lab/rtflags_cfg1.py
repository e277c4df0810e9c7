"""cfg1 (BASELINE configs[0]: the P graph at T = 256, every node tiny) replayed under whatever HIP runtime knobs the environment
sets: ms per step = the per-node floor of a replayed step.  tools/lab/r6_rtflags.sh runs it per setting."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
r = bench.side_config("cfg1", torch.device("cuda:0"), steps=40, warm=3)
print("%-42s cfg1 %.3f ms/step" % (sys.argv[1] if len(sys.argv) > 1 else "", r["ms_per_step"]), flush=True)
