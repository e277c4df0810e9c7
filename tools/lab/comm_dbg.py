import os, sys, torch
sys.path.insert(0, 'tests')
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
import test_dist_gpu as T
dev = torch.device("cuda:0")
from vilco_amd.graph import GraphedStep
orig = GraphedStep._capture
def cap(self, ent, inp, task_id):
    try:
        r = orig(self, ent, inp, task_id)
    except Exception as e:
        print("CAPTURE EXC", type(e).__name__, str(e)[:300]); raise
    print("captured: comm =", ent.get('comm'), "refused =", ent.get('comm_refused'), "fill", len(ent['fill']))
    return r
GraphedStep._capture = cap
for mode in ("plain", "graph", "graph_comm"):
    l, s, _ = T._train(dev, mode)
    print(mode, l)
dist.destroy_process_group()
