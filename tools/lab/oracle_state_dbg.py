"""which process event changes the CPU oracle's config-P gradients?  Runs the fp32 oracle (CPU masks, seed 0) after each stage."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import vilco_amd.modeling as vm
from oracle import mq_oracle
cfg = bench.p_config()
torch.manual_seed(0)
state = {k: v.detach().clone() for k, v in vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).state_dict().items()}
KEYS = ('backbone.branch.0.mlp.3.weight', 'backbone.branch.1.mlp.3.weight', 'reg_head.head.0.conv.weight', 'cls_head.head.0.conv.weight', 'backbone.stem.0.mlp.3.weight')
def run():
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}
    mq_oracle.DROP = mq_oracle.DropRandom(dropout=0.1, droppath=0.1, xl=0.1, seed=0)
    losses, _ = mq_oracle.forward_losses(p, cfg, bench.synth_batch(2, "cpu"))
    losses['final_loss'].backward()
    mq_oracle.DROP = None
    return {k: p[k].grad.detach().clone() for k in KEYS}
base = None
def stage(tag):
    global base
    t0 = time.time(); g = run()
    if base is None:
        base = g
    print(tag, "threads", torch.get_num_threads(), "%.0fs" % (time.time() - t0),
          {k.split('.')[-4] + '.' + k.split('.')[-2]: "%.2e" % float((g[k] - base[k]).abs().max() / base[k].abs().max()) for k in KEYS}, flush=True)
stage("fresh")
stage("again")
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
stage("after cuda init")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
stage("after nccl init + all_reduce")
dist.destroy_process_group()
stage("after destroy")
import pytest
pytest.main([os.path.join(ROOT, "tests/test_dist_gpu.py"), "-q", "-m", "gpu", "-k", "reducer_paths", "-p", "no:cacheprovider"])
stage("after reducer_paths")
pytest.main([os.path.join(ROOT, "tests/test_episode.py"), "-q", "-m", "gpu", "-k", "run_episodes_end_to_end and True", "-p", "no:cacheprovider"])
stage("after run_episodes[True]")
