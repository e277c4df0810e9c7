"""Round 6: the data-parallel training step at config P on a one-rank RCCL group, from one initial state over four rotating batches,
dropout off: (a) staged replay (buckets launched between stage graphs), (b) one-graph replay + exchange after it, (c) eager with hook
exchange, (d) no reducer, plain replay.  Prints the loss trajectories."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
import vilco_amd.modeling as vm
from vilco_amd.dist import GradReducer
from vilco_amd.graph import GraphedStep
from vilco_amd.utils.train_utils import make_optimizer
dev = torch.device("cuda:0")
cfg = bench.p_config(dropout=0.0, droppath=0.0)
batches = [bench.synth_batch(2, dev, seed=s) for s in range(4)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["staged", "onegraph", "eager_dp", "plain_replay"]
out = {}
for mode in modes:
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(dropout=0.0))).to(dev).train()
    opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
    red = GradReducer(model) if mode != "plain_replay" else None
    gs = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=2, reducer=red, enabled=(mode != "eager_dp"), segments=(mode == "staged"))
    losses = [gs(batches[it % 4])['final_loss'] for it in range(N)]
    torch.cuda.synchronize()
    out[mode] = torch.stack(losses).float().cpu()
    print("%-13s stats %s\n   %s" % (mode, gs.stats, " ".join("%.4f" % v for v in out[mode].tolist())), flush=True)
    if red is not None:
        red.remove()
    del model, opt, gs, red
    torch.cuda.empty_cache()
ref = out.get("eager_dp")
if ref is not None:
    for m, l in out.items():
        print("%-13s max relative distance to eager_dp: %.3e (first 8: %.3e)" % (m, float(((l - ref).abs() / ref.abs()).max()), float(((l[:8] - ref[:8]).abs() / ref[:8].abs()).max())))
dist.destroy_process_group()
