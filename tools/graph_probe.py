"""Eager vs hipGraph-replayed step (fwd+bwd, and a whole training iteration) of config P / cfg1: wall clock, host
enqueue time, and that the replayed losses follow the eager ones.  python tools/graph_probe.py [P|cfg1] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import vilco_amd.modeling as vm
from vilco_amd import ops
from vilco_amd.graph import GraphedStep
from vilco_amd.core.config import make_config
from vilco_amd.utils.train_utils import make_optimizer

import faulthandler
faulthandler.enable()
name = sys.argv[1] if len(sys.argv) > 1 else "P"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
if name == "P":
    cfg, xl, T, Cin = bench.p_config(), bench.P_XLNET, 2304, 2304
else:
    over = dict(dataset=dict(input_dim=512, num_classes=22, max_seq_len=256),
                model=dict(embd_dim=512, fpn_dim=512, head_dim=512, n_head=4, backbone_arch=(2, 2, 5), use_abs_pe=True,
                           use_cross_modal=True, n_txt_in=768, max_buffer_len_factor=1.0, use_xl=True),
                train_cfg=dict(init_loss_norm=100, dropout=0.1, droppath=0.1))
    cfg, T, Cin = make_config(**over)['model'], 256, 512
    xl = dict(bench.P_XLNET, d_model=512, n_head=4, d_head=128, d_inner=1024, dropout=0.1)
torch.manual_seed(0)
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=xl)).to(dev).train()
batch = bench.synth_batch(2, dev, seed=0, T=T, Cin=Cin)
params = list(model.parameters())


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3


def eager():
    for p in params:
        p.grad = None
    return model(batch, is_training=True)['final_loss'].backward()

if not os.environ.get("PROBE_SKIP_EAGER"):
    print(name, "eager fwd+bwd: %.2f ms/step (host enqueue %.2f ms)" % timed(eager, steps), flush=True)
if os.environ.get("PROBE_STAGES"):
    from vilco_amd import _lib
    import gc
    lib = _lib.load()
    g = torch.cuda.CUDAGraph()
    x = torch.randn(4096, device=dev); y = torch.empty_like(x)
    with torch.cuda.graph(g):
        _lib.check(lib.vilco_seed_word_bump(ops._stream()))
        _lib.check(lib.vilco_dropout(x.data_ptr(), y.data_ptr(), x.numel(), 0.5, 123, 0, ops._stream()))
    g.replay(); torch.cuda.synchronize(); a = y.clone(); g.replay(); torch.cuda.synchronize()
    print("stage 1 (kernel capture) ok; masks differ between replays:", bool((a != y).any()), flush=True)
    inp = model.prepare(batch, True, gt_pad=8)
    model._cat = None; gc.collect()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        with torch.cuda.graph(g):
            out = model.forward_prepared(inp, None, task_id=0)['final_loss']
    g.replay(); torch.cuda.synchronize()
    print("stage 2 (forward capture) ok, loss", float(out), flush=True)
    del out
    model._cat = None; gc.collect()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.forward_prepared(inp, None, task_id=0)['final_loss']
        out.backward()
    g.replay(); torch.cuda.synchronize()
    print("stage 3 (forward+backward capture) ok, loss", float(out), flush=True)
    del out, g
gs = GraphedStep(model, None, eager_steps=1)
l = [gs(batch) for _ in range(4)]
print("losses over 4 graphed calls:", [round(float(x['final_loss']), 5) for x in l], gs.stats, flush=True)
print(name, "graph fwd+bwd: %.2f ms/step (host enqueue %.2f ms)" % timed(lambda: gs(batch), steps), flush=True)
gn = [float(p.grad.abs().max()) for p in params if p.grad is not None]
print("grads finite:", all(x == x and x < 1e30 for x in gn), "n", len(gn))
if os.environ.get("PROBE_FB_ONLY"):
    sys.exit(0)
# training iteration
opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
def eager_it():
    eager()
    opt.step(clip_grad_l2norm=1.0)
print(name, "eager iteration: %.2f ms (host %.2f ms)" % timed(eager_it, steps), flush=True)
gt = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=1)
l = [float(gt(batch)['final_loss']) for _ in range(6)]
print("losses over 6 graphed training iterations:", [round(x, 5) for x in l], gt.stats, flush=True)
print(name, "graph iteration: %.2f ms (host %.2f ms)" % timed(lambda: gt(batch), steps), flush=True)
print("loss after:", float(gt(batch)['final_loss']))
