import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from parity_util import *
import vilco_amd
from vilco_amd import ops

def report(model, wgrads, tag):
    rows = []
    for k, p in model.named_parameters():
        w = wgrads.get(k)
        if w is None or p.grad is None:
            continue
        fro = float((p.grad.double().cpu() - w.double()).norm() / max(float(w.double().norm()), 1e-12))
        if float(w.abs().max()) < 1e-12:
            continue
        rows.append((rel_err(p.grad, w, GRAD_FLOOR), k, float(w.abs().max()), fro))
    rows.sort(reverse=True)
    print("==", tag)
    for r in rows[:18]:
        print("  %.3e  %-55s max|g|=%.2e fro=%.2e" % r)
    import numpy as np
    print("  median err %.2e ; fro: median %.2e max %.2e (excluding |g|<1e-12)" % (np.median([r[0] for r in rows]), np.median([r[3] for r in rows if r[2] > 1e-12]), max(r[3] for r in rows if r[2] > 1e-12)))

for name in [a for a in sys.argv[1:] if a != "none"]:
    gold = load_golden(name)
    want, wgrads, _ = oracle_run(gold, torch.float64)
    for prec in os.environ.get("DIAG_PREC", "split3").split(","):
        ops.set_precision(prec)
        model = build_hip_model(gold)
        model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
        losses = model(golden_inputs(gold), task_id=gold['task_id'], is_training=True)
        losses['final_loss'].backward()
        print(name, prec, {k: (float(v), float(want[k])) for k, v in losses.items()})
        report(model, wgrads, name + " " + prec)

if os.environ.get("DIAG_CFG1"):
    import vilco_amd.modeling as vm
    from oracle import mq_oracle as O
    from vilco_amd.core.config import make_config
    import cases
    over = cases.overrides(D=512, T=256, Cin=512, Ctxt=768, H=4, use_xl=True, droppath=0.1)
    cfg = make_config(**over)['model']
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=xlnet_json(512, 8, 2048)))
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if 'drop_path' in n_:
                p_.fill_(0.3)
    model.eval()
    vl = cases.video_list(256, 512, 768, 77)
    p64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    vl64 = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
    want, _ = O.forward_losses(p64, cfg, vl64)
    want['final_loss'].backward()
    model = model.to("cuda:0")
    wg = {k: v.grad for k, v in p64.items() if torch.is_tensor(v) and v.is_floating_point()}
    for prec in os.environ.get("DIAG_PREC", "split3").split(","):
        ops.set_precision(prec)
        model.zero_grad()
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        losses = model(vl, is_training=True)
        losses['final_loss'].backward()
        print(prec, {k: (float(v), float(want[k])) for k, v in losses.items()})
        report(model, wg, "cfg1 " + prec)
