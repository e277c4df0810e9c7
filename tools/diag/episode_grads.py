"""diagnostic: HIP vs fp64-oracle gradients on the episode case's task-1 state (reference state after task 0 + head growth)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity_util import cases, episode_full_state, load_episode_golden, rel_err, GRAD_FLOOR
from ref_import import xlnet_json
from vilco_amd.core.config import make_config
import vilco_amd.modeling as vm
from oracle import mq_oracle

gold = load_episode_golden()
cfg = make_config(**gold['overrides'])
dev = torch.device("cuda:0")
model = vm.make_meta_arch('LocPointTransformer', **dict(cfg['model'], xlnet_config=xlnet_json(cfg['model']['embd_dim'], cases.EP_H)))
st = episode_full_state(gold['tasks'][0]['state'])
model.load_state_dict(st)
model.augment_classification(cases.EP_NEW, 'cpu')
sd = model.state_dict()
for k, v in gold['tasks'][0]['post_augment'].items():
    sd[k].copy_(v)
model = model.to(dev).train()
model.n_known = 4
for bi in range(2):
    model.loss_normalizer = 100.0
    model.zero_grad(set_to_none=True)
    vl = cases.episode_batches(1)[bi]
    losses = model(vl, task_id=1)
    losses['final_loss'].backward()
    mcfg = dict(cfg['model']); mcfg['num_classes'] = 7
    p = {k: (v.detach().double().cpu().clone().requires_grad_(True) if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()}
    vl64 = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
    want, _ = mq_oracle.forward_losses(p, mcfg, vl64, loss_normalizer=100.0, task_id=1, n_known=4)
    want['final_loss'].backward()
    print("batch", bi, {k: (float(losses[k]), float(want[k])) for k in want})
    errs = []
    for k, q in model.named_parameters():
        if p[k].grad is not None and q.grad is not None:
            errs.append((rel_err(q.grad, p[k].grad, GRAD_FLOOR), k, float(p[k].grad.abs().max())))
    errs.sort(reverse=True)
    for e in errs[:12]:
        print("  %.3e %-60s max|g| %.3e" % e)
    for name in ('reg_head.norm.0.weight', 'reg_head.norm.0.bias'):
        q = dict(model.named_parameters())[name]
        g, w = q.grad.detach().double().cpu().reshape(-1), p[name].grad.reshape(-1)
        order = w.abs().argsort()
        print("  ", name, "n", g.numel(), "max|g| %.3e" % float(w.abs().max()))
        for i in order[:6].tolist():
            print("     elem %3d  hip % .6e  fp64 % .6e" % (i, float(g[i]), float(w[i])))
