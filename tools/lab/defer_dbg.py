import sys, os, torch
sys.path.insert(0, 'tests')
from parity_util import build_hip_model, golden_inputs, load_golden, golden_cfg
from vilco_amd import ops, _lib
name = sys.argv[1] if len(sys.argv) > 1 else 'xl'
gold = load_golden(name)
res = {}
for mode in (False, True):
    ops.defer_finish = mode
    model = build_hip_model(gold)
    model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
    losses = model(golden_inputs(gold), task_id=gold['task_id'], is_training=True)
    losses['final_loss'].backward()
    torch.cuda.synchronize()
    res[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
bad = [(k, float((res[True][k] - res[False][k]).abs().max())) for k in res[True] if not torch.equal(res[True][k], res[False][k])]
print(len(res[True]), "tensors;", len(bad), "differ")
for k, e in bad[:40]:
    print(k, e, tuple(res[True][k].shape))
