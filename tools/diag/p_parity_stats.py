"""HIP step vs the fp32 oracle at config P (the workload of the bench line, train mode, masks replayed): per-tensor statistics
of every parameter gradient, for the default arithmetic (single-part weight gradients for K >= 2048) and the strict one
(VILCO_DW_PRECISION=f16x2).  GPU box.  Writes gpurun_out/r04_p_parity_stats.json (copied to profiles/)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import vilco_amd.modeling as vm  # noqa: E402
from oracle import mq_oracle  # noqa: E402
from vilco_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.p_config()
out = {}
want_cache = None
which = sys.argv[1] if len(sys.argv) > 1 else "default"          # one arithmetic per process: the stochastic-depth pool is process state
for label, dwp in ((("default", 4),) if which == "default" else (("strict", None),)):
    ops.dw_precision = dwp
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
    batch = bench.synth_batch(2, dev)
    ops._drop_counter[0] = 0
    ops.dropout_log = []
    losses = model(batch, is_training=True)
    losses['final_loss'].backward()
    log = list(ops.dropout_log)
    ops.dropout_log = None
    torch.cuda.synchronize()
    got = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    got_losses = {k: float(v) for k, v in losses.items()}
    if want_cache is None:
        p = {k: (v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())) for k, v in model.state_dict().items()}
        vl = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.items()} for d in batch]
        mq_oracle.DROP = mq_oracle.DropReplay(log, lambda pr, seed, shape, site: ops.dropout_mask(pr, seed, shape, dev, site).cpu())
        want, _ = mq_oracle.forward_losses(p, cfg, vl)
        want['final_loss'].backward()
        mq_oracle.DROP = None
        want_cache = ({k: float(v) for k, v in want.items()}, {k: v.grad.clone() for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None})
    del model, losses
    torch.cuda.empty_cache()
    wl, wg = want_cache
    rows = []
    for k, g in got.items():
        if k in wg and not k.endswith(('key_norm.bias', '.key.bias')):
            w = wg[k]
            d = (g - w).abs()
            top = w.abs().max().clamp_min(1e-7)
            rows.append({"tensor": k, "max_rel": (d.max() / top).item(), "l2_rel": ((g - w).norm() / w.norm().clamp_min(1e-12)).item(),
                         "frac_beyond_1e-3": (d > 1e-3 * top).float().mean().item()})
    rows.sort(key=lambda r: -r["max_rel"])
    out[label] = {"losses": got_losses, "losses_oracle": wl, "tensors": len(rows),
                  "tensors_with_max_rel_above_1e-3": sum(1 for r in rows if r["max_rel"] > 1e-3),
                  "worst_max_rel": rows[0]["max_rel"], "worst_l2_rel": max(r["l2_rel"] for r in rows),
                  "worst_frac_beyond_1e-3": max(r["frac_beyond_1e-3"] for r in rows), "top": rows[:12]}
    print(label, {k: v for k, v in out[label].items() if k != "top"})
    for r in rows[:8]:
        print("   ", r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r04_p_parity_stats_%s.json" % which), "w") as f:
    json.dump(out, f, indent=1)
