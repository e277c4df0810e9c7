"""Sensitivity of the fp32 reference arithmetic at config P to a ONE-ULP perturbation of its inputs.

Two fp32 oracle runs (oracle/mq_oracle.py; same masks): the second with every floating-point parameter multiplied by
(1 + u), u uniform in +-2^-24 -- i.e. re-rounded by at most one unit in the last place.  Any arithmetic that differs from the
reference's in rounding (another summation order, MFMA with fp32 accumulation, ...) perturbs intermediate values at least
this much, so the per-tensor distance between these two runs is the floor below which element-wise agreement with the
fp32 reference carries no information.  Writes profiles/r04_oracle_perturbation.json.  CPU only, ~2 min."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


BITS = int(os.environ.get("PERT_BITS", "24"))          # relative perturbation 2^-BITS (24: one fp32 ulp; 22: the fp16 x2 operand format)


def run(perturb):
    import bench
    import vilco_amd.modeling as vm
    from oracle import mq_oracle
    cfg = bench.p_config()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet()))
    g = torch.Generator().manual_seed(123)
    p = {}
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            w = v.detach().clone()
            if perturb:
                w = w * (1.0 + (torch.rand(w.shape, generator=g) * 2 - 1) * 2.0 ** -BITS)
            p[k] = w.requires_grad_(True)
        else:
            p[k] = v
    del model
    vl = bench.synth_batch(2, "cpu")
    if perturb and BITS != 24:          # an arithmetic with BITS-bit operands also rounds the activations it consumes: the inputs here
        for d in vl:
            for k in ('feats', 'prompt_feature'):
                d[k] = d[k] * (1.0 + (torch.rand(d[k].shape, generator=g) * 2 - 1) * 2.0 ** -BITS)
    mq_oracle.DROP = mq_oracle.DropRandom(dropout=0.1, droppath=0.1, xl=0.1, seed=0)
    t0 = time.time()
    losses, _ = mq_oracle.forward_losses(p, cfg, vl)
    losses['final_loss'].backward()
    mq_oracle.DROP = None
    print({k: float(v.detach()) for k, v in losses.items()}, "%.0f s" % (time.time() - t0), flush=True)
    return {k: v.grad.detach().clone() for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    a, b = run(False), run(True)
    rows = []
    for k, w in a.items():
        if k.endswith(('key_norm.bias', '.key.bias')):
            continue
        g = b[k]
        d = (g - w).abs()
        top = w.abs().max().clamp_min(1e-7)
        rows.append({"tensor": k, "max_rel": (d.max() / top).item(), "l2_rel": ((g - w).norm() / w.norm().clamp_min(1e-12)).item(),
                     "frac_beyond_1e-3": (d > 1e-3 * top).float().mean().item()})
    rows.sort(key=lambda r: -r["max_rel"])
    out = {"what": "fp32 oracle vs fp32 oracle with every parameter%s multiplied by (1 + u), |u| <= 2^-%d; config P, 2 clips, train mode, same masks" % (" and input feature" if BITS != 24 else "", BITS),
           "tensors": len(rows), "tensors_with_max_rel_above_1e-3": sum(1 for r in rows if r["max_rel"] > 1e-3),
           "worst_max_rel": rows[0]["max_rel"], "worst_l2_rel": max(r["l2_rel"] for r in rows),
           "worst_frac_beyond_1e-3": max(r["frac_beyond_1e-3"] for r in rows), "top": rows[:12]}
    with open(os.path.join(ROOT, "profiles", "r04_oracle_perturbation%s.json" % ("" if BITS == 24 else "_2e-%d" % BITS)), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "top"}))
    for r in rows[:10]:
        print(r)


if __name__ == "__main__":
    main()
