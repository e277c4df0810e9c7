"""How far is the fp32 reference from an exact run of ITSELF at config P?  ("What's weak" #2.)

The full-size parity test (tests/test_fullsize_gpu.py) compares the HIP step with the fp32 oracle under the same dropout /
stochastic-depth masks and allows a few elements per tensor beyond 1e-3 of the tensor's maximum.  This script measures the
same statistics between the fp32 oracle and the fp64 oracle (oracle/mq_oracle.py, the pinned CPU restatement of the
reference's MQ path; identical parameters, inputs and masks): the element-wise self-distance of the reference arithmetic.
CPU only (build container: ~1 min fp32 + ~3 min fp64 on 8 cores, ~25 GB).  Writes profiles/r04_oracle_self_distance.json.

  python tools/diag/oracle_self_distance.py            # both runs + comparison
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


FULL = "--full-length" in sys.argv          # both clips T frames long (no padded rows): where does the distance come from?
NODROP = "--no-dropout" in sys.argv


def run(dtype):
    import bench
    import vilco_amd.modeling as vm
    from oracle import mq_oracle
    cfg = bench.p_config()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet()))
    p = {k: (v.detach().to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in model.state_dict().items()}
    del model
    clips = bench.synth_batch(2, "cpu")
    if FULL:
        g = torch.Generator().manual_seed(5)
        clips[1]['feats'] = torch.randn(2304, 2304, generator=g)
        clips[1]['segmentation_labels'] = torch.zeros(2304, 22)
    vl = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in clips]
    mq_oracle.DROP = None if NODROP else mq_oracle.DropRandom(dropout=0.1, droppath=0.1, xl=0.1, seed=0)
    t0 = time.time()
    losses, _ = mq_oracle.forward_losses(p, cfg, vl)
    losses['final_loss'].backward()
    mq_oracle.DROP = None
    print(dtype, {k: float(v) for k, v in losses.items()}, "%.0f s" % (time.time() - t0), flush=True)
    return {k: float(v) for k, v in losses.items()}, {k: v.grad.detach().double() for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def main():
    torch.set_num_threads(max(1, (os.cpu_count() or 1)))
    l32, g32 = run(torch.float32)
    torch.save(g32, "/tmp/_g32.pt")
    del g32
    l64, g64 = run(torch.float64)
    g32 = torch.load("/tmp/_g32.pt")
    os.remove("/tmp/_g32.pt")
    rows = []
    for k, w in g64.items():
        if k.endswith(('key_norm.bias', '.key.bias')):          # analytically zero gradients (softmax shift invariance)
            continue
        g = g32[k]
        d = (g - w).abs()
        top = w.abs().max().clamp_min(1e-7)
        rows.append({"tensor": k, "max_rel": (d.max() / top).item(), "l2_rel": ((g - w).norm() / w.norm().clamp_min(1e-12)).item(),
                     "frac_beyond_1e-3": (d > 1e-3 * top).double().mean().item(), "numel": w.numel()})
    rows.sort(key=lambda r: -r["max_rel"])
    out = {"what": "fp32 oracle vs fp64 oracle, config P, 2 clips, train mode, identical masks (DropRandom seed 0): per-tensor "
                   "max |g32 - g64| / max |g64|, L2 distance, fraction of elements beyond 1e-3 of the tensor's maximum",
           "losses_fp32": l32, "losses_fp64": l64,
           "tensors": len(rows), "tensors_with_max_rel_above_1e-3": sum(1 for r in rows if r["max_rel"] > 1e-3),
           "worst_max_rel": rows[0]["max_rel"], "worst_l2_rel": max(r["l2_rel"] for r in rows),
           "worst_frac_beyond_1e-3": max(r["frac_beyond_1e-3"] for r in rows), "top": rows[:12]}
    out["variant"] = {"full_length_clips": FULL, "dropout": not NODROP}
    path = os.path.join(ROOT, "profiles", "r04_oracle_self_distance%s%s.json" % ("_full" if FULL else "", "_nodrop" if NODROP else ""))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "top"}, indent=1))
    for r in rows[:12]:
        print(r)


if __name__ == "__main__":
    main()
