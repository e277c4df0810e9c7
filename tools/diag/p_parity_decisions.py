"""Round 6: the full-size parity comparison THREE ways per mask realisation -- HIP vs the fp64 oracle, the
fp32 oracle vs the fp64 oracle, HIP vs the fp32 oracle DIRECTLY -- and, for every realisation, a second fp64 oracle run that takes
the HIP step's LayerNorm -> ReLU sign decisions (oracle.mq_oracle.ReluReplay): where did the signs differ, how close to zero were
those pre-activations in exact arithmetic, and how far is HIP from that run.
  python tools/diag/p_parity_decisions.py [n_realisations]   ->  gpurun_out/r06_p_parity_decisions.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_util import p_step_three_ways, tensor_distance  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
res = []
skip = ('key_norm.bias', '.key.bias')
for r in range(n):
    t0 = time.time()
    hl, hg, orc = p_step_three_ways(dev, r, threads=min(64, os.cpu_count() or 1))
    l32, g32 = orc[torch.float32]
    l64, g64 = orc[torch.float64]
    lf, gf, events = orc['rerun'](torch.float64, orc['hip_relu'])
    l32f, g32f, events32 = orc['rerun'](torch.float32, orc['hip_relu'])
    rows = []
    for k, w in g64.items():
        if k.endswith(skip):
            continue
        dh, dr, dd, df, dd32 = (tensor_distance(hg[k], w), tensor_distance(g32[k], w), tensor_distance(hg[k], g32[k]),
                                tensor_distance(hg[k], gf[k]), tensor_distance(hg[k], g32f[k]))
        rows.append({"tensor": k, "hip_max": dh[0], "ref_max": dr[0], "hip_l2": dh[1], "ref_l2": dr[1], "hipref_max": dd[0], "hipref_l2": dd[1],
                     "hipforced_max": df[0], "hipforced_l2": df[1], "hipforced32_max": dd32[0], "hipforced32_l2": dd32[1]})
    def cnt(key, thr=1e-3):
        return sum(1 for x in rows if x[key] > thr)
    print("realisation %d: %.0f s" % (r, time.time() - t0), flush=True)
    print("  tensors %d | beyond 1e-3 (max / l2): HIP-fp64 %d / %d, fp32-fp64 %d / %d, HIP-fp32 %d / %d, HIP-forced64 %d / %d, HIP-forced32 %d / %d" %
          (len(rows), cnt("hip_max"), cnt("hip_l2"), cnt("ref_max"), cnt("ref_l2"), cnt("hipref_max"), cnt("hipref_l2"),
           cnt("hipforced_max"), cnt("hipforced_l2"), cnt("hipforced32_max"), cnt("hipforced32_l2")))
    print("  ReLU sign events vs fp64 (site, elements, max |x| / max): %s" % events)
    print("  ReLU sign events vs fp32: %s" % events32)
    for x in sorted(rows, key=lambda x: -x["hipref_l2"])[:8]:
        print("   %-58s HIP-fp32 max %.2e l2 %.2e | HIP-forced32 max %.2e l2 %.2e | HIP-fp64 l2 %.2e fp32-fp64 l2 %.2e" %
              (x["tensor"], x["hipref_max"], x["hipref_l2"], x["hipforced32_max"], x["hipforced32_l2"], x["hip_l2"], x["ref_l2"]))
    res.append({"realisation": r, "losses": {"hip": hl, "fp32": l32, "fp64": l64, "fp64_forced": lf, "fp32_forced": l32f},
                "relu_events_vs_fp64": events, "relu_events_vs_fp32": events32, "rows": rows})
    del hg, g32, g64, gf, g32f, orc
out = {"what": "config P train step: per mask realisation and tensor, max = max |a - b| / max |b|, l2 = ||a - b|| / ||b|| for the pairs (HIP, fp64 "
               "oracle), (fp32 oracle, fp64 oracle), (HIP, fp32 oracle), (HIP, fp64 oracle taking HIP's LayerNorm -> ReLU signs), (HIP, fp32 "
               "oracle taking them); relu events = (site, elements whose sign differed, largest |pre-activation| among them / max |pre-activation|)",
       "realisations": res}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", os.environ.get("VILCO_DIAG_OUT", "r06_p_parity_decisions.json")), "w") as f:
    json.dump(out, f)
