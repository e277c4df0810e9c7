"""Round 5: the full-size parity comparison over SEVERAL mask realisations, HIP and the fp32 oracle both
measured against the fp64 oracle under the same masks.  Prints / writes per realisation and tensor: d(HIP, fp64), d(fp32, fp64).
  python tools/diag/p_parity_realisations.py [n_realisations]   ->  profiles/r05_p_parity_realisations.json"""
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
for r in range(n):
    t0 = time.time()
    hl, hg, orc = p_step_three_ways(dev, r, threads=min(64, os.cpu_count() or 1))
    l32, g32 = orc[torch.float32]
    l64, g64 = orc[torch.float64]
    rows = []
    for k, w in g64.items():
        if k.endswith(('key_norm.bias', '.key.bias')):
            continue
        dh, dr = tensor_distance(hg[k], w), tensor_distance(g32[k], w)
        rows.append({"tensor": k, "hip_max": dh[0], "ref_max": dr[0], "hip_l2": dh[1], "ref_l2": dr[1], "hip_frac": dh[2], "ref_frac": dr[2]})
    rows.sort(key=lambda x: -x["hip_max"])
    viol = [x for x in rows if x["hip_max"] > max(1e-3, 2 * x["ref_max"])]
    violl2 = [x for x in rows if x["hip_l2"] > max(1e-3, 2 * x["ref_l2"])]
    print("realisation %d: %.0f s  losses hip %s | fp32 %s | fp64 %s" % (r, time.time() - t0, hl, l32, l64), flush=True)
    print("  tensors %d; hip beyond 1e-3: %d, fp32 beyond 1e-3: %d; per-tensor bound max(1e-3, 2 d_ref) violated: %d (max), %d (l2)" %
          (len(rows), sum(1 for x in rows if x["hip_max"] > 1e-3), sum(1 for x in rows if x["ref_max"] > 1e-3), len(viol), len(violl2)))
    for x in rows[:10]:
        print("   %-60s hip %.2e ref %.2e | l2 hip %.2e ref %.2e" % (x["tensor"], x["hip_max"], x["ref_max"], x["hip_l2"], x["ref_l2"]))
    res.append({"realisation": r, "losses_hip": hl, "losses_fp32": l32, "losses_fp64": l64, "violations_max": viol, "violations_l2": violl2,
                "top": rows[:20], "worst_ref": sorted(rows, key=lambda x: -x["ref_max"])[:10]})
out = {"what": "config P train step, HIP and fp32 oracle vs fp64 oracle under the same replayed masks, per mask realisation; "
               "max = max |g - g64| / max |g64| per tensor, l2 = ||g - g64|| / ||g64||", "realisations": res}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r05_p_parity_realisations.json"), "w") as f:
    json.dump(out, f, indent=1)
