"""NMS timing leg of the bench line (last bullet): the reference's compiled CPU extension
(oracle/_ref/nms_1d_cpu.so = MQ/libs/utils/csrc/nms_cpu.cpp:19-160, built by oracle/build_ref.py; per-class python loop as
MQ/libs/utils/nms.py:124-152) against the HIP kernels (vilco_nms_1d / vilco_softnms_1d, all classes in one launch) on the same
seeded candidates, N in {5 000, 30 000}, hard and soft (gaussian), one class and 22 classes.  Device times are HIP events
around the launch with the candidates resident in HBM; index outputs are compared (bit-exact) while timing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def candidates(n, seed, ncls):
    g = np.random.RandomState(seed)
    c = g.uniform(0, 2000.0, n).astype(np.float32)             # a 2000-unit timeline: overlap degree like a decoded pyramid
    w = g.uniform(0.5, 60, n).astype(np.float32)
    segs = np.stack([c - w / 2, c + w / 2], 1).astype(np.float32)
    scores = g.uniform(0.001, 1, n).astype(np.float32)
    cls = np.sort(g.randint(0, ncls, n)).astype(np.int64)
    return torch.from_numpy(segs), torch.from_numpy(scores), torch.from_numpy(cls)


def _ref_module():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    try:
        import nms_1d_cpu
        return nms_1d_cpu
    except Exception:
        return None


def cpu_time(ref, segs, scores, cls, soft, thr, sigma, min_score, reps=1):
    out = []
    t0 = time.perf_counter()
    for _ in range(reps):
        out = []
        for c in torch.unique(cls):
            m = cls == c
            s, sc = segs[m].contiguous(), scores[m].contiguous()
            if soft:
                dets = torch.zeros(s.shape[0], 3)
                out.append(ref.softnms(s, sc, dets, thr, sigma, min_score, 2))
            else:
                out.append(ref.nms(s, sc, thr))
    return (time.perf_counter() - t0) / reps * 1e3, out


def hip_time(segs, scores, cls, soft, thr, sigma, min_score, reps=3):
    from vilco_amd.utils import nms as N
    dev = torch.device("cuda:0")
    s, sc = segs.to(dev).contiguous(), scores.to(dev).contiguous()
    _, counts = torch.unique_consecutive(cls, return_counts=True)
    off = torch.zeros(counts.numel() + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(counts, 0)
    off_d = off.to(dev)
    nseg = counts.numel()

    def run():
        if soft:
            _, idx, cnt = N._run_soft(s, sc, off_d, nseg, thr, sigma, min_score, 2, 0)
        else:
            idx, cnt = N._run_hard(s, sc, off_d, nseg, thr)
        return idx, cnt
    idx, cnt = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        idx, cnt = run()
    e1.record()
    torch.cuda.synchronize()
    cnt = cnt.cpu()
    idx = idx.cpu()
    return e0.elapsed_time(e1) / reps, [idx[int(off[c]):int(off[c]) + int(cnt[c])] for c in range(nseg)]


def nms_timing(sizes=(5000, 30000), classes=(1, 22), thr_hard=0.5, thr_soft=0.1, sigma=0.75, min_score=0.001):
    ref = _ref_module()
    rows = []
    for n in sizes:
        for ncls in classes:
            segs, scores, cls = candidates(n, 1000 + n + ncls, ncls)
            for soft in (False, True):
                thr = thr_soft if soft else thr_hard
                hip_ms, hip_out = hip_time(segs, scores, cls, soft, thr, sigma, min_score)
                row = {"n": n, "classes": ncls, "kind": "soft" if soft else "hard", "hip_ms": round(hip_ms, 3),
                       "kept": int(sum(x.numel() for x in hip_out))}
                if ref is not None:
                    cpu_ms, cpu_out = cpu_time(ref, segs, scores, cls, soft, thr, sigma, min_score)
                    row["cpu_ms"] = round(cpu_ms, 3)
                    row["speedup"] = round(cpu_ms / hip_ms, 2)
                    row["indices_equal"] = bool(len(cpu_out) == len(hip_out) and all(torch.equal(a, b) for a, b in zip(cpu_out, hip_out)))
                rows.append(row)
    return {"cpu": "oracle/_ref nms_1d_cpu (the reference's nms_cpu.cpp, 1 thread, per-class python loop)" if ref is not None
            else "oracle/_ref not built", "hip": "vilco_nms_1d / vilco_softnms_1d, all classes in one launch, HIP events", "rows": rows}


if __name__ == "__main__":
    import json
    print(json.dumps(nms_timing(), indent=1))
