"""Golden vectors of the class-agnostic branch of the reference's batched_nms WITH segment voting (MQ/libs/utils/nms.py:67-101,
:161-180), from the IMPORTED reference and its compiled nms_1d_cpu (this container only).
Run:  python tests/golden/make_golden_nms_voting.py   ->  tests/golden/nms_voting_{soft,hard}.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def case(n, seed):
    g = np.random.RandomState(seed)
    c = g.uniform(0, 120.0, n).astype(np.float32)           # dense enough that every kept segment has voting neighbours
    w = g.uniform(2.0, 30, n).astype(np.float32)
    segs = np.stack([c - w / 2, c + w / 2], 1).astype(np.float32).reshape(n, 2)
    scores = g.uniform(0.001, 1, n).astype(np.float32)
    return segs, scores


if __name__ == "__main__":
    ref_import.setup()
    from libs.utils import batched_nms
    for soft in (True, False):
        n = 1200
        segs, scores = case(n, 31 + int(soft))
        cls = np.zeros(n, dtype=np.int64)
        s, sc, c = batched_nms(torch.from_numpy(segs), torch.from_numpy(scores), torch.from_numpy(cls), 0.1, 0.01, 100,
                               use_soft_nms=soft, multiclass=False, sigma=0.75, voting_thresh=0.75)
        np.savez_compressed(os.path.join(HERE, 'nms_voting_%s.npz' % ('soft' if soft else 'hard')), kind='voting', segs=segs,
                            scores=scores, cls=cls, thr=0.1, min_score=0.01, max_seg_num=100, soft=soft, sigma=0.75,
                            voting_thresh=0.75, out_segs=s.numpy(), out_scores=sc.numpy(), out_cls=c.numpy())
        print('voting', 'soft' if soft else 'hard', tuple(s.shape), float(np.abs(s.numpy()).max()))
