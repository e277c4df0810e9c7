"""Generates the golden vectors from the IMPORTED REFERENCE (this container only; the reference's
python never travels).  Run:  python tests/golden/make_golden.py

  model_<case>.pt : state_dict, loss dict, every parameter gradient (deterministic mode: model.eval()
                    + forward(is_training=True)), and the inference outputs of clip 0.
  nms_*.npz       : (segs, scores[, cls], params) -> exact indices / dets of the reference's compiled
                    nms_1d_cpu and its python batched_nms.
Inputs are regenerated from seeds by tests/golden/cases.py, so only outputs + weights are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402


def model_goldens(libs):
    from libs.modeling import make_meta_arch
    for name, (okw, L) in cases.CASES.items():
        over = cases.overrides(**okw)
        cfg = ref_import.make_cfg(libs, **over)
        torch.manual_seed(1234)
        model = make_meta_arch(cfg['model_name'], **cfg['model'])
        # the reference zero-inits nothing we need, but AffineDropPath scales start at 1e-4 and the
        # cls bias at -4.6: perturb every parameter a little so all paths carry signal
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if 'drop_path' in n_:
                    p_.copy_(0.5 + 0.1 * torch.randn(p_.shape, generator=g))
                elif p_.dim() <= 1 or 'norm' in n_ or n_.startswith(('mu', 'sigma')):
                    p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        model.eval()
        m = cfg['model']
        vl = cases.video_list(m['max_seq_len'], m['input_dim'], m['n_txt_in'], L)
        task_id = 1 if name == "prompt" else -1
        if name == "prompt":
            model.n_known = 22
        model.loss_normalizer = m['train_cfg']['init_loss_norm']
        losses = model(vl, task_id=task_id, is_training=True)
        losses['final_loss'].backward()
        grads = {n_: (p_.grad.clone() if p_.grad is not None else None) for n_, p_ in model.named_parameters()}
        with torch.no_grad():
            res = model([vl[0]], is_training=False)[0]
            raw = model([vl[0]], is_training=False, get_emb=True)
        out = {'overrides': over, 'L': L, 'task_id': task_id, 'n_known': model.n_known,
               'state_dict': {k: v.clone() for k, v in model.state_dict().items()},
               'losses': {k: v.detach().clone() for k, v in losses.items()},
               'loss_normalizer_after': model.loss_normalizer, 'grads': grads,
               'inference': {k: (v.clone() if torch.is_tensor(v) else v) for k, v in res.items()},
               'eval_cls_logits': [x.clone() for x in raw[0]], 'eval_offsets': [x.clone() for x in raw[1]]}
        path = os.path.join(HERE, 'model_%s.pt' % name)
        torch.save(out, path)
        print(name, {k: float(v) for k, v in losses.items()}, 'segs', tuple(res['segments'].shape),
              '%.1f KB' % (os.path.getsize(path) / 1e3))


def nms_goldens(libs):
    import nms_1d_cpu
    from libs.utils import batched_nms

    def case(n, seed, crafted=False):
        g = np.random.RandomState(seed)
        c = g.uniform(0, 200.0, n).astype(np.float32)
        w = g.uniform(0.5, 30, n).astype(np.float32)
        segs = np.stack([c - w / 2, c + w / 2], 1).astype(np.float32).reshape(n, 2)
        scores = g.uniform(0.001, 1, n).astype(np.float32)
        if crafted and n >= 8:                      # duplicates + a chain of overlaps around min_score
            segs[1] = segs[0]
            scores[2] = scores[3]
            segs[4:8] = segs[0] + np.arange(4, dtype=np.float32)[:, None] * 0.5
        return segs, scores
    for n in (0, 1, 8, 257, 5000):
        segs, scores = case(n, 100 + n)
        ts, tc = torch.from_numpy(segs).reshape(-1, 2), torch.from_numpy(scores)
        inds = nms_1d_cpu.nms(ts, tc, 0.5).numpy()
        np.savez_compressed(os.path.join(HERE, 'nms_hard_%d.npz' % n), kind='hard', segs=segs, scores=scores, thr=0.5, inds=inds)
        for sigma, ms in ((0.5, 0.001), (0.75, 0.01), (0.99, 0.2)):
            if n == 5000 and sigma != 0.75:
                continue
            s2, c2 = case(n, 200 + n, crafted=True)
            dets = torch.zeros(n, 3)
            inds = nms_1d_cpu.softnms(torch.from_numpy(s2).reshape(-1, 2), torch.from_numpy(c2), dets, 0.1, sigma, ms, 2).numpy()
            np.savez_compressed(os.path.join(HERE, 'nms_soft_%d_s%02d.npz' % (n, int(sigma * 100))), kind='soft', segs=s2,
                                scores=c2, thr=0.1, sigma=sigma, min_score=ms, inds=inds, dets=dets.numpy())
    for soft in (True, False):
        n = 3000
        segs, scores = case(n, 7)
        cls = np.random.RandomState(8).randint(0, 22, n).astype(np.int64)
        s, sc, c = batched_nms(torch.from_numpy(segs), torch.from_numpy(scores), torch.from_numpy(cls), 0.1, 0.01, 200,
                               use_soft_nms=soft, multiclass=True, sigma=0.75, voting_thresh=0.0)
        np.savez_compressed(os.path.join(HERE, 'nms_batched_%s.npz' % ('soft' if soft else 'hard')), kind='batched', segs=segs,
                            scores=scores, cls=cls, thr=0.1, min_score=0.01, max_seg_num=200, soft=soft, sigma=0.75,
                            out_segs=s.numpy(), out_scores=sc.numpy(), out_cls=c.numpy())
    print('nms goldens written')


if __name__ == "__main__":
    libs = ref_import.setup(extra_xlnet=((32, 4), (64, 4), (128, 4), (2304, 16)))
    nms_goldens(libs)
    model_goldens(libs)
