"""Golden vectors for the iCaRL / BiC distillation terms of the loss (meta_archs.py:1482-1519, BiC's bias layers :821-836),
from the IMPORTED REFERENCE (this container only): the reference model in deterministic mode (eval() + is_training=True)
with n_known > 0 and cached previous-model outputs, as train_cl.py:226-235 / train_utils.py:333-341 hand them over.
Run:  python tests/golden/make_golden_distill.py  ->  tests/golden/distill.pt

iCaRL gets the list over the batch's clips of per-level arrays (only clip 0's enter, :1505-1506); BiC indexes the list by
pyramid level (:1491), so it is given clip 0's per-level list -- the only form its code can run on."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402

N_KNOWN = 10


def prev_logits(T, levels, ncls, seed):
    """sigmoid-range stand-ins for the previous model's cached outputs: one [T_l, ncls] array per level"""
    g = np.random.RandomState(seed)
    return [g.uniform(0.02, 0.98, (T >> l, ncls)).astype(np.float32) for l in range(levels)]


def main():
    libs = ref_import.setup(extra_xlnet=((32, 4),))
    from libs.modeling import make_meta_arch
    from libs.modeling.meta_archs import BiasLayer
    out = {}
    for name in ('icarl', 'bic'):
        over = cases.overrides(use_xl=False, droppath=0.1, al_loss_weight=0.5, cl=dict(name=name))
        cfg = ref_import.make_cfg(libs, **over)
        torch.manual_seed(4321)
        model = make_meta_arch(cfg['model_name'], **cfg['model'])
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if 'drop_path' in n_:
                    p_.copy_(0.5 + 0.1 * torch.randn(p_.shape, generator=g))
                elif p_.dim() <= 1 or 'norm' in n_ or n_.startswith(('mu', 'sigma')):
                    p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        model.eval()
        model.n_known = N_KNOWN
        m = cfg['model']
        vl = cases.video_list(m['max_seq_len'], m['input_dim'], m['n_txt_in'], 11)
        levels = m['backbone_arch'][-1] + 1
        per_clip = [prev_logits(m['max_seq_len'], levels, cases.NCLS, 50 + b) for b in range(len(vl))]
        extra = {}
        if name == 'bic':
            model.list_splits = [N_KNOWN, cases.NCLS]
            model.list_bias_layers = [BiasLayer(), BiasLayer()]
            with torch.no_grad():
                for i, bl in enumerate(model.list_bias_layers):
                    bl.alpha.fill_(1.0 + 0.1 * (i + 1))
                    bl.beta.fill_(0.05 * (i + 1))
            extra = {'splits': model.list_splits, 'alphas': [float(b.alpha) for b in model.list_bias_layers],
                     'betas': [float(b.beta) for b in model.list_bias_layers]}
            prev = per_clip[0]
        else:
            prev = per_clip
        model.loss_normalizer = m['train_cfg']['init_loss_norm']
        losses = model(vl, task_id=-1, is_training=True, prev_out_cls_logits=prev)
        losses['final_loss'].backward()
        grads = {n_: (p_.grad.clone() if p_.grad is not None else None) for n_, p_ in model.named_parameters()}
        if name == 'bic':
            extra['bias_grads'] = [(b.alpha.grad.clone(), b.beta.grad.clone()) for b in model.list_bias_layers]
        out[name] = dict(overrides=over, L=11, n_known=N_KNOWN, levels=levels, prev_seeds=[50 + b for b in range(len(vl))],
                         state_dict={k: v.clone() for k, v in model.state_dict().items()},
                         losses={k: (v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in losses.items()},
                         grads=grads, **extra)
        print(name, {k: float(v) for k, v in losses.items()})
    path = os.path.join(HERE, 'distill.pt')
    torch.save(out, path)
    print('%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
