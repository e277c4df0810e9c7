"""Golden vectors for the continual-learning side functions, from the IMPORTED REFERENCE (this container only):
  * EWC / MAS penalties  (libs/cl_methods/EWC.py:6-22, MAS.py:5-21)  and the consolidation passes (:24-56 / :23-57)
  * the narration-SSL InfoNCE loss  PtTransformer.masked_contrastive_loss (meta_archs.py:1351-1372), called unbound
    on a stub holding a memory bank (its `.cuda()` calls are made no-ops: there is no GPU in this container)
Run:  python tests/golden/make_golden_cl.py  ->  tests/golden/cl_parts.pt"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402


def main():
    libs = ref_import.setup(extra_xlnet=((32, 4),))
    from libs.cl_methods.EWC import get_regularized_loss, on_task_update
    from libs.cl_methods.MAS import get_mas_regularized_loss, on_task_mas_update
    from libs.modeling.meta_archs import PtTransformer
    out = {}

    # ---- penalties: two consolidated tasks; `head.weight` grew from 4 to 7 rows after the first; a 'scale' is skipped
    for kind in ('ewc', 'mas'):
        model = cases.RegToy()
        g = torch.Generator().manual_seed(11)
        key = 'fisher' if kind == 'ewc' else 'importance'
        reg = {key: [], 'optpar': []}
        for t, rows in enumerate((4, 7)):
            imp, opt = {}, {}
            for n, p in model.named_parameters():
                if n == 'unused':
                    continue
                if p.dim() == 0:
                    imp[n], opt[n] = torch.rand((), generator=g), p.data.clone() + 0.1
                    continue
                shape = (rows,) + tuple(p.shape[1:]) if n.startswith('head') else tuple(p.shape)
                imp[n] = torch.rand(shape, generator=g)
                opt[n] = p.data[:shape[0]].clone() + 0.1 * torch.randn(shape, generator=g)
            reg[key].append(imp)
            reg['optpar'].append(opt)
        model.reg_params = reg
        base = cases.reg_toy_loss(model)
        fn = get_regularized_loss if kind == 'ewc' else get_mas_regularized_loss
        loss = fn(base, model, 0.37)
        loss.backward()
        out[kind] = {'reg_params': reg, 'base_loss': float(base), 'loss': float(loss),
                     'grads': {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                     'state': {k: v.clone() for k, v in model.state_dict().items()}}

        # consolidation pass over a 3-batch loader: keeps the LAST batch's gradient (zero_grad before every batch)
        model2 = cases.RegToy()
        model2.reg_params = {}
        opt2 = torch.optim.SGD(model2.parameters(), lr=0.1)
        up = on_task_update if kind == 'ewc' else on_task_mas_update
        reg2 = up(cases.reg_toy_loader(), 'cpu', opt2, model2)
        out[kind + '_update'] = {k: [{n: v.clone() for n, v in d.items()} for d in lst] for k, lst in reg2.items()}

    # ---- InfoNCE against a memory bank
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(5)
    text = torch.nn.functional.normalize(torch.randn(6, 16, generator=g), dim=1).requires_grad_(True)
    video = torch.nn.functional.normalize(torch.randn(6, 16, generator=g), dim=1).requires_grad_(True)
    mask = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.bool)
    mem = torch.randn(10, 16, generator=g)
    stub = types.SimpleNamespace(memory_bank=types.SimpleNamespace(get_all=lambda: mem))
    loss = PtTransformer.masked_contrastive_loss(stub, text, video, mask)
    loss.backward()
    out['ssl'] = {'text': text.detach().clone(), 'video': video.detach().clone(), 'mask': mask, 'memory': mem,
                  'loss': float(loss), 'dtext': text.grad.clone(), 'dvideo': video.grad.clone()}
    torch.save(out, os.path.join(HERE, 'cl_parts.pt'))
    print({k: (v.get('loss') if isinstance(v, dict) else None) for k, v in out.items()})


if __name__ == "__main__":
    main()
