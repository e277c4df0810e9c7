"""Golden vectors of the NLQ variant's blocks from the IMPORTED REFERENCE (NLQ/libs/modeling/blocks.py; this container
only).  Run:  python tests/golden/make_golden_nlq.py  ->  tests/golden/nlq_blocks.pt

blocks.py is imported on its own (with its two relative dependencies weight_init.py / adapter.py) through a synthetic
package whose __path__ is the reference directory: NLQ/libs/modeling/__init__.py pulls in backbones.py -> roberta.py,
which needs transformers-4.2x internals this image does not have.  The backbone case therefore drives the reference's
block classes in the order of backbones.py:551-615 (embedding convs, PE, text stem, video stem, branch) -- the arithmetic
is the reference's, the ten lines of orchestration are restated here and in oracle/nlq_oracle.py."""
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402


def ref_blocks():
    pkg = types.ModuleType('nlq_ref')
    pkg.__path__ = ['/root/reference/NLQ/libs/modeling']
    sys.modules['nlq_ref'] = pkg
    return importlib.import_module('nlq_ref.blocks')


def perturb(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if 'drop_path' in n:
                p.copy_(0.5 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() <= 1 or 'norm' in n or n.endswith(('ln1.weight', 'ln2.weight', 'ln3.weight', 'ln1.bias', 'ln2.bias', 'ln3.bias')):
                p.add_(0.05 * torch.randn(p.shape, generator=g))


def run(mod, args, wkey=7):
    outs = mod(*args)
    y = outs[0]
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(wkey))
    (y * w).sum().backward()
    return y.detach().clone(), outs[1].clone(), {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}


def main():
    B = ref_blocks()
    out = {}
    C, H, T, L = cases.NLQ_C, cases.NLQ_H, cases.NLQ_T, cases.NLQ_L
    x, mask, txt, tmask = cases.nlq_inputs()
    # ---- LocalMaskedMHCA, stride 1 and 2
    for stride in (1, 2):
        torch.manual_seed(10 + stride)
        m = B.LocalMaskedMHCA(C, H, window_size=cases.NLQ_WIN, n_qx_stride=stride, n_kv_stride=stride)
        perturb(m, 20 + stride)
        xx = x.clone().requires_grad_(True)
        y, om, grads = run(m, (xx, mask))
        out['local_s%d' % stride] = {'state': {k: v.clone() for k, v in m.state_dict().items()}, 'y': y, 'mask': om, 'grads': grads,
                                    'dx': xx.grad.clone()}
    # ---- TransformerBlock with a local window + cross attention, stride 1 and 2
    for stride in (1, 2):
        torch.manual_seed(30 + stride)
        m = B.TransformerBlock(C, H, n_ds_strides=(stride, stride), mha_win_size=cases.NLQ_WIN, path_pdrop=0.1, use_cross_modal=True).eval()
        perturb(m, 40 + stride)
        xx, tt = x.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        y, om, grads = run(m, (xx, mask, tt, tmask))
        out['block_s%d' % stride] = {'state': {k: v.clone() for k, v in m.state_dict().items()}, 'y': y, 'mask': om, 'grads': grads,
                                    'dx': xx.grad.clone(), 'dtxt': tt.grad.clone()}
    # ---- backbone composition (backbones.py:551-615) out of reference blocks
    cfg = cases.nlq_backbone_cfg()
    torch.manual_seed(50)
    mods = torch.nn.ModuleDict()
    mk = lambda n_in, ks: (torch.nn.ModuleList([B.MaskedConv1D(n_in if i == 0 else C, C, ks, stride=1, padding=ks // 2, bias=False) for i in range(cfg['arch'][0])]),
                           torch.nn.ModuleList([B.LayerNorm(C) for _ in range(cfg['arch'][0])]))
    mods['vid_embd'], mods['vid_embd_norm'] = mk(cases.NLQ_CV, 3)
    mods['txt_embd'], mods['txt_embd_norm'] = mk(cases.NLQ_CT, 1)
    blk = lambda s, win, cross: B.TransformerBlock(C, H, n_ds_strides=(s, s), path_pdrop=0.1, mha_win_size=win, use_cross_modal=cross)
    wins, arch = cfg['mha_win_size'], cfg['arch']
    mods['vid_stem'] = torch.nn.ModuleList([blk(1, wins[0], True) for _ in range(arch[2])])
    mods['txt_stem'] = torch.nn.ModuleList([blk(1, -1, False) for _ in range(arch[1])])
    mods['branch'] = torch.nn.ModuleList([blk(2, wins[1 + i], True) for i in range(arch[3])] + [blk(2, wins[1 + i], False) for i in range(arch[4])])
    mods.eval()
    perturb(mods, 60)
    vid, vmask, t2, t2mask = cases.nlq_backbone_inputs()
    v, vm, q, qm = vid, vmask, t2, t2mask
    for c, n in zip(mods['vid_embd'], mods['vid_embd_norm']):
        v, vm = c(v, vm)
        v = torch.relu(n(v))
    pe = B.get_sinusoid_encoding(cfg['max_len'], C) / (C ** 0.5)
    v = v + pe[:, :, :v.shape[-1]] * vm.to(v.dtype)
    for c, n in zip(mods['txt_embd'], mods['txt_embd_norm']):
        q, qm = c(q, qm)
        q = torch.relu(n(q))
    for b_ in mods['txt_stem']:
        q, qm = b_(q, qm)
    for b_ in mods['vid_stem']:
        v, vm = b_(v, vm, q, qm)
    feats = [v]
    for b_ in mods['branch']:
        v, vm = b_(v, vm, q, qm)
        feats.append(v)
    loss = sum((f * torch.randn(f.shape, generator=torch.Generator().manual_seed(70 + i))).sum() for i, f in enumerate(feats))
    loss.backward()
    out['backbone'] = {'state': {k: v_.clone() for k, v_ in mods.state_dict().items()}, 'feats': [f.detach().clone() for f in feats],
                       'grads': {n: p.grad.clone() for n, p in mods.named_parameters() if p.grad is not None}}
    torch.save(out, os.path.join(HERE, 'nlq_blocks.pt'))
    print({k: tuple(v['y'].shape) if 'y' in v else [tuple(f.shape) for f in v['feats']] for k, v in out.items()},
          '%.1f KB' % (os.path.getsize(os.path.join(HERE, 'nlq_blocks.pt')) / 1e3))


if __name__ == "__main__":
    main()
