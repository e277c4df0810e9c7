"""Golden vectors of the NLQ meta-architecture from the IMPORTED REFERENCE (NLQ/libs/modeling/meta_archs.py; this
container only).  Run:  python tests/golden/make_golden_nlq_model.py  ->  tests/golden/nlq_model.pt

meta_archs.py (heads, label assignment, losses, decode, post-processing), necks.py, loc_generators.py, losses.py,
blocks.py and libs/utils/nms.py are the reference's own files, imported through synthetic packages whose __init__ is
skipped: NLQ/libs/modeling/__init__.py pulls in backbones.py -> roberta.py, which needs transformers internals this
image does not have.  The backbone registered under 'convTransformer' is therefore composed of the reference's block
classes in the order of backbones.py:480-615 (the same composition tests/golden/make_golden_nlq.py pins the HIP backbone
against); parameter names follow backbones.py.  Two host-only shims: `PtTransformer.device` (hard-wired to cuda:0,
meta_archs.py:563-567) returns the CPU, and timm's ModelEmaV2 comes from tests/golden/_shims."""
import importlib
import os
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))       # nms_1d_cpu.so: the reference's own extension, built here
sys.path.insert(0, REPO)
import cases  # noqa: E402

REF = '/root/reference/NLQ/libs'


def ref_modules():
    from oracle import build_ref
    assert build_ref.build() is not None
    for name, path in (('nlq_libs', REF), ('nlq_libs.modeling', REF + '/modeling'), ('nlq_libs.utils', REF + '/utils')):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg
    nms = importlib.import_module('nlq_libs.utils.nms')
    sys.modules['nlq_libs.utils'].batched_nms = nms.batched_nms
    B = importlib.import_module('nlq_libs.modeling.blocks')
    models = importlib.import_module('nlq_libs.modeling.models')
    importlib.import_module('nlq_libs.modeling.necks')
    importlib.import_module('nlq_libs.modeling.loc_generators')

    @models.register_backbone('convTransformer')
    class ComposedBackbone(nn.Module):
        """backbones.py:480-615 out of the reference's blocks (see the module docstring)"""

        def __init__(self, n_vid_in, n_txt_in, n_embd, n_head, n_embd_ks, max_len, arch, mha_win_size, scale_factor, with_ln,
                     attn_pdrop, proj_pdrop, path_pdrop, use_abs_pe, use_rel_pe, use_adapter):
            super().__init__()
            assert with_ln and use_abs_pe and not use_rel_pe and not use_adapter
            C = n_embd
            mk = lambda n_in, ks: (nn.ModuleList([B.MaskedConv1D(n_in if i == 0 else C, C, ks, stride=1, padding=ks // 2, bias=False)
                                                  for i in range(arch[0])]), nn.ModuleList([B.LayerNorm(C) for _ in range(arch[0])]))
            self.vid_embd, self.vid_embd_norm = mk(n_vid_in, n_embd_ks)
            self.txt_embd, self.txt_embd_norm = mk(n_txt_in, 1)
            blk = lambda s, win, cross: B.TransformerBlock(C, n_head, n_ds_strides=(s, s), attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                                           path_pdrop=path_pdrop, mha_win_size=win, use_cross_modal=cross)
            self.vid_stem = nn.ModuleList([blk(1, mha_win_size[0], True) for _ in range(arch[2])])
            self.txt_stem = nn.ModuleList([blk(1, -1, False) for _ in range(arch[1])])
            self.branch = nn.ModuleList([blk(2, mha_win_size[1 + i], True) for i in range(arch[3])] +
                                        [blk(2, mha_win_size[1 + i], False) for i in range(arch[4])])
            self.register_buffer("pos_embd", B.get_sinusoid_encoding(max_len, C) / (C ** 0.5), persistent=False)

        def forward(self, v, vm, q, qm):
            for c, n in zip(self.vid_embd, self.vid_embd_norm):
                v, vm = c(v, vm)
                v = torch.relu(n(v))
            v = v + self.pos_embd[:, :, :v.shape[-1]] * vm.to(v.dtype)
            for c, n in zip(self.txt_embd, self.txt_embd_norm):
                q, qm = c(q, qm)
                q = torch.relu(n(q))
            for b_ in self.txt_stem:
                q, qm = b_(q, qm)
            for b_ in self.vid_stem:
                v, vm = b_(v, vm, q, qm)
            feats, masks = (v,), (vm,)
            for b_ in self.branch:
                v, vm = b_(v, vm, q, qm)
                feats += (v,)
                masks += (vm,)
            return feats, masks

    clm = types.ModuleType('nlq_libs.cl_methods')
    clm.__path__ = [REF + '/cl_methods']
    sys.modules['nlq_libs.cl_methods'] = clm
    clm.Prompt = importlib.import_module('nlq_libs.cl_methods.prompt').Prompt
    M = importlib.import_module('nlq_libs.modeling.meta_archs')
    M.PtTransformer.device = property(lambda self: torch.device('cpu'))
    return M


def perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if 'drop_path' in n:
                p.copy_(0.5 + 0.1 * torch.randn(p.shape, generator=g))
            elif 'cls_head.cls_head.conv.bias' in n:
                p.add_(1.5 + 0.2 * torch.randn(p.shape, generator=g))        # scores that survive the 1e-3 threshold
            elif p.dim() <= 1 or 'norm' in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g))


def main():
    M = ref_modules()
    torch.manual_seed(77)
    model = M.PtTransformer(**cases.nlq_model_cfg())
    perturb(model, 78)
    out = {'state': {k: v.clone() for k, v in model.state_dict().items()}}
    batch = cases.nlq_model_batch()
    # training step (dropout / droppath are 0 in the case's train_cfg: train mode is deterministic)
    model.train()
    losses = model([dict(x) for x in batch], is_training=True)
    losses['final_loss'].backward()
    out['losses'] = {k: float(v) for k, v in losses.items()}
    out['grads'] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out['loss_normalizer'] = float(model.loss_normalizer)
    # inference (one query at a time, :935): raw head outputs and the decoded + soft-NMS'd moments
    model.eval()
    out['eval'] = []
    for x in batch:
        with torch.no_grad():
            cls, off, masks = model([dict(x)], is_training=False, get_emb=True)
            res = model([dict(x)], is_training=False)[0]
        out['eval'].append({'cls_logits': [c.clone() for c in cls], 'offsets': [o.clone() for o in off], 'masks': [m.clone() for m in masks],
                            'segments': res['segments'].clone(), 'scores': res['scores'].clone(), 'labels': res['labels'].clone()})
    torch.save(out, os.path.join(HERE, 'nlq_model.pt'))
    print(out['losses'], len(out['grads']), [tuple(e['segments'].shape) for e in out['eval']],
          '%.1f KB' % (os.path.getsize(os.path.join(HERE, 'nlq_model.pt')) / 1e3))


if __name__ == "__main__":
    main()
