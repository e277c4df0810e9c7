"""Golden vectors for iCaRL's nearest-exemplar-mean classification at final validation (meta_archs.py:1061-1131
`classify`, its use in `inference` :1561-1562 and `inference_single_video` :1626-1643), from the IMPORTED REFERENCE.
Run:  python tests/golden/make_golden_icarl.py  ->  tests/golden/icarl.pt

`classify` hard-codes fpn_levels = 10 (:1065), so the reference can only run it on a 10-level pyramid: the case uses
backbone_arch (2, 2, 9) at max_seq_len 1024 (levels 1024 ... 2; a level of length 1 breaks its squeeze(), :1096).  The class-distance rows are matched with the class
logits by flat index (:1630-1633), which needs one exemplar class per output class: 4 classes, 2 exemplars each."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402


class Stub:
    """QILSetTask.get_dataloader(data_class, sample_frame=True) (cl_benchmark.py:128): batches of one clip"""

    def get_dataloader(self, data, batch_size=1, memory=None, sample_frame=False):
        return [[v] for vs in data.values() for v in vs]


def main():
    libs = ref_import.setup(extra_xlnet=((32, 4),))
    from libs.modeling import make_meta_arch
    over = cases.icarl_overrides()
    cfg = ref_import.make_cfg(libs, **over)
    torch.manual_seed(2468)
    model = make_meta_arch(cfg['model_name'], **cfg['model'])
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if 'drop_path' in n_:
                p_.copy_(0.5 + 0.1 * torch.randn(p_.shape, generator=g))
            elif p_.dim() <= 1 or 'norm' in n_ or n_.startswith(('mu', 'sigma')):
                p_.add_(0.05 * torch.randn(p_.shape, generator=g))
        model.cls_head.cls_head.conv.bias.add_(3.0)        # enough candidates above / below the distance threshold
    model.eval()
    model.memory = cases.icarl_memory()
    x = cases.icarl_clip(100)
    model.compute_means = True
    with torch.no_grad():
        dists = model.classify(x, Stub())
    means = [[m.clone() for m in lvl] for lvl in model.exemplar_means[:len(dists)]]
    model.compute_means = True
    with torch.no_grad():
        res = model([x], is_training=False, val_qilDatasetList=Stub())[0]
    assert model.compute_means is False
    with torch.no_grad():
        res2 = model([cases.icarl_clip(101)], is_training=False, val_qilDatasetList=Stub())[0]      # means computed: plain decode
    out = dict(overrides=over, state_dict={k: v.clone() for k, v in model.state_dict().items()},
               dists=[d.clone() for d in dists], means=means,
               inference={k: (v.clone() if torch.is_tensor(v) else v) for k, v in res.items()},
               inference_after={k: (v.clone() if torch.is_tensor(v) else v) for k, v in res2.items()})
    path = os.path.join(HERE, 'icarl.pt')
    torch.save(out, path)
    print('levels', len(dists), [tuple(d.shape) for d in dists], 'segs', tuple(res['segments'].shape), tuple(res2['segments'].shape),
          '%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
