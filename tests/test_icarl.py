"""iCaRL's nearest-exemplar-mean classification at final validation (MQ/libs/modeling/meta_archs.py:1061-1131 `classify`,
:1561-1562, :1626-1643) against tests/golden/icarl.pt, recorded from the imported reference on the only pyramid depth its
hard-coded `fpn_levels = 10` can run (tests/golden/make_golden_icarl.py).
CPU: the oracle restatement.  GPU: the HIP model's `classify` and the two inference calls of the recording."""
import os

import numpy as np
import pytest
import torch

from parity_util import HERE, cases, rel_err


class Stub:
    def get_dataloader(self, data, batch_size=1, memory=None, sample_frame=False):
        assert sample_frame is True
        return [[v] for vs in data.values() for v in vs]


def _gold():
    return torch.load(os.path.join(HERE, "golden", "icarl.pt"), weights_only=False)


def _cfg(g):
    from vilco_amd.core.config import make_config
    return make_config(**g['overrides'])['model']


def test_oracle_icarl_matches_reference():
    from oracle import mq_oracle as O, nms_oracle
    g = _gold()
    cfg, p = _cfg(g), g['state_dict']
    means = O.icarl_exemplar_means(p, cfg, cases.icarl_memory(), lambda d: Stub().get_dataloader(d, sample_frame=True))
    assert len(means) == 10
    for m, want in zip(means, g['means']):
        assert rel_err(m, torch.stack(want, 0)) < 2e-5
    x = cases.icarl_clip(100)
    dists = O.icarl_dists(p, cfg, means, x)
    for d, want in zip(dists, g['dists']):
        assert d.shape == want.shape and rel_err(d, want) < 2e-5
    # decode from the RECORDED distances (ranking by distance: an oracle-vs-reference rounding difference could swap ties)
    _, masks, cls, reg, _ = O.forward_network(p, cfg, [x], training=False)
    pts = O.points(cfg, [c.shape[1] for c in cls])
    segs, scores, labels = O.decode_single_video_icarl(cfg, pts, [m[0, 0] for m in masks], [c[0] for c in cls], [r[0] for r in reg],
                                                       g['dists'])
    tc = cfg['test_cfg']
    s, sc, lab = nms_oracle.batched_nms(segs.numpy(), scores.numpy(), labels.numpy(), tc['iou_threshold'], tc['min_score'],
                                        tc['max_seg_num'], tc['nms_method'] == 'soft', tc['multiclass_nms'], tc['nms_sigma'],
                                        tc['voting_thresh'])
    s = (s * x['feat_stride'] + 0.5 * x['feat_num_frames']) / x['fps']
    s = np.where(s <= 0.0, 0.0, s)
    s = np.where(s >= x['duration'], x['duration'], s)
    inf = g['inference']
    assert np.array_equal(lab, inf['labels'].numpy())
    np.testing.assert_allclose(sc, inf['scores'].numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(s, inf['segments'].numpy(), rtol=2e-4, atol=1e-5)


@pytest.mark.gpu
def test_hip_icarl_classify_and_final_validation(dev):
    import vilco_amd.modeling as vm
    g = _gold()
    cfg = _cfg(g)
    model = vm.make_meta_arch('LocPointTransformer', **cfg)
    model.load_state_dict(g['state_dict'])
    model = model.to(dev).eval()
    model.memory = cases.icarl_memory()
    assert model.compute_means is True and model.cl_name == 'icarl'
    x = cases.icarl_clip(100)
    dists = model.classify(x, Stub())
    assert model.compute_means is False and len(dists) == 10
    for lvl, (m, want) in enumerate(zip(model.exemplar_means, g['means'])):
        assert rel_err(m.permute(0, 2, 1), torch.stack(want, 0)) < 1e-3, lvl          # ours: [n_cls, T_l, C]
    for lvl, (d, want) in enumerate(zip(dists, g['dists'])):
        assert tuple(d.shape) == tuple(want.shape) and rel_err(d, want) < 1e-3, lvl
    # final validation: the first clip is decoded from class distances, the next one the ordinary way (compute_means is
    # cleared by the first classify call, meta_archs.py:1098)
    model.compute_means = True
    res = model([x], is_training=False, val_qilDatasetList=Stub())[0]
    res2 = model([cases.icarl_clip(101)], is_training=False, val_qilDatasetList=Stub())[0]
    for got, want in ((res, g['inference']), (res2, g['inference_after'])):
        assert got['segments'].shape == want['segments'].shape
        assert rel_err(got['scores'], want['scores']) < 5e-3
        assert (got['labels'] == want['labels']).float().mean().item() >= 0.9           # near-tied distances / scores swap neighbours
        same = got['labels'] == want['labels']
        assert rel_err(got['segments'][same], want['segments'][same]) < 5e-3
