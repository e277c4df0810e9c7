"""NLQ variant (BASELINE configs[3] scaled down): sliding-window attention (LocalMaskedMHCA), the NLQ
TransformerBlock and the two-stream backbone.  Goldens: tests/golden/nlq_blocks.pt, recorded from the imported
reference blocks (NLQ/libs/modeling/blocks.py; tests/golden/make_golden_nlq.py).
CPU: the banded-softmax oracle reproduces the reference's Longformer-chunked arithmetic.  GPU: the HIP modules
(flash kernels in mask mode 4) reproduce the goldens, forward and every gradient, and skip the key tiles outside the
window at full size."""
import os

import pytest
import torch

from parity_util import HERE, cases, rel_err

TOL = 1e-3
NOISE = ('key_norm.bias', '.key.bias', 'key.bias')      # analytically zero gradients (a shift of all keys cancels in softmax)


def _gold():
    return torch.load(os.path.join(HERE, "golden", "nlq_blocks.pt"), weights_only=False)


def _w(shape, key):
    return torch.randn(shape, generator=torch.Generator().manual_seed(key))


def _p64(state, pre=''):
    return {pre + k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}


@pytest.mark.parametrize("stride", [1, 2])
def test_oracle_local_attention_and_block_match_reference(stride):
    from oracle import nlq_oracle as N
    g = _gold()
    x, mask, txt, tmask = cases.nlq_inputs()
    c = g['local_s%d' % stride]
    p = _p64(c['state'], 'a.')
    xx = x.double().requires_grad_(True)
    y, om = N.local_mhca(p, 'a.', xx, mask, cases.NLQ_H, stride, cases.NLQ_WIN)
    (y * _w(y.shape, 7).double()).sum().backward()
    assert torch.equal(om, c['mask']) and rel_err(y, c['y']) < 1e-5 and rel_err(xx.grad, c['dx']) < 1e-4
    for k, w in c['grads'].items():
        if not k.endswith(NOISE):
            assert rel_err(p['a.' + k].grad, w, 1e-7) < 1e-4, k
    c = g['block_s%d' % stride]
    p = _p64(c['state'], 'b.')
    xx, tt = x.double().requires_grad_(True), txt.double().requires_grad_(True)
    y, om = N.transformer_block(p, 'b.', xx, mask, cases.NLQ_H, stride, cases.NLQ_WIN, tt, tmask.squeeze(1))
    (y * _w(y.shape, 7).double()).sum().backward()
    assert rel_err(y, c['y']) < 1e-5 and rel_err(xx.grad, c['dx']) < 1e-4 and rel_err(tt.grad, c['dtxt']) < 1e-4
    for k, w in c['grads'].items():
        if not k.endswith(NOISE):
            assert rel_err(p['b.' + k].grad, w, 1e-7) < 1e-4, k


def test_oracle_backbone_matches_reference_composition():
    from oracle import nlq_oracle as N
    c = _gold()['backbone']
    p = _p64(c['state'])
    vid, vmask, txt, tmask = cases.nlq_backbone_inputs()
    feats, masks = N.backbone(p, cases.nlq_backbone_cfg(), vid.double(), vmask, txt.double(), tmask)
    assert len(feats) == len(c['feats'])
    for a, b in zip(feats, c['feats']):
        assert rel_err(a, b) < 1e-5
    sum((f * _w(f.shape, 70 + i).double()).sum() for i, f in enumerate(feats)).backward()
    for k, w in c['grads'].items():
        if not k.endswith(NOISE):
            assert rel_err(p[k].grad, w, 1e-7) < 1e-4, k


def test_registry_and_state_dict_keys():
    import vilco_amd.modeling_nlq as nlq
    bb = nlq.make_backbone('convTransformer', **cases.nlq_backbone_cfg())
    want = _gold()['backbone']['state']
    assert sorted(bb.state_dict().keys()) == sorted(want.keys())
    for k, v in bb.state_dict().items():
        assert v.shape == want[k].shape, k
    bb.load_state_dict(want, strict=True)


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 2])
def test_local_attention_and_block_vs_reference_golden(dev, stride):
    import vilco_amd.modeling_nlq as nlq
    g = _gold()
    x, mask, txt, tmask = cases.nlq_inputs()
    c = g['local_s%d' % stride]
    m = nlq.LocalMaskedMHCA(cases.NLQ_C, cases.NLQ_H, window_size=cases.NLQ_WIN, n_qx_stride=stride, n_kv_stride=stride)
    m.load_state_dict(c['state'])
    m = m.to(dev)
    xx = x.to(dev).requires_grad_(True)
    y, om = m(xx, mask.to(dev))
    (y * _w(y.shape, 7).to(dev)).sum().backward()
    assert torch.equal(om.cpu(), c['mask']) and rel_err(y, c['y']) < TOL and rel_err(xx.grad, c['dx']) < TOL
    for k, w in c['grads'].items():
        if not k.endswith(NOISE):
            assert rel_err(dict(m.named_parameters())[k].grad, w, 1e-7) < TOL, k
    c = g['block_s%d' % stride]
    m = nlq.TransformerBlock(cases.NLQ_C, cases.NLQ_H, n_ds_strides=(stride, stride), mha_win_size=cases.NLQ_WIN, path_pdrop=0.1,
                             use_cross_modal=True)
    m.load_state_dict(c['state'])
    m = m.to(dev).eval()
    xx, tt = x.to(dev).requires_grad_(True), txt.to(dev).requires_grad_(True)
    y, om = m(xx, mask.to(dev), tt, tmask.to(dev))
    (y * _w(y.shape, 7).to(dev)).sum().backward()
    assert rel_err(y, c['y']) < TOL and rel_err(xx.grad, c['dx']) < TOL and rel_err(tt.grad, c['dtxt']) < TOL
    for k, w in c['grads'].items():
        if not k.endswith(NOISE):
            assert rel_err(dict(m.named_parameters())[k].grad, w, 1e-7) < TOL, k


@pytest.mark.gpu
def test_backbone_vs_reference_golden(dev):
    import vilco_amd.modeling_nlq as nlq
    c = _gold()['backbone']
    bb = nlq.make_backbone('convTransformer', **cases.nlq_backbone_cfg())
    bb.load_state_dict(c['state'], strict=True)
    bb = bb.to(dev).eval()
    vid, vmask, txt, tmask = [t.to(dev) for t in cases.nlq_backbone_inputs()]
    feats, masks = bb(vid, vmask, txt, tmask)
    for a, b in zip(feats, c['feats']):
        assert rel_err(a, b) < TOL
    sum((f * _w(f.shape, 70 + i).to(dev)).sum() for i, f in enumerate(feats)).backward()
    params = dict(bb.named_parameters())
    worst = max((rel_err(params[k].grad, w, 1e-7), k) for k, w in c['grads'].items() if not k.endswith(NOISE))
    assert worst[0] < TOL, worst


@pytest.mark.gpu
def test_local_attention_full_size_equals_dense_masked(dev):
    """NLQ sizes (T = 2560, window 19): the tile-skipping kernels = a dense attention under the same band mask, and the
    outputs are independent of everything outside the window (perturbing far-away keys changes nothing)."""
    from vilco_amd import ops
    torch.manual_seed(0)
    B, T, H, hd, w = 2, 2560, 4, 64, 9
    q, k, v = [torch.randn(B, T, H * hd, device=dev, requires_grad=True) for _ in range(3)]
    lens = torch.tensor([T, T - 301], dtype=torch.int32, device=dev)
    o = ops.attention(q, k, v, lens, H, mode=ops.MASK_LOCAL, window=w)
    wt = torch.randn_like(o)
    (o * wt).sum().backward()
    # dense reference in fp64 on one (b, h) slice per clip
    for b in range(B):
        L = int(lens[b])
        for h in (0, H - 1):
            sl = slice(h * hd, (h + 1) * hd)
            qq, kk, vv = [t[b, :, sl].detach().double().cpu() for t in (q, k, v)]
            att = (qq @ kk.t()) / hd ** 0.5
            idx = torch.arange(T)
            ok = ((idx[:, None] - idx[None, :]).abs() <= w) & (idx[None, :] < L)
            att = att.masked_fill(~ok, float('-inf')).softmax(-1)
            want = att @ vv
            assert rel_err(o[b, :L, sl], want[:L]) < 1e-4
    k2 = k.detach().clone()
    k2[:, 1000:1100] += 5.0                                    # keys 1000..1099 are outside every window of queries < 980
    o2 = ops.attention(q.detach(), k2, v.detach(), lens, H, mode=ops.MASK_LOCAL, window=w)
    assert torch.equal(o2[:, :980], o.detach()[:, :980])


# ------------------------------------------------------------------------------------------------------- meta-architecture
# tests/golden/nlq_model.pt: the reference's NLQ LocPointTransformer (heads, label assignment, losses, decode, soft-NMS are
# the imported meta_archs.py / nms.py; tests/golden/make_golden_nlq_model.py) on two query / clip pairs.
def _gold_model():
    return torch.load(os.path.join(HERE, "golden", "nlq_model.pt"), weights_only=False)


def test_oracle_meta_arch_matches_reference():
    from oracle import nlq_oracle as N
    g, cfg = _gold_model(), cases.nlq_model_cfg()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in g['state'].items()}
    losses, norm = N.forward_losses(p, cfg, cases.nlq_model_batch())
    for k, w in g['losses'].items():
        assert abs(float(losses[k]) - w) <= 2e-5 * max(abs(w), 1e-3), (k, float(losses[k]), w)
    assert abs(norm - g['loss_normalizer']) <= 1e-6 * g['loss_normalizer']
    losses['final_loss'].backward()
    worst = max((rel_err(p[k].grad, w, 1e-7), k) for k, w in g['grads'].items() if not k.endswith(NOISE))
    assert worst[0] < 5e-4, worst
    from vilco_amd.utils.nms import batched_nms            # the reference's post-processing call (:1361) on the oracle's decode
    tc = cfg['test_cfg']
    for x, e in zip(cases.nlq_model_batch(), g['eval']):
        with torch.no_grad():
            masks, cls, reg = N.forward_network({k: v.detach() for k, v in p.items()}, cfg, [x], False)
        for a, b in zip(cls, e['cls_logits']):
            assert rel_err(a, b) < 1e-5
        for a, b in zip(reg, e['offsets']):
            assert rel_err(a, b, 1e-6) < 1e-5
        segs, scores, labels = N.decode(cfg, masks, cls, reg)
        assert segs.shape[0] > tc['max_seg_num']             # the case's scores survive the pre-NMS threshold


def test_meta_arch_registry_and_state_dict_keys():
    import vilco_amd.modeling_nlq as nlq
    m = nlq.make_meta_arch('LocPointTransformer', **cases.nlq_model_cfg())
    want = _gold_model()['state']
    assert sorted(m.state_dict().keys()) == sorted(want.keys())
    for k, v in m.state_dict().items():
        assert v.shape == want[k].shape, k
    m.load_state_dict(want, strict=True)


@pytest.mark.gpu
def test_meta_arch_training_step_and_inference_vs_reference_golden(dev):
    """losses, every parameter gradient, raw head outputs and the decoded + soft-NMS'd moments of the HIP model"""
    import vilco_amd.modeling_nlq as nlq
    g, cfg = _gold_model(), cases.nlq_model_cfg()
    model = nlq.make_meta_arch('LocPointTransformer', **cfg)
    model.load_state_dict(g['state'], strict=True)
    model = model.to(dev).train()
    losses = model([dict(x) for x in cases.nlq_model_batch()], is_training=True)
    for k, w in g['losses'].items():
        assert abs(float(losses[k]) - w) <= TOL * max(abs(w), 1e-3), (k, float(losses[k]), w)
    assert abs(model.loss_normalizer - g['loss_normalizer']) <= 1e-5 * g['loss_normalizer']
    losses['final_loss'].backward()
    params = dict(model.named_parameters())
    assert set(params) == set(g['state']) - {k for k in g['state'] if k not in params}
    worst = max((rel_err(params[k].grad, w, 1e-7), k) for k, w in g['grads'].items() if not k.endswith(NOISE))
    assert worst[0] < TOL, worst
    model.eval()
    for x, e in zip(cases.nlq_model_batch(), g['eval']):
        with torch.no_grad():
            cls, off, masks = model([dict(x)], is_training=False, get_emb=True)
            res = model([dict(x)], is_training=False)[0]
        for a, b in zip(cls, e['cls_logits']):
            assert rel_err(a, b) < TOL
        for a, b in zip(off, e['offsets']):
            assert rel_err(a, b, 1e-6) < TOL
        for a, b in zip(masks, e['masks']):
            assert torch.equal(a.cpu(), b)
        assert res['segments'].shape == e['segments'].shape
        assert rel_err(res['scores'], e['scores']) < TOL and rel_err(res['segments'], e['segments']) < TOL
        assert torch.equal(res['labels'].cpu(), e['labels'])


@pytest.mark.gpu
def test_meta_arch_training_replayed_as_hipgraphs(dev):
    """BASELINE configs[3]'s model under vilco_amd.graph.GraphedStep: the replayed training iterations (sliding-window
    attention, two-stream backbone, fused labels / losses, AdamW) equal the eager ones"""
    import vilco_amd.modeling_nlq as nlq
    from vilco_amd.graph import GraphedStep
    from vilco_amd.utils.train_utils import make_optimizer
    g, cfg = _gold_model(), cases.nlq_model_cfg()
    runs = []
    for use_graph in (False, True):
        model = nlq.make_meta_arch('LocPointTransformer', **cfg)
        model.load_state_dict(g['state'], strict=True)
        model = model.to(dev).train()
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-3))
        step = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=1, enabled=use_graph)
        batch = [dict(x) for x in cases.nlq_model_batch()]
        losses = [float(step(batch)['final_loss']) for _ in range(4)]
        if use_graph:
            assert step.stats['replayed'] == 3 and step.stats['captured'] == 1, step.stats
        runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (la, sa), (lb, sb) = runs
    assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(la, lb)), (la, lb)
    assert la[-1] < la[0]                                       # it trains
    for k in sa:
        if sa[k].is_floating_point() and not k.endswith(NOISE):
            assert torch.equal(sa[k], sb[k]) or rel_err(sb[k], sa[k], 1e-7) < 1e-5, k
