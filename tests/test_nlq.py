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


# ------------------------------------------------------------------------------------------------------- episode (f-2)
def _gold_episode():
    return torch.load(os.path.join(HERE, "golden", "nlq_episode.pt"), weights_only=False)


def test_param_groups_and_schedule_vs_reference():
    """NLQ's make_optimizer in both modes (default: decay / no-decay / encoder twins; head_backbone_group: four head /
    backbone groups with their own learning rates) against the membership lists, decay and LR values recorded from the
    imported NLQ/libs/utils/train_utils.py, and the per-iteration LR sequence of every group over the first task"""
    import json
    import vilco_amd.modeling_nlq as nlq
    from vilco_amd.utils import train_utils_nlq as tu
    with open(os.path.join(HERE, "golden", "nlq_train_glue.json")) as f:
        glue = json.load(f)
    model = nlq.make_meta_arch('LocPointTransformer', **cases.nlq_model_cfg())
    names = {id(p): n for n, p in model.named_parameters()}
    for mode, (hb, w) in (('default', (False, 1)), ('head_backbone', (True, 0.5))):
        opt = tu.make_optimizer(model, cases.nlq_episode_opt(w), head_backbone_group=hb)
        assert len(opt.param_groups) == len(glue[mode]) == 4
        for g, want in zip(opt.param_groups, glue[mode]):
            assert [names[id(p)] for p in g['params']] == want['names']
            assert g['weight_decay'] == want['weight_decay'] and abs(g['lr'] - want['lr']) <= 1e-12
    opt = tu.make_optimizer(model, cases.nlq_episode_opt(0.5), head_backbone_group=True)
    gold = _gold_episode()['tasks'][0]
    sch = tu.make_scheduler(opt, cases.nlq_episode_opt(0.5), gold['n_batches'])
    for want in gold['lrs']:
        got = [g['lr'] for g in opt.param_groups]
        assert all(abs(a - b) <= 1e-12 + 1e-9 * abs(b) for a, b in zip(got, want)), (got, want)
        sch.step()
    # a parameter no rule classifies: the head / backbone mode refuses it (:199-202), the default mode leaves it out (:208-213)
    model.stray = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(AssertionError):
        tu.param_groups(model, True)
    assert all('stray' not in n for g in tu.param_groups(model, False) for n in g)
    # the "constant" schedule of NLQ's make_scheduler: linear warm-up, then the base rate
    del model.stray
    opt = tu.make_optimizer(model, cases.nlq_episode_opt(1))
    sch = tu.make_scheduler(opt, dict(cases.nlq_episode_opt(1), schedule_type="constant", warmup_epochs=2), 2)
    seq = []
    for _ in range(6):
        seq.append(opt.param_groups[0]['lr'])
        sch.step()
    assert all(abs(a - b) < 1e-12 for a, b in zip(seq, [0.0, 1e-3 / 3, 2e-3 / 3, 1e-3, 1e-3, 1e-3])), seq


class _ValTasks:
    def get_valSet_by_taskNum(self, n):
        return [([[q] for q in list(cases.nlq_episode_data(k).values())[0]], 1) for k in range(n)]


class _Recorder:
    dataset = "ego4d_cl"

    def __init__(self):
        self.calls = []

    def evaluate(self, results, verbose=True):
        import numpy as np
        self.calls.append([dict(r) for r in results])
        return np.array([[cases.nlq_episode_metric(results)]]), ""


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_nlq_episode_reproduces_reference(dev, tmp_path, use_graph):
    """BASELINE configs[3] scaled down: three query-template tasks through vilco_amd.train_cl.run_episodes_nlq (head / backbone
    optimizer groups, warm-up + cosine per iteration, validation records, best-checkpoint files, replay memory) against the
    recording of the imported reference driving its own make_optimizer / train_one_epoch / valid_one_epoch_cl_single_gpu /
    final_validate (tests/golden/make_golden_nlq_episode.py)"""
    import random
    import vilco_amd.modeling_nlq as nlq
    from parity_util import delta_err
    from vilco_amd.train_cl import run_episodes_nlq
    from vilco_amd.utils.cl_stream import InMemoryQILStream
    gold = _gold_episode()
    mcfg = cases.nlq_model_cfg()
    cfg = {'opt': cases.nlq_episode_opt(0.5), 'train_cfg': mcfg['train_cfg'],
           'cl_cfg': dict(mcfg['cl_cfg'], memory_size=cases.NLQ_EP_MEMORY, path_memory='mem.pkl')}
    model = nlq.make_meta_arch('LocPointTransformer', **mcfg)
    model.load_state_dict(gold['init_state'], strict=True)
    model = model.to(dev)
    stream = InMemoryQILStream([cases.nlq_episode_data(j) for j in range(cases.NLQ_EP_TASKS)], batch_size=cases.NLQ_EP_BATCH,
                               shuffle=False)
    rec = _Recorder()
    last_epoch = cfg['opt']['epochs'] + cfg['opt']['warmup_epochs'] - 1

    def on_validate(kind, j, epoch, r1):
        if kind == 'epoch' and epoch == last_epoch:
            random.seed(1000 + j)                        # the memory shuffle that follows (the recording seeds it the same way)

    from torch.optim.lr_scheduler import LRScheduler
    step0 = LRScheduler.step
    lrs = []

    def rec_step(self, *a, **k):                         # (patched on the class: an instance attribute would end up in the
        if self._step_count >= 1:                        # scheduler's state_dict and the checkpoint could not pickle it)
            lrs.append([g['lr'] for g in self.optimizer.param_groups])    # the rates the iteration that just ran was stepped with
        return step0(self, *a, **k)
    LRScheduler.step = rec_step
    try:
        model, opt, sch, log = run_episodes_nlq(cfg, model, stream, _ValTasks(), rec, ckpt_folder=str(tmp_path), ckpt_freq=2,
                                                use_graph=use_graph, on_validate=on_validate)
    finally:
        LRScheduler.step = step0
    assert len(log) == cases.NLQ_EP_TASKS
    init = gold['init_state']
    call = 0
    for j, (entry, want) in enumerate(zip(log, gold['tasks'])):
        assert abs(entry['init_R1'] - want['init_R1']) <= 2e-2 * abs(want['init_R1']), (j, entry['init_R1'], want['init_R1'])
        for e, (hist, wl) in enumerate(zip(entry['history'], want['losses'])):
            assert len(hist) == len(wl)
            for i, (h, w) in enumerate(zip(hist, wl)):
                for k in w:
                    assert abs(float(h[k]) - w[k]) <= 5e-3 * max(abs(w[k]), 1e-3), (j, e, i, k, float(h[k]), w[k])
        assert [ep for ep, _ in entry['R1']] == list(range(len(want['R1'])))
        for (_, a), b in zip(entry['R1'], want['R1']):
            assert abs(a - b) <= 2e-2 * abs(b), (j, a, b)
        assert entry['best_epoch'] == want['best_epoch']
        assert os.path.exists(os.path.join(str(tmp_path), 'Best_task_%02d.pth.tar' % j))
        ck = torch.load(os.path.join(str(tmp_path), 'Best_task_%02d.pth.tar' % j), weights_only=False)
        assert set(ck) == {'epoch', 'state_dict', 'scheduler', 'optimizer', 'current_task', 'reg_params'} and ck['current_task'] == j
        # the task's best state (what every later task starts from) against the reference's, as the error of the UPDATE
        errs = sorted(((delta_err(ck['state_dict'][k].cpu(), init[k], w, init[k]), k) for k, w in want['state'].items()
                       if w.is_floating_point() and not k.endswith(NOISE)), reverse=True)
        assert errs[0][0] < 0.2 and errs[len(errs) // 2][0] < 2e-2, (j, errs[:6], errs[len(errs) // 2])
    # learning rates of all four groups, every optimisation step of the episode
    got_lrs = lrs
    want_lrs = [x for t in gold['tasks'] for x in t['lrs']]
    assert len(got_lrs) == len(want_lrs)
    assert all(abs(a - b) <= 1e-12 + 1e-9 * abs(b) for g, w in zip(got_lrs, want_lrs) for a, b in zip(g, w))
    # replay memory after the last task and the records of the last final validation (the evaluator's input format)
    assert {c: [v['query_id'] for v in vs] for c, vs in model.memory.items()} == gold['tasks'][-1]['memory_ids']
    assert model.n_known == cases.NLQ_EP_TASKS
    got, want = rec.calls[-1], gold['tasks'][-1]['results']
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert set(a) == set(b) == {'query_idx', 'annotation_uid', 'predicted_times', 'clip_uid'}
        assert (a['query_idx'], a['annotation_uid'], a['clip_uid']) == (b['query_idx'], b['annotation_uid'], b['clip_uid'])
        assert len(a['predicted_times']) == len(b['predicted_times']) and all(len(r) == 3 for r in a['predicted_times'])
        assert abs(a['predicted_times'][0][2] - b['predicted_times'][0][2]) <= 5e-2 * abs(b['predicted_times'][0][2])
