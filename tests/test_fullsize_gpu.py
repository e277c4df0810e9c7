"""Full-size (BASELINE configs[1], "P": T=2304, Cin=2304, D=1024, H=16, XLNet on) checks through
size-independent properties -- the oracle is too slow at this size for direct comparison in a test."""
import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model_and_batch(dev):
    import bench
    import vilco_amd.modeling as vm
    cfg = bench.p_config(dropout=0.0, droppath=0.0)      # the properties below need a deterministic step
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(0.0))).to(dev).train()
    return model, bench.synth_batch(2, dev), cfg


def test_p_config_step_properties(dev):
    """(1) fused and materialised attention agree at T=2304 (two independent code paths: flash kernels vs
    batched GEMM + softmax); (2) the loss is independent of batch order; (3) padding frames of the short clip
    do not influence the loss (mask invariance); (4) every used parameter gets a finite gradient and the 107
    never-used tensors (SURVEY.md 7 'Unused parameters') get none."""
    from vilco_amd import ops
    model, batch, cfg = _model_and_batch(dev)

    def run(b):
        model.zero_grad(set_to_none=True)
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        out = model(b, is_training=True)
        out['final_loss'].backward()
        return {k: float(v) for k, v in out.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run(batch)
    assert all(np.isfinite(v) for v in l1.values())
    n_none = sum(1 for p in model.parameters() if p.grad is None)
    assert n_none == 107, n_none
    assert all(torch.isfinite(g).all() for g in g1.values())

    ops.use_flash = False
    try:
        l2, g2 = run(batch)
    finally:
        ops.use_flash = True
    for k in l1:
        assert abs(l1[k] - l2[k]) <= 1e-5 * max(1.0, abs(l2[k])), (k, l1[k], l2[k])
    worst = max(((g1[k] - g2[k]).abs().max() / g2[k].abs().max().clamp_min(1e-7)).item() for k in g1)
    assert worst < 1e-3, worst

    l3, _ = run(batch[::-1])
    for k in ('cls_loss', 'reg_loss', 'final_loss'):
        assert abs(l1[k] - l3[k]) <= 2e-5 * max(1.0, abs(l1[k])), (k, l1[k], l3[k])

    # clip 1 has 2287 valid frames: trimming nothing but changing what lies beyond the mask must not matter.
    # (feats are [C, t]; preprocessing pads with zeros -- replace the pad region by garbage via a longer clip)
    b4 = [dict(d) for d in batch]
    extra = torch.randn(b4[1]['feats'].shape[0], 17, device=dev) * 50
    long = torch.cat([b4[1]['feats'], extra], dim=1)
    # same tensor, but tell the model only the first 2287 frames are valid by slicing a view of the long one
    b4[1]['feats'] = long[:, :2287].contiguous()
    l4, _ = run(b4)
    assert abs(l1['final_loss'] - l4['final_loss']) <= 1e-6 * max(1.0, abs(l1['final_loss']))


def test_w_config_step_properties(dev):
    """VERDICT r04 (weak 3): config W (P with D = 2304: hd = 144, no XLNet layer, stem[0] applied twice; bench.py side line) had
    parity only at hd = 144 / D = 288.  The size-independent properties at FULL size: (1) fused (hd <= 160 kernels) and
    materialised attention agree; (2) batch order does not matter; (3) what lies beyond the mask does not matter; (4) finite
    gradients on every used parameter; (5) the debug-mode range check (ops.range_check = "warn") finds the channel
    attention's outlier-row gradient by itself when the model's static mark is taken off."""
    import warnings
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd import ops
    from vilco_amd.core.config import make_config
    over = dict(dataset=dict(input_dim=2304, num_classes=22, max_seq_len=2304),
                model=dict(embd_dim=2304, fpn_dim=2304, head_dim=2304, n_head=16, backbone_arch=(2, 2, 5), use_abs_pe=True,
                           use_cross_modal=True, n_txt_in=768, max_buffer_len_factor=1.0, use_xl=False),
                train_cfg=dict(init_loss_norm=100, dropout=0.0, droppath=0.0))
    cfg = make_config(**over)['model']
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **cfg).to(dev).train()
    batch = bench.synth_batch(2, dev)

    def run(b):
        model.zero_grad(set_to_none=True)
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        out = model(b, is_training=True)
        out['final_loss'].backward()
        return {k: float(v) for k, v in out.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    ops.relu_log, ops.relu_log_values = [], True          # the LayerNorm -> ReLU outputs of both runs: which side of zero, by how much
    try:
        l1, g1 = run(batch)
        log_fused, ops.relu_log = ops.relu_log, []
        assert all(np.isfinite(v) for v in l1.values())
        assert all(torch.isfinite(g).all() for g in g1.values()) and len(g1) > 300
        ops.use_flash = False
        try:
            l2, g2 = run(batch)
        finally:
            ops.use_flash = True
        log_mat = ops.relu_log
    finally:
        ops.relu_log, ops.relu_log_values = None, False
    # where the two runs took different sides of a ReLU, and how far from zero the pre-activation that did pass was
    assert len(log_fused) == len(log_mat) >= 6
    flips = []
    for i, ((_, ma, ya), (_, mb, yb)) in enumerate(zip(log_fused, log_mat)):
        d = ma != mb
        n = int(d.sum())
        if n:
            flips.append((i, n, float(torch.maximum(ya, yb)[d].max() / torch.maximum(ya.max(), yb.max()).clamp_min(1e-30))))
    del log_fused, log_mat
    print("W fused vs materialised: ReLU sign differences (call, elements, passed value / max):", flips)
    for k in l1:
        assert abs(l1[k] - l2[k]) <= 1e-5 * max(1.0, abs(l2[k])), (k, l1[k], l2[k])
    # Two arithmetics that differ in the last bits (flash vs materialised softmax) can put ONE LayerNorm -> ReLU pre-activation
    # of a head trunk on different sides of zero; the gradient term of that (token, channel) then appears in one run only, and
    # everything upstream of it moves with it.  Measured here (round 5): reg_head.head.0.conv.weight 1.3e-2 of its maximum and,
    # behind it, a uniform ~1e-3 L2 shift of the trunk's gradients (median over the 322 tensors 1.0e-3, largest 2.3e-3) -- the
    # signature of one such flip (tests/test_fullsize_gpu.py::test_p_config_train_step_vs_oracles_over_mask_realisations sees
    # the same event between HIP and the oracle at config P under one of its realisations).  Bounds: 5e-3 in L2 and 5e-2 of
    # the maximum on every tensor.  Round 6: the event is LOCATED, not assumed -- both runs log their LayerNorm -> ReLU outputs
    # (ops.relu_log_values) and the loose bound applies only when elements really changed side, few of them, each a value within
    # 1e-5 of zero (measured: 21 elements in the four head-trunk calls, <= 2.7e-7 of their tensor's maximum); without such an
    # element the bar is 1e-3 in both norms.
    dist = sorted(((((g1[k] - g2[k]).abs().max() / g2[k].abs().max().clamp_min(1e-7)).item(),
                    ((g1[k] - g2[k]).norm() / g2[k].norm().clamp_min(1e-12)).item(), k) for k in g1
                   if not k.endswith(('key_norm.bias', '.key.bias'))), reverse=True)     # (analytically zero: softmax shift)
    print("W fused vs materialised: worst max-norm / L2 distances:", dist[0][:2], max(d[1] for d in dist))
    if flips:
        # (round 6) the loose bound is only available WITH its cause located: a handful of elements changed side, every one of them a
        # value within 1e-5 of zero relative to its tensor's maximum
        assert sum(n for _, n, _ in flips) <= 64 and all(rel <= 1e-5 for _, _, rel in flips), flips
        assert max(d[1] for d in dist) < 5e-3, sorted(dist, key=lambda d: -d[1])[:6]
        assert dist[0][0] < 5e-2, dist[:8]
    else:
        assert max(d[1] for d in dist) < 1e-3 and dist[0][0] < 1e-3, dist[:8]
    del g2
    l3, _ = run(batch[::-1])
    for k in ('cls_loss', 'reg_loss', 'final_loss'):
        assert abs(l1[k] - l3[k]) <= 2e-5 * max(1.0, abs(l1[k])), (k, l1[k], l3[k])
    b4 = [dict(d) for d in batch]
    long = torch.cat([b4[1]['feats'], torch.randn(b4[1]['feats'].shape[0], 17, device=dev) * 50], dim=1)
    b4[1]['feats'] = long[:, :2287].contiguous()
    l4, _ = run(b4)
    assert abs(l1['final_loss'] - l4['final_loss']) <= 1e-6 * max(1.0, abs(l1['final_loss']))
    # (5) the detector against the static mark
    ca = model.backbone.stem[0].channel_attn.attn
    assert ca.wide_range
    ca.wide_range = False
    ops.range_events.clear()
    ops._range_warned.clear()
    old = ops.range_check
    ops.range_check = "warn"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            run(batch)
    finally:
        ops.range_check = old
        ca.wide_range = True
    D = cfg['embd_dim']
    hit = [e for e in ops.range_events if e[1] in (D, 3 * D) and e[2] == D]       # the block's proj / qkv Linears
    assert hit, ops.range_events[:8]


def test_p_config_inference_runs(dev):
    model, batch, cfg = _model_and_batch(dev)
    model.eval()
    with torch.no_grad():
        res = model([batch[0]], is_training=False)
    r = res[0]
    assert r['segments'].shape[0] == r['scores'].shape[0] == r['labels'].shape[0] <= cfg['test_cfg']['max_seg_num']
    assert (r['segments'][:, 1] >= r['segments'][:, 0]).all()
    assert (r['scores'][:-1] >= r['scores'][1:]).all()            # sortedness of the final ranking
    assert (r['segments'] >= 0).all() and (r['segments'] <= batch[0]['duration']).all()


def test_nms_full_size_vs_reference_build(dev):
    """N = 30 000 candidates in one class (the worst case of SURVEY.md 6): indices bit-exact vs the reference
    extension when it is available, idempotence otherwise."""
    from oracle import build_ref
    from vilco_amd.utils.nms import nms_1d_cpu
    g = np.random.RandomState(3)
    n = 30000
    c = g.uniform(0, 2304.0, n).astype(np.float32)
    w = g.uniform(0.5, 200, n).astype(np.float32)
    segs = torch.from_numpy(np.stack([c - w / 2, c + w / 2], 1).astype(np.float32))
    scores = torch.from_numpy(g.uniform(0.001, 1, n).astype(np.float32))
    keep = nms_1d_cpu.nms(segs, scores, 0.5)
    again = nms_1d_cpu.nms(segs[keep].contiguous(), scores[keep].contiguous(), 0.5)
    assert torch.equal(again, torch.arange(keep.numel()))          # NMS of an NMS output keeps everything, in order
    dets = torch.zeros(n, 3)
    sidx = nms_1d_cpu.softnms(segs, scores, dets, 0.1, 0.75, 0.01, 2)
    k = sidx.numel()
    assert (dets[:k - 1, 2] >= dets[1:k, 2] - 1e-7).all() and (dets[:k, 2] >= 0.01 - 1e-7).all()
    ref = build_ref.load_ref()
    if ref is not None:
        assert torch.equal(keep, ref.nms(segs, scores, 0.5))
        rdets = torch.zeros(n, 3)
        assert torch.equal(sidx, ref.softnms(segs, scores, rdets, 0.1, 0.75, 0.01, 2))


def test_p_config_step_with_reference_dropout(dev):
    """the benchmark workload itself (dropout 0.1, droppath 0.1, XLNet dropout 0.1 as in mq_vilco.yaml /
    xlnet_config_1024.json): finite losses and gradients, masks differ between steps, eval mode is deterministic."""
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd import ops
    cfg = bench.p_config()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
    batch = bench.synth_batch(2, dev)
    vals = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        ops.dropout_log = []
        try:
            out = model(batch, is_training=True)
            out['final_loss'].backward()
            sites = [e[0] for e in ops.dropout_log]
        finally:
            ops.dropout_log = None
        assert sites.count('attn_prob') >= 1 and 'xl_pos_emb' in sites and 'proj_drop' in sites and 'mlp_drop' in sites
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        vals.append(float(out['final_loss']))
    assert np.isfinite(vals).all() and vals[0] != vals[1]


def test_single_part_weight_gradients_at_full_size(dev):
    """Weight-gradient products with long contractions (K = B*T >= 2048) multiply only the leading fp16 parts of the
    operand planes (ops.dw_precision, 1 MFMA instead of 3).  At config P every gradient must stay within 5e-4 (max
    abs / max abs) of the three-MFMA result, losses identical (the forward is untouched)."""
    from vilco_amd import ops
    model, batch, cfg = _model_and_batch(dev)

    def run():
        model.zero_grad(set_to_none=True)
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        out = model(batch, is_training=True)
        out['final_loss'].backward()
        return {k: float(v) for k, v in out.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    saved = ops.dw_precision
    try:
        ops.dw_precision = 4
        l1, g1 = run()
        ops.dw_precision = None
        l2, g2 = run()
    finally:
        ops.dw_precision = saved
    assert l1 == l2
    errs = sorted((((g1[k] - g2[k]).abs().max() / g2[k].abs().max().clamp_min(1e-7)).item(), k) for k in g1)
    assert errs[-1][0] < 5e-4, errs[-5:]          # measured: 3.5e-4 worst (branch.0 key.weight), 2e-4 typical
    assert sum(1 for e, _ in errs if e > 0) > 30          # the fast mode is actually in use (48 weight tensors at P)


def test_p_config_train_step_vs_oracles_over_mask_realisations(dev):
    """The benchmark workload itself -- config P, two clips, train mode with dropout 0.1 / stochastic depth 0.1 / XLNet dropout
    0.1, single-part weight-gradient products active -- over THREE realisations of the masks, each compared with the oracle
    (the CPU restatement of the reference) replaying exactly the masks the HIP step drew, in fp32 AND in fp64 (~60 s each).

    (A) THE BAR (BASELINE.json: "within 1e-3 rel fp32"): every gradient tensor within 1e-3 of the REFERENCE ARITHMETIC -- the
        fp32 oracle -- in the max norm (max|g - w| / max|w|) AND in the L2 norm (||g - w|| / ||w||); losses within 1e-5.
        Measured over four realisations (round 6, profiles/r06_p_parity_decisions.json): worst 7.2e-4 / 3.3e-4, no tensor
        beyond 1e-3 in either norm under any of them.
    (B) sanity against exact arithmetic: d(HIP, fp64) <= max(1e-3, 2 d(fp32 oracle, fp64)) per tensor and norm.  (An fp64 run
        is NOT the reference: the step contains discrete decisions -- stride-2 max-pools routing a gradient element to one of
        two near-tied tokens, LayerNorm -> ReLU pre-activations within rounding of zero -- so the reference's own fp32 is
        beyond 1e-3 of an fp64 run of itself on 88-100 (max norm) / 182-191 (L2) of the 355 tensors under every realisation,
        worst 0.3; HIP sits at the same distances to 3-6 digits because it takes the fp32 oracle's side of those decisions.)
    A realisation meeting (A) and (B) on every tensor is "clean".
    (C) a realisation that does not is accepted ONLY WITH A LOCATED CAUSE, never on a looser bound (round 5
        accepted one such realisation of three on 1e-2 / 5e-2).  The HIP step's LayerNorm -> ReLU sign decisions (ops.relu_log:
        embeddings and head trunks, ~50 M elements) are replayed into a second fp32 oracle run (oracle.mq_oracle.ReluReplay:
        y = x * [HIP took the positive side]).  That run reports every element where its own sign differs from HIP's -- 10-15
        sites with 1-5 elements each under every realisation, all of them pre-activations within 6e-7 of their tensor's
        maximum of zero: a coin flip for any arithmetic, and an element that happens to carry a large loss gradient moves
        everything upstream by a discrete amount.  Each such element must be within 1e-5 of zero, and with the decisions
        shared every tensor must meet (A) against that run.  At most one realisation of three may need (C).
    (Seeds fix the realisations; both arithmetics are bit-reproducible given the masks.)"""
    from parity_util import p_step_three_ways, tensor_distance
    from vilco_amd import ops
    assert ops.dw_precision == 4 and ops.get_precision() == 3
    threads = min(64, os.cpu_count() or 1)
    kinds, report = [], []
    zero_by_symmetry = ('key_norm.bias', '.key.bias')                # analytically zero (softmax shift)
    for r in range(3):
        hl, hg, orc = p_step_three_ways(dev, r, threads=threads)
        l32, g32 = orc[torch.float32]
        l64, g64 = orc[torch.float64]
        for k in ('cls_loss', 'reg_loss', 'final_loss'):
            assert abs(hl[k] - l32[k]) <= 1e-5 * abs(l32[k]), (r, k, hl[k], l32[k], l64[k])
            assert abs(hl[k] - l64[k]) <= 1e-5 * abs(l64[k]), (r, k, hl[k], l32[k], l64[k])
        rows = []
        for k, w in g64.items():
            if k.endswith(zero_by_symmetry):
                continue
            dh, dr, dd = tensor_distance(hg[k], w), tensor_distance(g32[k], w), tensor_distance(hg[k], g32[k])
            rows.append((k, dh[0], dr[0], dh[1], dr[1], dd[0], dd[1]))
        assert len(rows) > 300
        direct = [x for x in rows if x[5] > 1e-3 or x[6] > 1e-3]                                           # (A)
        viol = [x for x in rows if x[1] > max(1e-3, 2 * x[2]) or x[3] > max(1e-3, 2 * x[4])]              # (B)
        report.append(dict(realisation=r, tensors=len(rows), beyond_bar_vs_fp32=len(direct), beyond_relative_bar_vs_fp64=len(viol),
                           hip_vs_fp32_worst_max=max(x[5] for x in rows), hip_vs_fp32_worst_l2=max(x[6] for x in rows),
                           hip_beyond_1e3_of_fp64=sum(1 for x in rows if x[1] > 1e-3), fp32_beyond_1e3_of_fp64=sum(1 for x in rows if x[2] > 1e-3)))
        if not direct and not viol:
            kinds.append("clean")
            del hg, g32, g64, orc
            continue
        # ---- (C) not clean: locate the cause or fail
        kinds.append("flip")
        lf, gf, events = orc['rerun'](torch.float32, orc['hip_relu'])
        assert events, ("realisation %d misses the bar without any ReLU sign difference between HIP and the fp32 oracle" % r,
                        sorted(direct + viol, key=lambda x: -x[6])[:6])
        for site, n, rel in events:
            assert rel <= 1e-5, ("a LayerNorm -> ReLU pre-activation that is NOT within rounding of zero changed sign", r, site, n, rel)
        for k in ('cls_loss', 'reg_loss', 'final_loss'):
            assert abs(hl[k] - lf[k]) <= 1e-5 * abs(lf[k]), (r, k, hl[k], lf[k])
        bad = []
        for k, w in gf.items():
            if k.endswith(zero_by_symmetry):
                continue
            d = tensor_distance(hg[k], w)
            if d[0] > 1e-3 or d[1] > 1e-3:
                bad.append((k, d[0], d[1]))
        report[-1].update(relu_sign_events=[(s_, n, rel) for s_, n, rel in events], beyond_bar_after_replay=len(bad))
        assert not bad, (r, events, sorted(bad, key=lambda x: -x[2])[:8])
        del hg, g32, g64, gf, orc
    print("full-size parity report:", report)
    assert kinds.count("clean") >= 2, (kinds, report)


def test_p_config_replayed_steps_equal_eager_steps_on_changing_batches(dev):
    """Round 6: at FULL size every replay of the captured step must produce the gradients of the batch it was handed -- checked
    against eager steps of the same model on the same batches, over replays that alternate between two different batches.  (What
    this pins: on torch 2.10 + ROCm 7 an ATen multi-block reduction captured in a hipGraph writes its result on the first replay
    only -- tools/lab/sum_graph_probe.py -- and up to round 5 XLNet's r_w_bias / r_r_bias gradients, the backward of `q + bias`,
    were such reductions: they stood still in replayed steps at this size and nowhere smaller.  No test compared a replayed
    full-size step with an eager one on a DIFFERENT batch than the captured one.)"""
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd.graph import GraphedStep
    cfg = bench.p_config(dropout=0.0, droppath=0.0)
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(dropout=0.0))).to(dev).train()
    batches = [bench.synth_batch(2, dev, seed=s) for s in (0, 1)]

    def grads():
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    want = []
    for b in batches:
        for p in model.parameters():
            p.grad = None
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        model(b, is_training=True)['final_loss'].backward()
        want.append(grads())
    k = 'backbone.xlnet.layer.0.rel_attn.r_w_bias'
    assert rel(want[0][k], want[1][k]) > 1e-3                    # the two batches really ask for different gradients
    gs = GraphedStep(model, None, eager_steps=1)
    order = [0, 0, 1, 0, 1, 1]                                   # eager, capture + replay, then replays on alternating batches
    for call, bi in enumerate(order):
        model.loss_normalizer = cfg['train_cfg']['init_loss_norm']
        gs(batches[bi])
        torch.cuda.synchronize()
        got = grads()
        assert set(got) == set(want[bi])
        bad = [(n, rel(got[n], want[bi][n])) for n in got if not n.endswith(('key_norm.bias', '.key.bias'))
               and not torch.equal(got[n], want[bi][n]) and rel(got[n], want[bi][n]) > 1e-5]
        assert not bad, (call, bi, sorted(bad, key=lambda x: -x[1])[:6])
    assert gs.stats['replayed'] == len(order) - 1


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_p_config_replayed_training_follows_eager_training(dev):
    """Round 6: 40 TRAINING iterations at full size (forward + backward + clip + AdamW; replayed as two hipGraphs vs launched eagerly)
    from the same initial state over four rotating batches, dropout off: the same loss trajectory -- every loss within 1e-4
    (relative; bit-equal in practice until the float atomics of the loss kernel's mu / sigma gradients have gone through Adam a few
    times) -- and falling.  tools/lab/train_soak.py is the 300-iteration form (profiles/r06_train_soak.txt)."""
    import bench
    import vilco_amd.modeling as vm
    from vilco_amd.graph import GraphedStep
    from vilco_amd.utils.train_utils import make_optimizer
    cfg = bench.p_config(dropout=0.0, droppath=0.0)
    batches = [bench.synth_batch(2, dev, seed=s) for s in range(4)]
    runs = []
    for replay in (True, False):
        torch.manual_seed(0)
        model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet(dropout=0.0))).to(dev).train()
        opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
        gs = GraphedStep(model, opt, clip_grad_l2norm=1.0, eager_steps=1, enabled=replay)
        losses = [gs(batches[it % 4])['final_loss'] for it in range(40)]
        torch.cuda.synchronize()
        assert (gs.stats['replayed'] == 39) if replay else (gs.stats['replayed'] == 0)
        runs.append(torch.stack(losses).float().cpu())
        del model, opt, gs
        torch.cuda.empty_cache()
    a, b = runs
    assert bool(torch.isfinite(a).all()) and float(a[-4:].mean()) < 0.7 * float(a[:4].mean())
    assert float(((a - b).abs() / b.abs()).max()) < 1e-4, ((a - b).abs() / b.abs()).max()
