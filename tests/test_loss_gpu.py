"""ops.mq_loss (vilco_mq_loss_fwd / _bwd: point labelling + focal / DIoU / al losses + loss_normalizer EMA in two
launches each way) against the fp64 oracle restatement of meta_archs.py:1253-1344 / 1374-1447 (oracle.label_points_single,
oracle.losses -- themselves pinned to the reference goldens through tests/test_oracle_model.py)."""
import pytest
import torch

from parity_util import rel_err

pytestmark = pytest.mark.gpu


def _cfg(ncls, T, radius=1.5, smoothing=0.0, loss_weight=1.0, al=0.2):
    from vilco_amd.core.config import make_config
    return make_config(dataset=dict(input_dim=8, num_classes=ncls, max_seq_len=T),
                       model=dict(embd_dim=8, fpn_dim=8, head_dim=8, n_head=2, backbone_arch=(2, 2, 5), use_xl=False),
                       train_cfg=dict(init_loss_norm=100, loss_weight=loss_weight, al_loss_weight=al, label_smoothing=smoothing,
                                      center_sample='radius' if radius > 0 else 'none', center_sample_radius=max(radius, 0.1)))['model']


def _case(dev, ncls, T, gts, lens, gaps, radius=1.5, smoothing=0.0, seed=0, al=0.2, loss_weight=1.0):
    from oracle import mq_oracle
    from vilco_amd import ops
    torch.manual_seed(seed)
    cfg = _cfg(ncls, T, radius, smoothing, loss_weight, al)
    L = 6
    Ts = [T >> l for l in range(L)]
    B = len(gts)
    pts = mq_oracle.points(cfg, Ts, torch.float64)
    # per-level validity from the clip lengths (stride-2 pyramid: ceil)
    lvl_len = torch.tensor([[-(-ln // (1 << l)) for l in range(L)] for ln in lens], dtype=torch.int32)
    masks = [(torch.arange(Ts[l])[None, :] < lvl_len[:, l:l + 1]).unsqueeze(1) for l in range(L)]
    logits = [2.0 * torch.randn(B, Ts[l], ncls, dtype=torch.float64) - 2.0 for l in range(L)]
    raw = [torch.randn(B, Ts[l], 2, dtype=torch.float64) * 3 for l in range(L)]
    scales = torch.rand(L, dtype=torch.float64) + 0.5
    p = {'mu': 0.2 * torch.randn(ncls, 1, dtype=torch.float64), 'sigma': 1 + 0.2 * torch.rand(ncls, 1, dtype=torch.float64),
         'mu_reg_left': -0.5 + 0.1 * torch.randn(ncls, 1, dtype=torch.float64), 'sigma_reg_left': 1 + 0.2 * torch.rand(ncls, 1, dtype=torch.float64),
         'mu_reg_right': 0.5 + 0.1 * torch.randn(ncls, 1, dtype=torch.float64), 'sigma_reg_right': 1 + 0.2 * torch.rand(ncls, 1, dtype=torch.float64)}
    for t in logits + raw + [scales] + list(p.values()):
        t.requires_grad_(True)
    offs = [torch.relu(r * scales[l]) for l, r in enumerate(raw)]
    segs = [torch.tensor(g[0], dtype=torch.float64).reshape(-1, 2) for g in gts]
    labs = [torch.tensor(g[1], dtype=torch.long) for g in gts]
    want, want_norm = mq_oracle.losses(p, cfg, masks, logits, offs, segs, labs, 100.0)
    gw = torch.tensor([0.3, -0.2, 0.5, 1.0], dtype=torch.float64)        # all four outputs carry upstream gradient
    (gw[0] * want['cls_loss'] + gw[1] * want['reg_loss'] + gw[2] * want['al_loss'].sum() + gw[3] * want['final_loss'].sum()).backward()

    # device side: LevelCat-style rows with separator rows when `gaps`
    rows_l, rows_o, tab_p, tab_l, tab_q = [], [], [], [], []
    for l in range(L):
        if l and gaps:
            rows_l.append(torch.zeros(B, 1, ncls, dtype=torch.float64)); rows_o.append(torch.zeros(B, 1, 2, dtype=torch.float64))
            tab_p.append(torch.zeros(1, 4, dtype=torch.float64)); tab_l.append(l); tab_q.append(1 << 30)
        rows_l.append(logits[l].detach()); rows_o.append(raw[l].detach()); tab_p.append(pts[l])
        tab_l += [l] * Ts[l]; tab_q += list(range(Ts[l]))
    lg = torch.cat(rows_l, 1).float().to(dev).requires_grad_(True)
    of = torch.cat(rows_o, 1).float().to(dev).requires_grad_(True)
    sc = scales.detach().float().to(dev).requires_grad_(True)
    gauss = torch.cat([p[k].detach() for k in ('mu', 'sigma', 'mu_reg_left', 'sigma_reg_left', 'mu_reg_right', 'sigma_reg_right')],
                      dim=1).t().contiguous().float().to(dev).requires_grad_(True)
    tables = (torch.cat(tab_p).float().contiguous().to(dev), torch.tensor(tab_l, dtype=torch.int32, device=dev),
              torch.tensor(tab_q, dtype=torch.int32, device=dev))
    nmax = max(len(g[1]) for g in gts)
    gt = torch.zeros(B, 3 * nmax + 1)
    for b, g in enumerate(gts):
        n = len(g[1])
        gt[b, :2 * n] = torch.tensor(g[0], dtype=torch.float32).reshape(-1)
        gt[b, 2 * nmax:2 * nmax + n] = torch.tensor(g[1], dtype=torch.float32)
        gt[b, 3 * nmax] = n
    norm = torch.full((1,), 100.0, device=dev)
    got = ops.mq_loss(lg, of, sc, gauss, tables, lvl_len.to(dev), gt.to(dev), norm, radius, smoothing, 0.9, loss_weight, al, ncls != 1)
    (gw[0] * got[0] + gw[1] * got[1] + gw[2] * got[2] + gw[3] * got[3]).backward()
    assert abs(float(norm) - want_norm) <= 1e-5 * want_norm
    for i, k in enumerate(('cls_loss', 'reg_loss', 'al_loss', 'final_loss')):
        assert abs(float(got[i]) - float(want[k].sum())) <= 2e-5 * max(abs(float(want[k].sum())), 1e-3), (k, float(got[i]), float(want[k].sum()))

    def uncat(t, width):
        out, o = [], 0
        for l in range(L):
            if l and gaps:
                assert float(t[:, o].abs().max()) == 0.0          # separator rows get zero gradient
                o += 1
            out.append(t[:, o:o + Ts[l]])
            o += Ts[l]
        return out
    for l, (a, b_) in enumerate(zip(uncat(lg.grad, ncls), logits)):
        assert rel_err(a, b_.grad, 1e-9) < 1e-4, ("dlogits", l)
    for l, (a, b_) in enumerate(zip(uncat(of.grad, 2), raw)):
        assert rel_err(a, b_.grad, 1e-9) < 1e-4, ("doffsets", l)
    assert rel_err(sc.grad, scales.grad if scales.grad is not None else torch.zeros_like(scales), 1e-9) < 1e-4
    wg = torch.cat([p[k].grad if p[k].grad is not None else torch.zeros_like(p[k]) for k in ('mu', 'sigma', 'mu_reg_left', 'sigma_reg_left', 'mu_reg_right', 'sigma_reg_right')], dim=1).t()
    assert rel_err(gauss.grad, wg, 1e-9) < 1e-4
    # the packed-maximum workspace is left clean: a second call gives the same value
    norm2 = torch.full((1,), 100.0, device=dev)
    again = ops.mq_loss(lg.detach(), of.detach(), sc.detach(), gauss.detach(), tables, lvl_len.to(dev), gt.to(dev), norm2, radius,
                        smoothing, 0.9, loss_weight, al, ncls != 1)
    assert float(again[3]) == float(got[3])


GT2 = [([[10.0, 40.0], [60.5, 130.25]], [1, 5]), ([[10.0, 40.0], [60.5, 130.25]], [1, 5])]


@pytest.mark.parametrize("gaps", [True, False])
def test_loss_basic(dev, gaps):
    _case(dev, 22, 256, GT2, [256, 239], gaps)


def test_loss_ragged_gt_counts_ties_and_smoothing(dev):
    gts = [([[5.0, 25.0], [5.0, 25.0005], [100.0, 101.0], [30.0, 200.0]], [3, 7, 3, 0]),      # two GT within 1e-3 of each other: multi-hot
           ([[50.0, 58.0]], [2]),
           ([[0.0, 256.0], [120.0, 136.0], [121.0, 135.0]], [9, 9, 1])]
    _case(dev, 10, 256, gts, [256, 100, 255], True, smoothing=0.1, seed=3)


def test_loss_no_positive_anywhere(dev):
    """GT far outside every regression range / beyond the clip: num_pos = 0 -> normaliser uses max(num_pos, 1), reg loss 0"""
    gts = [([[300.0, 300.5]], [0]), ([[400.0, 400.2]], [1])]
    _case(dev, 4, 64, gts, [64, 40], True, seed=5)


def test_loss_center_sample_none_and_many_classes(dev):
    _case(dev, 110, 128, [([[3.0, 90.0], [20.0, 30.0]], [109, 64]), ([[40.0, 44.0]], [33])], [128, 128], False, radius=0.0,
          seed=7, al=0.5, loss_weight=2.0)


def test_model_fused_equals_tensor_expression_path(dev):
    """the whole model with the fused loss kernels vs VILCO_FUSED_LOSS=0 (the tensor-expression losses the round-1
    goldens were checked with): same losses, same gradients"""
    from parity_util import GRAD_FLOOR, build_hip_model, golden_cfg, golden_inputs, load_golden
    gold = load_golden("noxl")
    outs = []
    for fused in (True, False):
        model = build_hip_model(gold)
        model.fused_loss = fused
        model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
        losses = model(golden_inputs(gold), task_id=gold['task_id'], is_training=True)
        losses['final_loss'].backward()
        outs.append((losses, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, model.loss_normalizer))
    (la, ga, na), (lb, gb, nb) = outs
    assert abs(na - nb) < 1e-4
    for k in ('cls_loss', 'reg_loss', 'al_loss', 'final_loss'):
        assert rel_err(la[k], lb[k]) < 1e-5, k
    assert sorted(ga) == sorted(gb)
    for k in ga:
        if not k.endswith(('key_norm.bias', '.key.bias')):          # analytically zero: 1e-11-level noise both ways
            assert rel_err(ga[k], gb[k], GRAD_FLOOR) < 1e-4, k


def test_parameter_gradients_of_the_loss_are_the_same_bits_on_every_launch(dev):
    """Round 6: the gradients of the per-level regression scales and of the gaussian-weight parameters (mu / sigma) leave
    vilco_mq_loss_bwd through per-row shares and a fixed-order second launch (loss_bwd_finish_kernel) instead of float atomicAdds --
    the last sums of a step whose order changed from launch to launch.  Eight backward passes of the golden model from one state:
    every gradient tensor bit-equal to the first pass's, the atomics' former targets included."""
    from parity_util import build_hip_model, golden_cfg, golden_inputs, load_golden
    gold = load_golden("xl")
    model = build_hip_model(gold)
    ref = None
    for it in range(8):
        model.zero_grad(set_to_none=True)
        model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
        model(golden_inputs(gold), task_id=gold['task_id'], is_training=True)['final_loss'].backward()
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        if ref is None:
            ref = g
            assert any(k.startswith(('mu', 'sigma')) for k in g) and any(k.startswith('reg_head.scale.') for k in g)
            assert any(float(g[k].abs().max()) > 0 for k in g if k.startswith('reg_head.scale.'))      # (levels without a positive point: 0)
        else:
            assert sorted(g) == sorted(ref)
            for k in g:
                assert torch.equal(g[k], ref[k]), (it, k)
