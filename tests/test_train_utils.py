"""Step glue.  CPU: parameter grouping and LR sequences vs goldens produced by the
reference's make_optimizer / make_scheduler.  GPU: the fused multi-tensor clip + AdamW / SGD kernels vs torch.optim."""
import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "train_glue.json")))


def _model(name):
    from parity_util import xlnet_json
    import vilco_amd.modeling as vm
    from vilco_amd.core.config import make_config
    over = GOLD[name]["overrides"]
    over['model']['backbone_arch'] = tuple(over['model']['backbone_arch'])
    m = make_config(**over)['model']
    kw = dict(m, xlnet_config=xlnet_json(m['embd_dim'], 4)) if m['use_xl'] else m
    return vm.make_meta_arch('LocPointTransformer', **kw)


@pytest.mark.parametrize("name", ["xl", "prompt"])
def test_param_groups_match_reference(name):
    from vilco_amd.utils.train_utils import make_optimizer
    model = _model(name)
    opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=1e-4))
    by_id = {}
    for n, p in model.named_parameters(remove_duplicate=False):
        by_id.setdefault(id(p), []).append(n)
    groups = [[sorted(by_id[id(p)])[0] for p in g['params']] for g in opt.param_groups]
    assert [len(g) for g in groups] == [len(g) for g in GOLD[name]["groups"]]
    assert groups == GOLD[name]["groups"]
    assert [g['weight_decay'] for g in opt.param_groups] == GOLD[name]["weight_decay"]


@pytest.mark.parametrize("tag", ["cosine", "multistep"])
def test_lr_sequences_match_reference(tag):
    from vilco_amd.utils.train_utils import make_scheduler
    g = GOLD["lr"][tag]
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=g["base_lr"])
    sch = make_scheduler(opt, g["cfg"], g["iters_per_epoch"])
    for want in g["lrs"]:
        got = opt.param_groups[0]['lr']
        assert abs(got - want) <= 1e-12 + 1e-9 * abs(want), (got, want)
        opt.step()
        sch.step()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["AdamW", "SGD"])
@pytest.mark.parametrize("clip", [-1.0, 0.5])
def test_fused_optimizer_vs_torch(dev, kind, clip):
    from vilco_amd.utils.train_utils import FusedOptimizer
    torch.manual_seed(0)
    shapes = [(300, 17), (5,), (70000,), (1, 64, 1), (33, 3, 3)]
    ref_p = [torch.randn(s, dtype=torch.float64, requires_grad=True) for s in shapes]
    hip_p = [p.detach().float().to(dev).requires_grad_(True) for p in ref_p]
    mk = lambda ps: [{"params": ps[:2], "weight_decay": 0.05}, {"params": ps[2:4], "weight_decay": 0.0},
                     {"params": ps[4:], "weight_decay": 0.05}]
    if kind == "AdamW":
        ref = torch.optim.AdamW(mk(ref_p), lr=1e-2)
    else:
        ref = torch.optim.SGD(mk(ref_p), lr=1e-2, momentum=0.9)
    hip = FusedOptimizer(mk(hip_p), lr=1e-2, kind=kind, momentum=0.9)
    for step in range(4):
        for a, b in zip(ref_p, hip_p):
            g = torch.randn(a.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(10 * step + a.numel()))
            a.grad, b.grad = g.clone(), g.float().to(dev)
        hip_p[1].grad = None if step == 2 else hip_p[1].grad          # a parameter without a gradient is skipped
        ref_p[1].grad = None if step == 2 else ref_p[1].grad
        if clip > 0:
            want_norm = torch.nn.utils.clip_grad_norm_([p for p in ref_p if p.grad is not None], clip)
        for g in hip.param_groups + ref.param_groups:
            g['lr'] = 1e-2 * (1 + step)
        ref.step()
        hip.step(clip_grad_l2norm=clip)
        if clip > 0:
            assert abs(float(hip.last_grad_norm[0]) - float(want_norm)) < 1e-4 * float(want_norm)
        for a, b in zip(ref_p, hip_p):
            err = (b.detach().cpu().double() - a.detach()).abs().max() / a.detach().abs().max()
            assert err < 2e-6, (step, tuple(a.shape), float(err))


@pytest.mark.gpu
def test_fused_optimizer_duplicate_param_uses_clipped_grad(dev):
    """a parameter listed twice (the `pets.*` adapter aliases, train_utils.py:108-113) is stepped twice per
    iteration by torch.optim, both times with the gradient clip_grad_norm_ scaled in place"""
    import warnings
    from vilco_amd.utils.train_utils import FusedOptimizer
    torch.manual_seed(1)
    shapes = [(40, 9), (130,), (7, 3)]
    ref_p = [torch.randn(s, dtype=torch.float64, requires_grad=True) for s in shapes]
    hip_p = [p.detach().float().to(dev).requires_grad_(True) for p in ref_p]
    mk = lambda ps: [{"params": [ps[0], ps[1], ps[1]], "weight_decay": 0.05}, {"params": [ps[2]], "weight_decay": 0.0}]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = torch.optim.AdamW(mk(ref_p), lr=1e-2)
        hip = FusedOptimizer(mk(hip_p), lr=1e-2, kind="AdamW")
    for step in range(3):
        for a, b in zip(ref_p, hip_p):
            g = 5.0 * torch.randn(a.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(7 * step + a.numel()))
            a.grad, b.grad = g.clone(), g.float().to(dev)
        torch.nn.utils.clip_grad_norm_(ref_p, 0.5)
        ref.step()
        hip.step(clip_grad_l2norm=0.5)
        for a, b in zip(ref_p, hip_p):
            err = (b.detach().cpu().double() - a.detach()).abs().max() / a.detach().abs().max()
            assert err < 2e-6, (step, tuple(a.shape), float(err))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["AdamW", "SGD"])
def test_optimizer_leaves_weight_amax_for_the_packs(dev, kind):
    """the update kernels emit max|p| per chunk (vilco_optim_step_amax); FusedOptimizer tags every matrix parameter with
    its slice; the weight pack built from the tag is bit-identical to the pack that runs its own amax pass; a
    parameter stepped twice carries the partials of its LAST update; any later write invalidates the tag"""
    import warnings
    from vilco_amd import ops
    from vilco_amd.utils.train_utils import FusedOptimizer
    torch.manual_seed(3)
    shapes = [(1024, 512), (70,), (96, 40, 3), (3, 5)]
    ps = [(torch.randn(s) * (10.0 ** i)).to(dev).requires_grad_(True) for i, s in enumerate(shapes)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt = FusedOptimizer([{"params": [ps[0], ps[1], ps[2], ps[2]], "weight_decay": 0.05},
                              {"params": [ps[3]], "weight_decay": 0.0}], lr=1e-2, kind=kind)
    for step in range(2):
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step(clip_grad_l2norm=1.0)
        for p in ps:
            tag = getattr(p, "_vilco_wamax", None)
            if p.dim() < 2:
                assert tag is None
                continue
            parts, n = ops._weight_amax(p)
            assert n == -(-p.numel() // 16384) and parts.numel() == n
            assert float(parts.max()) == float(p.detach().abs().max()), (step, tuple(p.shape))
        w = ps[0]
        rows, cols = w.shape
        tagged = ops.weight_planes(w, rows, cols)
        plain = ops.pack(w.detach().clone(), rows, cols)
        # buffer = [1024 amax partials | {1/s, s} | pad to 4608 B | planes]: same scale, same planes
        assert torch.equal(tagged[4096:4104], plain[4096:4104]) and torch.equal(tagged[4608:], plain[4608:])
    assert ops._weight_amax(ps[0][:512])[0] is None              # a slice of the parameter: the owner's maximum is not its own
    ops.weights_changed()                                        # somebody wrote parameter memory behind our back
    assert ops._weight_amax(ps[0])[0] is None
    opt.step()
    with torch.no_grad():
        ps[0].mul_(2.0)                                          # an in-place edit moves the version counter
    assert ops._weight_amax(ps[0])[0] is None


@pytest.mark.gpu
def test_train_step_runs_and_descends(dev):
    """three iterations of the glue on a golden-size model: loss decreases with a plain AdamW schedule"""
    from parity_util import golden_inputs, load_golden, build_hip_model, golden_cfg
    from vilco_amd.utils.train_utils import make_optimizer, make_scheduler, train_step
    gold = load_golden("noxl")
    model = build_hip_model(gold).train()
    model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
    opt = make_optimizer(model, dict(type="AdamW", momentum=0.9, weight_decay=0.05, learning_rate=2e-3))
    sch = make_scheduler(opt, dict(warmup=False, epochs=2, schedule_type="cosine", schedule_steps=[], schedule_gamma=0.1), 10)
    vl = golden_inputs(gold)
    losses = [float(train_step(model, opt, sch, vl, clip_grad_l2norm=1.0)['final_loss']) for _ in range(6)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0], losses


def _reference_nlq_saves(init_r1, r1_of_epoch, max_epochs, ckpt_freq):
    """The save decisions of one task of NLQ/train_cl.py:216-292, restated flag for flag: validation only at the last epoch or at
    multiples of ckpt_freq (:216-222), `is_best = R1 >= best_R1; best_R1 = max(R1, best_R1)` inside it (:250-251), and the
    `if is_best: save_checkpoint(... 'Best_task_XX')` block OUTSIDE it (:283), once per epoch."""
    best_r1, saves, state = init_r1, [], {}
    for epoch in range(max_epochs):
        if (epoch == max_epochs - 1) or ((ckpt_freq > 0) and (epoch % ckpt_freq == 0)):
            r1 = r1_of_epoch[epoch]
            state['is_best'] = r1 >= best_r1
            best_r1 = max(r1, best_r1)
        if state['is_best']:
            saves.append(epoch)
    return saves, best_r1


@pytest.mark.parametrize("ckpt_freq,max_epochs,r1s,init", [
    (2, 5, [0.3, 0.9, 0.2, 0.9, 0.25], 0.1),      # epoch 0 reaches the bar: epochs 0 AND 1 (unvalidated) write the file
    (2, 6, [0.1, 0.0, 0.4, 0.0, 0.5, 0.5], 0.2),  # misses at 0, hits at 2 (3 rides along), 4 (and the last epoch validates too)
    (1, 4, [0.5, 0.4, 0.6, 0.6], 0.45),           # every epoch validated: sticky == plain
    (3, 7, [0.9, 0.0, 0.0, 0.1, 0.0, 0.0, 0.95], 0.5),
])
def test_nlq_best_checkpoint_is_sticky_like_the_reference(ckpt_freq, max_epochs, r1s, init):
    """run_episodes_nlq wrote Best_task_XX only at validated epochs that reached the bar; the reference
    keeps overwriting it at the following unvalidated epochs (sticky is_best).  train_cl.StickyBest carries that rule."""
    from vilco_amd.train_cl import StickyBest
    want_saves, want_best = _reference_nlq_saves(init, r1s, max_epochs, ckpt_freq)
    t = StickyBest()
    t.new_task(init)
    saves = []
    for epoch in range(max_epochs):
        if StickyBest.validates(epoch, max_epochs, ckpt_freq):
            t.validated(epoch, r1s[epoch])
        if t.is_best:
            saves.append(epoch)
    assert saves == want_saves and t.best == want_best
    assert any(not StickyBest.validates(e, max_epochs, ckpt_freq) for e in saves) == (ckpt_freq > 1)
