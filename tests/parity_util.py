"""Helpers shared by the parity tests, smoke() and bench.py (test infrastructure)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
from ref_import import xlnet_json  # noqa: E402  (pure helper; does not touch /root/reference)


def load_golden(name):
    return torch.load(os.path.join(HERE, "golden", "model_%s.pt" % name), weights_only=False)


def golden_cfg(gold):
    """cfg['model'] dict for a golden case (vilco_amd.core.config applies the DEFAULTS)."""
    from vilco_amd.core.config import make_config
    return make_config(**gold['overrides'])['model']


def golden_inputs(gold):
    m = golden_cfg(gold)
    return cases.video_list(m['max_seq_len'], m['input_dim'], m['n_txt_in'], gold['L'])


def oracle_run(gold, dtype=torch.float64, with_grads=True):
    """oracle losses (+ grads wrt every parameter) for a golden case."""
    from oracle import mq_oracle
    cfg = golden_cfg(gold)
    p = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and with_grads)
         for k, v in gold['state_dict'].items()}
    vl = golden_inputs(gold)
    vl = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
    losses, ln = mq_oracle.forward_losses(p, cfg, vl, task_id=gold['task_id'], n_known=gold['n_known'])
    grads = None
    if with_grads:
        losses['final_loss'].backward()
        grads = {k: v.grad for k, v in p.items()}
    return losses, grads, ln


def rel_err(got, want, floor=1e-12):
    """max |got - want| / max(max |want|, floor).  `floor` keeps analytically-zero gradients (e.g. the
    key LayerNorm bias: a constant shift of all keys cancels in softmax) from dividing noise by noise."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(want.abs().max().item(), floor)).item()


GRAD_FLOOR = 1e-7


def build_hip_model(gold, device="cuda:0", xl_heads=4):
    import vilco_amd.modeling as vm
    m = golden_cfg(gold)
    kw = dict(m)
    if m['use_xl']:
        kw['xlnet_config'] = xlnet_json(m['embd_dim'], xl_heads)
    model = vm.make_meta_arch('LocPointTransformer', **kw)
    model.load_state_dict(gold['state_dict'])
    model.n_known = gold['n_known']
    return model.to(device).eval()


def run_smoke():
    """one tiny fwd+bwd of the MQ model on cuda:0 through the HIP path, checked against the oracle."""
    gold = load_golden("xl")
    model = build_hip_model(gold)
    model.loss_normalizer = golden_cfg(gold)['train_cfg']['init_loss_norm']
    losses = model(golden_inputs(gold), task_id=gold['task_id'], is_training=True)
    losses['final_loss'].backward()
    torch.cuda.synchronize()
    want, wgrads, _ = oracle_run(gold, torch.float64)
    for k in ('cls_loss', 'reg_loss', 'final_loss'):
        e = rel_err(losses[k], want[k])
        assert e < 1e-3, "smoke: %s differs from the oracle by %.3e" % (k, e)
    worst = 0.0
    for n_, p_ in model.named_parameters():
        if wgrads[n_] is not None and p_.grad is not None:
            worst = max(worst, rel_err(p_.grad, wgrads[n_], GRAD_FLOOR))
    assert worst < 1e-3, "smoke: worst gradient rel err %.3e" % worst
    print("smoke ok: final_loss %.6f (oracle %.6f), worst grad rel err %.2e" %
          (float(losses['final_loss']), float(want['final_loss']), worst))


# ------------------------------------------------------------------------------------------ episode case
def load_episode_golden():
    return torch.load(os.path.join(HERE, "golden", "episode_vilco.pt"), weights_only=False)


def episode_full_state(gold_state):
    """golden snapshot -> full state_dict: small tensors as stored, the adapters' large Linear weights regenerated
    (cases.seeded_tensor) under every alias (pets.*, backbone.branch.b.adapters.attn.*, pets_emas.0.module.*)."""
    out = {}
    for k, v in gold_state.items():
        if isinstance(v, dict):
            tail = k.split('layer.')[-1]                                  # '0.weight' / '2.weight'
            if k.startswith('pets_emas.'):
                i = k.split('.module.')[1].split('.')[0]
            elif k.startswith('pets.'):
                i = k.split('.')[1]
            else:
                i = str(cases.episode_overrides()['cl_cfg']['adapt_blocks'].index(int(k.split('.')[2])))
            name = 'pets.%s.layer.%s' % (i, tail)
            out[k] = cases.seeded_tensor(name, v['shape'], 0.02 if 'layer.0' in name else 0.01)
        else:
            out[k] = v.clone()
    return out


def compact_err(got, want):
    """relative error of a tensor against a golden entry (full tensor or cases.compact sample)"""
    if isinstance(want, dict):
        g = got.detach().reshape(-1).double().cpu()
        return max(rel_err(g[::cases.SAMPLE_STRIDE], want['sample']),
                   abs(float(g.norm()) - want['l2']) / max(want['l2'], 1e-12))
    return rel_err(got, want)


def delta_err(got_after, got_init, want_after, want_init):
    """|| (got_after - got_init) - (want_after - want_init) ||_2 / || want_after - want_init ||_2 over the stored
    elements: the error of the UPDATE a training phase made to a tensor (robust to single near-zero-gradient elements,
    where Adam's m / sqrt(v) is a coin flip in any arithmetic)."""
    def vec(t):
        if isinstance(t, dict):
            return t['sample'].double()
        return t.detach().reshape(-1).double().cpu()
    big = isinstance(want_after, dict)
    g, gi = vec(got_after), vec(got_init)
    if big:
        g, gi = g[::cases.SAMPLE_STRIDE], gi[::cases.SAMPLE_STRIDE]
    w, wi = vec(want_after), vec(want_init)
    ref = (w - wi).norm().item()
    if ref < 1e-9:
        return (g - gi).norm().item()
    return ((g - gi) - (w - wi)).norm().item() / ref


def oracle_episode_trajectory(gold, dtype=torch.float64):
    """The episode case stepped by the ORACLE (CPU) under torch.optim.AdamW + clip_grad_norm_ + the reference's parameter
    groups (incl. the duplicated adapter aliases) and scheduler: returns per task (loss list, state after the task,
    state before it).  In float32 this reproduces the reference golden bit for bit; in float64 it is the exact-arithmetic
    trajectory the HIP path is held to (Adam turns single fp32 rounding events of the reference into visible update
    differences on a few tensors, which fp64 and the HIP path do not share)."""
    import warnings
    import vilco_amd.modeling as vm
    from oracle import mq_oracle
    from vilco_amd.core.config import make_config
    from vilco_amd.utils.train_utils import make_scheduler, param_groups
    cfg = make_config(**gold['overrides'])
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg['model'], xlnet_config=xlnet_json(cfg['model']['embd_dim'], cases.EP_H)))
    model.load_state_dict(episode_full_state(gold['init_state']))
    model = model.to(dtype)
    norm = float(cfg['model']['train_cfg']['init_loss_norm'])

    def mkopt():
        decay, no_decay, remain = param_groups(model)
        pd = dict(model.named_parameters())
        for mn, m in model.named_modules():
            for pn, p in m.named_parameters():
                full = '%s.%s' % (mn, pn) if mn else pn
                if 'pets' in full:
                    pd[full] = p
        wd = cfg['opt']['weight_decay']
        groups = [{"params": [pd[n] for n in decay], "weight_decay": wd}, {"params": [pd[n] for n in no_decay], "weight_decay": 0.0},
                  {"params": [pd[n] for n in remain], "weight_decay": wd}]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return torch.optim.AdamW(groups, lr=cfg['opt']['learning_rate'])
    out = []
    ema = {k: v.detach().clone() for k, v in model.pets.state_dict().items()}      # adapter EMA (meta_archs.py:702-707)
    for task in range(2):
        opt = mkopt()
        sch = make_scheduler(opt, cfg['opt'], len(cases.episode_batches(task)))
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        mcfg = dict(cfg['model'], num_classes=model.num_classes)
        losses = []
        for vl in cases.episode_batches(task):
            opt.zero_grad(set_to_none=True)
            vlt = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
            ls, norm = mq_oracle.forward_losses(dict(model.state_dict(keep_vars=True)), mcfg, vlt, loss_normalizer=norm,
                                                task_id=task, n_known=model.n_known)
            ls['final_loss'].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), cfg['train_cfg']['clip_grad_l2norm'])
            opt.step()
            sch.step()
            for k, v in model.pets.state_dict().items():
                ema[k] = 0.999 * ema[k] + 0.001 * v.detach()
            losses.append({k: float(v.detach().sum()) for k, v in ls.items()})
        # eval-mode forward of the task's first clip with the adapter-EMA ensemble (meta_archs.py:854-881)
        clip = cases.episode_batches(task)[0][0]
        clip = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in clip.items()}
        with torch.no_grad():
            sd = dict(model.state_dict())
            _, _, cls_a, off_a, _ = mq_oracle.forward_network(sd, mcfg, [clip], training=False, task_id=task)
            sd_e = dict(sd)
            for k, v in ema.items():
                sd_e['pets.' + k] = v
            _, _, cls_b, off_b, _ = mq_oracle.forward_network(sd_e, mcfg, [clip], training=False, task_id=task)
        evals = ([(a + b) / 2 for a, b in zip(cls_a, cls_b)], [(a + b) / 2 for a, b in zip(off_a, off_b)])
        out.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, before, evals))
        if task == 0:
            model.n_known = gold['tasks'][0]['n_known']
            torch.manual_seed(99)
            model.augment_classification(cases.EP_NEW, 'cpu')
            model = model.to(dtype)
    return out


# ------------------------------------------------------------------------------------------------------------------
# Full-size (config P) step under ONE realisation of the dropout / stochastic-depth masks: the HIP step, then the oracle
# replaying exactly those masks in fp32 and in fp64 (tests/test_fullsize_gpu.py, tools/diag/p_parity_realisations.py).
def p_step_three_ways(dev, realisation, oracle_dtypes=(torch.float32, torch.float64), threads=None):
    """-> (hip_losses, hip_grads, {dtype: (losses, grads)}).  `realisation` seeds the device RNG (stochastic depth), the
    dropout seed counter and the device step word, so every value gives different masks whatever ran before in the process."""
    import bench
    import vilco_amd.modeling as vm
    from oracle import mq_oracle
    from vilco_amd import _lib, ops
    from vilco_amd.modeling import blocks
    cfg = bench.p_config()
    torch.manual_seed(0)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg, xlnet_config=bench.p_xlnet())).to(dev).train()
    for name, mod in model.named_modules():          # names for the LayerNorm -> ReLU sign-decision log (ops.relu_log)
        if isinstance(mod, blocks.LayerNorm):
            mod._site = name
    batch = bench.synth_batch(2, dev)
    torch.cuda.manual_seed_all(1000 + realisation)
    blocks.reset_drop_pool()
    ops._drop_counter[0] = 7919 * realisation
    _lib.check(_lib.load().vilco_seed_word_set(realisation, None))
    ops.dropout_log = []
    ops.relu_log = []
    try:
        losses = model(batch, is_training=True)
        losses['final_loss'].backward()
        log = list(ops.dropout_log)
        relu_sites = [(e[0], e[1].cpu()) for e in ops.relu_log]
    finally:
        ops.dropout_log = None
        ops.relu_log = None
    torch.cuda.synchronize()
    level_T = [int(cfg['max_seq_len']) // int(cfg['scale_factor']) ** l for l in range(int(cfg['backbone_arch'][-1]) + 1)]
    hip_grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    hip_losses = {k: float(v) for k, v in losses.items()}
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    del model, losses
    torch.cuda.empty_cache()
    if threads:
        torch.set_num_threads(threads)
    out = {}
    # the masks are regenerated by the kernels that drew them: the device step word stays at `realisation` until they exist
    masks = [(e[0], e[1]) if len(e) == 2 else (e[0], ops.dropout_mask(e[1], e[2], e[3], dev, e[0]).cpu()) for e in log]
    torch.cuda.synchronize()
    _lib.check(_lib.load().vilco_seed_word_set(0, None))
    def run_oracle(dt, relu_masks=None):
        """one oracle step under the HIP step's masks; relu_masks: the HIP step's LayerNorm -> ReLU sign decisions in the
        oracle's layout (hip_relu_masks) -> the run takes the same sides and reports where its own would have differed"""
        p = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}
        vl = [{k: ((v.cpu().to(dt) if v.is_floating_point() else v.cpu()) if torch.is_tensor(v) else v) for k, v in d.items()} for d in batch]
        ctx = mq_oracle.DropReplay(masks, None)
        mq_oracle.DROP = ctx
        rr = mq_oracle.RELU = mq_oracle.ReluReplay(relu_masks) if relu_masks is not None else None
        try:
            want, _ = mq_oracle.forward_losses(p, cfg, vl)
            want['final_loss'].backward()
        finally:
            mq_oracle.DROP = None
            mq_oracle.RELU = None
        assert ctx.leftover() == {}, ctx.leftover()
        res = ({k: float(v) for k, v in want.items()},
               {k: v.grad.detach().double() for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None})
        if rr is not None:
            assert rr.seen == set(relu_masks), (sorted(set(relu_masks) - rr.seen))
            res = res + (rr.events,)
        return res

    for dt in oracle_dtypes:
        out[dt] = run_oracle(dt)
    out['rerun'] = run_oracle
    out['hip_relu'] = hip_relu_masks(relu_sites, level_T)
    return hip_losses, hip_grads, out


def hip_relu_masks(relu_sites, level_T):
    """(module name, y > 0 in the HIP layout) of every LayerNorm -> ReLU call of a step -> {oracle site: bool [B, C, T]}.
    HIP layouts: token-major [B, T, C]; the heads run all pyramid levels of a clip as ONE token sequence with a zero row between
    neighbours (vilco_amd/modeling/meta_archs.py: LevelCat) -- split back into the oracle's per-level calls 'norm.i@level'."""
    out = {}
    for site, m in relu_sites:
        assert site is not None and m.dim() == 3, site
        if site.startswith(('cls_head.norm.', 'reg_head.norm.')):
            assert m.shape[1] == sum(level_T) + len(level_T) - 1, (site, tuple(m.shape), level_T)
            o = 0
            for l, T in enumerate(level_T):
                out['%s@%d' % (site, l)] = m[:, o:o + T].permute(0, 2, 1).contiguous()
                o += T + 1
        else:
            out[site] = m.permute(0, 2, 1).contiguous()
    return out


def tensor_distance(g, w):
    """(max |g - w| / max |w|, L2 distance / ||w||, fraction of elements beyond 1e-3 of max |w|)"""
    d = (g - w).abs()
    top = w.abs().max().clamp_min(1e-7)
    return (d.max() / top).item(), ((g - w).norm() / w.norm().clamp_min(1e-12)).item(), (d > 1e-3 * top).double().mean().item()
