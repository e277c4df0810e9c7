"""BASELINE configs[2] as a tested config: the episode path (adapters + their EMA + L2P prompts + replay memory +
class-head growth + a new optimizer per task) against golden vectors recorded from the imported reference driving
its own train_one_epoch / augment_classification (tests/golden/make_golden_episode.py).

CPU part: the oracle and the host-side logic (memory sampling, head growth, the in-memory task stream).
GPU part: the HIP model + FusedOptimizer + train_one_epoch reproduce the reference's loss / LR sequences, parameter
updates, EMA adapters, post-augment tensors and the EMA-ensemble inference output."""
import os
import random

import pytest
import torch

from parity_util import (GRAD_FLOOR, cases, compact_err, delta_err, episode_full_state, load_episode_golden,
                         oracle_episode_trajectory, rel_err)


# a constant shift of every key cancels in softmax: the gradient of these biases is analytically zero, the
# reference's fp32 gradient is 1e-12-level noise, and Adam's m / sqrt(v) turns that noise into +-lr steps
NOISE_GRADS = ('key_norm.bias', '.key.bias')


def _cfg(gold):
    from vilco_amd.core.config import make_config
    return make_config(**gold['overrides'])


def _xl(cfg):
    from ref_import import xlnet_json
    return xlnet_json(cfg['model']['embd_dim'], cases.EP_H)


def _task_data(task):
    data = {}
    for b in cases.episode_batches(task):
        for v in b:
            for c in v['labels'].tolist():
                if (task == 0 and c < cases.EP_NCLS0) or (task == 1 and c >= cases.EP_NCLS0):
                    data.setdefault(c, []).append(v)
    return data


def test_oracle_matches_reference_first_iteration():
    """the oracle (with time adapters and prompts) on the episode case's initial state = the reference's first loss"""
    from oracle import mq_oracle
    gold = load_episode_golden()
    cfg = _cfg(gold)['model']
    p = {k: v.double() if v.is_floating_point() else v for k, v in episode_full_state(gold['init_state']).items()}
    vl = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
          for d in cases.episode_batches(0)[0]]
    losses, ln = mq_oracle.forward_losses(p, cfg, vl, task_id=0, n_known=0)
    want = gold['tasks'][0]['losses'][0]
    for k in ('cls_loss', 'reg_loss', 'al_loss', 'final_loss'):
        assert abs(float(losses[k]) - want[k]) <= 2e-5 * max(abs(want[k]), 1e-3), (k, float(losses[k]), want[k])


def test_oracle_trajectory_reproduces_reference_episode():
    """oracle (fp32, CPU) + torch AdamW / clip / scheduler over both tasks = the reference's recorded losses and
    parameter updates: the oracle is pinned through eight optimisation steps, head growth included"""
    gold = load_episode_golden()
    traj = oracle_episode_trajectory(gold, torch.float32)
    want_init = gold['init_state']
    for task, (losses, after, before, evals) in enumerate(traj):
        want = gold['tasks'][task]
        for a, b in zip(evals[0], want['eval_cls_logits']):          # EMA-ensemble eval forward (prompt selection incl.)
            assert rel_err(a, b) < 1e-4
        for a, b in zip(evals[1], want['eval_offsets']):
            assert rel_err(a, b, 1e-6) < 1e-4
        for i, (h, w) in enumerate(zip(losses, want['losses'])):
            for k in w:
                assert abs(h[k] - w[k]) <= 1e-5 * max(abs(w[k]), 1e-3), (task, i, k, h[k], w[k])
        errs = sorted(((delta_err(after[k], before[k], w, want_init[k]), k) for k, w in want['state'].items()
                       if after[k].is_floating_point() and not k.endswith(NOISE_GRADS) and not k.startswith('pets_emas.')),
                      reverse=True)
        assert errs[0][0] < 1e-3, (task, errs[:5])
        want_init = dict(want['state'])
        want_init.update(want.get('post_augment', {}))


def test_memory_sampling_and_head_growth_match_reference():
    """add_samples_to_mem (random replay sampling, m per class) and augment_classification on the CPU: host logic"""
    import vilco_amd.modeling as vm
    gold = load_episode_golden()
    cfg = _cfg(gold)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg['model'], xlnet_config=_xl(cfg)))
    state = episode_full_state(gold['tasks'][0]['state'])
    model.load_state_dict(state, strict=True)                 # reference keys, incl. the adapter aliases and the EMA
    random.seed(0)
    n_cls = model.cls_head.cls_head.conv.out_channels
    assert n_cls == cases.EP_NCLS0
    from vilco_amd.train_cl import memory_quota
    model.add_samples_to_mem(None, _task_data(0), memory_quota(cfg['cl_cfg']['memory_size'], n_cls))
    assert {c: [v['video_id'] for v in vs] for c, vs in model.memory.items()} == gold['tasks'][0]['memory_ids']
    torch.manual_seed(99)
    model.augment_classification(cases.EP_NEW, 'cpu')
    sd = model.state_dict()
    for k, want in gold['tasks'][0]['post_augment'].items():
        assert sd[k].shape == want.shape and torch.equal(sd[k], want), k
    assert model.num_classes == cases.EP_NCLS0 + cases.EP_NEW


def test_in_memory_stream_contract():
    from vilco_amd.utils.cl_stream import InMemoryQILStream
    s = InMemoryQILStream([_task_data(0), _task_data(1)], batch_size=2, seed=3)
    it = iter(s)
    d0, l0, nxt = next(it)
    assert nxt == cases.EP_NEW and sorted(d0) == [0, 1, 2, 3] and len(l0) == 4
    ids0 = sorted(v['video_id'] for b in l0 for v in b)
    assert ids0 == sorted({v['video_id'] for vs in d0.values() for v in vs})       # every clip once per epoch
    order0 = [v['video_id'] for b in l0 for v in b]
    l0.sampler.set_epoch(1)
    assert sorted(v['video_id'] for b in l0 for v in b) == ids0 and [v['video_id'] for b in l0 for v in b] != order0
    s.memory = {0: d0[0][:1]}
    d1, l1, nxt = next(it)
    assert nxt is None
    assert [v['is_memory'] for v in l1.items] == [True] + [False] * (len(l1.items) - 1)    # replayed clips first
    assert len(l1) == len(l1.items) // 2                                                   # drop_last, as the reference's loader
    # two ranks see disjoint halves of the same permutation
    a = InMemoryQILStream([_task_data(0)], 2, seed=3, rank=0, world=2).__next__()[1]
    b = InMemoryQILStream([_task_data(0)], 2, seed=3, rank=1, world=2).__next__()[1]
    ia, ib = {v['video_id'] for x in a for v in x}, {v['video_id'] for x in b for v in x}
    assert len(a) == len(b) == 2 and not (ia & ib)


# ------------------------------------------------------------------------------------------------------- GPU
def _build(gold, dev):
    import vilco_amd.modeling as vm
    cfg = _cfg(gold)
    model = vm.make_meta_arch('LocPointTransformer', **dict(cfg['model'], xlnet_config=_xl(cfg)))
    model.load_state_dict(episode_full_state(gold['init_state']), strict=True)
    model = model.to(dev)
    model.loss_normalizer = cfg['model']['train_cfg']['init_loss_norm']
    return cfg, model


@pytest.mark.gpu
def test_episode_reproduces_reference(dev):
    """two tasks x four iterations through vilco_amd.utils.train_utils.train_one_epoch on the HIP path"""
    from vilco_amd.utils.train_utils import make_optimizer, make_scheduler, train_one_epoch
    gold = load_episode_golden()
    exact = oracle_episode_trajectory(gold, torch.float64)        # the exact-arithmetic trajectory (CPU, ~20 s)
    cfg, model = _build(gold, dev)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    want_init = gold['init_state']
    opt = make_optimizer(model, cfg['opt'])
    sch = make_scheduler(opt, cfg['opt'], len(cases.episode_batches(0)))
    for task in range(2):
        want = gold['tasks'][task]
        lrs = []
        step0 = sch.step

        def rec_step(*a, **k):
            lrs.append(opt.param_groups[0]['lr'])
            return step0(*a, **k)
        sch.step = rec_step
        model.pre_train_epoch(task_id=task, current_epoch=0)
        hist = train_one_epoch(cases.episode_batches(task), model, opt, sch, 0, 1,
                               clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], cl_name=cfg['cl_cfg']['name'],
                               reg_lambda=cfg['cl_cfg']['reg_lambda'], prev_out_cls_logits_dict={}, current_task_id=task)
        assert len(hist) == len(want['losses'])
        for i, (h, w) in enumerate(zip(hist, want['losses'])):
            for k in w:
                assert abs(float(h[k]) - w[k]) <= 1e-3 * max(abs(w[k]), 1e-3), (task, i, k, float(h[k]), w[k])
        assert all(abs(a - b) <= 1e-12 + 1e-9 * abs(b) for a, b in zip(lrs, want['lrs'])), (lrs, want['lrs'])
        assert abs(model.loss_normalizer - want['loss_normalizer']) <= 1e-4 * want['loss_normalizer']
        # parameter updates, adapter EMA included (pets_emas.* keys)
        sd = model.state_dict()
        # The reference's own fp32 run and the exact (fp64) trajectory are BOTH legitimate executions of the reference
        # algorithm, and they differ from each other by up to 0.14 (L2, per tensor) on the regression head in task 1:
        # several LayerNorm->ReLU channels there are alive on a handful of tokens only (their gradient is exactly 0 on
        # some batches, tools/diag/episode_grads.py), one pre-activation within rounding of zero switches such a
        # channel's gradient on or off, and a fresh Adam turns that into a +-lr step.  The fp32 oracle reproduces the
        # recording bit for bit (test_oracle_trajectory_reproduces_reference_episode), so the restatement is not in
        # question.  Every tensor's update must therefore be within 5e-2 of one of the two trajectories and within
        # 0.2 of both.
        _, ex_after, ex_before, ex_eval = exact[task]
        keys = [k for k in want['state'] if sd[k].is_floating_point() and not k.endswith(NOISE_GRADS)
                and not k.startswith('pets_emas.')]
        e64 = {k: delta_err(sd[k], init[k], ex_after[k], ex_before[k]) for k in keys}
        e32 = {k: delta_err(sd[k], init[k], want['state'][k], want_init[k]) for k in keys}
        errs = sorted(((min(e64[k], e32[k]), e64[k], e32[k], k) for k in keys), reverse=True)
        assert errs[0][0] < 5e-2, "task %d: updates (min, vs fp64, vs reference): %s" % (task, errs[:8])
        assert max(max(e64.values()), max(e32.values())) < 0.2, (task, errs[:8])
        ema_keys = [k for k in want['state'] if k.startswith('pets_emas.')]
        assert ema_keys and max(compact_err(sd[k], want['state'][k]) for k in ema_keys) < 1e-3

        # eval: EMA-ensemble forward (meta_archs.py:854-881), decode, soft-NMS
        model.eval()
        clip = cases.episode_batches(task)[0][0]
        with torch.no_grad():
            raw = model([clip], task_id=task, is_training=False, get_emb=True)
            res = model([clip], task_id=task, is_training=False)[0]
        # outputs of a TRAINED model: the two legitimate executions (see above) differ from each other by `spread`;
        # the HIP path's outputs must be no further from either than 1.5 x that (or 1e-2 where they coincide)
        for a, b, c in zip(raw[0], want['eval_cls_logits'], ex_eval[0]):
            e, spread = (rel_err(a, c), rel_err(a, b)), rel_err(b, c)
            assert max(e) < max(1e-2, 1.5 * spread) and max(e) < 1e-1, (e, spread)
        for a, b, c in zip(raw[1], want['eval_offsets'], ex_eval[1]):
            e, spread = (rel_err(a, c, 1e-6), rel_err(a, b, 1e-6)), rel_err(b, c, 1e-6)
            assert max(e) < max(1e-2, 1.5 * spread) and max(e) < 1e-1, (e, spread)
        wi = want['inference']
        assert res['segments'].shape == wi['segments'].shape
        assert rel_err(res['scores'], wi['scores']) < 5e-2
        agree = (res['labels'] == wi['labels']).float().mean().item()
        assert agree >= 0.9, agree           # near-tied scores swap neighbours

        # between the tasks: memory, n_known, head growth, NEW optimizer + scheduler (train_cl.py:343-389)
        random.seed(0)
        model.add_samples_to_mem(None, _task_data(task), cfg['cl_cfg']['memory_size'] // model.cls_head.cls_head.conv.out_channels)
        model.n_known = len(model.memory)
        assert model.n_known == want['n_known']
        if task == 0:
            torch.manual_seed(99)
            model.augment_classification(cases.EP_NEW, dev)
            sd = model.state_dict()
            for k, w in want['post_augment'].items():
                assert sd[k].shape == w.shape
                n_old = cases.EP_NCLS0
                assert rel_err(sd[k][:n_old], w[:n_old]) < 2e-2 if 'cls_head' in k else True     # trained rows
                assert torch.equal(sd[k][n_old:].cpu(), w[n_old:]), k                            # fresh rows: same init stream
            init = {k: v.detach().clone() for k, v in model.state_dict().items()}
            want_init = dict(want['state'])
            want_init.update(want['post_augment'])
            opt = make_optimizer(model, cfg['opt'])
            sch = make_scheduler(opt, cfg['opt'], len(cases.episode_batches(1)))
            model.train()


@pytest.mark.gpu
def test_single_part_weight_gradients_vs_fp64_oracle(dev):
    """The arithmetic the benchmark runs: weight-gradient products whose contraction spans >= 2048 tokens use the leading
    fp16 part of both operands (ops.dw_precision = 4, one MFMA per product).  The episode case has B * T = 2 * 1024 = 2048
    rows, so its first iteration takes that path in every Linear / 1x1-conv layer of the stem and the XLNet layer: every
    gradient against the float64 ORACLE at the 1e-3 bar of the north star (not against the HIP path's own 3-MFMA result)."""
    from oracle import mq_oracle
    from vilco_amd import ops
    gold = load_episode_golden()
    cfg, model = _build(gold, dev)
    model.train()
    assert ops.dw_precision == 4 and ops.DW_FAST_MIN_K <= 2048 and ops.get_precision() == 3
    fast = []
    real = ops.gemm

    def spy(A, B, Cc, M, N, K, *a, **k):
        if k.get("precision") == 4:
            fast.append((M, N, K))
        return real(A, B, Cc, M, N, K, *a, **k)
    ops.gemm = spy
    try:
        vl = cases.episode_batches(0)[0]
        losses = model(vl, task_id=0, is_training=True)
        losses['final_loss'].backward()
    finally:
        ops.gemm = real
    assert len(fast) >= 20 and all(K >= 2048 for _, _, K in fast), fast[:5]
    p = {k: (v.double() if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point())
         for k, v in episode_full_state(gold['init_state']).items()}
    vl64 = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()} for d in vl]
    want, _ = mq_oracle.forward_losses(p, cfg['model'], vl64, task_id=0, n_known=0)
    want['final_loss'].backward()
    for k in ('cls_loss', 'reg_loss', 'al_loss', 'final_loss'):
        assert rel_err(losses[k], want[k]) < 1e-3, (k, float(losses[k]), float(want[k]))
    errs = []
    for k, q in model.named_parameters():
        if p[k].grad is not None and q.grad is not None and not k.endswith(NOISE_GRADS):
            errs.append((rel_err(q.grad, p[k].grad, GRAD_FLOOR), k))
    errs.sort(reverse=True)
    assert len(errs) > 250 and errs[0][0] < 1e-3, errs[:6]


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_run_episodes_end_to_end(dev, tmp_path, use_graph):
    """vilco_amd.train_cl.run_episodes itself over the two-task stream (MQ/train_cl.py:206-389): initial / in-epoch / final
    validation calls, best-checkpoint files with the reference's keys (:300-307), the memory pickle (:355-361), the
    best-checkpoint reload (:363; here the best epoch is NOT the last, so the reload changes the weights), class-head
    growth and a new optimizer for task 1 -- eager and replayed as hipGraphs."""
    import pickle
    from vilco_amd.train_cl import run_episodes
    from vilco_amd.utils.cl_stream import InMemoryQILStream
    gold = load_episode_golden()
    cfg, model = _build(gold, dev)
    cfg = dict(cfg, opt=dict(cfg['opt'], epochs=2, warmup_epochs=1))            # 3 epochs per task, validation from epoch 1
    cfg['cl_cfg'] = dict(cfg['cl_cfg'], path_memory='memory.pkl')
    stream = InMemoryQILStream([_task_data(0), _task_data(1)], batch_size=2, seed=3)
    random.seed(0)
    calls, snaps = [], {}

    def validate(m, epoch, task):
        calls.append((task, epoch))
        # epoch 1 scores best: the checkpoint written then must be what the task ends with
        if len([c for c in calls if c[0] == task]) == 2:          # (first call of a task is the incoming-model validation)
            snaps[task] = {k: v.detach().clone() for k, v in m.state_dict().items()}
            return 1.0
        return 0.1
    folder = str(tmp_path)
    model, opt, sch, log = run_episodes(cfg, model, stream, validate=validate, ckpt_folder=folder, gpu_id=0,
                                        use_graph=use_graph, keep_history=True)
    assert [t for t, _ in calls] == [0] * 4 + [1] * 4 and [e for _, e in calls] == [0, 1, 2, 2, 0, 1, 2, 2], calls
    assert [e['best_epoch'] for e in log] == [1, 1] and [e['best_metric'] for e in log] == [1.0, 1.0]
    for task in (0, 1):
        ck = torch.load(os.path.join(folder, 'best_task_%03d_performance.pth.tar' % task), weights_only=False)
        assert sorted(ck) == ['epoch', 'optimizer', 'reg_params', 'scheduler', 'state_dict', 'task']
        assert ck['task'] == task and ck['epoch'] == 1
        assert sorted(ck['optimizer']) == ['param_groups', 'state'] and len(ck['optimizer']['param_groups']) == 3
        steps = {float(s['step']) for s in ck['optimizer']['state'].values()}
        n_it = len(log[task]['history'][0])
        assert min(steps) == 2 * n_it, (steps, n_it)             # two epochs of updates at the time of the checkpoint
        for k, v in snaps[task].items():
            assert torch.equal(ck['state_dict'][k].to(dev), v), k
    # the model the run ends with is task 1's BEST state (epoch 1), not its last epoch's
    sd = model.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in snaps[1].items())
    with open(os.path.join(folder, 'memory.pkl'), 'rb') as h:
        mem = pickle.load(h)
    assert sorted(mem) == sorted(model.memory) and model.n_known == len(mem) == cases.EP_NCLS0 + cases.EP_NEW
    assert all(len(v) <= max(1, cfg['cl_cfg']['memory_size'] // model.cls_head.cls_head.conv.out_channels) for v in mem.values())
    assert model.cls_head.cls_head.conv.out_channels == cases.EP_NCLS0 + cases.EP_NEW
    hist = [h for e in log for ep in e['history'] for h in ep]
    assert len(hist) == 3 * (len(log[0]['history'][0]) + len(log[1]['history'][0]))
    assert all(bool(torch.isfinite(h['final_loss'])) for h in hist)
