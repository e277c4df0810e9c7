"""Golden vectors of the NLQ EPISODE path from the IMPORTED REFERENCE (this container only).
Run:  python tests/golden/make_golden_nlq_episode.py   ->  tests/golden/nlq_episode.pt, nlq_train_glue.json

The reference's own functions are driven as NLQ/train_cl.py:113-342 drives them, over the three-task case of
cases.nlq_episode_*: `make_optimizer(model, opt, head_backbone_group=True)` (NLQ/libs/utils/train_utils.py:63-240; the
branch train_cl.py:115-118 takes when backbone_lr_weight != 1), `make_scheduler`, the reference `train_one_epoch`
(:376-521) for every epoch, `valid_one_epoch_cl_single_gpu` (:705-781) with a recording evaluator, best-state bookkeeping
(R1 >= best_R1, :265-266), `add_samples_to_mem` with m = memory_size // 13 (:293-303), n_known = j + 1, reload of the
best state (:311, :321), and a NEW optimizer + scheduler per task (:331-336).

Recorded: parameter-group membership and hyper-parameters of both optimizer modes (nlq_train_glue.json), per task every
iteration's loss dict and per-group learning rates, the validation records handed to the evaluator (the reference's
`results` list of {query_idx, annotation_uid, predicted_times, clip_uid}), R1 sequence / best epoch, memory ids, the
model state after the task."""
import copy
import importlib
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import make_golden_nlq_model as mgm  # noqa: E402

NLQ = '/root/reference/NLQ'


class Loader:
    """what train_one_epoch touches of a DataLoader: len, iteration, sampler.set_epoch"""

    class _S:
        def set_epoch(self, e):
            self.epoch = e

    def __init__(self, batches):
        self.batches, self.sampler = batches, Loader._S()

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter([[dict(x) for x in b] for b in self.batches])


def flatten(comp):
    seen, out = set(), []
    for videos in comp.values():
        for v in videos:
            if v['query_id'] not in seen:
                seen.add(v['query_id'])
                out.append(v)
    return out


def batches_of(items, bs):
    return [items[i:i + bs] for i in range(0, len(items) - bs + 1, bs)]


class ValTasks:
    """get_valSet_by_taskNum contract (cl_benchmark.py:60-74): [(loader over template k's queries, batch 1, #templates)]"""

    def get_valSet_by_taskNum(self, n):
        return [(Loader([[q] for q in list(cases.nlq_episode_data(k).values())[0]]), 1) for k in range(n)]


class Recorder:
    dataset = "ego4d_cl"

    def __init__(self):
        self.calls = []

    def evaluate(self, results, verbose=True):
        self.calls.append(copy.deepcopy(results))
        return np.array([[cases.nlq_episode_metric(results)]]), ""


def ref_train_utils(M):
    sys.path.insert(0, NLQ)                                   # basic_utils.py
    B = sys.modules['nlq_libs.modeling.blocks']
    for n in ('MaskedConv1D', 'Scale', 'AffineDropPath', 'LayerNorm'):
        setattr(sys.modules['nlq_libs.modeling'], n, getattr(B, n))
    return importlib.import_module('nlq_libs.utils.train_utils')


def group_dump(model, opt):
    names = {id(p): n for n, p in model.named_parameters()}
    return [{'names': [names[id(p)] for p in g['params']], 'weight_decay': g['weight_decay'], 'lr': g['lr']} for g in opt.param_groups]


def main():
    os.environ["LOCAL_RANK"] = "0"
    M = mgm.ref_modules()
    TU = ref_train_utils(M)
    cfg = cases.nlq_model_cfg()
    torch.manual_seed(177)
    model = M.PtTransformer(**cfg)
    mgm.perturb(model, 178)
    model.use_adapter = False
    glue = {}
    for name, (hb, w) in (('default', (False, 1)), ('head_backbone', (True, 0.5))):
        opt = TU.make_optimizer(model, cases.nlq_episode_opt(w), head_backbone_group=hb)
        glue[name] = group_dump(model, opt)
    with open(os.path.join(HERE, 'nlq_train_glue.json'), 'w') as f:
        json.dump(glue, f, indent=1)

    opt_cfg = cases.nlq_episode_opt(0.5)
    out = {'init_state': {k: v.clone() for k, v in model.state_dict().items()}, 'tasks': []}
    seen = []
    orig_forward = model.forward

    def recording_forward(*a, **k):
        r = orig_forward(*a, **k)
        if isinstance(r, dict) and 'final_loss' in r:
            seen.append({kk: float(v) for kk, v in r.items()})
        return r
    model.forward = recording_forward

    val_tasks, memory = ValTasks(), {}
    optimizer = TU.make_optimizer(model, opt_cfg, head_backbone_group=True)
    iters = len(batches_of(flatten(cases.nlq_episode_data(0)), cases.NLQ_EP_BATCH))
    scheduler = TU.make_scheduler(optimizer, opt_cfg, iters)
    max_epochs = opt_cfg['epochs'] + opt_cfg['warmup_epochs']
    for j in range(cases.NLQ_EP_TASKS):
        data = cases.nlq_episode_data(j)
        loader = Loader(batches_of(flatten({**memory, **data}), cases.NLQ_EP_BATCH))
        rec_eval = Recorder()
        with torch.no_grad():
            best = TU.valid_one_epoch_cl_single_gpu(val_tasks, model, 0, j, evaluator=rec_eval, print_freq=1000)
        torch.set_grad_enabled(True)
        rec = {'init_R1': float(best), 'losses': [], 'lrs': [], 'R1': [], 'n_batches': len(loader)}
        best_state, best_epoch = None, -1
        lrs = []
        orig_step = scheduler.step

        def step_rec(*a, **k):
            lrs.append([g['lr'] for g in optimizer.param_groups])
            return orig_step(*a, **k)
        scheduler.step = step_rec
        for epoch in range(max_epochs):
            seen.clear()
            TU.train_one_epoch(loader, model, optimizer, scheduler, epoch, model_ema=None, clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'],
                               tb_writer=None, print_freq=1000, cl_name=cfg['cl_cfg']['name'], reg_lambda=0.0,
                               prev_out_cls_logits_dict={}, current_task_id=j)
            rec['losses'].append(list(seen))
            with torch.no_grad():
                r1 = TU.valid_one_epoch_cl_single_gpu(val_tasks, model, epoch, j, evaluator=rec_eval, print_freq=1000)
            torch.set_grad_enabled(True)
            rec['R1'].append(float(r1))
            if r1 >= best:
                best, best_epoch = r1, epoch
                best_state = {k: v.clone() for k, v in model.state_dict().items()}
        rec['lrs'] = lrs
        rec['best_epoch'] = best_epoch
        random.seed(1000 + j)
        model.add_samples_to_mem(val_tasks, {k: list(v) for k, v in data.items()}, cases.NLQ_EP_MEMORY // 13)
        memory = model.memory
        model.n_known = j + 1
        rec['memory_ids'] = {c: [v['query_id'] for v in vs] for c, vs in model.memory.items()}
        if best_state is not None:
            model.load_state_dict(best_state)
        rec['state'] = {k: v.clone() for k, v in model.state_dict().items()}
        final = Recorder()
        with torch.no_grad():
            TU.final_validate(val_tasks, model, max_epochs - 1, j, evaluator=final, print_freq=1000,
                              list_val_recall_ii={'val': [0.0] * (j + 1), 'test': []}, type_val='val')
        torch.set_grad_enabled(True)
        rec['results'] = final.calls[-1]
        out['tasks'].append(rec)
        if j + 1 < cases.NLQ_EP_TASKS:
            optimizer = TU.make_optimizer(model, opt_cfg, head_backbone_group=True)
            scheduler = TU.make_scheduler(optimizer, opt_cfg, iters)
        print('task', j, 'R1', rec['init_R1'], rec['R1'], 'best', best_epoch, 'loss', [[round(l['final_loss'], 4) for l in e] for e in rec['losses']],
              'mem', rec['memory_ids'])
    path = os.path.join(HERE, 'nlq_episode.pt')
    torch.save(out, path)
    print('%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
