"""Episode (task) loop of query-incremental continual learning on the HIP path -- our counterpart of the
reference driver MQ/train_cl.py:206-389 (+ load_best_checkpoint :31-40), written around the drop-in model:

  for every task j of the stream:
      validate the incoming model                                   (train_cl.py:209-219)
      [type_sampling 'icarl': cache sigmoid(logits) of the task's clips, :226-235]
      for every epoch: sampler.set_epoch, pre_train_epoch (adapters), train_one_epoch, validate after the first
          third of the epochs, keep the best checkpoint             (:237-316)
      replay memory: m = memory_size // #classes per class, add_samples_to_mem, n_known = len(memory), memory
          pickle                                                    (:343-361)
      reload the task's best checkpoint, final validation           (:363-365)
      if another task follows: augment_classification(num_next_classes), EWC / MAS consolidation, NEW optimizer
          and scheduler                                             (:373-389)

What is ours rather than the reference's: the stream is any iterable with QILSetTask's contract
(`(data, loader, num_next_classes)` per task, a `.memory` attribute; utils/cl_stream.py has an in-memory one --
the Ego4D readers are outside the hot path), validation is a callable (the evaluator is outside the hot path;
`train_utils.collect_results` produces its input format), gradients of data-parallel replicas go through
`dist.GradReducer`, and the per-iteration work is `train_utils.train_one_epoch` (fused clip + AdamW, no host sync).
Checkpoints keep the reference's keys: task, epoch, state_dict, scheduler, optimizer, reg_params (:300-307).
"""
import os
import pickle

import numpy as np
import torch

from .cl_methods import regularizers
from .utils.train_utils import make_optimizer, make_scheduler, save_checkpoint, train_one_epoch


def load_best_checkpoint(model, file_folder, file_name, current_task, gpu_id):
    """train_cl.py:31-40: restore state_dict + reg_params when the file exists; the model is returned either way."""
    path = os.path.join(file_folder, file_name)
    if os.path.exists(path):
        ck = torch.load(path, map_location=lambda storage, loc: storage.cuda(gpu_id), weights_only=False)
        model.load_state_dict(ck['state_dict'])
        model.reg_params = ck['reg_params']
        model = model.cuda(gpu_id)
    return model


def memory_quota(memory_size, n_classes):
    """videos kept per class after a task (train_cl.py:343-349)"""
    return 'ALL' if memory_size == 'ALL' else memory_size // n_classes


@torch.no_grad()
def cache_prev_logits(model, loader, task_id, as_numpy=False):
    """{video_id: [sigmoid(cls logits) per pyramid level]} of the incoming model (train_cl.py:226-235): the distillation
    targets of iCaRL / BiC.  They stay ON THE DEVICE (one small tensor per level and clip; the reference round-trips them
    through numpy and uploads them again in every iteration, meta_archs.py:1492,1508); as_numpy=True gives the
    reference's format."""
    out = {}
    for video_list in loader:
        cls_logits, _, _ = model(video_list, task_id=task_id, get_emb=True)
        for i, v in enumerate(video_list):
            probs = [torch.sigmoid(lvl[i]).clone() for lvl in cls_logits]
            out[v['video_id']] = [np.array(t.cpu().numpy()) for t in probs] if as_numpy else probs
    return out


def _barrier(reducer):
    if reducer is not None:
        import torch.distributed as dist
        dist.barrier(group=reducer.group)


def run_episodes(cfg, model, train_stream, validate=None, ckpt_folder=None, gpu_id=0, start_task=0, start_epoch=0,
                 combine_train=False, reducer=None, logger=None, print_freq=20, on_step_history=None, use_graph=False,
                 keep_history=True):
    """cfg: the merged config dict (opt / train_cfg / cl_cfg as in libs/core/config.py).
    validate(model, epoch, task) -> mAP-like float (higher is better), or None to skip validation (every epoch's
    state then counts as "best", so the last epoch is what gets reloaded).  `validate` must be RANK-LOCAL (no
    collectives): inside the epoch loop only the main rank calls it (train_cl.py:283), the others wait at a barrier.
    use_graph: replay the iterations as hipGraphs (vilco_amd/graph.py; a fresh GraphedStep per optimizer, i.e. per task).
    keep_history: keep every iteration's loss dict (device scalars) in the log; off for long runs.
    Returns (model, optimizer, scheduler, log) with log = list of per-task dicts."""
    is_main = int(os.environ.get("LOCAL_RANK", "0")) == 0
    optimizer = make_optimizer(model, cfg['opt'])

    def make_graph(opt):
        if not use_graph:
            return None
        from .graph import GraphedStep
        return GraphedStep(model, opt, clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], reducer=reducer)
    graph = make_graph(optimizer)
    it = iter(train_stream)
    num_tasks = train_stream.num_tasks
    data, loader, num_next = next(it)
    iters_per_epoch = len(loader)
    scheduler = make_scheduler(optimizer, cfg['opt'], iters_per_epoch)
    max_epochs = cfg['opt'].get('early_stop_epochs', cfg['opt']['epochs'] + cfg['opt']['warmup_epochs'])
    memory_size = cfg['cl_cfg']['memory_size']
    log = []
    for j in range(start_task, num_tasks):
        if j != 0:
            data, loader, num_next = next(it)
        entry = {'task': j, 'init_metric': None, 'best_metric': None, 'best_epoch': -1, 'history': []}
        if validate is not None:
            entry['init_metric'] = validate(model, 0, j)
        best, best_epoch = -10000.0, -1
        prev_logits = cache_prev_logits(model, loader, j) if model.type_sampling == 'icarl' else {}
        ck_name = 'best_task_{:03d}_performance.pth.tar'.format(j)
        for epoch in range(start_epoch, max_epochs):
            sampler = getattr(loader, 'sampler', None)
            if sampler is not None and hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)
            if model.use_adapt:
                model.pre_train_epoch(task_id=j, current_epoch=epoch)
            hist = train_one_epoch(loader, model, optimizer, scheduler, epoch, 1, model_ema=None,
                                   clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], print_freq=print_freq,
                                   logger=logger, cl_name=cfg['cl_cfg']['name'], reg_lambda=cfg['cl_cfg']['reg_lambda'],
                                   prev_out_cls_logits_dict=prev_logits, current_task_id=j, reducer=reducer, graph=graph,
                                   keep_history=keep_history or on_step_history is not None)
            if keep_history:
                entry['history'].append(hist)
            if on_step_history is not None:
                on_step_history(j, epoch, hist)
            if combine_train or epoch < max_epochs // 3:
                continue
            if is_main:
                metric = validate(model, epoch, j) if validate is not None else float(epoch)
                if metric > best:
                    best, best_epoch = metric, epoch
                    if ckpt_folder is not None:
                        save_checkpoint({'task': j, 'epoch': epoch, 'state_dict': model.state_dict(),
                                         'scheduler': scheduler.state_dict(), 'optimizer': optimizer.state_dict(),
                                         'reg_params': model.reg_params}, file_folder=ckpt_folder, file_name=ck_name)
            _barrier(reducer)          # the other ranks do not run ahead into the next epoch's collectives while rank 0
                                       # validates and writes the checkpoint
        entry['best_metric'], entry['best_epoch'] = best, best_epoch

        # replay memory for the next task (train_cl.py:343-361)
        n_cls = model.cls_head.cls_head.conv.out_channels
        if memory_size != 0:
            model.add_samples_to_mem(None, data, memory_quota(memory_size, n_cls))
        train_stream.memory = model.memory
        model.n_known = len(model.memory)
        if ckpt_folder is not None:
            if is_main:
                os.makedirs(ckpt_folder, exist_ok=True)
                with open(os.path.join(ckpt_folder, cfg['cl_cfg']['path_memory']), 'wb') as h:
                    pickle.dump(model.memory, h)
            _barrier(reducer)          # the checkpoint rank 0 wrote is complete before anybody reads it
            # EVERY rank reloads the task's best state (train_cl.py:363 runs on every rank): replicas that kept their
            # last-epoch weights would never meet the others again -- the gradient average does not re-align weights
            model = load_best_checkpoint(model, ckpt_folder, ck_name, j, gpu_id)
        if validate is not None:
            entry['final_metric'] = validate(model, max_epochs - 1, j)
        log.append(entry)

        if num_next is not None:
            model.augment_classification(num_next, torch.device('cuda', gpu_id))
            if cfg['cl_cfg']['name'] == 'ewc':
                model.reg_params = regularizers.on_task_update(loader, gpu_id, optimizer, model, kind='ewc', group=getattr(reducer, 'group', None), data_parallel=reducer is not None)
            elif cfg['cl_cfg']['name'] == 'mas':
                model.reg_params = regularizers.on_task_update(loader, gpu_id, optimizer, model, kind='mas', group=getattr(reducer, 'group', None), data_parallel=reducer is not None)
            optimizer = make_optimizer(model, cfg['opt'])
            scheduler = make_scheduler(optimizer, cfg['opt'], iters_per_epoch)
            if reducer is not None:
                reducer.rebuild()          # the class head and the gaussian parameters are new tensors
            graph = make_graph(optimizer)  # new parameters, new optimizer: the old task's graphs are dropped
    return model, optimizer, scheduler, log


class StickyBest:
    """Which epochs write 'Best_task_XX' in the NLQ driver (NLQ/train_cl.py:216-292).  `is_best = R1 >= best_R1` is assigned
    only at validated epochs (:250) but the save block tests it at EVERY epoch (:283), and nothing resets it between epochs or
    tasks: after a validated epoch that reached the bar every following epoch overwrites the file until a validation misses.
    (The reference leaves is_best undefined before the first validation; epoch 0 validates whenever ckpt_freq > 0.)"""

    def __init__(self):
        self.is_best = False
        self.best = None
        self.best_epoch = -1

    def new_task(self, init_r1):
        self.best, self.best_epoch = init_r1, -1          # (is_best carries over, as in the reference)

    def validated(self, epoch, r1):
        self.is_best = r1 >= self.best
        if self.is_best:
            self.best, self.best_epoch = r1, epoch

    @staticmethod
    def validates(epoch, max_epochs, ckpt_freq):
        return epoch == max_epochs - 1 or (ckpt_freq > 0 and epoch % ckpt_freq == 0)


def run_episodes_nlq(cfg, model, train_stream, val_stream, evaluator, ckpt_folder=None, gpu_id=0, start_task=0,
                     start_epoch=0, ckpt_freq=2, reducer=None, print_freq=100, use_graph=False, keep_history=True,
                     on_validate=None):
    """Episode loop of the NLQ driver (BASELINE configs[3]; NLQ/train_cl.py:113-342), which differs from the MQ loop in
    what surrounds the training iterations:

      optimizer: NLQ's make_optimizer, with the head / backbone learning-rate groups when opt.backbone_lr_weight != 1
          (:115-118), created anew for every task (:331-336); the scheduler keeps the FIRST task's iterations per epoch (:123)
      per task j: R@1 of the incoming model over templates 0..j (:181-183) is the bar the epochs have to reach
          (is_best = R1 >= best_R1, :265); validation after epoch e when e is the last one or e % ckpt_freq == 0 (:216-222);
          the best state goes to 'Best_task_{j:02d}.pth.tar' (keys epoch / state_dict / scheduler / optimizer / current_task /
          reg_params, :269-276)
      replay memory: m = memory_size // 13 queries per template (:293-299 -- 13 is the benchmark's template count,
          hard-wired there), add_samples_to_mem, stream.memory = model.memory, n_known = j + 1 (:303-308), memory pickle
      reload of the task's best checkpoint (:315), final validation over all learnt templates (:317-318), and -- when
          another task follows -- EWC / MAS consolidation (:325-329); no class-head growth (one query class)

    val_stream: `get_valSet_by_taskNum(n)` -> [(loader with batch size 1, #templates), ...] (cl_benchmark.py:60-74);
    evaluator: `.dataset`, `.evaluate(records, verbose) -> (performance[[R@1, ...]], str)` (libs/utils/metrics.py
    ReferringRecall; outside the hot path).  on_validate(kind, task, epoch, r1) is called after every validation.
    Returns (model, optimizer, scheduler, log)."""
    from .utils import train_utils_nlq as tu
    is_main = int(os.environ.get("LOCAL_RANK", "0")) == 0
    hb = cfg['opt']["backbone_lr_weight"] != 1

    def new_optimizer():
        return tu.make_optimizer(model, cfg['opt'], head_backbone_group=hb)

    def make_graph(opt):
        if not use_graph:
            return None
        from .graph import GraphedStep
        return GraphedStep(model, opt, clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], reducer=reducer)
    optimizer = new_optimizer()
    graph = make_graph(optimizer)
    it = iter(train_stream)
    num_tasks = train_stream.num_tasks
    data, loader, num_next = next(it)
    iters_per_epoch = len(loader)
    scheduler = tu.make_scheduler(optimizer, cfg['opt'], iters_per_epoch)
    max_epochs = cfg['opt'].get('early_stop_epochs', cfg['opt']['epochs'] + cfg['opt']['warmup_epochs'])
    memory_size = cfg['cl_cfg']['memory_size']
    recalls = {'val': [], 'test': []}
    log = []
    tracker = StickyBest()

    def validate(kind, j, epoch):
        fn = tu.final_validate if kind == 'final' else tu.valid_one_epoch_cl_single_gpu
        kw = dict(list_val_recall_ii=recalls, type_val='val') if kind == 'final' else {}
        r1 = fn(val_stream, model, epoch, j, evaluator=evaluator, print_freq=print_freq, **kw)
        model.train()
        if on_validate is not None:
            on_validate(kind, j, epoch, r1)
        return r1

    for j in range(start_task, num_tasks):
        if j != 0:
            data, loader, num_next = next(it)
        entry = {'task': j, 'history': [], 'R1': []}
        entry['init_R1'] = validate('init', j, 0)
        tracker.new_task(entry['init_R1'])
        prev_logits = cache_prev_logits(model, loader, j, as_numpy=True) if model.type_sampling == 'icarl' else {}
        ck_name = 'Best_task_{:02d}.pth.tar'.format(j)
        for epoch in range(start_epoch, max_epochs):
            sampler = getattr(loader, 'sampler', None)
            if sampler is not None and hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)
            if model.use_adapter:
                model.pre_train_epoch(task_id=j, current_epoch=epoch)
            hist = tu.train_one_epoch(loader, model, optimizer, scheduler, epoch, model_ema=None,
                                      clip_grad_l2norm=cfg['train_cfg']['clip_grad_l2norm'], print_freq=print_freq,
                                      cl_name=cfg['cl_cfg']['name'], reg_lambda=cfg['cl_cfg']['reg_lambda'],
                                      prev_out_cls_logits_dict=prev_logits, current_task_id=j, reducer=reducer, graph=graph,
                                      keep_history=keep_history)
            if keep_history:
                entry['history'].append(hist)
            validated = StickyBest.validates(epoch, max_epochs, ckpt_freq)
            if validated:
                r1 = validate('epoch', j, epoch)
                entry['R1'].append((epoch, r1))
                tracker.validated(epoch, r1)
            # The reference's save block sits OUTSIDE the validation `if` and `is_best` is sticky (StickyBest): with ckpt_freq = 2
            # the checkpoint reloaded for final validation, memory selection and the next task is one (unvalidated) epoch newer
            # than the best validated one.  Reproduced here; `saved_epoch` records which epoch the file holds.
            is_best = tracker.is_best
            if is_best:
                entry['saved_epoch'] = epoch
                if is_main and ckpt_folder is not None:
                    save_checkpoint({'epoch': epoch, 'state_dict': model.state_dict(), 'scheduler': scheduler.state_dict(),
                                     'optimizer': optimizer.state_dict(), 'current_task': j, 'reg_params': model.reg_params},
                                    file_folder=ckpt_folder, file_name=ck_name)
            if validated or is_best:
                _barrier(reducer)
        entry['best_R1'], entry['best_epoch'] = tracker.best, tracker.best_epoch
        if memory_size != 0:
            model.add_samples_to_mem(val_stream, data, 'ALL' if memory_size == 'ALL' else memory_size // 13)
        train_stream.memory = model.memory
        model.n_known = j + 1
        if ckpt_folder is not None:
            if is_main:
                os.makedirs(ckpt_folder, exist_ok=True)
                with open(os.path.join(ckpt_folder, cfg['cl_cfg']['path_memory']), 'wb') as h:
                    pickle.dump(model.memory, h)
            _barrier(reducer)
            model = load_best_checkpoint(model, ckpt_folder, ck_name, j, gpu_id)
        entry['final_R1'] = validate('final', j, max_epochs - 1)
        log.append(entry)
        if num_next is not None:
            if cfg['cl_cfg']['name'] in ('ewc', 'mas'):
                model.reg_params = regularizers.on_task_update(loader, gpu_id, optimizer, model, kind=cfg['cl_cfg']['name'],
                                                               group=getattr(reducer, 'group', None), data_parallel=reducer is not None)
            optimizer = new_optimizer()
            scheduler = tu.make_scheduler(optimizer, cfg['opt'], iters_per_epoch)
            graph = make_graph(optimizer)
    return model, optimizer, scheduler, log
