"""Step glue of the MQ training loop on the HIP path (reference:
MQ/libs/utils/train_utils.py -- make_optimizer :68-144, make_scheduler :147-211, train_one_epoch :318-357,
save_checkpoint :54-59, fix_random_seed :33-51).

`make_optimizer` reproduces the reference's parameter grouping (decay / no-decay / "remain") and returns a
`FusedOptimizer`: an ordinary torch.optim.Optimizer (param_groups, state_dict with torch.optim.AdamW's layout,
LR schedulers attach to it) whose `step()` is three multi-tensor HIP launches -- global-norm clip coefficient
(kept on the device) + AdamW/SGD update -- instead of ~1500 small kernels."""
import ctypes as C
import os
import random

import numpy as np
import torch
from torch import optim

from .. import _lib, ops
from ..modeling.blocks import AffineDropPath, LayerNorm, MaskedConv1D, Scale
from .lr_schedulers import LinearWarmupCosineAnnealingLR, LinearWarmupMultiStepLR

CHUNK = 16384
OPT_AMAX = os.environ.get("VILCO_OPT_AMAX", "1") != "0"      # the update kernels leave max|w| per chunk for the weight packs


def fix_random_seed(seed, include_cuda=True):
    rng = torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if include_cuda and torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return rng


def save_checkpoint(state, file_folder, file_name='checkpoint.pth.tar'):
    os.makedirs(file_folder, exist_ok=True)
    torch.save(state, os.path.join(file_folder, file_name))


def param_groups(model):
    """(decay, no_decay, remain) sorted name lists, following the rule chain of train_utils.py:74-103.
    `m.named_parameters()` is recursive there, so a parameter is classified by every (ancestor module,
    relative name) pair that matches a rule; what matches nothing ends up in `remain` (weight-decayed)."""
    decay, no_decay = set(), set()
    white = (torch.nn.Linear, torch.nn.Conv1d, MaskedConv1D)
    black = (LayerNorm, torch.nn.GroupNorm)
    for mn, m in model.named_modules():
        for pn, _ in m.named_parameters():
            full = '%s.%s' % (mn, pn) if mn else pn
            if pn.endswith('bias'):
                no_decay.add(full)
            elif 'xlnet' in pn:
                (no_decay if 'norm' in pn else decay).add(full)
            elif pn.endswith('weight') and isinstance(m, white):
                decay.add(full)
            elif pn.endswith('weight') and isinstance(m, black):
                no_decay.add(full)
            elif pn.endswith('scale') and isinstance(m, (Scale, AffineDropPath)):
                no_decay.add(full)
            elif pn.endswith('rel_pe'):
                no_decay.add(full)
    names = {pn for pn, _ in model.named_parameters()}
    remain = names - (decay | no_decay)
    return sorted(decay), sorted(no_decay), sorted(remain)


class FusedOptimizer(optim.Optimizer):
    """AdamW (torch.optim.AdamW defaults: betas (0.9, 0.999), eps 1e-8) or SGD+momentum over fp32 CUDA
    parameters, executed by vilco_grad_norm / vilco_optim_step.

    Everything a step needs lives on the device between steps: the chunk tables, the [4, n] pointer table (rebuilt only when
    a parameter, gradient or state tensor moved), the per-tensor step counts (incremented by a device add) and -- when a
    `lr_dev` tensor is passed -- the learning rates.  A step is then three launches with no host -> device copy, which is
    what lets vilco_amd/graph.py capture it as a hipGraph; `state[p]['step']` (torch.optim's layout) is brought up to date
    lazily (`state_dict()`, plan changes)."""

    def __init__(self, params, lr, kind="AdamW", momentum=0.9, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        assert kind in ("AdamW", "SGD")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, momentum=momentum))
        self.kind = kind
        self._plans = None              # list over passes, valid for self._plans_key
        self._plans_key = None
        self.last_grad_norm = None      # device tensor [norm, clip coef] of the last step

    def _passes(self):
        """[(param, group index)] lists; a parameter listed k times (the adapter aliases of train_utils.py:108-113)
        is stepped k times per iteration like torch.optim would: k-th occurrences go to pass k."""
        seen, passes = {}, []
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if p.grad is None:
                    continue
                k = seen.get(id(p), 0)
                seen[id(p)] = k + 1
                while len(passes) <= k:
                    passes.append([])
                passes[k].append((p, gi))
        return passes, seen

    def _build_plan(self, k, items, occurrences):
        numel = [p.numel() for p, _ in items]
        ct, co, first = [], [], []
        for i, n in enumerate(numel):
            first.append(len(ct))
            for off in range(0, n, CHUNK):
                ct.append(i)
                co.append(off)
        first.append(len(ct))
        dev = items[0][0].device
        amax = torch.empty(max(len(ct), 1), dtype=torch.float32, device=dev)
        inc = [float(occurrences[id(p)]) for p, _ in items]
        base = [float(self.state[p]['step']) for p, _ in items]
        # step count handed to the kernel = count AFTER this pass's increment: a parameter listed m times goes
        # s -> s+1 (pass 0) -> s+2 (pass 1) ... within one iteration and by m from iteration to iteration
        tstep0 = [b + (k + 1) - m for b, m in zip(base, inc)]
        uniform = all(m == 1.0 for m in inc)
        return dict(items=items, first=first, amax=amax,
                    amax_views=[amax[first[i]:first[i + 1]] for i in range(len(numel))],
                    numel=torch.tensor(numel, dtype=torch.int64, device=dev),
                    chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                    chunk_off=torch.tensor(co, dtype=torch.int64, device=dev),
                    group=torch.tensor([gi for _, gi in items], dtype=torch.int32, device=dev),
                    partial=torch.empty(max(len(ct), 1), dtype=torch.float32, device=dev), nchunks=len(ct),
                    tstep=torch.tensor(tstep0, dtype=torch.float32, device=dev),
                    inc=None if uniform else torch.tensor(inc, dtype=torch.float32, device=dev),
                    inc_host=inc, base=base, count=0, tables={})

    def _sync_steps(self):
        """bring state[p]['step'] (host tensors, torch.optim's layout) up to date with the device-side counts"""
        if self._plans:
            pl = self._plans[0]                  # pass 0 lists every stepped parameter exactly once
            if pl['count']:
                for (p, _), b, m in zip(pl['items'], pl['base'], pl['inc_host']):
                    self.state[p]['step'] = torch.tensor(b + m * pl['count'])

    def _flush_steps(self):
        self._sync_steps()
        self._plans, self._plans_key = None, None

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._plans, self._plans_key = None, None
        return super().load_state_dict(state_dict)

    def _ensure_state(self, items):
        for p, _ in items:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("FusedOptimizer needs contiguous fp32 parameters on the HIP device")
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0)
                if self.kind == "AdamW":
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                else:
                    st['momentum_buffer'] = torch.zeros_like(p)

    MAX_EAGER_TABLES = 2        # pointer tables kept per pass outside captured graphs

    def prepare_step(self, pin=False):
        """(re)build whatever the next step() needs that costs a host -> device copy: chunk plans when the set of stepped
        parameters changed, pointer tables when a tensor moved.  step() calls it itself; a caller about to CAPTURE step()
        calls it first, outside the capture, with pin=True: a captured launch holds the table's address, so pinned tables
        are never evicted.  Unpinned tables (plain eager training, where gradient addresses change with the allocation
        pattern, e.g. ragged text lengths) are a small LRU; every table is keyed by the tuple of all four pointer rows
        and uploaded from page-locked memory without blocking the host."""
        passes, occ = self._passes()
        key = tuple(tuple(id(p) for p, _ in items) for items in passes)
        if key != self._plans_key:
            self._flush_steps()
            for items in passes:
                self._ensure_state(items)
            self._plans = [self._build_plan(k, items, occ) for k, items in enumerate(passes)]
            self._plans_key = key
        s1 = 'exp_avg' if self.kind == "AdamW" else 'momentum_buffer'
        adam = self.kind == "AdamW"
        tables = []
        for pl in self._plans:
            items = pl['items']
            tkey = (tuple(p.data_ptr() for p, _ in items), tuple(p.grad.data_ptr() for p, _ in items),
                    tuple(self.state[p][s1].data_ptr() for p, _ in items),
                    tuple(self.state[p]['exp_avg_sq'].data_ptr() if adam else 0 for p, _ in items))
            cache = pl['tables']
            ent = cache.pop(tkey, None)
            if ent is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedOptimizer.step() under stream capture needs prepare_step() before the capture")
                host = torch.tensor(tkey, dtype=torch.int64)
                dev = items[0][0].device
                if dev.type == 'cuda':
                    host = host.pin_memory()
                ent = {'dev': host.to(dev, non_blocking=True), 'host': host, 'pinned': False}
            ent['pinned'] = ent['pinned'] or bool(pin)
            cache[tkey] = ent                     # (re)inserted last: dict order is the LRU order
            loose = [k for k, e in cache.items() if not e['pinned']]
            for k in loose[:max(len(loose) - self.MAX_EAGER_TABLES, 0)]:
                del cache[k]
            tables.append(ent['dev'])
        return tables

    @torch.no_grad()
    def step(self, closure=None, clip_grad_l2norm=-1.0, lr_dev=None):
        """lr_dev: optional device float tensor [len(param_groups)] the kernels read the learning rates from (a captured
        step); default: the param_groups' values, passed by value."""
        lib = _lib.load()
        stream = ops._stream()
        tables = self.prepare_step()
        ng = len(self.param_groups)
        lr = (C.c_float * ng)(*[float(g['lr']) for g in self.param_groups])
        wd = (C.c_float * ng)(*[float(g['weight_decay']) for g in self.param_groups])
        g0 = self.param_groups[0]
        emitted = {}     # id(p) -> (p, partials view, count) of the LAST pass that wrote p
        coef = None      # clip coefficient of pass 0, reused by every pass: clip_grad_norm_ scales p.grad in place
                         # (train_utils.py:343-347), so a parameter listed twice is stepped twice with the CLIPPED gradient
        for k, (plan, ptrs) in enumerate(zip(self._plans, tables)):
            items = plan['items']
            n = len(items)
            if k == 0:
                coef = plan.get('coef')
                if coef is None:
                    coef = plan['coef'] = torch.empty(2, dtype=torch.float32, device=items[0][0].device)
                _lib.check(lib.vilco_grad_norm(ptrs.data_ptr(), plan['numel'].data_ptr(), plan['chunk_tensor'].data_ptr(),
                                               plan['chunk_off'].data_ptr(), n, plan['nchunks'], CHUNK,
                                               float(clip_grad_l2norm), plan['partial'].data_ptr(), coef.data_ptr(), stream))
                self.last_grad_norm = coef
            if plan['inc'] is None:
                plan['tstep'].add_(1.0)              # torch.optim keeps the step per parameter
            else:
                plan['tstep'].add_(plan['inc'])
            plan['count'] += 1
            # the update also leaves max|p| of every chunk it wrote: the scale of next step's fp16 x2 weight planes
            _lib.check(lib.vilco_optim_step_dev(0 if self.kind == "AdamW" else 1, ptrs.data_ptr(), plan['numel'].data_ptr(),
                                                plan['chunk_tensor'].data_ptr(), plan['chunk_off'].data_ptr(),
                                                plan['group'].data_ptr(), n, plan['nchunks'], CHUNK, lr, wd, ng,
                                                g0['betas'][0], g0['betas'][1], g0['eps'], g0['momentum'], plan['tstep'].data_ptr(),
                                                None if coef is None or clip_grad_l2norm <= 0 else coef.data_ptr(),
                                                plan['amax'].data_ptr() if OPT_AMAX else None,
                                                None if lr_dev is None else lr_dev.data_ptr(), stream))
            first = plan['first']
            for i, (p, _) in enumerate(items):
                if p.dim() >= 2 and OPT_AMAX:    # matrices: the tensors that get packed
                    emitted[id(p)] = (p, plan['amax_views'][i], first[i + 1] - first[i])
        ops.weights_changed()          # parameter memory was written behind autograd's back: cached weight planes are stale
        for p, parts, cnt in emitted.values():
            ops.tag_weight_amax(p, parts, cnt)
        return None

    def note_replays(self, n=1):
        """a captured step() was replayed n times: account for the step counts the replays advanced on the device"""
        for pl in self._plans or ():
            pl['count'] += n
        ops.weights_changed()


def make_optimizer(model, optimizer_config):
    """same groups as the reference; the duplicated adapter aliases (`pets.*`) are listed again when
    model.use_adapt, as train_utils.py:108-113 does."""
    decay, no_decay, remain = param_groups(model)
    pd = dict(model.named_parameters())
    if getattr(model, "use_adapt", False):
        for mn, m in model.named_modules():
            for pn, p in m.named_parameters():
                full = '%s.%s' % (mn, pn) if mn else pn
                if 'pets' in full:
                    pd[full] = p
    groups = [{"params": [pd[n] for n in decay], "weight_decay": optimizer_config['weight_decay']},
              {"params": [pd[n] for n in no_decay], "weight_decay": 0.0},
              {"params": [pd[n] for n in remain], "weight_decay": optimizer_config['weight_decay']}]
    if optimizer_config["type"] not in ("SGD", "AdamW"):
        raise TypeError("Unsupported optimizer!")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # duplicate-parameter warning: intended, see _passes()
        return FusedOptimizer(groups, lr=optimizer_config["learning_rate"], kind=optimizer_config["type"],
                              momentum=optimizer_config["momentum"],
                              weight_decay=0.01 if optimizer_config["type"] == "AdamW" else 0.0)


def make_scheduler(optimizer, optimizer_config, num_iters_per_epoch, last_epoch=-1):
    """per-iteration schedulers, as train_utils.py:147-211"""
    c = optimizer_config
    if c["warmup"]:
        max_steps = (c["epochs"] + c["warmup_epochs"]) * num_iters_per_epoch
        warmup_steps = c["warmup_epochs"] * num_iters_per_epoch
        if c["schedule_type"] == "cosine":
            return LinearWarmupCosineAnnealingLR(optimizer, warmup_steps, max_steps, last_epoch=last_epoch)
        if c["schedule_type"] == "multistep":
            steps = [num_iters_per_epoch * s for s in c["schedule_steps"]]
            return LinearWarmupMultiStepLR(optimizer, warmup_steps, steps, gamma=c["schedule_gamma"], last_epoch=last_epoch)
        raise TypeError("Unsupported scheduler!")
    max_steps = c["epochs"] * num_iters_per_epoch
    if c["schedule_type"] == "cosine":
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, max_steps, last_epoch=last_epoch)
    if c["schedule_type"] == "multistep":
        steps = [num_iters_per_epoch * s for s in c["schedule_steps"]]
        return optim.lr_scheduler.MultiStepLR(optimizer, steps, gamma=c["schedule_gamma"], last_epoch=last_epoch)
    raise TypeError("Unsupported scheduler!")


def train_step(model, optimizer, scheduler, video_list, task_id=0, prev_out_cls_logits=None,
               clip_grad_l2norm=-1.0, reducer=None):
    """one iteration of train_one_epoch (train_utils.py:322-357): zero_grad, forward, backward,
    (gradient all-reduce), clip, optimizer / scheduler step, adapter EMA."""
    optimizer.zero_grad(set_to_none=True)
    if reducer is not None:
        reducer.begin()
    losses = model(video_list, task_id=task_id, prev_out_cls_logits=prev_out_cls_logits)
    losses['final_loss'].backward()
    if reducer is not None:
        reducer.finish()
    optimizer.step(clip_grad_l2norm=clip_grad_l2norm)
    if scheduler is not None:
        scheduler.step()
    if getattr(model, "use_adapt", False):
        model.post_train_step()
    return losses


class AverageMeter(object):
    """running value / average (train_utils.py:216-244)"""

    def __init__(self):
        self.initialized, self.val, self.avg, self.sum, self.count = False, None, None, None, None

    def update(self, val, n=1):
        if not self.initialized:
            self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * n, n, True
        else:
            self.val = val
            self.sum += val * n
            self.count += n
            self.avg = self.sum / self.count


def check_grid_sync():
    """With VILCO_GRID_SYNC=1 the two-stage reductions finish inside their launch behind a device-wide barrier whose spin
    is bounded (common.h: vilco_grid_barrier): a barrier that gave up leaves sums built from incomplete partials.  The
    count of such give-ups is checked once per epoch (it synchronises the device) and is fatal."""
    if os.environ.get("VILCO_GRID_SYNC", "0") != "1":
        return
    from .. import _lib
    n = _lib.load().vilco_sync_timeouts_read()
    if n != 0:
        raise RuntimeError("vilco: %d in-launch grid barrier(s) timed out -- results of this epoch are invalid "
                           "(run without VILCO_GRID_SYNC=1)" % n)


def train_one_epoch(train_loader, model, optimizer, scheduler, curr_epoch, n_gpu=1, model_ema=None,
                    clip_grad_l2norm=-1, tb_writer=None, print_freq=20, logger=None, cl_name=None, reg_lambda=0.0,
                    prev_out_cls_logits_dict=None, current_task_id=0, reducer=None, graph=None, keep_history=True):
    """One epoch of the continual-learning training loop (train_utils.py:278-423), same arguments.  Per iteration:
    zero_grad, forward with `task_id` and the cached iCaRL/BiC logits of the batch's videos, backward of
    final_loss (+ the EWC / MAS penalty of cl_name, applied as one multi-tensor kernel that adds its gradient to
    p.grad), [gradient all-reduce], fused clip + optimizer step, scheduler step, adapter EMA (`post_train_step`).
    Nothing in the loop reads a device value unless a log line is due (every `print_freq` iterations).
    graph: a vilco_amd.graph.GraphedStep built over (model, optimizer, clip_grad_l2norm[, reducer]) -- iterations whose
    device half can run alone are then replayed as hipGraphs instead of launched kernel by kernel; the others (and the
    first iterations of every new batch shape) take the eager path inside it.
    Returns the list of per-iteration loss dicts (device scalars; empty with keep_history=False), which the reference
    does not -- harmless."""
    from ..cl_methods import regularizers
    model.train()
    model.compute_means = model.cl_name == 'icarl'
    tracker, history = {}, []
    if graph is not None:
        assert graph.model is model and graph.optimizer is optimizer, "the GraphedStep was built over another model / optimizer"
        assert graph.reducer is reducer, "pass the reducer to the GraphedStep as well"
        graph.clip = float(clip_grad_l2norm)
    last_pen = [None]

    def penalty():
        # loss + lambda * sum_i sum_p F_i (theta*_i - theta)^2 (EWC.py:6-22 / MAS.py:5-21): value added to the
        # reported loss, gradient added straight into p.grad.  The penalty is the same on every rank, so adding it
        # after the gradient average equals averaging it.
        last_pen[0] = regularizers.apply_penalty(model, reg_lambda, kind=cl_name) if cl_name in ('ewc', 'mas') else None

    for iter_idx, video_list in enumerate(train_loader, 0):
        prev = []
        for v in video_list:
            if prev_out_cls_logits_dict is not None and v['video_id'] in prev_out_cls_logits_dict:
                prev.append(prev_out_cls_logits_dict[v['video_id']])
        if graph is not None:
            graph.between = penalty
            losses = graph(video_list, task_id=current_task_id, prev_out_cls_logits=prev)
        else:
            optimizer.zero_grad(set_to_none=True)
            if reducer is not None:
                reducer.begin()
            losses = model(video_list, task_id=current_task_id, prev_out_cls_logits=prev)
            losses['final_loss'].backward()
            if reducer is not None:
                reducer.finish()
            penalty()
            optimizer.step(clip_grad_l2norm=clip_grad_l2norm)
        if last_pen[0] is not None:
            losses['final_loss'] = losses['final_loss'].detach() + last_pen[0]
        scheduler.step()
        if model.use_adapt:
            model.post_train_step()
        if keep_history:
            history.append({k: v.detach() for k, v in losses.items()})
        if iter_idx != 0 and iter_idx % print_freq == 0 and logger is not None:
            for k, v in losses.items():
                tracker.setdefault(k, AverageMeter()).update(float(v))
            logger.info('Epoch: [{:03d}][{:05d}/{:05d}]\tLoss {:.2f} ({:.2f})'.format(
                curr_epoch, iter_idx, len(train_loader), tracker['final_loss'].val, tracker['final_loss'].avg))
    if logger is not None:
        logger.info("[Train]: Epoch {:d} finished with lr={:.8f}\n".format(curr_epoch, scheduler.get_last_lr()[0]))
    check_grid_sync()
    return history


@torch.no_grad()
def collect_results(val_loader, model, task_id=0, restore_mode=True):
    """the evaluator's input format (train_utils.py:1049-1098 / 749-752): a dict of flat per-segment columns
    {'video-id': [...], 't-start', 't-end', 'label', 'score': numpy arrays}, from eval-mode forwards (batch size 1).
    restore_mode=False leaves the model in eval mode, as the reference's validation functions do."""
    was_training = model.training
    model.eval()
    res = {'video-id': [], 't-start': [], 't-end': [], 'label': [], 'score': []}
    for video_list in val_loader:
        for out in model(video_list, task_id=task_id, is_training=False):
            n = out['segments'].shape[0]
            if n > 0:
                res['video-id'].extend([out['video_id']] * n)
                res['t-start'].append(out['segments'][:, 0])
                res['t-end'].append(out['segments'][:, 1])
                res['label'].append(out['labels'])
                res['score'].append(out['scores'])
    for k, dt in (('t-start', torch.float32), ('t-end', torch.float32), ('label', torch.int64), ('score', torch.float32)):
        res[k] = torch.cat(res[k]).numpy() if res[k] else torch.zeros(0, dtype=dt).numpy()
    if restore_mode:
        model.train(was_training)
    return res


def results_to_anet_json(results, idx_classes=None, version="1.0"):
    """the ActivityNet-style object the reference dumps for the retrieval metric (train_utils.py:1118-1127, :771-774):
    {"version": "1.0", "external_data": "", "results": {video_id: [{"segment": [s, e], "score", "label"}, ...]}} with the
    rows of a video in result order.  idx_classes: {label id: class name} -- the reference labels rows with the NAME from
    the Ego4D MQ class table hard-wired there (:1104); that table is dataset metadata, so it is an argument here (None keeps
    the integer id)."""
    out = {}
    for vid, s, e, l, sc in zip(results['video-id'], results['t-start'], results['t-end'], results['label'],
                                results['score']):
        out.setdefault(vid, []).append({"segment": [float(s), float(e)], "score": float(sc),
                                        "label": int(l) if idx_classes is None else idx_classes[int(l)]})
    return {"version": version, "external_data": "", "results": out}


def _validate_tasks(val_qilDatasetList, model, current_task_id, evaluator, retrieval_eval, idx_classes, ext_score_file,
                    logger, dataset_name):
    """the loop valid_one_epoch_cl_single_gpu and final_validate share (train_utils.py:1049-1160 / 1213-1323): every task
    learnt so far is validated on its own loader; per task the records go (a) as the ANet JSON object to the retrieval
    metric and (b) as the flat result dict to the evaluator.  Yields (n_task, num_queries, recall table [5 tIoU, 2 ranks]
    or None, avg_mAP)."""
    model.eval()
    for b in getattr(model, 'list_bias_layers', ()):
        b.eval()
    for n_task, (val_loader, num_queries) in enumerate(val_qilDatasetList.get_valSet_by_taskNum(current_task_id + 1)):
        results = collect_results(val_loader, model, task_id=current_task_id, restore_mode=False)
        eval_result = None
        if dataset_name in ("ego4d", "ego4d_cl") and retrieval_eval is not None:
            eval_result = retrieval_eval(results_to_anet_json(results, idx_classes), current_task_id=n_task)
            if logger is not None:
                for i, t in enumerate((0.1, 0.2, 0.3, 0.4, 0.5)):
                    for j, r in enumerate((1, 5)):
                        logger.info(f'Task {n_task} Rank {r}x @ tIoU {t} is {eval_result[i, j]}')
        assert evaluator is not None
        if ext_score_file is not None and isinstance(ext_score_file, str):
            raise NotImplementedError("external classification scores (postprocess_results) are outside the hot path")
        mAP, avg_mAP, tious = evaluator.evaluate(results, current_task_id=current_task_id, verbose=False)
        if logger is not None:
            for tiou, m in zip(tious, mAP):
                logger.info(f'Task {n_task} tIoU = {tiou:.1f}: mAP = {m * 100:.2f} %')
            logger.info(f'Task {n_task} Average Map is :{avg_mAP * 100: .2f} %')
        yield n_task, num_queries, eval_result, avg_mAP


def _meters():
    return {k: AverageMeter() for k in ('R1_0_3', 'R5_0_3', 'R1_0_5', 'R5_0_5', 'mAP')}


def _update(m, eval_result, avg_mAP, n):
    if eval_result is not None:
        for k, (i, j) in (('R1_0_3', (2, 0)), ('R5_0_3', (2, 1)), ('R1_0_5', (4, 0)), ('R5_0_5', (4, 1))):
            m[k].update(eval_result[i, j], n)
    m['mAP'].update(avg_mAP, n)


@torch.no_grad()
def valid_one_epoch_cl_single_gpu(val_qilDatasetList, model, curr_epoch, current_task_id, ext_score_file=None, evaluator=None,
                                  output_file=None, tb_writer=None, print_freq=20, logger=None, dataset_name='ego4d_cl',
                                  retrieval_eval=None, idx_classes=None):
    """Validation over the tasks learnt so far, the reference's signature and return value (train_utils.py:1016-1173):
    query-weighted means (R1@0.3, R5@0.3, R1@0.5, R5@0.5, mAP).  Two hooks stand where the reference reaches outside the
    path: `retrieval_eval(json_obj, current_task_id=n_task) -> recall[5, 2]` takes the place of the JSON file +
    `evaluation_retrieval` round trip (:1128-1143: the object is the file's content), `idx_classes` is the class-name table."""
    assert (evaluator is not None) or (output_file is not None)
    m = _meters()
    for n_task, nq, er, avg_mAP in _validate_tasks(val_qilDatasetList, model, current_task_id, evaluator, retrieval_eval,
                                                   idx_classes, ext_score_file, logger, dataset_name):
        _update(m, er, avg_mAP, nq)
    return m['R1_0_3'].avg, m['R5_0_3'].avg, m['R1_0_5'].avg, m['R5_0_5'].avg, m['mAP'].avg


@torch.no_grad()
def final_validate(val_qilDatasetList, model, curr_epoch, current_task_id, ext_score_file=None, evaluator=None, output_file=None,
                   tb_writer=None, print_freq=20, logger=None, dataset_name='ego4d_cl', list_val_recall_ii=None,
                   list_val_mAP_ii=None, type_val='val', retrieval_eval=None, idx_classes=None):
    """train_utils.py:1176-1352: as above plus the forgetting bookkeeping -- the newest task's R1@0.5 / mAP are appended to
    list_val_recall_ii / list_val_mAP_ii[type_val], older tasks contribute (value when learnt - value now) to the
    backward-forgetting means.  Returns the five means + (BWF R1@0.5, BWF mAP)."""
    assert (evaluator is not None) or (output_file is not None)
    list_val_recall_ii = {'val': []} if list_val_recall_ii is None else list_val_recall_ii
    list_val_mAP_ii = {'val': []} if list_val_mAP_ii is None else list_val_mAP_ii
    m, bwf_r, bwf_m = _meters(), AverageMeter(), AverageMeter()
    for n_task, nq, er, avg_mAP in _validate_tasks(val_qilDatasetList, model, current_task_id, evaluator, retrieval_eval,
                                                   idx_classes, ext_score_file, logger, dataset_name):
        _update(m, er, avg_mAP, nq)
        if n_task == current_task_id:
            list_val_recall_ii[type_val].append(er[4, 0])
            list_val_mAP_ii[type_val].append(avg_mAP)
        elif n_task < current_task_id:
            bwf_r.update(list_val_recall_ii[type_val][n_task] - er[4, 0], nq)
            bwf_m.update(list_val_mAP_ii[type_val][n_task] - avg_mAP, nq)
    return (m['R1_0_3'].avg, m['R5_0_3'].avg, m['R1_0_5'].avg, m['R5_0_5'].avg, m['mAP'].avg, bwf_r.avg, bwf_m.avg)


def merge_results(parts):
    """concatenate evaluator-format result dicts (the output of collect_results) in the given order"""
    out = {'video-id': [], 't-start': [], 't-end': [], 'label': [], 'score': []}
    for p in parts:
        out['video-id'].extend(p['video-id'])
    for k in ('t-start', 't-end', 'label', 'score'):
        out[k] = np.concatenate([np.asarray(p[k]) for p in parts]) if parts else np.zeros(0)
    return out


def collect_results_sharded(val_batches, model, task_id=0, rank=0, world=1, group=None):
    """Validation sharded over data-parallel ranks (the reference validates on rank 0 only, train_cl.py:283): rank r
    runs batches r, r + world, ... through `collect_results`, the per-rank result dicts travel once
    (all_gather_object) and every rank returns the merged dict in rank order -- the same rows a single-rank pass
    produces, up to that order (the evaluator groups by video id)."""
    val_batches = list(val_batches)
    mine = collect_results(val_batches[rank::world], model, task_id=task_id)
    if world == 1:
        return mine
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return merge_results(parts)
