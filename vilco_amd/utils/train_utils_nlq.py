"""Step glue of the NLQ training loop on the HIP path (BASELINE configs[3]; reference:
NLQ/libs/utils/train_utils.py -- make_optimizer :63-250, make_scheduler :253-315, train_one_epoch :376-521,
valid_one_epoch_cl_single_gpu :705-781, final_validate :784-874).

What differs from the MQ glue (utils/train_utils.py):
  * `make_optimizer(model, cfg, head_backbone_group)`: the default mode has FOUR groups -- decay / no-decay and their
    'encoder.' twins at `learning_rate * coef_lr` (the RoBERTa text encoder of the non-CL model; empty for the CL model) --
    and silently leaves parameters no rule matches OUT of the optimizer (:208-213; MQ puts them in a decayed "remain"
    group); `head_backbone_group=True` (train_cl.py:115-118, taken when backbone_lr_weight != 1) splits by `"head" in name`
    into head decay / head no-decay at `learning_rate` and backbone decay / no-decay at
    `learning_rate * backbone_lr_weight`, and asserts that every parameter was classified (:199-202).
  * `make_scheduler` knows the "constant" schedule (WarmupLRScheduler).
  * validation produces the evaluator's NLQ record format: a list of
    {'query_idx', 'annotation_uid', 'predicted_times': [[start, end, score], ...], 'clip_uid'} for the ego4d datasets,
    {'query_id', 'predicted_times', 'video_id'} otherwise (:735-757).
The optimizer is the same `FusedOptimizer` (multi-tensor clip + AdamW / SGD kernels, per-group learning rates and decay)."""
import warnings

import torch
from torch import optim

from ..modeling.blocks import AffineDropPath, LayerNorm, MaskedConv1D, Scale
from . import train_utils as mq
from .lr_schedulers import LinearWarmupCosineAnnealingLR, LinearWarmupMultiStepLR, WarmupLRScheduler
from .train_utils import AverageMeter, FusedOptimizer, fix_random_seed, save_checkpoint  # noqa: F401

_WHITE = (torch.nn.Linear, torch.nn.Conv1d, MaskedConv1D)
_BLACK = (LayerNorm, torch.nn.GroupNorm)


def _rule(m, pn):
    """'decay' / 'no_decay' / None for parameter `pn` (relative name) seen from module m: the chain of :89-103"""
    if pn.endswith('bias'):
        return 'no_decay'
    if pn.endswith('weight') and isinstance(m, _WHITE):
        return 'decay'
    if pn.endswith('weight') and isinstance(m, _BLACK):
        return 'no_decay'
    if pn.endswith('scale') and isinstance(m, (Scale, AffineDropPath)):
        return 'no_decay'
    if pn.endswith('rel_pe'):
        return 'no_decay'
    return None


def param_groups(model, head_backbone_group=False):
    """sorted name lists per group, in the order of the reference's `optim_groups`:
    default -> (decay, no_decay, encoder_decay, encoder_no_decay); head_backbone_group -> (head_decay, head_no_decay,
    backbone_decay, backbone_no_decay).  `m.named_parameters()` is recursive, so a parameter is classified from every
    ancestor module; a name may therefore sit in a decay AND a no-decay set only if two ancestors disagree, which the
    reference asserts against in the head / backbone mode (:193-198)."""
    sets = {k: set() for k in ('decay', 'no_decay', 'encoder_decay', 'encoder_no_decay', 'head_decay', 'head_no_decay',
                               'backbone_decay', 'backbone_no_decay')}
    for mn, m in model.named_modules():
        for pn, _ in m.named_parameters():
            full = '%s.%s' % (mn, pn) if mn else pn
            if 'encoder.' in full:
                r = _rule(m, pn)
                if r is None:                                   # :100-103: any other weight decays, the rest does not
                    r = 'decay' if pn.endswith('weight') else 'no_decay'
                sets['encoder_' + r].add(full)
            else:
                r = _rule(m, pn)
                if r is not None:
                    sets[r].add(full)
            if head_backbone_group:
                r = _rule(m, pn)
                if r is not None:
                    sets[('head_' if 'head' in full else 'backbone_') + r].add(full)
    if head_backbone_group:
        names = {n for n, _ in model.named_parameters()}
        hd, hn, bd, bn = (sets[k] for k in ('head_decay', 'head_no_decay', 'backbone_decay', 'backbone_no_decay'))
        assert not (hd & bd) and not (hn & bn) and not (bd & bn), "a parameter made it into two groups"
        missing = names - (hd | hn | bd | bn)
        assert not missing, "parameters %s were not separated into either decay/no_decay set!" % (sorted(missing),)
        return tuple(sorted(s) for s in (hd, hn, bd, bn))
    return tuple(sorted(sets[k]) for k in ('decay', 'no_decay', 'encoder_decay', 'encoder_no_decay'))


def make_optimizer(model, optimizer_config, head_backbone_group=False):
    c = optimizer_config
    pd = dict(model.named_parameters())
    if getattr(model, "use_adapter", False):                    # :172-177: the adapter aliases under `pets.*`
        for mn, m in model.named_modules():
            for pn, p in m.named_parameters():
                full = '%s.%s' % (mn, pn) if mn else pn
                if 'pets' in full:
                    pd[full] = p
    g = param_groups(model, head_backbone_group)
    lr, wd = c["learning_rate"], c['weight_decay']
    if head_backbone_group:
        blr = lr * c["backbone_lr_weight"]
        groups = [{"params": [pd[n] for n in g[0]], "weight_decay": wd, "lr": lr},
                  {"params": [pd[n] for n in g[1]], "weight_decay": 0.0, "lr": lr},
                  {"params": [pd[n] for n in g[2]], "weight_decay": wd, "lr": blr},
                  {"params": [pd[n] for n in g[3]], "weight_decay": 0.0, "lr": blr}]
    else:
        elr = lr * c.get("coef_lr", 1)
        groups = [{"params": [pd[n] for n in g[0]], "weight_decay": wd},
                  {"params": [pd[n] for n in g[1]], "weight_decay": 0.0},
                  {"params": [pd[n] for n in g[2]], "weight_decay": wd, "lr": elr},
                  {"params": [pd[n] for n in g[3]], "weight_decay": 0.0, "lr": elr}]
    if c["type"] not in ("SGD", "AdamW"):
        raise TypeError("Unsupported optimizer!")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return FusedOptimizer(groups, lr=lr, kind=c["type"], momentum=c["momentum"],
                              weight_decay=0.01 if c["type"] == "AdamW" else 0.0)


def make_scheduler(optimizer, optimizer_config, num_iters_per_epoch, last_epoch=-1):
    """per-iteration schedulers, as :253-315 (cosine / constant / multistep behind a linear warm-up; cosine without)"""
    c = optimizer_config
    if c["warmup"]:
        max_steps = (c["epochs"] + c["warmup_epochs"]) * num_iters_per_epoch
        warmup_steps = c["warmup_epochs"] * num_iters_per_epoch
        if c["schedule_type"] == "cosine":
            return LinearWarmupCosineAnnealingLR(optimizer, warmup_steps, max_steps, last_epoch=last_epoch)
        if c["schedule_type"] == "constant":
            return WarmupLRScheduler(optimizer, warmup_steps, last_epoch=last_epoch)
        if c["schedule_type"] == "multistep":
            steps = [num_iters_per_epoch * s for s in c["schedule_steps"]]
            return LinearWarmupMultiStepLR(optimizer, warmup_steps, steps, gamma=c["schedule_gamma"], last_epoch=last_epoch)
        raise TypeError("Unsupported scheduler!")
    if c["schedule_type"] == "cosine":
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, c["epochs"] * num_iters_per_epoch, last_epoch=last_epoch)
    raise TypeError("Unsupported scheduler!")


def train_one_epoch(train_loader, model, optimizer, scheduler, curr_epoch, model_ema=None, clip_grad_l2norm=-1,
                    tb_writer=None, print_freq=20, cl_name=None, reg_lambda=0.0, prev_out_cls_logits_dict=None,
                    current_task_id=0, reducer=None, graph=None, keep_history=True, logger=None):
    """the reference's signature (:376-390); the iteration itself is the MQ one (same sequence: zero_grad, forward with
    task_id, [EWC / MAS penalty], backward, clip, optimizer and scheduler step, adapter EMA)"""
    if not hasattr(model, 'use_adapt'):
        model.use_adapt = getattr(model, 'use_adapter', False)
    return mq.train_one_epoch(train_loader, model, optimizer, scheduler, curr_epoch, 1, model_ema=model_ema,
                              clip_grad_l2norm=clip_grad_l2norm, tb_writer=tb_writer, print_freq=print_freq, logger=logger,
                              cl_name=cl_name, reg_lambda=reg_lambda, prev_out_cls_logits_dict=prev_out_cls_logits_dict,
                              current_task_id=current_task_id, reducer=reducer, graph=graph, keep_history=keep_history)


def prediction_records(video_list, output, dataset="ego4d_cl"):
    """the evaluator records of one validation batch (:735-757); ONE host copy per clip (segments and scores together)"""
    recs = []
    for v, o in zip(video_list, output):
        assert o['segments'].shape[0] > 0
        rows = torch.cat([o['segments'].float(), o['scores'].float()[:, None]], dim=1).cpu().tolist()
        if dataset in ("ego4d", "ego4d_cl"):
            uid, idx = v['query_id'].split("_")[:2]
            recs.append({'query_idx': int(idx), 'annotation_uid': uid, 'predicted_times': rows, 'clip_uid': v['video_id']})
        else:
            recs.append({'query_id': v['query_id'], 'predicted_times': rows, 'video_id': v['video_id']})
    return recs


def _eval_mode(model):
    model.eval()
    for b in getattr(model, 'list_bias_layers', ()):
        b.eval()


@torch.no_grad()
def valid_one_epoch_cl_single_gpu(val_qilDatasetList, model, current_epoch, current_task_id, evaluator=None,
                                  output_file=None, tb_writer=None, print_freq=20, dataset_name='ego4d_cl'):
    """:705-781 -- every template learnt so far is validated in turn; the records ACCUMULATE over the templates and the
    evaluator sees the running list each time; the returned R@1 is that of the LAST evaluation (the reference updates its
    meter once, after the loop: :776)."""
    _eval_mode(model)
    results, performance, num_queries = [], None, 1
    for val_loader, num_queries in val_qilDatasetList.get_valSet_by_taskNum(current_task_id + 1):
        for video_list in val_loader:
            output = model(video_list, task_id=current_task_id, is_training=False)
            results.extend(prediction_records(video_list, output, evaluator.dataset))
        performance, _ = evaluator.evaluate(results, verbose=True)
    total = AverageMeter()
    total.update(performance[0, 0], num_queries)
    return total.avg


@torch.no_grad()
def final_validate(val_qilDatasetList, model, current_epoch, current_task_id, evaluator=None, output_file=None,
                   tb_writer=None, print_freq=20, dataset_name='ego4d_cl', list_val_recall_ii=None, type_val='val'):
    """:784-874 -- as above, plus the per-template bookkeeping: R@1 of the newest template is appended to
    list_val_recall_ii[type_val], older templates contribute their drop to the backward-forgetting meter.
    Returns the query-weighted mean R@1 (the reference prints it; returning it is harmless)."""
    if list_val_recall_ii is None:
        list_val_recall_ii = {'val': [], 'test': []}
    _eval_mode(model)
    total, bwf, results = AverageMeter(), AverageMeter(), []
    for n_task, (val_loader, num_queries) in enumerate(val_qilDatasetList.get_valSet_by_taskNum(current_task_id + 1)):
        for video_list in val_loader:
            output = model(video_list, task_id=current_task_id, is_training=False, val_qilDatasetList=val_qilDatasetList)
            results.extend(prediction_records(video_list, output, evaluator.dataset))
        assert evaluator.dataset == "ego4d_cl"
        performance, _ = evaluator.evaluate(results, verbose=True)
        if n_task == current_task_id:
            list_val_recall_ii[type_val].append(performance[0, 0])
        elif n_task < current_task_id:
            bwf.update(list_val_recall_ii[type_val][n_task] - performance[0, 0], num_queries)
        total.update(performance[0, 0], num_queries)
    return total.avg
