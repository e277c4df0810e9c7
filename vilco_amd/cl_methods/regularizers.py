"""EWC / MAS continual-learning regularisers on the HIP path (reference: MQ/libs/cl_methods/EWC.py:6-56,
MAS.py:5-57; called from train_utils.py:337-344 and train_cl.py:381-385).

`model.reg_params` keeps the reference's layout -- {'fisher' | 'importance': [ {name: tensor}, ... one dict per
finished task ], 'optpar': [ {name: tensor}, ... ]} -- so checkpoints ('reg_params' key, train_cl.py:306) interchange.

The penalty  lambda * sum_tasks sum_names sum_elems F (theta* - theta[:len(theta*)])^2  and its gradient are ONE
multi-tensor launch (vilco_cl_penalty, csrc/optim.hip) over a chunk table of every (task, name) pair, instead of
one autograd graph node per pair per step: the value comes back as a device scalar, the gradient
-2 lambda F (theta* - theta) is accumulated straight into p.grad (after backward, before clipping, which is where the
reference's autograd puts it too)."""
import torch

from .. import _lib

CHUNK = 16384


def _entries(model, kind):
    reg = getattr(model, 'reg_params', None) or {}
    key = 'fisher' if kind == 'ewc' else 'importance'
    if key not in reg or 'optpar' not in reg:
        return []
    out = []
    for imp_d, opt_d in zip(reg[key], reg['optpar']):
        for name, p in model.named_parameters():
            if 'scale' not in name and name in imp_d:           # EWC.py:15 / MAS.py:14
                imp, opt = imp_d[name], opt_d[name]
                assert imp.shape == opt.shape and opt.numel() <= p.numel() and opt.shape[1:] == p.shape[1:]
                out.append((p, imp, opt))
    return out


def apply_penalty(model, reg_lambda, kind='ewc'):
    """adds the penalty gradient to p.grad (allocating zeros where a parameter has none) and returns the penalty value
    as a device scalar, or None when no task has been consolidated yet."""
    items = _entries(model, kind)
    if not items:
        return None
    lib = _lib.load()
    dev = items[0][0].device
    for p, imp, opt in items:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if not (p.is_contiguous() and p.grad.is_contiguous() and imp.is_contiguous() and opt.is_contiguous()
                and imp.is_cuda and opt.is_cuda and imp.dtype == torch.float32):
            raise RuntimeError("vilco_cl_penalty needs contiguous fp32 tensors on the HIP device")
    numel = [opt.numel() for _, _, opt in items]                 # prefix of the (possibly grown) parameter
    ct, co = [], []
    for i, n in enumerate(numel):
        for off in range(0, n, CHUNK):
            ct.append(i)
            co.append(off)
    ptrs = torch.tensor([[p.data_ptr() for p, _, _ in items], [p.grad.data_ptr() for p, _, _ in items],
                         [imp.data_ptr() for _, imp, _ in items], [opt.data_ptr() for _, _, opt in items]],
                        dtype=torch.int64).to(dev, non_blocking=True)
    t_numel = torch.tensor(numel, dtype=torch.int64).to(dev, non_blocking=True)
    t_ct = torch.tensor(ct, dtype=torch.int32).to(dev, non_blocking=True)
    t_co = torch.tensor(co, dtype=torch.int64).to(dev, non_blocking=True)
    partial = torch.empty(max(len(ct), 1), dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(lib.vilco_cl_penalty(ptrs.data_ptr(), t_numel.data_ptr(), t_ct.data_ptr(), t_co.data_ptr(), len(items),
                                    len(ct), CHUNK, float(reg_lambda), int(len({id(p) for p, _, _ in items}) < len(items)),
                                    partial.data_ptr(), out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream))
    return out[0]


def get_regularized_loss(loss, model, reg_lambda, kind='ewc'):
    """reference-shaped entry (EWC.py:6 / MAS.py:5) for callers that want the autograd form: loss + penalty with the
    penalty as ordinary tensor expressions.  train_one_epoch uses `apply_penalty` instead."""
    for p, imp, opt in _entries(model, kind):
        loss = loss + (imp * (opt - p[:opt.size(0)]).pow(2)).sum() * reg_lambda
    return loss


def on_task_update(loader_task, device, optimizer, model, kind='ewc', group=None, data_parallel=False):
    """importance of the weights after a task (EWC.py:24-56 / MAS.py:23-57): one pass over the task's loader with
    zero_grad before every batch -- so, as in the reference, what is kept is the LAST batch's gradient (squared for
    EWC, absolute for MAS) -- plus a copy of the parameters.
    data_parallel / group: the run is data parallel over `group` (None = the default group).  Every rank sees the last batch of ITS shard, so the importances
    are averaged over the ranks (the reference wraps this pass in no DDP hook and lets them differ; the penalty is
    applied after the gradient exchange, so different importances would pull the replicas apart)."""
    model.train()
    reg = model.reg_params
    key = 'fisher' if kind == 'ewc' else 'importance'
    if not (key in reg and 'optpar' in reg):
        reg[key], reg['optpar'] = [], []
    for video_list in loader_task:
        optimizer.zero_grad(set_to_none=True)
        model(video_list)['final_loss'].backward()
    imp_d, opt_d = {}, {}
    for name, p in model.named_parameters():
        if p.grad is not None:
            opt_d[name] = p.data.clone()
            g = p.grad.data.clone()
            imp_d[name] = g.pow(2) if kind == 'ewc' else g.abs()
    if data_parallel and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        ws = float(torch.distributed.get_world_size(group))
        for name in sorted(imp_d):                      # the same order on every rank
            torch.distributed.all_reduce(imp_d[name], group=group)
            imp_d[name].div_(ws)
    reg[key].append(imp_d)
    reg['optpar'].append(opt_d)
    return reg


def on_task_mas_update(loader_task, device, optimizer, model):
    return on_task_update(loader_task, device, optimizer, model, kind='mas')


def get_mas_regularized_loss(loss, model, reg_lambda):
    return get_regularized_loss(loss, model, reg_lambda, kind='mas')
