"""Losses of the MQ heads as device tensor expressions (reference: MQ/libs/modeling/losses.py:5-168).
The training step does NOT go through this file: labelling + focal / DIoU / action-localisation losses run as the fused
kernels `vilco_mq_loss_fwd / _bwd` (csrc/loss.hip, ops.mq_loss; DESIGN_LOG.md 3.5).  These functions are the
`VILCO_FUSED_LOSS=0` path the fused kernels are tested equal to (tests/test_loss_gpu.py) and what iCaRL / BiC
distillation terms and the narration SSL loss are built from."""
import torch
from torch.nn import functional as F


def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2.0, reduction: str = "none"):
    inputs, targets = inputs.float(), targets.float()
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def _iou_terms(inp, tgt, eps, check=True):
    inp, tgt = inp.float(), tgt.float()
    if check:      # the reference's asserts (losses.py:96-97): two host syncs; the dense training path skips them
        assert (inp >= 0.0).all(), "predicted offsets must be non-negative"
        assert (tgt >= 0.0).all(), "GT offsets must be non-negative"
    lp, rp, lg, rg = inp[:, 0], inp[:, 1], tgt[:, 0], tgt[:, 1]
    inter = torch.min(rp, rg) + torch.min(lp, lg)
    union = (lp + rp) + (lg + rg) - inter
    return lp, rp, lg, rg, inter / union.clamp(min=eps)


def _reduce(loss, reduction):
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


def ctr_giou_loss_1d(input_offsets, target_offsets, reduction: str = 'none', eps: float = 1e-8):
    _, _, _, _, iou = _iou_terms(input_offsets, target_offsets, eps)
    return _reduce(1.0 - iou, reduction)


def ctr_diou_loss_1d(input_offsets, target_offsets, reduction: str = 'none', eps: float = 1e-8, check: bool = True):
    lp, rp, lg, rg, iou = _iou_terms(input_offsets, target_offsets, eps, check)
    len_c = torch.max(lp, lg) + torch.max(rp, rg)
    rho = 0.5 * (rp - lp - rg + lg)
    return _reduce(1.0 - iou + torch.square(rho / len_c.clamp(min=eps)), reduction)
